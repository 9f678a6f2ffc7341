#!/bin/bash
export GPU_MAX_HW_QUEUES=8
bash tools/profile_config.sh r02 c4 > gpurun_out/prof_c4.log 2>&1; tail -1 gpurun_out/prof_c4.log | cut -c1-200
