#!/bin/bash
export GPU_MAX_HW_QUEUES=8
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash tools/profile_config.sh r02 c4 > gpurun_out/prof_c4.log 2>&1; tail -1 gpurun_out/prof_c4.log | cut -c1-200
timeout 600 python tools/lbvh_probe.py 2>&1 | tee gpurun_out/r02_lbvh_probe.txt
