#!/bin/bash
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r2_batch2
timeout 900 python tools/batch_probe.py 2>&1 | tee gpurun_out/r2_batch2/batch.txt
