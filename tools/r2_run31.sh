#!/bin/bash
export GPU_MAX_HW_QUEUES=16
for n in 1 2 4 8; do HJ_COMM_SHARED_GPU=1 timeout 300 python bench.py --inproc --gpus $n --steps 3 2>&1 | grep -o '"value": [0-9.]*, "unit": "Mrays/s", "n_gpus": [0-9]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo; done
