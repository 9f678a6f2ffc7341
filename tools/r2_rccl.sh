#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_rccl; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rccl or comm or contexts" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 $out/pytest.log
