#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_lbvh; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "device_built or config4_million" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $out/pytest.log
for c in 64 16 256 0; do echo "== cluster $c"; HJ_LBVH_CLUSTER=$c timeout 600 python tools/lbvh_probe.py 2>&1 | tee $out/lbvh_$c.txt; done
