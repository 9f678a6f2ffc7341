#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_lbvh3; mkdir -p $out; rm -f $out/lbvh.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "device_built or config4_million" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $out/pytest.log
for v in "HJ_LBVH_SAH=0" "HJ_LBVH_SAH=1" "HJ_LBVH_SAH=1 HJ_LBVH_CLUSTER=48"; do echo "== $v" | tee -a $out/lbvh.txt; env $v timeout 600 python tools/lbvh_probe.py 2>&1 | tee -a $out/lbvh.txt; done
