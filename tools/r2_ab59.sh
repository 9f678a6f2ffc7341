#!/bin/bash
# child order in the device-built tree (host top tree + cluster kernel): LBVH probe with and without, device-BVH parity tests
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab59; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "device_built or config4 or large_mesh" > $out/pytest.log 2>&1; rc=$?; tail -2 $out/pytest.log
[ $rc -ne 0 ] && exit 1
( echo "== default (fewer shapes first)"; timeout 600 python tools/lbvh_probe.py 2>&1; echo "== HJ_BVH_CHILD_ORDER=0"; HJ_BVH_CHILD_ORDER=0 timeout 600 python tools/lbvh_probe.py 2>&1 ) | tee $out/lbvh.txt
