#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_lbvh4; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_obj_and_image_io.py -m gpu -x -q -k "device_built or config4_million or device_bvh or cli" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
HJ_LBVH_TIMING=1 timeout 600 python tools/lbvh_probe.py 2>&1 | tail -9 | tee $out/lbvh.txt
HJ_LBVH_SAH=0 timeout 600 python tools/lbvh_probe.py 2>&1 | tail -2 | tee -a $out/lbvh.txt
