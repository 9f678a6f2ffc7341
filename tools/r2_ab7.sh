#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab7; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config4 or golden or config1 or device_built" > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log; tail -4 $out/pytest.log
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 32" tools/ab_variants.sh cur:HJ_TREELET_SLOTS=0 cur cur:HJ_TREELET_SLOTS=8 cur:HJ_TREELET_SLOTS=16 w7 w8 2>&1 | tee $out/c4.txt
echo "== C2"; tools/ab_variants.sh cur:HJ_TREELET_SLOTS=0 cur cur:HJ_TREELET_SLOTS=8 w7 2>&1 | tee $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh cur:HJ_TREELET_SLOTS=0 cur w7 2>&1 | tee $out/c3.txt
