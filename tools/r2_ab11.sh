#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab11; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or config1 or random or divergent" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
echo "== C2"; tools/ab_variants.sh fixedburst cur cur:HJ_BURST_MAX=6 cur:HJ_BURST_MAX=8 cur:HJ_BURST_MAX=12 cur:HJ_BURST_MAX=8,HJ_LEAF_GO=16 cur:HJ_BURST_MAX=8,HJ_LEAF_GO=32 cur:HJ_BURST_MAX=8,HJ_STEP_MIN=24 cur:HJ_BURST_MAX=8,HJ_STEP_MIN=8 cur:HJ_INNER_BURST=2,HJ_BURST_MAX=8 cur:HJ_INNER_BURST=3,HJ_BURST_MAX=8 2>&1 | tee $out/c2.txt
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 32" tools/ab_variants.sh fixedburst cur cur:HJ_BURST_MAX=8 cur:HJ_BURST_MAX=12 cur:HJ_BURST_MAX=12,HJ_LEAF_GO=16 cur:HJ_BURST_MAX=16,HJ_LEAF_GO=32 2>&1 | tee $out/c4.txt
