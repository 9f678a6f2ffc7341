#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab23; mkdir -p $out
echo "== C4 (pairs)"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 32" tools/ab_variants.sh cur cur:HJ_INNER_BURST=6 cur:HJ_INNER_BURST=8 cur:HJ_INNER_BURST=12 cur:HJ_INNER_BURST=3 cur:HJ_REFILL_MIN=24 cur:HJ_REFILL_MIN=40 cur:HJ_POOL=16384 2>&1 | tee $out/c4.txt
