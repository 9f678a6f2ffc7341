#!/bin/bash
# knobs again on the rotated, child-ordered trees
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab63; mkdir -p $out
V="cur cur:HJ_INNER_BURST=4 cur:HJ_INNER_BURST=6 cur:HJ_REFILL_MIN=24 cur:HJ_REFILL_MIN=40 cur:HJ_BVH_CHILD_ORDER=2 cur:HJ_BVH_ROTATE=16 cur:HJ_COLLAPSE_PCT=40 cur:HJ_COLLAPSE_PCT=60"
echo "== C2"; PROBE_ARGS="" tools/ab_variants.sh $V 2>&1 | tee $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh $V 2>&1 | tee $out/c3.txt
V="cur cur:HJ_INNER_BURST=6 cur:HJ_INNER_BURST=10 cur:HJ_REFILL_MIN=24 cur:HJ_REFILL_MIN=40 cur:HJ_BVH_CHILD_ORDER=2 cur:HJ_BVH_ROTATE=16 cur:HJ_COLLAPSE_PCT=40 cur:HJ_COLLAPSE_PCT=60"
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 64" tools/ab_variants.sh $V 2>&1 | tee $out/c4.txt
