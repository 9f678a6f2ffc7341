#!/bin/bash
# sweep of the walk's knobs on the merged-step kernel
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab47; mkdir -p $out
V="cur cur:HJ_INNER_BURST=3 cur:HJ_INNER_BURST=5 cur:HJ_INNER_BURST=6 cur:HJ_REFILL_MIN=24 cur:HJ_REFILL_MIN=40 cur:HJ_REFILL_MIN=48 w8 cur:HJ_PAIR_LEAVES=1 cur:HJ_PAIR_LEAVES=1,HJ_INNER_BURST=4 cur:HJ_PAIR_LEAVES=1,HJ_INNER_BURST=6 cur:HJ_PAIR_LEAVES=1,HJ_INNER_BURST=4,HJ_NODE_ORDER=0"
echo "== C2"; PROBE_ARGS="" tools/ab_variants.sh $V 2>&1 | tee $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh $V 2>&1 | tee $out/c3.txt
V="cur cur:HJ_INNER_BURST=6 cur:HJ_INNER_BURST=10 cur:HJ_INNER_BURST=12 cur:HJ_REFILL_MIN=24 cur:HJ_REFILL_MIN=40 w8"
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 64" tools/ab_variants.sh $V 2>&1 | tee $out/c4.txt
