#!/bin/bash
# where small-tree rounds (7 steps, refill at 24) stop paying: 20 k / 60 k / 200 k triangles
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab71; mkdir -p $out
V="cur:HJ_INNER_BURST=8,HJ_REFILL_MIN=32 cur:HJ_INNER_BURST=7,HJ_REFILL_MIN=24 cur:HJ_INNER_BURST=8,HJ_REFILL_MIN=24 cur:HJ_INNER_BURST=7,HJ_REFILL_MIN=32"
echo "== 20k"; PROBE_ARGS="--kind 2 --tris 20000 --size 1024 --spp 256" tools/ab_variants.sh $V 2>&1 | tee $out/c20k.txt
echo "== 60k"; PROBE_ARGS="--kind 2 --tris 60000 --size 1024 --spp 256" tools/ab_variants.sh $V 2>&1 | tee $out/c60k.txt
echo "== 200k"; PROBE_ARGS="--kind 2 --tris 200000 --size 2048 --spp 64" tools/ab_variants.sh $V 2>&1 | tee $out/c200k.txt
