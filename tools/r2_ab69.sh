#!/bin/bash
# steps per round x refill threshold together (shorter rays after the tree passes)
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab69; mkdir -p $out
V="cur cur:HJ_INNER_BURST=6,HJ_REFILL_MIN=24 cur:HJ_INNER_BURST=6,HJ_REFILL_MIN=20 cur:HJ_INNER_BURST=6,HJ_REFILL_MIN=28 cur:HJ_INNER_BURST=7,HJ_REFILL_MIN=24 cur:HJ_INNER_BURST=8,HJ_REFILL_MIN=24 cur:HJ_INNER_BURST=7,HJ_REFILL_MIN=20 cur:HJ_INNER_BURST=6,HJ_REFILL_MIN=16"
echo "== C2"; PROBE_ARGS="" tools/ab_variants.sh $V 2>&1 | tee $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh $V 2>&1 | tee $out/c3.txt
V="cur cur:HJ_INNER_BURST=8,HJ_REFILL_MIN=24 cur:HJ_INNER_BURST=10,HJ_REFILL_MIN=24 cur:HJ_INNER_BURST=8,HJ_REFILL_MIN=20 cur:HJ_INNER_BURST=6,HJ_REFILL_MIN=24"
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 64" tools/ab_variants.sh $V 2>&1 | tee $out/c4.txt
