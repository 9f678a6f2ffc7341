#!/bin/bash
# steps per round 7 on small trees? (repeat, three rounds)
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab70; mkdir -p $out
V="cur cur:HJ_INNER_BURST=7 cur:HJ_INNER_BURST=7,HJ_REFILL_MIN=24 cur:HJ_INNER_BURST=7,HJ_REFILL_MIN=28 cur:HJ_INNER_BURST=8 cur:HJ_INNER_BURST=6"
for r in 1 2; do
echo "== C2"; PROBE_ARGS="" tools/ab_variants.sh $V 2>&1 | tee -a $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh $V 2>&1 | tee -a $out/c3.txt
done
