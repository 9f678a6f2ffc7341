#!/bin/bash
# variants of the merged first step: a merge only, b + shadow carry, c + carry + select fetch, d + select fetch, e + service merge, f + service merge + carry
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab43; mkdir -p $out
V="base a b c d e f"
echo "== C2"; PROBE_ARGS="" tools/ab_variants.sh $V 2>&1 | tee $out/c2.txt
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 64" tools/ab_variants.sh $V 2>&1 | tee $out/c4.txt
