#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab34; mkdir -p $out
V="cur nts ntst"
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 128" tools/ab_variants.sh $V 2>&1 | tee $out/c4.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 512" tools/ab_variants.sh $V 2>&1 | tee $out/c3.txt
echo "== C2"; tools/ab_variants.sh $V 2>&1 | tee $out/c2.txt
