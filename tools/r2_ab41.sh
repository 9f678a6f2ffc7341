#!/bin/bash
# merged leaf/box first step against the committed build (HEAD), same box
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab41; mkdir -p $out
echo "== C2"; PROBE_ARGS="" tools/ab_variants.sh base cur 2>&1 | tee $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh base cur 2>&1 | tee $out/c3.txt
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 64" tools/ab_variants.sh base m2 2>&1 | tee $out/c4.txt
