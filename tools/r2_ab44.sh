#!/bin/bash
# merged-first-step variants at 6 and 5 waves per SIMD (80 / 96 VGPRs: no spills inside the walk)
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab44; mkdir -p $out
V="base a6 b6 f6 b5 f5 g5"
echo "== C2"; PROBE_ARGS="" tools/ab_variants.sh $V 2>&1 | tee $out/c2.txt
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 64" tools/ab_variants.sh $V 2>&1 | tee $out/c4.txt
