#!/bin/bash
# does the VALU work of the leaf tests matter?  64 / 128 extra VALU instructions in the leaf branch of the merged step (a pair test has ~130)
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab55; mkdir -p $out
V="cur ps8"
echo "== C2"; PROBE_ARGS="" tools/ab_variants.sh $V 2>&1 | tee $out/c2.txt
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 64" tools/ab_variants.sh $V 2>&1 | tee $out/c4.txt
