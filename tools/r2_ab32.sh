#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab32; mkdir -p $out
V="cur cur:HJ_WG_PER_CU=7 cur:HJ_WG_PER_CU=14 cur:HJ_WG_PER_CU=21"
echo "== C2"; tools/ab_variants.sh $V 2>&1 | tee $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh $V 2>&1 | tee $out/c3.txt
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 32" tools/ab_variants.sh $V 2>&1 | tee $out/c4.txt
