#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab28; mkdir -p $out
echo "== C2"; tools/ab_variants.sh cur cur:HJ_WG_PER_CU=7 cur:HJ_WG_PER_CU=14 cur:HJ_REFILL_MIN=28 cur:HJ_REFILL_MIN=36 cur:HJ_SLOTS=4 cur:HJ_POOL=12288 2>&1 | tee $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh cur cur:HJ_WG_PER_CU=7 cur:HJ_WG_PER_CU=14 cur:HJ_SLOTS=4 2>&1 | tee $out/c3.txt
