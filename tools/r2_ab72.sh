#!/bin/bash
# last sweep at the final defaults: hot nodes, pool, workgroups per CU, collapse
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab72; mkdir -p $out
V="cur h256 h512 cur:HJ_POOL=4096 cur:HJ_POOL=16384 cur:HJ_WG_PER_CU=7 cur:HJ_COLLAPSE_PCT=40 cur:HJ_COLLAPSE_PCT=60"
echo "== C2"; PROBE_ARGS="" tools/ab_variants.sh $V 2>&1 | tee $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh $V 2>&1 | tee $out/c3.txt
