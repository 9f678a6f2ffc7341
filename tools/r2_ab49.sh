#!/bin/bash
# workgroups per CU, hot-node count, pool on the merged-step kernel with pair nodes everywhere
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab49; mkdir -p $out
V="cur cur:HJ_WG_PER_CU=7 cur:HJ_WG_PER_CU=14 cur:HJ_POOL=4096 cur:HJ_POOL=16384 h512 h256 cur:HJ_REFILL_MIN=24 cur:HJ_REFILL_MIN=40"
echo "== C2"; PROBE_ARGS="" tools/ab_variants.sh $V 2>&1 | tee $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh $V 2>&1 | tee $out/c3.txt
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 64" tools/ab_variants.sh $V 2>&1 | tee $out/c4.txt
