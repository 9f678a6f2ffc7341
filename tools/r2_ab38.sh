#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab38; mkdir -p $out
V="cur cur:HJ_NODE_ORDER=1 cur:HJ_NODE_ORDER=1,HJ_NODE_ORDER_SA=1"
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 128" tools/ab_variants.sh $V 2>&1 | tee $out/c4.txt
echo "== 200k"; PROBE_ARGS="--kind 2 --tris 200000 --size 2048 --spp 64" tools/ab_variants.sh $V 2>&1 | tee $out/c200k.txt
