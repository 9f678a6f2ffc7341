#!/bin/bash
# batch size / batch slots on the final kernel
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab53; mkdir -p $out
V="cur cur:HJ_BATCH_CAP=4096 cur:HJ_BATCH_CAP=5462 cur:HJ_BATCH_CAP=6554 cur:HJ_SLOTS=4 cur:HJ_SLOTS=4,HJ_BATCH_CAP=4096 cur:HJ_SLOTS=2"
echo "== C2"; PROBE_ARGS="" tools/ab_variants.sh $V 2>&1 | tee $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 1024" tools/ab_variants.sh $V 2>&1 | tee $out/c3.txt
