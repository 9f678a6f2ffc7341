#!/bin/bash
# child order in the device-built tree: same-box A/B at the probe's batch sizes
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab60; mkdir -p $out
V="cur:HJ_BVH_CHILD_ORDER=0 cur"
echo "== C2 device tree"; PROBE_ARGS="--device-bvh 1" tools/ab_variants.sh $V 2>&1 | tee $out/c2.txt
echo "== C3 device tree"; PROBE_ARGS="--kind 1 --spp 256 --device-bvh 1" tools/ab_variants.sh $V 2>&1 | tee $out/c3.txt
echo "== C4 device tree"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 64 --device-bvh 1" tools/ab_variants.sh $V 2>&1 | tee $out/c4.txt
echo "== C4 host tree"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 64" tools/ab_variants.sh $V 2>&1 | tee $out/c4h.txt
