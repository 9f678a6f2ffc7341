#!/bin/bash
# pool of 16384 / 32768 positions per workgroup at the final defaults; rank 0 of 8
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab73; mkdir -p $out
V="cur cur:HJ_POOL=16384 cur:HJ_POOL=32768 cur:HJ_POOL=12288"
echo "== C2"; PROBE_ARGS="" tools/ab_variants.sh $V 2>&1 | tee $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh $V 2>&1 | tee $out/c3.txt
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 64" tools/ab_variants.sh $V 2>&1 | tee $out/c4.txt
for e in "HJ_POOL=8192" "HJ_POOL=16384"; do echo -n "world 8 $e: "; env $e timeout 200 python tools/pipeline_probe.py --world 8 --frames 16 2>&1 | grep serial | sort -k5 -n | tail -1; done | tee $out/w8.txt
