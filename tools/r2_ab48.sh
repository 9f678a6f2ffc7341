#!/bin/bash
# pair nodes on small trees (now without the streaming accesses), burst and node order; mid-size trees
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab48; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or config1 or divergent or random or pair_nodes or config4_million" > $out/pytest.log 2>&1; rc=$?; tail -2 $out/pytest.log
[ $rc -ne 0 ] && exit 1
P="HJ_PAIR_LEAVES=1"
V="cur cur:$P,HJ_INNER_BURST=5 cur:$P,HJ_INNER_BURST=6 cur:$P,HJ_INNER_BURST=7 cur:$P,HJ_INNER_BURST=6,HJ_NODE_ORDER=0 cur:$P,HJ_INNER_BURST=6,HJ_STREAM_STATE=1"
echo "== C2"; PROBE_ARGS="" tools/ab_variants.sh $V 2>&1 | tee $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh $V 2>&1 | tee $out/c3.txt
V="cur:HJ_PAIR_LEAVES=0,HJ_STREAM_STATE=0 cur:$P,HJ_INNER_BURST=6,HJ_STREAM_STATE=0 cur:$P,HJ_INNER_BURST=8,HJ_STREAM_STATE=0 cur:$P,HJ_INNER_BURST=6,HJ_STREAM_STATE=1 cur:$P,HJ_INNER_BURST=8,HJ_STREAM_STATE=1"
echo "== 60k"; PROBE_ARGS="--kind 2 --tris 60000 --size 1024 --spp 256" tools/ab_variants.sh $V 2>&1 | tee $out/c60k.txt
echo "== 200k"; PROBE_ARGS="--kind 2 --tris 200000 --size 2048 --spp 64" tools/ab_variants.sh $V 2>&1 | tee $out/c200k.txt
V="cur cur:HJ_INNER_BURST=6 cur:HJ_INNER_BURST=7"
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 64" tools/ab_variants.sh $V 2>&1 | tee $out/c4.txt
