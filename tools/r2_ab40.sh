#!/bin/bash
# merged leaf/box first step of a walk round (HJ_MERGE_LEAF): parity subset first, then same-box A/B
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab40; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or config1 or divergent or tinted or quads or random or linear or ragged or traversal or pair_nodes or edge_inputs" > $out/pytest.log 2>&1; rc=$?; tail -5 $out/pytest.log
[ $rc -ne 0 ] && exit 1
V="m0 cur cur:HJ_INNER_BURST=5 cur:HJ_INNER_BURST=3"
echo "== C2"; PROBE_ARGS="" tools/ab_variants.sh $V 2>&1 | tee $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh m0 cur cur:HJ_INNER_BURST=5 2>&1 | tee $out/c3.txt
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 64" tools/ab_variants.sh m0 m2 m2:HJ_INNER_BURST=9 2>&1 | tee $out/c4.txt
