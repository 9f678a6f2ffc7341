#!/bin/bash
# rotations with the cross exchanges of grandchildren: parity subset, then against no rotations
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab65; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or config1 or divergent or tinted or quads or random or linear or ragged or traversal or pair_nodes or edge_inputs or config4_million or inconsistent" > $out/pytest.log 2>&1; rc=$?; tail -2 $out/pytest.log
[ $rc -ne 0 ] && exit 1
V="cur:HJ_BVH_ROTATE=0 cur"
echo "== C2"; PROBE_ARGS="" tools/ab_variants.sh $V 2>&1 | tee $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh $V 2>&1 | tee $out/c3.txt
echo "== 60k"; PROBE_ARGS="--kind 2 --tris 60000 --size 1024 --spp 256" tools/ab_variants.sh $V 2>&1 | tee $out/c60k.txt
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 64" tools/ab_variants.sh $V 2>&1 | tee $out/c4.txt
