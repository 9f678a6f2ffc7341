#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab10; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log; tail -5 $out/pytest.log
echo "== C2"; tools/ab_variants.sh nospec cur cur:HJ_LEAF_MIN=16 cur:HJ_LEAF_MIN=24 cur:HJ_LEAF_MIN=48 cur:HJ_INNER_BURST=8 2>&1 | tee $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh nospec cur cur:HJ_LEAF_MIN=16 cur:HJ_LEAF_MIN=48 2>&1 | tee $out/c3.txt
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 32" tools/ab_variants.sh nospec cur cur:HJ_LEAF_MIN=16 cur:HJ_LEAF_MIN=48 cur:HJ_INNER_BURST=8 2>&1 | tee $out/c4.txt
HJ_STATS_SPP=512 timeout 200 python tools/walk_stats.py 0 2>&1 | head -6 | tee $out/walk_c2.txt
HJ_STATS_SPP=32 HJ_STATS_SIZE=2048 HJ_STATS_TRIS=1000000 timeout 300 python tools/walk_stats.py 2 2>&1 | head -6 | tee $out/walk_c4.txt
