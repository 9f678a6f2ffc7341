#!/bin/bash
# round-2 A/B #1: split hot/cold walk vs the FLAT walk, burst knobs, hot-set size; parity first
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab1; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log; tail -3 $out/pytest.log
echo "== C2"; tools/ab_variants.sh flat cur cur:HJ_COLD_BURST=1 cur:HJ_COLD_BURST=4 cur:HJ_INNER_BURST=8 cur:HJ_INNER_BURST=2 hot512 2>&1 | tee $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh flat cur cur:HJ_COLD_BURST=4 hot512 2>&1 | tee $out/c3.txt
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 32" tools/ab_variants.sh flat cur cur:HJ_COLD_BURST=4 cur:HJ_COLD_BURST=8 2>&1 | tee $out/c4.txt
echo "== stats"; timeout 120 python tools/walk_stats.py 0 2>&1 | tee $out/walk_stats_c2.txt
