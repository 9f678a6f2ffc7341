#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab17; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
echo "== C2"; tools/ab_variants.sh base cur 2>&1 | tee $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh base cur 2>&1 | tee $out/c3.txt
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 32" tools/ab_variants.sh base cur 2>&1 | tee $out/c4.txt
HJ_STATS_SPP=512 timeout 200 python tools/walk_stats.py 0 2>&1 | head -5
