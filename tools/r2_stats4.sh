#!/bin/bash
# walk statistics with and without tree rotations
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_stats4; mkdir -p $out
for rot in 0 8; do
  echo "== rotate $rot: cbox"; HJ_BVH_ROTATE=$rot HJ_STATS_SPP=128 timeout 300 python tools/walk_stats.py 0 2>&1 | head -4
  echo "== rotate $rot: 60k mesh"; HJ_BVH_ROTATE=$rot HJ_STATS_SPP=64 HJ_STATS_TRIS=60000 timeout 300 python tools/walk_stats.py 2 2>&1 | head -4
done | tee $out/stats.txt
