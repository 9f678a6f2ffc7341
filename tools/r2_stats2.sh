#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_stats2; mkdir -p $out
timeout 300 python tools/walk_stats.py 0 2>&1 | tee $out/c2.txt
HJ_STATS_SPP=1024 timeout 300 python tools/walk_stats.py 1 2>&1 | tee $out/c3.txt
HJ_STATS_SPP=256 HJ_STATS_SIZE=2048 HJ_STATS_TRIS=1000000 timeout 600 python tools/walk_stats.py 2 2>&1 | tee $out/c4.txt
