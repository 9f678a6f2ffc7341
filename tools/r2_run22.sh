#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_run22; mkdir -p $out
HJ_STATS_SPP=256 HJ_STATS_SIZE=2048 HJ_STATS_TRIS=1000000 timeout 400 python3 tools/walk_stats.py 2 --json $out/c4.json > $out/c4.txt 2>&1; head -4 $out/c4.txt
