#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_stats2; mkdir -p $out
HJ_STATS_SPP=512 timeout 300 python3 tools/walk_stats.py 0 --json $out/c2.json > $out/c2.txt 2>&1
HJ_STATS_SPP=1024 timeout 300 python3 tools/walk_stats.py 1 --json $out/c3.json > $out/c3.txt 2>&1
HJ_STATS_SPP=256 HJ_STATS_SIZE=2048 HJ_STATS_TRIS=1000000 timeout 400 python3 tools/walk_stats.py 2 --json $out/c4.json > $out/c4.txt 2>&1
tail -8 $out/c3.txt
