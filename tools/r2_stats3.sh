#!/bin/bash
# walk statistics of the merged-step kernel with pair nodes everywhere (diagnostic build), c2 / c3 / c4
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_stats3; mkdir -p $out
timeout 300 python tools/walk_stats.py 0 --json $out/c2.json > $out/c2.txt 2>&1; cat $out/c2.txt
HJ_STATS_SPP=256 timeout 300 python tools/walk_stats.py 1 --json $out/c3.json > $out/c3.txt 2>&1; head -8 $out/c3.txt
HJ_STATS_SPP=64 HJ_STATS_SIZE=2048 HJ_STATS_TRIS=1000000 timeout 400 python tools/walk_stats.py 2 --json $out/c4.json > $out/c4.txt 2>&1; head -8 $out/c4.txt
