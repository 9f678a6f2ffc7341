#!/bin/bash
export GPU_MAX_HW_QUEUES=8
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/perf_probe.py --spp 512 --reps 3 2>&1 | grep -o "[0-9.]* Mpaths/s" | sort -n | tail -1
HJ_STATS_SPP=64 timeout 200 python tools/walk_stats.py 0 2>&1 | head -4
