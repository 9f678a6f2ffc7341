#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_stats; mkdir -p $out
HJ_STATS_SPP=512 timeout 200 python tools/walk_stats.py 0 --json $out/walk_c2.json 2>&1 | tee $out/walk_c2_512spp.txt
HJ_STATS_SPP=256 timeout 200 python tools/walk_stats.py 1 --json $out/walk_c3.json 2>&1 | tee $out/walk_c3_256spp.txt
HJ_STATS_SPP=32 HJ_STATS_SIZE=2048 HJ_STATS_TRIS=1000000 timeout 300 python tools/walk_stats.py 2 --json $out/walk_c4.json 2>&1 | tee $out/walk_c4_32spp.txt
echo "== batch 4096"; for p in 8192 16384 32768; do echo -n "pool $p batch 4096: "; HJ_POOL=$p timeout 100 python tools/perf_probe.py --spp 512 --reps 3 --batch 4096 2>&1 | grep -o "[0-9.]* Mpaths/s" | sort -n | tail -1; done
echo -n "pool 8192 batch 2048: "; timeout 100 python tools/perf_probe.py --spp 512 --reps 3 2>&1 | grep -o "[0-9.]* Mpaths/s" | sort -n | tail -1
for s in 2 3 4; do echo -n "slots $s: "; HJ_SLOTS=$s timeout 100 python tools/perf_probe.py --spp 512 --reps 3 2>&1 | grep -o "[0-9.]* Mpaths/s" | sort -n | tail -1; done
