#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_batch; mkdir -p $out
for c in c2 c3 c4; do for cap in 2048 4096 8192; do echo -n "$c cap $cap: "; HJ_BATCH_CAP=$cap timeout 600 python bench.py --config $c --steps 4 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*'; done; done 2>&1 | tee $out/batch.txt
for c in c3 c4; do for p in 4096 16384; do echo -n "$c cap 4096 pool $p: "; HJ_POOL=$p timeout 600 python bench.py --config $c --steps 4 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*'; done; done 2>&1 | tee -a $out/batch.txt
