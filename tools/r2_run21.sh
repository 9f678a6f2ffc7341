#!/bin/bash
export GPU_MAX_HW_QUEUES=8
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash tools/profile_config.sh r02 c4 > gpurun_out/prof_c4.log 2>&1; tail -1 gpurun_out/prof_c4.log | cut -c1-200
for c in c2 c3; do timeout 600 python bench.py --config $c --steps 5 2>/dev/null | grep '^{' | tee gpurun_out/r02_final_bench_$c.json | cut -c1-160; done
