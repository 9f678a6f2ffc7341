#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_check2; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
timeout 600 python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_c4.json 2> $out/bench_c4.err; cut -c1-300 $out/bench_c4.json
