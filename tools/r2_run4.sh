#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_run4; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log; tail -25 $out/pytest.log
timeout 300 python bench.py --steps 5 2>$out/bench.err | tee $out/bench_c2.json
bash tools/profile_config.sh r02a c2 2>&1 | tail -40
