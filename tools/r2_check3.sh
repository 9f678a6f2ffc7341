#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_check3; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
timeout 600 python tools/shard_probe.py 2>&1 | tee $out/shard.txt
for c in c2 c3 c4; do timeout 600 python bench.py --config $c --steps 5 --no-cpu-baseline 2>$out/bench_$c.err | grep '^{' > $out/bench_$c.json; cut -c1-150 $out/bench_$c.json; done
