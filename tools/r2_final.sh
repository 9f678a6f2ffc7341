#!/bin/bash
# end-of-round check: whole GPU suite, smoke, the three bench lines (un-profiled)
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_final; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log; tail -12 $out/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
for c in c2 c3 c4; do timeout 600 python bench.py --config $c --steps 5 2>$out/bench_$c.err | grep '^{' > $out/bench_$c.json; cut -c1-150 $out/bench_$c.json; done
