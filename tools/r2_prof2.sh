#!/bin/bash
export GPU_MAX_HW_QUEUES=8
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2_prof2_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2_prof2_pytest.log
for c in c2 c3 c4; do bash tools/profile_config.sh r02 $c > gpurun_out/prof_$c.log 2>&1; tail -1 gpurun_out/prof_$c.log | cut -c1-200; done
timeout 600 python tools/lbvh_probe.py 2>&1 | tee gpurun_out/r02_lbvh_probe.txt
timeout 600 python tools/shard_probe.py 2>&1 | tee gpurun_out/r02_shard_probe.txt
