#!/bin/bash
# round-2 evidence at the final commit: GPU suite, the three per-config profiles, c5 on one GPU, device-BVH and sharding probes
export GPU_MAX_HW_QUEUES=8
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2_prof3_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2_prof3_pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
for c in c2 c3 c4; do bash tools/profile_config.sh r02 $c > gpurun_out/prof_$c.log 2>&1; tail -1 gpurun_out/prof_$c.log | cut -c1-200; done
timeout 600 python bench.py --config c5 --steps 1 --warmup 0 2> gpurun_out/r02_c5_bench.err | grep '^{' > gpurun_out/r02_c5_bench.json; cut -c1-250 gpurun_out/r02_c5_bench.json
( echo "== default"; HJ_LBVH_TIMING=1 timeout 600 python tools/lbvh_probe.py 2>&1; echo "== HJ_LBVH_SAH=0 (Morton splits inside the clusters)"; HJ_LBVH_SAH=0 timeout 600 python tools/lbvh_probe.py 2>&1 ) | tee gpurun_out/r02_lbvh_probe.txt
timeout 600 python tools/shard_probe.py 2>&1 | tee gpurun_out/r02_shard_probe.txt
