#!/bin/bash
# round-2 A/B #3: compacted path arrays + regeneration (pool sweep) vs the round-1 kernel; parity first
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab3; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log; tail -15 $out/pytest.log
echo "== C2"; tools/ab_variants.sh base cur:HJ_POOL=1024 cur:HJ_POOL=2048 cur cur:HJ_POOL=8192 cur:HJ_POOL=16384 w5 2>&1 | tee $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh base cur:HJ_POOL=1024 cur:HJ_POOL=2048 cur cur:HJ_POOL=8192 cur:HJ_POOL=16384 w5 2>&1 | tee $out/c3.txt
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 32" tools/ab_variants.sh base cur:HJ_POOL=2048 cur cur:HJ_POOL=16384 2>&1 | tee $out/c4.txt
