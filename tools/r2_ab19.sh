#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab19; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
echo "== C2 (no pairs by default)"; tools/ab_variants.sh base cur 2>&1 | tee $out/c2.txt
for tr in 20000 60000 200000 1000000; do echo "== mesh $tr"; PROBE_ARGS="--kind 2 --tris $tr --size 2048 --spp 32" tools/ab_variants.sh cur:HJ_PAIR_LEAVES=0 cur:HJ_PAIR_LEAVES=1 2>&1 | tee $out/mesh_$tr.txt; done
