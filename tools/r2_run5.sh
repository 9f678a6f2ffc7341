#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_run5; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 --deselect tests/test_converged.py::test_hip_matches_converged_float64_image > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log; tail -25 $out/pytest.log
timeout 300 python bench.py --inproc --gpus 1 --steps 3 2>&1 | tail -2 | cut -c1-300
timeout 100 hijiki_amd/bin/hijiki-hip synthetic:cbox --use-bvh -w 512 -h 512 -s 64 -o /tmp/x.pfm 2>&1 | tr '\r' '\n' | tail -6
echo "== sweeps C2"; tools/ab_variants.sh cur cur:HJ_INNER_BURST=3 cur:HJ_INNER_BURST=6 cur:HJ_INNER_BURST=8 cur:HJ_REFILL_MIN=24 cur:HJ_REFILL_MIN=40 cur:HJ_REFILL_MIN=48 cur:HJ_SLOTS=2 cur:HJ_WG_PER_CU=6 2>&1 | tee $out/c2_sweeps.txt
echo "== sweeps C3"; PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh cur cur:HJ_INNER_BURST=6 cur:HJ_REFILL_MIN=24 cur:HJ_REFILL_MIN=48 cur:HJ_SLOTS=2 cur:HJ_POOL=16384 2>&1 | tee $out/c3_sweeps.txt
