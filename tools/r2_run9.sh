#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_run9; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 --deselect tests/test_converged.py::test_hip_matches_converged_float64_image > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log; tail -12 $out/pytest.log
timeout 600 python tools/lbvh_probe.py 2>&1 | tee $out/lbvh.txt
HJ_LBVH_BIG_PCT=0 timeout 600 python tools/lbvh_probe.py 2>&1 | tee $out/lbvh_nobig.txt
timeout 100 hijiki_amd/bin/hijiki-hip synthetic:cbox --use-bvh -w 512 -h 512 -s 64 -o /tmp/x.pfm > $out/cli.txt 2>&1; tr '\r' '\n' < $out/cli.txt | tail -8
