#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab37; mkdir -p $out
HJ_NODE_ORDER=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
V="cur cur:HJ_NODE_ORDER=1"
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 128" tools/ab_variants.sh $V 2>&1 | tee $out/c4.txt
echo "== C2"; tools/ab_variants.sh $V 2>&1 | tee $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 512" tools/ab_variants.sh $V 2>&1 | tee $out/c3.txt
