#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab16; mkdir -p $out
HJ_XCD_DEAL=1 timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest (XCD deal) rc=$?"; tail -3 $out/pytest.log
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 32" tools/ab_variants.sh cur cur:HJ_XCD_DEAL=1 2>&1 | tee $out/c4.txt
echo "== C2"; tools/ab_variants.sh cur cur:HJ_XCD_DEAL=1 2>&1 | tee $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh cur cur:HJ_XCD_DEAL=1 2>&1 | tee $out/c3.txt
for c in c4; do for x in 0 1; do echo -n "$c bench xcd $x: "; HJ_XCD_DEAL=$x timeout 600 python bench.py --config $c --steps 4 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*'; done; done
