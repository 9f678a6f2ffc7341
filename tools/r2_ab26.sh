#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab26; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== C2"; tools/ab_variants.sh nopf cur 2>&1 | tee $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh nopf cur 2>&1 | tee $out/c3.txt
echo "== mesh60k"; PROBE_ARGS="--kind 2 --tris 60000 --size 2048 --spp 32" tools/ab_variants.sh nopf cur 2>&1 | tee $out/m60k.txt
