#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab24; mkdir -p $out
echo "== C2"; tools/ab_variants.sh cur cur:HJ_PAIR_LEAVES=1,HJ_INNER_BURST=4 cur:HJ_PAIR_LEAVES=1,HJ_INNER_BURST=6 cur:HJ_PAIR_LEAVES=1 cur:HJ_PAIR_LEAVES=1,HJ_INNER_BURST=12 2>&1 | tee $out/c2.txt
echo "== mesh 200k"; PROBE_ARGS="--kind 2 --tris 200000 --size 2048 --spp 32" tools/ab_variants.sh cur cur:HJ_PAIR_LEAVES=1,HJ_INNER_BURST=6 cur:HJ_PAIR_LEAVES=1 cur:HJ_PAIR_LEAVES=1,HJ_INNER_BURST=12 2>&1 | tee $out/m200k.txt
echo "== C4 bench"; for b in 8 16; do echo -n "burst $b: "; HJ_INNER_BURST=$b timeout 600 python bench.py --config c4 --steps 4 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*'; done
