#!/bin/bash
# burst steps on LDS-resident nodes only (HJ_HOT_BURST): cold nodes wait for the merged first step of the next round
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab52; mkdir -p $out
HIJIKI_HIP_LIB=hijiki_amd/lib/var_hb.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or config1 or divergent or tinted or quads or random or linear or ragged or traversal or pair_nodes or edge_inputs or split_kernel" > $out/pytest.log 2>&1; rc=$?; tail -2 $out/pytest.log
[ $rc -ne 0 ] && exit 1
V="cur hb hb:HJ_INNER_BURST=8 hb:HJ_INNER_BURST=12 hb:HJ_INNER_BURST=16 hb:HJ_INNER_BURST=32"
echo "== C2"; PROBE_ARGS="" tools/ab_variants.sh $V 2>&1 | tee $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh $V 2>&1 | tee $out/c3.txt
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 64" tools/ab_variants.sh $V 2>&1 | tee $out/c4.txt
