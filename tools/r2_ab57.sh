#!/bin/bash
# a node fetched together with the sibling record behind it (HJ_DUAL_FETCH): when its box fails the sibling's test follows in the same trip
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab57; mkdir -p $out
HIJIKI_HIP_LIB=hijiki_amd/lib/var_df.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or config1 or divergent or tinted or quads or random or linear or ragged or traversal or pair_nodes or edge_inputs or split_kernel or config4_million or inconsistent or device_built" > $out/pytest.log 2>&1; rc=$?; tail -2 $out/pytest.log
[ $rc -ne 0 ] && exit 1
V="cur df df:HJ_INNER_BURST=4 df:HJ_INNER_BURST=6"
echo "== C2"; PROBE_ARGS="" tools/ab_variants.sh $V 2>&1 | tee $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh $V 2>&1 | tee $out/c3.txt
V="cur df df:HJ_INNER_BURST=6 df:HJ_INNER_BURST=10"
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 64" tools/ab_variants.sh $V 2>&1 | tee $out/c4.txt
