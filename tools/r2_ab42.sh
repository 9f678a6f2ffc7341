#!/bin/bash
# merged first step + shadow contribution carried in registers (cur), + first-step loads inside the service phase (ms), against HEAD (base)
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab42; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or config1 or divergent or tinted or quads or random or linear or ragged or traversal or pair_nodes or edge_inputs or split_kernel" > $out/pytest.log 2>&1; rc=$?; tail -5 $out/pytest.log
[ $rc -ne 0 ] && exit 1
HIJIKI_HIP_LIB=hijiki_amd/lib/var_ms.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or config1 or divergent or tinted or quads or random or pair_nodes" > $out/pytest_ms.log 2>&1; rc=$?; tail -3 $out/pytest_ms.log
[ $rc -ne 0 ] && exit 1
echo "== C2"; PROBE_ARGS="" tools/ab_variants.sh base cur ms 2>&1 | tee $out/c2.txt
echo "== C3"; PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh base cur ms 2>&1 | tee $out/c3.txt
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 64" tools/ab_variants.sh base cur ms 2>&1 | tee $out/c4.txt
