#!/bin/bash
# workgroup size of the path kernel again (called stages, merged step): 128 / 512 threads, 512 with 768 hot nodes
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab68; mkdir -p $out
HIJIKI_HIP_LIB=hijiki_amd/lib/var_bt512h.so HJ_WG_PER_CU=4 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or config1 or divergent or random" > $out/pytest.log 2>&1; rc=$?; tail -2 $out/pytest.log
[ $rc -ne 0 ] && exit 1
V="cur bt512:HJ_WG_PER_CU=4 bt512h:HJ_WG_PER_CU=4 bt512:HJ_WG_PER_CU=4,HJ_POOL=16384 bt128:HJ_WG_PER_CU=16 bt128:HJ_WG_PER_CU=16,HJ_POOL=4096"
echo "== C2"; PROBE_ARGS="" tools/ab_variants.sh $V 2>&1 | tee $out/c2.txt
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 64" tools/ab_variants.sh $V 2>&1 | tee $out/c4.txt
