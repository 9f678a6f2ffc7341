#!/bin/bash
# rank 0's share of the c2 frame at 8 and 4 ranks: pool size (path regeneration generations per workgroup) and batch size
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_pool8; mkdir -p $out
for w in 8 4; do
for e in "HJ_POOL=8192" "HJ_POOL=4096" "HJ_POOL=2048" "HJ_POOL=1024" "HJ_POOL=2048 HJ_BATCH_CAP=2048" "HJ_POOL=4096 HJ_BATCH_CAP=2048" "HJ_POOL=2048 HJ_WG_PER_CU=7"; do
  echo -n "world $w $e: "; env $e timeout 200 python tools/pipeline_probe.py --world $w --frames 16 2>&1 | grep serial | sort -k5 -n | tail -1
done; done 2>&1 | tee $out/pool.txt
