#!/bin/bash
export GPU_MAX_HW_QUEUES=8
export HIJIKI_DIST_BACKEND=gloo
for n in 2 4; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $n --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -v "^W\|Warning\|warn" | tail -3 | cut -c1-400
done
