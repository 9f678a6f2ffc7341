#!/bin/bash
# two frames in flight (two contexts, hj_render_frame_async) against frames one after the other
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_pipe; mkdir -p $out
( echo "== c2"; timeout 200 python tools/pipeline_probe.py
  echo "== c2 rank 0 of 8"; timeout 200 python tools/pipeline_probe.py --world 8 --frames 16
  echo "== c2 rank 0 of 4"; timeout 200 python tools/pipeline_probe.py --world 4 --frames 12
  echo "== c3"; timeout 300 python tools/pipeline_probe.py --kind 1 --spp 1024 --frames 4 ) 2>&1 | tee $out/pipe.txt
