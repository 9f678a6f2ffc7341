#!/bin/bash
# kernel timeline of two c2 frames (how the batch slots overlap)
export GPU_MAX_HW_QUEUES=8
root=$(pwd); out=$root/gpurun_out/r2_timeline; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $out/tr -- python3 $root/bench.py --config c2 --steps 1 --warmup 1 --no-cpu-baseline > $out/log.txt 2>&1
python3 $root/tools/launch_timeline.py "$(find $out/tr -name '*kernel_trace.csv' | head -1)" 60 | tee $out/timeline.txt
rm -rf $out/tr
