#!/bin/bash
# kernel timeline of rank 0's share of the c2 frame at 8 ranks (what the fixed part of a small frame consists of)
export GPU_MAX_HW_QUEUES=8
root=$(pwd); out=$root/gpurun_out/r2_timeline8; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $out/tr -- python3 $root/tools/pipeline_probe.py --world 8 --frames 3 > $out/log.txt 2>&1
python3 $root/tools/launch_timeline.py "$(find $out/tr -name '*kernel_trace.csv' | head -1)" 400 > $out/timeline.txt
rm -rf $out/tr
grep serial $out/log.txt | head -2
# the frames of the first "serial" loop: rows after the two warm-up frames (8 path launches)
awk 'NR>16 && NR<=46' $out/timeline.txt
