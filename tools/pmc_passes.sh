#!/bin/bash
# pmc_passes.sh OUTDIR "C1 C2 C3" "C4 C5" ... : one rocprofv3 --pmc pass per counter group over a 64-spp C2 frame (perf_probe, 1 rep),
# each under its own timeout; prints per-kernel sums.  Run from the repo root on the GPU box.
export GPU_MAX_HW_QUEUES=8   # before rocprofv3 / python start: the tool library initialises HIP first, later settings are ignored
out=$1; shift; mkdir -p $out
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $root/$out/p$i -- python3 $root/tools/perf_probe.py --spp 64 --reps 1 --time-kernels 0 > $root/$out/p$i.log 2>&1 || echo "pass $i ($grp) failed/timeout"
  f=$(find $root/$out/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $root/tools/pmc_summary.py $f | grep -E "k_path_wavefront"
done
