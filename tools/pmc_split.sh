#!/bin/bash
# pmc_split.sh OUT "C1 C2" ... : like pmc_passes.sh but over the SPLIT stage kernels (one launch per stage per bounce), so that
# counters can be attributed to k_shade / k_trace_closest / k_trace_shadow separately.  64-spp cbox frame.
export GPU_MAX_HW_QUEUES=8   # before rocprofv3 / python start: the tool library initialises HIP first, later settings are ignored
out=$1; shift
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $root/$out/p$i -- python3 $root/tools/perf_probe.py --spp 64 --reps 1 --time-kernels 0 --split 1 > $root/$out/p$i.log 2>&1 || echo "pass $i ($grp) failed/timeout"
  f=$(find $root/$out/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $root/tools/pmc_summary.py $f | grep -E "k_shade|k_trace|k_gen"
  t=$(find $root/$out/p$i -name "*kernel_trace.csv" | head -1)
  [ $i = 1 ] && [ -n "$t" ] && python3 - "$t" <<'PY'
import csv, sys, collections
d = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    d[k][0] += 1; d[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (n, t) in sorted(d.items()):
    if "hj::" in k: print(f"{k}: {n} launches, {t/1e6:.2f} ms total")
PY
done
