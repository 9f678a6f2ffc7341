#!/bin/bash
# profile_round.sh TAG : the three rocprofv3 runs behind profiles/<TAG>_* (kernel stats, FETCH_SIZE, WRITE_SIZE), each under a timeout.
# Run from the repo root on the GPU box; results land in gpurun_out/<TAG>/ (copy what you want judged into profiles/).
export GPU_MAX_HW_QUEUES=8   # before rocprofv3 / python start: the tool library initialises HIP first, later settings are ignored
tag=$1
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/stats.log 2>&1 || echo "stats pass failed"
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv 2>/dev/null
# the timed region of that run = the last 2 frames = the last 32 k_path_wavefront launches: their average is what bench.py's
# HIP events measure (the --stats average above also contains the warm-up frame)
python3 - "$(find $out/stats -name "*kernel_trace.csv" | head -1)" > $out/timed_region_kernel_avg.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_path_wavefront" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
n = 32 if len(d) >= 32 else len(d)
print(f"k_path_wavefront launches {len(d)}; all: avg {sum(d)/len(d):.4f} ms; last {n} (timed region): avg {sum(d[-n:])/n:.4f} ms min {min(d[-n:]):.3f} max {max(d[-n:]):.3f}")
PY
cat $out/timed_region_kernel_avg.txt
grep '^{' $out/stats.log > $out/bench_under_rocprof.json
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/$c.log 2>&1 || echo "$c pass failed"
  f=$(find $out/$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$c" > $out/pmc_$c.csv <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if r["Counter_Name"] != sys.argv[2]: continue
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
print("kernel,launches,%s_sum" % sys.argv[2])
for k, (n, v) in sorted(agg.items()): print(f"{k},{n},{v:.0f}")
PY
done
cd $root
timeout 300 python3 bench.py --steps 5 > $out/bench.json 2> $out/bench.err || echo "bench failed"
rm -rf $out/stats $out/FETCH_SIZE $out/WRITE_SIZE
cat $out/kernel_stats.csv | head -8; cat $out/pmc_FETCH_SIZE.csv $out/pmc_WRITE_SIZE.csv; tail -1 $out/bench.json | cut -c1-200
