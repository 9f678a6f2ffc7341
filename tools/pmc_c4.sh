#!/bin/bash
# HBM traffic of the path kernel on the 1 M-triangle mesh (config 4 scene), 2048^2 x 16 spp: FETCH_SIZE and WRITE_SIZE passes
export GPU_MAX_HW_QUEUES=8   # before rocprofv3 / python start: the tool library initialises HIP first, later settings are ignored
root=$(pwd); out=$root/gpurun_out/c4; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python3 $root/tools/perf_probe.py --kind 2 --tris 1000000 --size 2048 --spp 16 --reps 1 --time-kernels 0 > $out/$c.log 2>&1 || echo "$c failed"
  f=$(find $out/$c -name "*counter_collection.csv" | head -1)
  t=$(find $out/$c -name "*kernel_trace.csv" | head -1)
  python3 - "$f" "$t" "$c" <<'PY'
import csv, sys
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    if "k_path_wavefront" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[3]: tot += float(r["Counter_Value"])
dur = 0
for r in csv.DictReader(open(sys.argv[2])):
    if "k_path_wavefront" in r["Kernel_Name"]: dur += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
paths = 2048 * 2048 * 16
print(f"{sys.argv[3]}: {tot*1024/1e9:.1f} GB over the frame = {tot*1024/paths:.0f} B/path; path kernels {dur/1e6:.1f} ms in total (serialised by the profiler)")
PY
done
rm -rf $out/FETCH_SIZE $out/WRITE_SIZE
