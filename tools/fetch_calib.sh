#!/bin/bash
# FETCH_SIZE calibration for 16 / 32 / 48-byte gathers (tools/micro/fetch_calib.hip) -> gpurun_out/fetch_calib.txt
out=$(pwd)/gpurun_out; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/fetch_calib.hip -o $out/fetch_calib || exit 1
cd /tmp && export TMPDIR=/tmp
: > $out/fetch_calib.txt
for f4 in 1 2 3; do
  rm -rf $out/fc; timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fc -- $out/fetch_calib $f4 > $out/fc.log 2>&1
  req=$(grep requested_bytes $out/fc.log | awk '{print $4}')
  python3 - "$(find $out/fc -name '*counter_collection.csv' | head -1)" $req $f4 >> $out/fetch_calib.txt <<'PY'
import csv, sys
v = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == "FETCH_SIZE" and "gather" in r["Kernel_Name"])
req = float(sys.argv[2])
print(f"record {int(sys.argv[3])*16} B: requested {req/1e9:.3f} GB, FETCH_SIZE {v*1024/1e9:.3f} GB (KiB x 1024) = {v*1024/req:.3f} x requested")
PY
done
rm -rf $out/fc; cat $out/fetch_calib.txt
