#!/bin/bash
# A/B on one box: base_hip.so (previous build) vs the current library, alternating.
export GPU_MAX_HW_QUEUES=8   # before rocprofv3 / python start: the tool library initialises HIP first, later settings are ignored
for i in 1 2 3; do
  echo -n "base: "; HIJIKI_HIP_LIB=hijiki_amd/lib/base_hip.so timeout 100 python tools/perf_probe.py --spp 512 --reps 3 "$@" 2>&1 | grep -o "[0-9.]* Mpaths/s" | sort -n | tail -1
  echo -n "new:  "; timeout 100 python tools/perf_probe.py --spp 512 --reps 3 "$@" 2>&1 | grep -o "[0-9.]* Mpaths/s" | sort -n | tail -1
done
