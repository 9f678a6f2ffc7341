#!/bin/bash
# ab_configs.sh LIB_A LIB_B [rounds] : same-box, interleaved A/B of two builds of libhijiki_hip.so on c2 / c3 / c4 at their own
# sizes (blocking frames, tools/perf_probe.py); prints the best frame of every run.  LIB = path of a .so, or "head" for the tree's library.
export GPU_MAX_HW_QUEUES=8
A=$1; B=$2; R=${3:-3}
run() {  # lib label args...
  local lib=$1 label=$2; shift 2
  if [ "$lib" = head ]; then unset HIJIKI_HIP_LIB; else export HIJIKI_HIP_LIB=$lib; fi
  echo -n "$label: "; timeout -k 10 200 python tools/perf_probe.py --reps 3 "$@" 2>&1 | grep -o "[0-9.]* Mpaths/s" | sort -n | tail -1
}
for i in $(seq $R); do
  for cfg in "c2 --spp 512" "c3 --spp 1024 --kind 1" "c4 --spp 256 --size 2048 --kind 2 --tris 1000000"; do
    set -- $cfg; name=$1; shift
    run $A "$name A" "$@"
    run $B "$name B" "$@"
  done
done
