#!/bin/bash
# ab_trees.sh [rounds]: same-box, interleaved comparison of host-tree variants (environment of the scene compiler) on c2 / c3 / c4
# at their own sizes (blocking frames, tools/perf_probe.py): best frame of every run.
export GPU_MAX_HW_QUEUES=8
R=${1:-2}
VARIANTS=("base:HJ_BVH_CHILD_ORDER=3" "vote:HJ_BVH_CHILD_ORDER=4" "vote+reins3:HJ_BVH_CHILD_ORDER=4 HJ_BVH_REINSERT=3")
for i in $(seq $R); do
  for cfg in "c2 --spp 512" "c3 --spp 1024 --kind 1" "c4 --spp 256 --size 2048 --kind 2 --tris 1000000"; do
    set -- $cfg; name=$1; shift
    for v in "${VARIANTS[@]}"; do
      label=${v%%:*}; envs=${v#*:}
      t0=$(date +%s.%N)
      out=$(env $envs timeout -k 10 400 python tools/perf_probe.py --reps 3 "$@" 2>&1 | grep -o "[0-9.]* Mpaths/s" | sort -n | tail -1)
      t1=$(date +%s.%N)
      echo "$name $label: $out  (process $(echo "$t1 - $t0" | bc) s)"
    done
  done
done
