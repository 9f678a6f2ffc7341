#!/bin/bash
# ab_env.sh "ENV_A" "ENV_B" [rounds]: same-box, interleaved A/B of two environments (switches of the libraries) on c2 / c3 / c4 at
# their own sizes (blocking frames, tools/perf_probe.py): best frame of every run + the frame's shadow-ray statistics.
export GPU_MAX_HW_QUEUES=8
A=$1; B=$2; R=${3:-2}
for i in $(seq $R); do
  for cfg in "c2 --spp 512" "c3 --spp 1024 --kind 1" "c4 --spp 256 --size 2048 --kind 2 --tris 1000000"; do
    set -- $cfg; name=$1; shift
    for v in "A:$A" "B:$B"; do
      label=${v%%:*}; envs=${v#*:}
      out=$(env $envs timeout -k 10 400 python tools/perf_probe.py --reps 3 "$@" 2>&1 | grep -o "[0-9.]* Mpaths/s" | sort -n | tail -1)
      echo "$name $label [$envs]: $out"
    done
  done
done
