#!/bin/bash
# sweep_env.sh CONFIG "ENV1" "ENV2" ...: best blocking frame (tools/perf_probe.py, 3 frames) of CONFIG (c2 | c3 | c4) under each environment, twice, interleaved
export GPU_MAX_HW_QUEUES=8
cfg=$1; shift
case $cfg in
  c2) args="--spp 512" ;;
  c3) args="--spp 1024 --kind 1" ;;
  c4) args="--spp 256 --size 2048 --kind 2 --tris 1000000" ;;
esac
for i in 1 2; do
  for envs in "$@"; do
    out=$(env $envs timeout -k 10 400 python tools/perf_probe.py --reps 3 $args 2>&1 | grep -o "[0-9.]* Mpaths/s" | sort -n | tail -1)
    echo "$cfg [$envs]: $out"
  done
done
