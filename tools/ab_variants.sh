#!/bin/bash
# ab_variants.sh CONFIG NAME...: best blocking frame of CONFIG (c2 | c3 | c4) for the tree's library ("head") and for each
# build/variants/var_NAME.so, twice, interleaved on one box
export GPU_MAX_HW_QUEUES=8
cfg=$1; shift
case $cfg in
  c2) args="--spp 512" ;;
  c3) args="--spp 1024 --kind 1" ;;
  c4) args="--spp 256 --size 2048 --kind 2 --tris 1000000" ;;
esac
for i in 1 2; do
  for v in head "$@"; do
    if [ "$v" = head ]; then unset HIJIKI_HIP_LIB; else export HIJIKI_HIP_LIB=build/variants/var_$v.so; fi
    out=$(timeout -k 10 400 python tools/perf_probe.py --reps 3 $args 2>&1 | grep -o "[0-9.]* Mpaths/s" | sort -n | tail -1)
    echo "$cfg $v: $out"
  done
done
