#!/bin/bash
# ab_variants.sh "name[:ENV=VAL,...]" ... : best of 3 C2 frames per variant library (build/variants/var_NAME.so; "cur" = the built library), two rounds
export GPU_MAX_HW_QUEUES=8   # before rocprofv3 / python start: the tool library initialises HIP first, later settings are ignored
for round in 1 2; do
  for spec in "$@"; do
    name=${spec%%:*}; envs=""; [[ "$spec" == *:* ]] && envs=$(echo "${spec#*:}" | tr ',' ' ')
    lib=build/variants/var_$name.so; [ "$name" = cur ] && lib=hijiki_amd/lib/libhijiki_hip.so
    echo -n "$spec: "; env HIJIKI_HIP_LIB=$lib $envs timeout 100 python tools/perf_probe.py --spp 512 --reps 3 $PROBE_ARGS 2>&1 | grep -o "[0-9.]* Mpaths/s" | sort -n | tail -1
  done
done
