#!/bin/bash
# device-built tree against the host's (rotated) SAH tree at the probe's batch sizes
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab67; mkdir -p $out
for cfg in "c2:" "c3:--kind 1 --spp 256" "c4:--kind 2 --tris 1000000 --size 2048 --spp 64"; do
  name=${cfg%%:*}; args=${cfg#*:}
  for rep in 1 2; do
    echo -n "$name host:   "; timeout 200 python tools/perf_probe.py --spp 512 --reps 3 $args 2>&1 | grep -o "[0-9.]* Mpaths/s" | sort -n | tail -1
    echo -n "$name device: "; timeout 200 python tools/perf_probe.py --spp 512 --reps 3 $args --device-bvh 1 2>&1 | grep -o "[0-9.]* Mpaths/s" | sort -n | tail -1
  done
done | tee $out/dev_vs_host.txt
