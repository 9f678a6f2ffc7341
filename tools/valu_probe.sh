#!/bin/bash
# valu_probe.sh TAG [extra]: is the shipped kernel VALU-bound?  Same-box, interleaved A/B of the tree's library against a build
# with EXTRA (default 16) more VALU instructions in every plain box step of the walk (-DHJ_VALU_PROBE=EXTRA), on c2 / c3 / c4 at
# their own sizes; writes gpurun_out/TAG_<config>_valu_probe.json (copy into profiles/: bench.py's `limited_by` reads the newest).
#   slope = (% more frame time) / (% more VALU instructions); the VALU increase is EXTRA x box wave-steps over the frame's
#   SQ_INSTS_VALU, both from the newest profiles/rNN_<config>_roofline_inputs.json.
# Needs build/variants/var_valuprobe.so (tools/build_variant.sh valuprobe -DHJ_VALU_PROBE=16), built in the build container.
export GPU_MAX_HW_QUEUES=8
tag=$1; extra=${2:-16}
mkdir -p gpurun_out
best() { grep -o "[0-9.]* Mpaths/s" | sort -n | tail -1 | cut -d' ' -f1; }
for cfg in "c2 --spp 512" "c3 --spp 1024 --kind 1" "c4 --spp 256 --size 2048 --kind 2 --tris 1000000"; do
  set -- $cfg; name=$1; shift
  a=(); b=()
  for i in 1 2 3; do
    a+=($(timeout -k 10 300 python tools/perf_probe.py --reps 3 "$@" 2>&1 | best))
    b+=($(HIJIKI_HIP_LIB=build/variants/var_valuprobe.so timeout -k 10 300 python tools/perf_probe.py --reps 3 "$@" 2>&1 | best))
  done
  python3 - "$name" "$extra" "$tag" "${a[*]}" "${b[*]}" <<'PY'
import sys, json, glob, os
name, extra, tag = sys.argv[1], int(sys.argv[2]), sys.argv[3]
a = [float(x) for x in sys.argv[4].split()]; b = [float(x) for x in sys.argv[5].split()]
inp = sorted(glob.glob(f"profiles/r*_{name}_roofline_inputs.json"))
d = json.load(open(inp[-1])) if inp else {}
w = d.get("walk", {})
rays, insts = w.get("rays"), d.get("counters", {}).get("SQ_INSTS_VALU")
steps = w.get("merged_box_wave_steps")
if steps is None and w.get("lanes_per_box_step"):
    steps = w["box_lane_steps_per_ray"] * rays / w["lanes_per_box_step"]
valu_pct = None if not (steps and insts) else 100.0 * extra * steps / insts
ra, rb = max(a), max(b)
time_pct = 100.0 * (ra / rb - 1.0)
out = {"config": name, "extra_valu_per_box_step": extra, "shipped_mrays": a, "probe_mrays": b, "time_increase_pct": round(time_pct, 2),
       "valu_increase_pct": None if valu_pct is None else round(valu_pct, 1),
       "slope_time_pct_per_valu_pct": None if not valu_pct else round(max(0.0, time_pct) / valu_pct, 3),
       "inputs": os.path.basename(inp[-1]) if inp else None,
       "method": "tools/valu_probe.sh: best blocking frame of 3 runs x 3 frames each, interleaved on one box; probe build = -DHJ_VALU_PROBE"}
json.dump(out, open(f"gpurun_out/{tag}_{name}_valu_probe.json", "w"), indent=1)
print(json.dumps(out))
PY
done
