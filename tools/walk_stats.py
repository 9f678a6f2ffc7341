"""Wave-level occupancy of the walk's phases (needs the diagnostic build: tools/build_variant.sh stats -DHJ_WALK_STATS).

    python tools/walk_stats.py KIND [MAX_BOUNCES] [--json OUT]     env: HJ_STATS_SPP (512 = the batch sizes of the benchmark; small frames give small batches and emptier waves), HJ_STATS_SIZE (1024), HJ_STATS_TRIS
"""
import sys, os, json, ctypes as C
os.environ.setdefault("HIJIKI_HIP_LIB", "build/variants/var_stats.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hijiki_amd import host, device
argv = [a for a in sys.argv[1:]]
js = None
if "--json" in argv:
    js = argv[argv.index("--json") + 1]
    del argv[argv.index("--json"):argv.index("--json") + 2]
kind = int(argv[0]) if len(argv) > 0 else host.SYNTH_CBOX
max_bounces = int(argv[1]) if len(argv) > 1 else 0     # 1 = camera rays and their shadow rays only
spp = int(os.environ.get("HJ_STATS_SPP", "512")); size = int(os.environ.get("HJ_STATS_SIZE", "1024"))
cs = host.Scene.synthetic(kind, mesh_triangles=int(os.environ.get("HJ_STATS_TRIS", "0"))).compile()
r = device.Renderer(0)
if os.environ.get("HJ_STATS_DEVICE_BVH") == "1":        # the tree bench.py's c4 uses
    cs.set_bvh(r.build_bvh(cs))
r.upload_scene(cs); r.create_framebuffer(size, size)
L = device.lib()
out = (C.c_ulonglong * 16)()
L.hj_debug_walk_stats(out, 1)
rs = (C.c_ulonglong * 32)()
L.hj_debug_round_stats(rs, 1)
o = device.default_opts()
if max_bounces:
    o.max_bounces = max_bounces
st = r.render_frame(spp, 1, opts=o)
L.hj_debug_walk_stats(out, 1)
o = list(out)
rays = st["closest_rays"] + st["shadow_rays"]
steps = o[1] + o[8]
print(f"rays {rays/1e6:.1f} M; outer iterations {o[0]/1e6:.2f} M, active lanes at their start {o[7]/max(1,o[0]):.1f}")
print(f"box wave-steps {steps/1e6:.2f} M with {(o[2]+o[9])/max(1,steps):.1f} lanes  ({(o[2]+o[9])/rays:.1f} lane-steps per ray, {o[14]/rays:.2f} of them on nodes outside the LDS copy; {steps/max(1,o[0]):.2f} wave-steps per outer iteration)")
print(f"  of these, merged walk (continuing + shadow rays): {o[1]/1e6:.2f} M wave-steps with {o[2]/max(1,o[1]):.1f} lanes; camera packets (scalar-cache walk, "
      f"one node per wave-step): {o[8]/1e6:.2f} M wave-steps with {o[9]/max(1,o[8]):.1f} live lanes = {o[9]/max(1,st['paths']):.1f} lane-steps per camera ray")
print(f"leaf phases {o[3]/1e6:.2f} M with {o[4]/max(1,o[3]):.1f} lanes  ({o[4]/rays:.2f} leaf stops and {o[15]/rays:.2f} shape records per ray; in {100*o[3]/max(1,o[0]):.0f} % of outer iterations)")
tot = max(1, o[13])
print(f"wave cycles in the walk: service {100*o[10]/tot:.1f} %, box steps {100*o[11]/tot:.1f} %, leaf tests {100*o[12]/tot:.1f} %; per outer iteration {o[13]/max(1,o[0]):.0f} cycles; per box wave-step {o[11]/max(1,steps):.0f}; per leaf phase {o[12]/max(1,o[3]):.0f}; per refill {o[10]/max(1,o[5]):.0f}")
print(f"refills {o[5]/1e6:.2f} M with {o[6]/max(1,o[5]):.1f} rays each")
L.hj_debug_round_stats(rs, 1)
rs = list(rs)
tot_c = sum(rs[16:24]) or 1
tot_r = sum(rs[8:16]) or 1
print("rounds of the fused kernel by size (rays in the round): share of rays vs share of wave-time")
lo = 0
hist = []
for b in range(8):
    hi = 16 << (2 * b)
    if rs[b]:
        print(f"  [{lo:6d}, {hi if b < 7 else 10**9:>10d}): {rs[b]/1e3:9.1f} k rounds, {100*rs[8+b]/tot_r:5.1f} % of rays, {100*rs[16+b]/tot_c:5.1f} % of wave-time")
        hist.append({"rays_lo": lo, "rays_hi": hi, "rounds": rs[b], "share_of_rays": rs[8+b]/tot_r, "share_of_wave_time": rs[16+b]/tot_c})
    lo = hi
stage = rs[24:29]
stage_tot = sum(stage) or 1
names = ("top-up (camera rays)", "walk", "hit compaction", "shade", "round bookkeeping")
print(f"waves waiting at the barrier that ends the walk: {100*rs[29]/max(1,rs[25]):.1f} % of the walk's wave time")
print("wall time of the workgroups by stage: " + ", ".join(f"{n} {100*v/stage_tot:.1f} %" for n, v in zip(names, stage)))
if js:
    json.dump({"kind": kind, "size": size, "spp": spp, "rays": rays, "paths": st["paths"],
               "box_lane_steps_per_ray": (o[2] + o[9]) / rays, "cold_node_steps_per_ray": o[14] / rays,
               "leaf_tests_per_ray": o[4] / rays, "triangle_records_per_ray": o[15] / rays, "lanes_per_box_step": (o[2] + o[9]) / max(1, steps),
               "lanes_per_leaf_phase": o[4] / max(1, o[3]), "active_lanes": o[7] / max(1, o[0]),
               "cycle_share": {"service": o[10] / tot, "box": o[11] / tot, "leaf": o[12] / tot},
               "stage_share": {n: v / stage_tot for n, v in zip(("top_up", "walk", "compaction", "shade", "bookkeeping"), stage)},
               "merged_box_wave_steps": o[1], "merged_box_lane_steps": o[2], "packet_wave_steps": o[8], "packet_lane_steps": o[9],
               "packet_cold_wave_steps": rs[30], "leaf_phases": o[3], "outer_iterations": o[0],
               "build": "the shipped kernel (camera packets on) compiled with -DHJ_WALK_STATS: counters and clock reads added, nothing else changed",
               "rounds": hist}, open(js, "w"), indent=1)
