"""Wave-level occupancy of the walk's phases (needs the diagnostic build: tools/build_variant.sh stats -DHJ_WALK_STATS)."""
import sys, os, ctypes as C
os.environ.setdefault("HIJIKI_HIP_LIB", "hijiki_amd/lib/var_stats.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hijiki_amd import host, device
kind = int(sys.argv[1]) if len(sys.argv) > 1 else host.SYNTH_CBOX
max_bounces = int(sys.argv[2]) if len(sys.argv) > 2 else 0     # 1 = camera rays and their shadow rays only
cs = host.Scene.synthetic(kind).compile()
r = device.Renderer(0); r.upload_scene(cs); r.create_framebuffer(1024, 1024)
L = device.lib()
out = (C.c_ulonglong * 8)()
L.hj_debug_walk_stats(out, 1)
rs = (C.c_ulonglong * 24)()
L.hj_debug_round_stats(rs, 1)
o = device.default_opts()
if max_bounces:
    o.max_bounces = max_bounces
st = r.render_frame(16, 1, opts=o)
L.hj_debug_walk_stats(out, 1)
o = list(out)
rays = st["closest_rays"] + st["shadow_rays"]
print(f"rays {rays/1e6:.1f} M; outer iterations {o[0]/1e6:.2f} M, active lanes at their start {o[7]/max(1,o[0]):.1f}")
print(f"inner wave-steps {o[1]/1e6:.2f} M with {o[2]/max(1,o[1]):.1f} lanes  ({o[2]/rays:.1f} lane-steps per ray; {o[1]/max(1,o[0]):.2f} wave-steps per outer iteration)")
print(f"leaf phases {o[3]/1e6:.2f} M with {o[4]/max(1,o[3]):.1f} lanes  ({o[4]/rays:.2f} leaf tests per ray; in {100*o[3]/max(1,o[0]):.0f} % of outer iterations)")
print(f"refills {o[5]/1e6:.2f} M with {o[6]/max(1,o[5]):.1f} rays each")
L.hj_debug_round_stats(rs, 1)
rs = list(rs)
tot_c = sum(rs[16:24]) or 1
tot_r = sum(rs[8:16]) or 1
print("rounds of the fused kernel by size (rays in the round): share of rays vs share of wave-time")
lo = 0
for b in range(8):
    hi = 16 << (2 * b)
    if rs[b]:
        print(f"  [{lo:6d}, {hi if b < 7 else 10**9:>10d}): {rs[b]/1e3:9.1f} k rounds, {100*rs[8+b]/tot_r:5.1f} % of rays, {100*rs[16+b]/tot_c:5.1f} % of wave-time")
    lo = hi
