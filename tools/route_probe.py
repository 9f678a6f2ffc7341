"""Three routes to a large scene's tree, same box: (a) the host's compile (binned SAH + rotations + batched reinsertion + ray vote),
(b) hj_build_bvh_device (Morton clusters + SAH re-split + ray vote, 30 ms), (c) the device's tree with the host's tree passes run
on it (hjh_compiled_tune_bvh: batched reinsertion + vote) - build time and frame rate (best blocking frame of 3) of each.

    python tools/route_probe.py [TRIS] [SIZE SPP]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hijiki_amd import host, device
tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
size = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 256
s = host.Scene.synthetic(host.SYNTH_CBOX_MESH, mesh_triangles=tris)
r = device.Renderer(0)


def rate(cs):
    r.upload_scene(cs)
    r.create_framebuffer(size, size)
    best = 0.0
    for _ in range(3):
        r.clear()
        t = time.time()
        r.render_frame(spp, 1)
        best = max(best, size * size * spp / (time.time() - t) / 1e6)
    return best


for rnd in range(2):
    t = time.time(); cs = s.compile(); ta = time.time() - t
    print(f"(a) host compile            {1e3 * ta:8.1f} ms  {rate(cs):7.1f} Mpaths/s", flush=True)
    t = time.time(); cs2 = s.compile(with_tree=False); cs2.set_bvh(r.build_bvh(cs2)); tb = time.time() - t
    print(f"(b) device build            {1e3 * tb:8.1f} ms  {rate(cs2):7.1f} Mpaths/s", flush=True)
    for passes in (2, 6):
        t = time.time(); cs3 = s.compile(with_tree=False); cs3.set_bvh(r.build_bvh(cs3)); cs3.tune_bvh(reinsert_passes=passes, vote_paths=60000); tc = time.time() - t
        print(f"(c) device build + {passes} passes {1e3 * tc:8.1f} ms  {rate(cs3):7.1f} Mpaths/s", flush=True)
