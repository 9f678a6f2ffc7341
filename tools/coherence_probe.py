"""How much faster are COHERENT rays?  Frames with max_bounces = 1 (camera rays, walked as packets, and the shadow rays of their
hits: neighbouring origins, one small light) against whole frames, in rays per second, on c2 / c3 / c4's scenes.

    python tools/coherence_probe.py
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hijiki_amd import host, device
r = device.Renderer(0)
for name, kind, tris, size, spp in (("c2 cbox", host.SYNTH_CBOX, 0, 1024, 512), ("c3 spheres", host.SYNTH_CBOX_SPHERES, 0, 1024, 512),
                                    ("c4 1M mesh", host.SYNTH_CBOX_MESH, 1000000, 2048, 128)):
    cs = host.Scene.synthetic(kind, mesh_triangles=tris).compile()
    r.upload_scene(cs); r.create_framebuffer(size, size)
    for label, mb in (("whole paths", 1000), ("bounce 0 only", 1), ("bounces 0-1", 2)):
        o = device.default_opts(); o.max_bounces = mb
        best, st = 1e9, None
        for _ in range(3):
            r.clear(); t = time.time(); st = r.render_frame(spp, 1, opts=o); best = min(best, time.time() - t)
        rays = st["closest_rays"] + st["shadow_rays"]
        print(f"{name:11s} {label:14s}: {rays / st['paths']:.2f} rays/path, {rays / best / 1e9:6.2f} G rays/s, {st['paths'] / best / 1e6:7.0f} Mpaths/s", flush=True)
