"""Device-built LBVH vs host-built SAH tree: build time and frame rate (cbox and the 1 M-triangle mesh)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hijiki_amd import host, device
r = device.Renderer(0)
for name, kind, tris, size, spp in (("cbox", host.SYNTH_CBOX, 0, 1024, 64), ("1 M-triangle mesh", host.SYNTH_CBOX_MESH, 1000000, 2048, 16)):
    s = host.Scene.synthetic(kind, mesh_triangles=tris)
    t = time.time(); cs = s.compile(); t_host = time.time() - t
    r.build_bvh(cs)                                   # warm-up (module load, allocations)
    t = time.time(); nodes = r.build_bvh(cs); t_dev = time.time() - t
    rates = {}
    for label in ("SAH (host)", "LBVH (device)"):
        if label.startswith("LBVH"):
            cs.set_bvh(nodes)
        r.upload_scene(cs); r.create_framebuffer(size, size)
        best = 1e9
        for _ in range(3):
            r.clear(); t = time.time(); st = r.render_frame(spp, 1); best = min(best, time.time() - t)
        rates[label] = (size * size * spp / best / 1e6, (st["closest_rays"] + st["shadow_rays"]) / st["paths"])
    print(f"{name}: {cs.desc.num_bvh_nodes} nodes; host compile {t_host*1e3:.0f} ms, device build {t_dev*1e3:.1f} ms; " +
          "; ".join(f"{k}: {v[0]:.0f} Mpaths/s" for k, v in rates.items()), flush=True)
