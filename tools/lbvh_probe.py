"""Device-built LBVH vs host-built SAH tree: build time and frame rate (cbox and the 1 M-triangle mesh).

    python tools/lbvh_probe.py [--variants]      --variants: the device build with / without the rotation passes over its
                                                 host-built top (HJ_LBVH_TOP_ROTATE)
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hijiki_amd import host, device
r = device.Renderer(0)
variants = [("default", {})]
if "--variants" in sys.argv:
    variants += [("no rotations", {"HJ_LBVH_TOP_ROTATE": "0"}), ("rotations", {"HJ_LBVH_TOP_ROTATE": "8"})]
# (the configurations' own sample counts: small frames run smaller batches, and the trees rank differently there - at 32 spp the
# device-built tree of the mesh is 2 % AHEAD of the host's, at the benchmark's 256 spp 2.5 % behind)
for name, kind, tris, size, spp in (("cbox", host.SYNTH_CBOX, 0, 1024, 512), ("1 M-triangle mesh", host.SYNTH_CBOX_MESH, 1000000, 2048, 256)):
    s = host.Scene.synthetic(kind, mesh_triangles=tris)
    t = time.time(); cs = s.compile(); t_host = time.time() - t          # (the host tree: built with the environment as it is)
    host_nodes = cs.bvh.copy()

    def rate():
        r.upload_scene(cs); r.create_framebuffer(size, size)
        best = 1e9
        for _ in range(3):
            r.clear(); t = time.time(); r.render_frame(spp, 1); best = min(best, time.time() - t)
        return size * size * spp / best / 1e6

    base = rate()
    print(f"{name}: {cs.desc.num_bvh_nodes} nodes; host compile {t_host*1e3:.0f} ms: {base:.0f} Mpaths/s", flush=True)
    for label, env in variants:
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        r.build_bvh(cs)                                   # warm-up (module load, allocations)
        t = time.time(); nodes = r.build_bvh(cs); t_dev = time.time() - t
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
        cs.set_bvh(nodes)
        v = rate()
        t = time.time(); cs.tune_bvh(0, 60000); t_tune = time.time() - t      # the ray-voted child order on the device-built tree
        v2 = rate()
        cs.set_bvh(host_nodes)
        print(f"  device build ({label}): {t_dev*1e3:.1f} ms, {v:.0f} Mpaths/s = {v / base:.3f} of the host tree; + hjh_compiled_tune_bvh "
              f"(vote, {t_tune*1e3:.0f} ms): {v2:.0f} Mpaths/s = {v2 / base:.3f}", flush=True)
