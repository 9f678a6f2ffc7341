"""The ray-voted child order on the device (hj_tune_bvh_device, kernels/hj_vote.h) against the host's (hjh_compiled_tune_bvh):
oracle walk counters on a small frame, bit-exactness of the HIP path on the voted tree, time of the pass, frame rate at the
configuration's own size; and hj_build_bvh_device with and without the vote at its end.

    python tools/vote_probe.py [c2 c3 m60k c4 ...]
"""
import os, sys, time
os.environ["HJ_BVH_CHILD_ORDER"] = "3"          # the host builder's tree WITHOUT its own vote: the starting point of both
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hijiki_amd import host, device
from oracle import hj_oracle as O

CFG = {"c2": (host.SYNTH_CBOX, 0, 1024, 512), "c3": (host.SYNTH_CBOX_SPHERES, 0, 1024, 1024),
       "m60k": (host.SYNTH_CBOX_MESH, 60_000, 1024, 256), "c4": (host.SYNTH_CBOX_MESH, 1_000_000, 2048, 256)}


def counters(cs, size=128, spp=2):
    blocks = host.make_blocks(size, size, spp, 1)
    O.lib().hjo_set_shadow_anyhit(1)
    img, c, _ = O.render_blocks(cs, blocks, size, size)
    cc, sc = max(1, c["closest_calls"]), max(1, c["shadow_calls"])
    return img, f"nodes per closest ray {c['nodes'] / cc:.2f}, per shadow ray {c['shadow_nodes'] / sc:.2f}", blocks


def main():
    names = sys.argv[1:] or ["c2", "c3", "m60k", "c4"]
    r = device.Renderer(0)
    for name in names:
        kind, tris, size, spp = CFG[name]
        cs = host.Scene.synthetic(kind, mesh_triangles=tris).compile()
        base = cs.bvh.copy()

        def rate():
            r.upload_scene(cs); r.create_framebuffer(size, size)
            best = 1e9
            for _ in range(3):
                r.clear(); t = time.time(); r.render_frame(spp, 1); best = min(best, time.time() - t)
            return size * size * spp / best / 1e6

        def check(label, ms):
            img, txt, blocks = counters(cs)
            r.upload_scene(cs); r.create_framebuffer(128, 128)
            r.render_blocks(blocks)
            same = np.array_equal(r.read().view(np.uint32), np.ascontiguousarray(img, np.float32).view(np.uint32))
            print(f"  {label:28s} {ms:8.1f} ms   {txt}   HIP == oracle: {same}   {rate():.0f} Mpaths/s", flush=True)

        print(f"{name}: {len(base)} nodes", flush=True)
        check("host tree, order by shapes", 0.0)
        t = time.time(); cs.tune_bvh(0, 60000); ms = (time.time() - t) * 1e3
        check("+ host vote", ms)
        host_voted = cs.bvh.copy()
        cs.set_bvh(base)
        r.tune_bvh_device(cs)                             # (module load)
        t = time.time(); nodes = r.tune_bvh_device(cs); ms = (time.time() - t) * 1e3
        same_leaves = np.array_equal(np.sort(nodes[:, 3]), np.sort(base[:, 3]))
        agree = float(np.mean(nodes[:, 3] == host_voted[:, 3]))
        cs.set_bvh(nodes)
        check("+ device vote", ms)
        print(f"  same leaves {same_leaves}; records whose shape word equals the host-voted array's: {agree:.3f}", flush=True)
        for paths in (0, 60000):
            os.environ["HJ_LBVH_VOTE_PATHS"] = str(paths)
            r.build_bvh(cs)
            t = time.time(); nodes = r.build_bvh(cs); ms = (time.time() - t) * 1e3
            cs.set_bvh(nodes)
            check(f"device build, vote {paths}", ms)
        os.environ.pop("HJ_LBVH_VOTE_PATHS")


if __name__ == "__main__":
    main()
