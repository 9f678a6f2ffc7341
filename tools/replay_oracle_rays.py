"""Ray-level parity: EVERY ray the oracle traces for a small frame (oracle.logged_rays: closest-hit rays of every bounce and
next-event shadow rays, with the directions the reference's arithmetic really produces - not always unit vectors) replayed through
the tree as hj_scene_upload re-laid it out (collapse, pair nodes, guard nodes, hot-first order): hj_debug_trace must find the
same shape and the same t bits, the any-hit walk the same boolean.  This is the probe that found round 5's sphere-guard bug
(a padded box in front of a sphere leaf is not exact for |d| != 1); run it after any change to the upload's re-layout.

    python tools/replay_oracle_rays.py [first_seed count] [--device-tree]
        the three synthetic scenes, three clusters of small spheres, `count` random scenes; --device-tree: every scene on the tree
        hj_build_bvh_device builds for it instead of the compiled one (the oracle walks that tree, too)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import scenes
from hijiki_amd import host, device
from oracle import hj_oracle as O

device_tree = "--device-tree" in sys.argv
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
first = int(argv[0]) if len(argv) > 0 else 700
count = int(argv[1]) if len(argv) > 1 else 20


def sphere_cluster(n, radius_scale, seed):
    s = host.Scene()
    s.set_camera_cbox()
    m, e, g, mi = s.add_diffuse((0.6, 0.6, 0.6)), s.add_emissive((9, 9, 9)), s.add_dielectric(1.5, (0.1, 0.2, 0.3)), s.add_mirror()
    rng = np.random.default_rng(seed)
    for k in range(n):
        s.add_sphere(tuple(rng.uniform(-0.9, 0.9, 3) + (0, 1, 0)), radius_scale * (0.01 + 0.02 * rng.random()), (m, m, g, mi)[k % 4])
    s.add_quad((-0.3, 1.98, -0.3), (0.6, 0, 0), (0, 0, 0.6), e)
    return s.compile()


r = device.Renderer(0)
W, H, spp = 96, 64, 2
todo = [("cbox", host.Scene.synthetic(host.SYNTH_CBOX).compile()), ("cbox + spheres", host.Scene.synthetic(host.SYNTH_CBOX_SPHERES).compile()),
        ("mesh 20 k", host.Scene.synthetic(host.SYNTH_CBOX_MESH, mesh_triangles=20000).compile())]
todo += [("20 spheres", sphere_cluster(20, 5.0, 11)), ("300 spheres", sphere_cluster(300, 2.0, 12)), ("3000 spheres", sphere_cluster(3000, 1.0, 13))]
todo += [(f"random scene {s}", scenes.random_scene(s)) for s in range(first, first + count)]
todo += [(f"random cluster scene {s}", scenes.random_cluster_scene(s)) for s in range(first, first + count // 2)]
todo += [(f"nasty scene {s}", scenes.nasty_scene(s)) for s in range(first, first + count // 2)]
bad_total = 0
for name, cs in todo:
    blocks = host.make_blocks(W, H, spp, 3)
    if device_tree and cs.num_shapes >= 2:
        cs.set_bvh(r.build_bvh(cs))
    log = O.logged_rays(cs, blocks)
    rays = np.ascontiguousarray(log[:, 0:8])
    want = log[:, 9].astype(np.int32)
    r.upload_scene(cs)
    ids, t, _, _ = r.trace(rays)
    oi, ot, _, _ = O.intersect(cs, rays)
    anyhit, *_ = r.trace(rays, any_hit=True)
    bad = int((ids != want).sum()) + int((t.view(np.uint32) != ot.view(np.uint32))[want >= 0].sum()) + int(((anyhit >= 0) != (want >= 0)).sum())
    length = np.linalg.norm(log[:, 3:6].astype(np.float64), axis=1)
    bad_total += bad
    print(f"{name}: {len(log)} rays ({int((log[:, 8] == 1).sum())} shadow), direction lengths {length.min():.6f} .. {length.max():.6f}, "
          f"mismatches {bad}", flush=True)
print(f"{len(todo)} scenes, {bad_total} mismatches")
sys.exit(1 if bad_total else 0)
