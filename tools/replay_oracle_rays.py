"""Ray-level parity: EVERY ray the oracle traces for a small frame (oracle.logged_rays: closest-hit rays of every bounce and
next-event shadow rays, with the directions the reference's arithmetic really produces - not always unit vectors) replayed through
the tree as hj_scene_upload re-laid it out (collapse, pair nodes, guard nodes, hot-first order): hj_debug_trace must find the
same shape and the same t bits, the any-hit walk the same boolean.  This is the probe that found round 5's sphere-guard bug
(a padded box in front of a sphere leaf is not exact for |d| != 1); run it after any change to the upload's re-layout.

    python tools/replay_oracle_rays.py [first_seed count]        # the three synthetic scenes + `count` random scenes
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import scenes
from hijiki_amd import host, device
from oracle import hj_oracle as O

first = int(sys.argv[1]) if len(sys.argv) > 1 else 700
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
r = device.Renderer(0)
W, H, spp = 96, 64, 2
todo = [("cbox", host.Scene.synthetic(host.SYNTH_CBOX).compile()), ("cbox + spheres", host.Scene.synthetic(host.SYNTH_CBOX_SPHERES).compile()),
        ("mesh 20 k", host.Scene.synthetic(host.SYNTH_CBOX_MESH, mesh_triangles=20000).compile())]
todo += [(f"random scene {s}", scenes.random_scene(s)) for s in range(first, first + count)]
bad_total = 0
for name, cs in todo:
    blocks = host.make_blocks(W, H, spp, 3)
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
