"""Parity soak: N randomised scenes (tests/scenes.py: random mixes of triangles, spheres, quads, all five materials, several lights of
different shape kinds, random cameras) through the HIP path with every exact shortcut on (light-shaft grid, leaf guards, pair nodes,
collapse, camera packets) against the oracle, bit for bit; the share of shadow rays the grid proved free is printed per scene.

    python tools/soak_random_scenes.py [first_seed] [count] [--device-tree] [--clusters | --smooth]

--clusters: tests/scenes.py random_cluster_scene instead of random_scene (hundreds of small spheres of all materials, small triangles,
three lights: the class of scene in which round 5's sphere guards were wrong).
--smooth: tests/scenes.py smooth_mesh_scene (tessellated smooth bodies in a box, one to three lights: what the light-shaft grid's cells on
meshes and in corners are for; the share of proven rays is high here).
--device-tree: every scene a second time on the tree hj_build_bvh_device builds for it (Morton clusters, SAH re-split, host top, the
ray vote of kernels/hj_vote.h on the device), which must be a valid tree the oracle and the HIP path walk to the same bits.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import scenes
from hijiki_amd import host, device
from oracle import hj_oracle as O

device_tree = "--device-tree" in sys.argv
clusters = "--clusters" in sys.argv
smooth = "--smooth" in sys.argv
scale = float(os.environ.get("SOAK_SCALE", "1"))              # with --clusters: the scene magnified (the reference's epsilons are not)
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
first = int(argv[0]) if len(argv) > 0 else 100
count = int(argv[1]) if len(argv) > 1 else 60
r = device.Renderer(0)
W, H = 160, 96
bad_total = proven = shadow = 0
for seed in range(first, first + count):
    cs = scenes.smooth_mesh_scene(seed) if smooth else scenes.random_cluster_scene(seed, scale=scale) if clusters else scenes.random_scene(seed)
    blocks = host.make_blocks(W, H, 3, seed)
    want, ctr, _ = O.render_blocks(cs, blocks, W, H)
    r.upload_scene(cs); r.create_framebuffer(W, H)
    st = r.render_blocks(blocks)
    got = r.read()
    bad = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=-1).sum())
    ok = bad == 0 and st["closest_rays"] == ctr["closest_calls"] and st["shadow_rays"] == ctr["shadow_calls"] and st["hits"] == ctr["hits"]
    bad_total += 0 if ok else 1
    proven += st["shadow_rays_proven_free"]; shadow += st["shadow_rays"]
    print(f"seed {seed}: {'ok ' if ok else 'FAIL'} differing pixels {bad}, shapes {cs.num_shapes}, shadow rays {st['shadow_rays']}, proven free {st['shadow_rays_proven_free']}", flush=True)
    if device_tree and cs.num_shapes >= 2:
        cs.set_bvh(r.build_bvh(cs))
        want, ctr, _ = O.render_blocks(cs, blocks, W, H)
        r.upload_scene(cs); r.create_framebuffer(W, H)          # (the upload validates the links)
        st = r.render_blocks(blocks)
        bad = int((r.read().view(np.uint32) != want.view(np.uint32)).any(axis=-1).sum())
        ok = bad == 0 and st["closest_rays"] == ctr["closest_calls"] and st["shadow_rays"] == ctr["shadow_calls"] and st["hits"] == ctr["hits"]
        bad_total += 0 if ok else 1
        print(f"seed {seed}, device-built tree: {'ok ' if ok else 'FAIL'} differing pixels {bad}", flush=True)
print(f"{count} scenes, {bad_total} failures; {proven} of {shadow} shadow rays proven free ({100.0 * proven / max(1, shadow):.1f} %)")
sys.exit(1 if bad_total else 0)
