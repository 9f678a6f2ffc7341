"""CPU-only attack on the light-shaft grid (api/light_grid.cpp), cells on meshes included (HJ_LIGHT_GRID_MESH=1): for many scenes,
(a) every next-event shadow ray the ORACLE traces for a frame, looked up in the grid as the kernel does - a proven ray that the
    oracle finds occluded would be a wrong pixel;
(b) crafted rays: points on EVERY flat shape (uniform per shape, not by area: the small mesh triangles count as much as the walls),
    moved +-1e-5 off their plane and slightly across their edges (planar bits), or slid along a random incoming direction as far as
    the shade stage's check of the hit point admits (bits of cells on meshes and in corners), towards corners, edges and random
    points of the emitter - every ray that counts as proven must come back unoccluded from the oracle's closest-hit walk.

    HJ_LIGHT_GRID_MESH=1 python tools/light_grid_attack.py [first_seed count] [--smooth]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch  # noqa: F401
import scenes
import test_light_grid as TL
from hijiki_amd import host
from oracle import hj_oracle as O

smooth_only = "--smooth" in sys.argv                          # only tests/scenes.py smooth_mesh_scene (tessellated smooth bodies in a box)
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
first = int(argv[0]) if len(argv) > 0 else 0
count = int(argv[1]) if len(argv) > 1 else 20
RES = TL.RES


def attack(cs, rng, per_shape):
    grid = TL.Grid(cs, RES)
    if grid.got == 0:
        return 0, 0, 0, 0, 0, 0
    _, _, _, em, _, _ = TL.scene_arrays(cs)
    # (a) the oracle's own shadow rays, each with the hit it starts from (the shade stage's check of the hit point needs it)
    sh, ids, d, u, v = TL.oracle_shadow_rays(cs, O, host.make_blocks(128, 96, 2, 4))
    bad_a = proven_a = mesh_a = 0
    if len(sh):
        e = sh[:, 10].astype(np.int64)
        planar = grid.proven(sh[:, 0:3], e)
        proven = grid.proven(sh[:, 0:3], e, ids, d, u, v)
        bad_a, proven_a, mesh_a = int((proven & (sh[:, 9] >= 0)).sum()), int(proven.sum()), int((proven & ~planar).sum())
    # (b) crafted rays: for the planar bits any point of the cell near the plane (points on the shapes, a hair across their edges,
    # +-1e-5 off them); for the bits of cells on meshes and in corners hit points that pass the check (on the shape, slid along a
    # random incoming direction)
    p, nrm = TL.shape_points(cs, rng, per_shape)
    p = (p + nrm * rng.uniform(-1e-5, 1e-5, (len(p), 1))).astype(np.float32)
    hits = TL.crafted_hits(cs, grid, rng, per_shape)
    tried = bad_b = mesh_b = 0
    for e, shape in enumerate(em[:8]):
        o = p[grid.proven(p, np.full(len(p), e))]
        hsel = grid.proven(hits[1], np.full(len(hits[1]), e), hits[0], hits[2], hits[3], hits[4]) & ~grid.proven(hits[1], np.full(len(hits[1]), e))
        mesh_b += int(hsel.sum())
        o = np.concatenate([o, hits[1][hsel]])
        if not len(o):
            continue
        y = TL.emitter_points(cs, int(shape), rng, len(o))
        d = y - o
        dist = np.sqrt((d.astype(np.float32) ** 2).sum(1, dtype=np.float32)).astype(np.float32)
        d = (d / dist[:, None]).astype(np.float32)
        rays = np.zeros((len(o), 8), np.float32)
        rays[:, 0:3], rays[:, 3:6], rays[:, 6], rays[:, 7] = o, d, np.float32(2e-4), dist - np.float32(1e-4)
        ids, _, _, _ = O.intersect(cs, rays, use_bvh=True)
        tried += len(rays); bad_b += int((ids >= 0).sum())
    return proven_a, bad_a, tried, bad_b, mesh_a, mesh_b


def scene_list(first, count):
    if smooth_only:
        for s in range(first, first + count):
            yield f"smooth-mesh scene {s}", scenes.smooth_mesh_scene(s), 8
        return
    for k, tris in enumerate((320, 1280, 6320, 20000)):
        yield f"cbox, object of {tris} triangles", host.Scene.synthetic(host.SYNTH_CBOX, mesh_triangles=tris).compile(), 12
        yield f"cbox + spheres, {tris} triangles", host.Scene.synthetic(host.SYNTH_CBOX_SPHERES, mesh_triangles=tris).compile(), 12
        yield f"cbox with the dense mesh, {tris} triangles", host.Scene.synthetic(host.SYNTH_CBOX_MESH, mesh_triangles=tris).compile(), 12
    yield "rich scene", scenes.rich_scene(), 200
    for s in range(first, first + count):
        yield f"random scene {s}", scenes.random_scene(s), 200
        yield f"cluster scene {s}", scenes.random_cluster_scene(s, scale=(1.0, 0.1, 7.0)[s % 3]), 40
        yield f"nasty scene {s}", scenes.nasty_scene(s), 200
        yield f"smooth-mesh scene {s}", scenes.smooth_mesh_scene(s), 8


rng = np.random.default_rng(first)
tot = [0, 0, 0, 0, 0, 0]
t0 = time.time()
for name, cs, per_shape in scene_list(first, count):
    r = attack(cs, rng, per_shape)
    for k in range(6): tot[k] += r[k]
    flag = "" if r[1] == 0 and r[3] == 0 else "   <-- WRONG"
    print(f"{name}: oracle's shadow rays proven {r[0]} ({r[4]} by bundle proofs; occluded among them {r[1]}); crafted rays from proven cells {r[2]} ({r[5]} by bundle proofs; occluded {r[3]}){flag}", flush=True)
print(f"total: {tot[0]} proven oracle rays ({tot[4]} by bundle proofs), {tot[1]} occluded; {tot[2]} crafted rays ({tot[5]} by bundle proofs), {tot[3]} occluded; {time.time() - t0:.0f} s")
sys.exit(1 if tot[1] or tot[3] else 0)
