"""The K-wide device tree (hijiki_amd/csrc/wide_tree.h) keeps the reference's visiting order: per ray the same shapes are
tested, in the same order, each with the same tMax, as by the binary skip-link walk of reference shader/scene.glsl:97-133
(CPU check of the tree transformation and of the wide walk's rules; the kernels themselves are compared with the oracle
bit for bit under -m gpu)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import scenes
from hijiki_amd import host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("wide") / "wide_check.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "tests", "native", "wide_check.cpp")])
    L = C.CDLL(so)
    L.wide_check.restype = C.c_long
    L.wide_check.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                             C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_void_p]
    return L


def random_rays(cs, n, seed):
    rng = np.random.default_rng(seed)
    v = cs.vertices[:, :3] if len(cs.vertices) else np.zeros((1, 3), np.float32)
    lo, hi = v.min(0) - 0.3, v.max(0) + 0.3
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3] = rng.uniform(lo, hi, (n, 3))
    d = rng.normal(size=(n, 3))
    rays[:, 3:6] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 6] = 1e-4
    rays[:, 7] = np.where(rng.uniform(size=n) < 0.5, np.inf, rng.uniform(0.2, 3.0, n))
    # degenerate directions: exactly zero components (inf - inf in the slab test), axis-parallel rays, origins on box planes
    k = n // 8
    rays[:k, 3] = 0.0
    rays[k:2 * k, 3:6] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, k)] * rng.choice([-1.0, 1.0], (k, 1))
    rays[2 * k:3 * k, 4] = -0.0
    rays[3 * k:4 * k, 0] = 0.0
    return rays


def run(checker, cs, rays, anyhit, K):
    out = (C.c_uint64 * 3)()
    b = np.ascontiguousarray(cs.bvh)
    sp, qd = np.ascontiguousarray(cs.spheres), np.ascontiguousarray(cs.quads)
    tr, vx = np.ascontiguousarray(cs.triangles), np.ascontiguousarray(cs.vertices)
    rays = np.ascontiguousarray(rays, np.float32)
    ptr = lambda a: a.ctypes.data if a.size else None
    bad = checker.wide_check(ptr(b), len(b), ptr(sp), len(sp), ptr(qd), len(qd), ptr(tr), len(tr), ptr(vx), ptr(rays), len(rays),
                             int(anyhit), K, out)
    return bad, list(out)


@pytest.mark.parametrize("K", [2, 3, 4])
@pytest.mark.parametrize("scene", ["cbox", "spheres", "rich", "mesh20k"])
def test_wide_walk_is_the_reference_walk(checker, scene, K):
    cs = {"cbox": lambda: host.Scene.synthetic(host.SYNTH_CBOX).compile(),
          "spheres": lambda: host.Scene.synthetic(host.SYNTH_CBOX_SPHERES).compile(),
          "rich": lambda: scenes.rich_scene(),
          "mesh20k": lambda: host.Scene.synthetic(host.SYNTH_CBOX_MESH, mesh_triangles=20000).compile()}[scene]()
    rays = random_rays(cs, 20000, 11)
    for anyhit in (False, True):
        bad, (nodes, steps, events) = run(checker, cs, rays, anyhit, K)
        assert bad == 0, f"{scene} K={K} anyhit={anyhit}: {bad} rays visit other shapes / another order / with another tMax"
        assert events > len(rays) // 4 and nodes > 1


def test_malformed_trees_are_refused(checker):
    """hj_scene_upload accepts any forward-linked array; the wide conversion needs a proper binary tree and says so."""
    cs = host.Scene.synthetic(host.SYNTH_CBOX, mesh_triangles=320).compile()
    rays = random_rays(cs, 100, 3)
    nodes = cs.bvh                                   # a view of the compiled scene's own array
    saved = nodes[5, 7]
    nodes[5, 7] = nodes[5, 7] + 1 if nodes[5, 7] + 1 < len(nodes) else nodes[5, 7] - 1
    try:
        bad, _ = run(checker, cs, rays, False, 4)
        assert bad == -1
    finally:
        nodes[5, 7] = saved


def test_boxes_that_stick_out_keep_their_node(checker):
    """Shrunk inner boxes break containment (tests/test_gpu_parity.py test_uploaded_tree_with_inconsistent_boxes): such nodes
    are not dissolved and the walk stays the reference's, box for box."""
    cs = host.Scene.synthetic(host.SYNTH_CBOX, mesh_triangles=1280).compile()
    nodes, f = cs.bvh, cs.bvh_f32
    inner = np.nonzero(nodes[:, 3] == 0xFFFFFFFF)[0]
    rng = np.random.default_rng(5)
    for i in rng.choice(inner[1:], size=len(inner) // 3, replace=False):
        c = 0.5 * (f[i, 0:3] + f[i, 4:7])
        f[i, 0:3] = c + (f[i, 0:3] - c) * 0.8
        f[i, 4:7] = c + (f[i, 4:7] - c) * 0.8
    rays = random_rays(cs, 20000, 4)
    for K in (3, 4):
        bad, _ = run(checker, cs, rays, False, K)
        assert bad == 0
