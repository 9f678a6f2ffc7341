"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

All arithmetic on the path is IEEE binary32 under the pinned numeric contract HJ-NUM-1, so the bar is EXACT
equality of the RGBA32F accumulation buffer (tolerance 0 ulp), not BASELINE.json's per-pixel L2 < 1e-4.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import scenes
from hijiki_amd import abi, device, host

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def render(r, cs, W, H, blocks, opts=None):
    r.upload_scene(cs)
    r.create_framebuffer(W, H)
    st = r.render_blocks(blocks, opts)
    return r.read(), st


def assert_same(got, want, what):
    bad = (bits(got) != bits(want)).any(axis=-1)
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} pixels differ, max |d| = {np.nanmax(np.abs(got - want))}"


@pytest.mark.parametrize("name", ["cbox_64x64x4", "cbox_spheres_64x64x4", "cbox_cboard_96x40x3"])
def test_golden_fixtures(gpu_renderer, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    W, H, spp, seed, kind = (int(g[k]) for k in ("width", "height", "spp", "seed", "kind"))
    cs = host.Scene.synthetic(kind).compile()
    blocks = host.make_blocks(W, H, spp, seed)
    got, st = render(gpu_renderer, cs, W, H, blocks)
    assert_same(got, g["accum"], name)
    ctr = dict(zip(g["counter_names"].tolist(), g["counters"].tolist()))
    assert st["paths"] == ctr["paths"] and st["closest_rays"] == ctr["closest_calls"] and st["shadow_rays"] == ctr["shadow_calls"]
    smp = gpu_renderer.samples(blocks[0])
    assert (bits(smp) == bits(g["samples0"])).all()


def test_config1_cbox_256x256_4spp(gpu_renderer, oracle, cbox):
    """BASELINE.json configs[0]: the CPU-runnable case, whole image, exact."""
    W = H = 256
    blocks = host.make_blocks(W, H, 4, 1)
    want, ctr, _ = oracle.render_blocks(cbox, blocks, W, H)
    got, st = render(gpu_renderer, cbox, W, H, blocks)
    assert_same(got, want, "C1")
    assert st["closest_rays"] == ctr["closest_calls"] and st["shadow_rays"] == ctr["shadow_calls"]
    a, b = oracle.resolve(want), oracle.resolve(got)
    assert float(np.sqrt(np.mean((a - b) ** 2))) < 1e-4       # the bar BASELINE.json states; actual: 0


def test_render_frame_entry_point_against_the_oracle(gpu_renderer, oracle, cbox):
    """hj_render_frame - the entry point bench.py times - against the oracle: the frame's ImageBlock list is generated
    inside the library (BlockGrid::make; src/main.rs:619-682 made deterministic), the oracle renders host.make_blocks' list.
    Whole frame bit for bit; then three ranks (rank / world as bench.py --gpus N passes them), each against the oracle on
    ITS block list (hj_block_owner), and their float64 sum against the frame."""
    from hijiki_amd import dist as hjdist
    W = H = 384
    spp, seed = 5, 7
    r = gpu_renderer
    r.upload_scene(cbox)
    r.create_framebuffer(W, H)
    st = r.render_frame(spp, seed)
    all_blocks = host.make_blocks(W, H, spp, seed)
    want, ctr, _ = oracle.render_blocks(cbox, all_blocks, W, H)
    assert_same(r.read(), want, "hj_render_frame")
    assert st["paths"] == W * H * spp and st["closest_rays"] == ctr["closest_calls"] and st["shadow_rays"] == ctr["shadow_calls"]
    per, world = host.blocks_per_pass(W, H), 3
    total, owned = np.zeros((H, W, 4), np.float64), 0
    for rank in range(world):
        mine = [all_blocks[p * per + j] for p in range(spp) for j in hjdist.owned_blocks(W, H, rank, world, p)]
        owned += len(mine)
        want_r, _, _ = oracle.render_blocks(cbox, (abi.ImageBlock * len(mine))(*mine), W, H)
        r.clear()
        r.render_frame(spp, seed, rank=rank, world=world)
        got_r = r.read()
        assert_same(got_r, want_r, f"rank {rank} of {world}")
        total += got_r
    assert owned == len(all_blocks)
    np.testing.assert_allclose(total, want, rtol=5e-6, atol=1e-6)


def test_config3_scene_512x512x16_against_the_oracle(gpu_renderer, oracle, cbox_spheres):
    """BASELINE.json configs[2]'s scene at a size the oracle takes seconds for: 4.2 M paths through the mirror and the
    dielectric sphere (long specular chains, total internal reflection, 5-way material sort), bit for bit."""
    W = H = 512
    blocks = host.make_blocks(W, H, 16, 3)
    want, ctr, _ = oracle.render_blocks(cbox_spheres, blocks, W, H)
    got, st = render(gpu_renderer, cbox_spheres, W, H, blocks)
    assert_same(got, want, "C3 scene 512x512x16")
    assert st["closest_rays"] == ctr["closest_calls"] and st["shadow_rays"] == ctr["shadow_calls"] and st["hits"] == ctr["hits"]
    assert ctr["closest_calls"] > 3.3 * ctr["paths"]


def test_divergent_materials_and_many_bounces(gpu_renderer, oracle, cbox_spheres):
    """configs[2] shape (mirror + dielectric spheres): long specular chains, 5-way material sort."""
    W, H = 256, 128
    blocks = host.make_blocks(W, H, 6, 11)
    want, ctr, _ = oracle.render_blocks(cbox_spheres, blocks, W, H)
    got, st = render(gpu_renderer, cbox_spheres, W, H, blocks)
    assert_same(got, want, "spheres")
    assert ctr["sphere_tests"] > 0 and st["batches"] == 1


def test_tinted_dielectric_extinction(gpu_renderer, oracle):
    s = host.Scene.synthetic(host.SYNTH_CBOX, mesh_triangles=320)
    m = s.add_dielectric(1.5, extinction=(0.5, 1.5, 3.0))       # DielectricMaterial::tinted (src/main.rs:135-139)
    s.add_sphere((0.3, 1.0, 0.2), 0.3, m)
    cs = s.compile()
    W = H = 128
    blocks = host.make_blocks(W, H, 8, 5)
    want, _, _ = oracle.render_blocks(cs, blocks, W, H)
    got, _ = render(gpu_renderer, cs, W, H, blocks)
    assert_same(got, want, "tinted")


def test_quads_and_emissive_sphere(gpu_renderer, oracle):
    """Shape-API completeness: quad shapes (never produced by the OBJ path) and a sphere light."""
    s = host.Scene()
    s.set_camera_cbox()
    white, red = s.add_diffuse((0.7, 0.7, 0.7)), s.add_diffuse((0.6, 0.1, 0.1))
    lamp = s.add_emissive((20, 18, 15))
    cb = s.add_diffuse_cboard((0.9, 0.9, 0.2), 0.13, (0.1, 0.2, 0.8), 0.21)
    s.add_quad((-1, 0, 1), (2, 0, 0), (0, 0, -2), white)        # floor, normal +y
    s.add_quad((-1, 0, -1), (2, 0, 0), (0, 1.6, 0), red)        # back wall, normal +z
    s.add_quad((-1, 0, 1), (0, 0, -2), (0, 1.6, 0), cb)         # left wall (quad uv = hit uv)
    s.add_sphere((0.2, 1.3, 0.0), 0.15, lamp)
    s.add_sphere((-0.3, 0.3, 0.2), 0.3, cb)
    s.add_sphere((0.5, 0.25, 0.4), 0.25, s.add_mirror())
    cs = s.compile()
    W, H = 192, 128
    blocks = host.make_blocks(W, H, 5, 21)
    want, ctr, _ = oracle.render_blocks(cs, blocks, W, H)
    got, _ = render(gpu_renderer, cs, W, H, blocks)
    assert ctr["quad_tests"] > 0 and ctr["sphere_tests"] > 0
    assert_same(got, want, "quads")


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_scenes_bit_exact(gpu_renderer, oracle, seed):
    """Randomised scenes through the whole API: random mixes of triangles (a small random soup with random shading
    normals), spheres and quads, all five material kinds (dielectrics with and without extinction, several lights of
    different shape kinds), random camera; GPU accumulation buffer == oracle, bit for bit."""
    cs = scenes.random_scene(seed)
    W, H = 160, 96
    blocks = host.make_blocks(W, H, 3, seed)
    want, ctr, _ = oracle.render_blocks(cs, blocks, W, H)
    got, st = render(gpu_renderer, cs, W, H, blocks)
    assert st["closest_rays"] == ctr["closest_calls"] and st["shadow_rays"] == ctr["shadow_calls"]
    assert ctr["tri_tests"] > 0 and ctr["sphere_tests"] > 0 and ctr["quad_tests"] > 0
    assert_same(got, want, f"random scene {seed}")


def test_split_kernel_mode_is_identical(gpu_renderer, oracle, cbox_spheres):
    """HJ_RENDER_SPLIT_KERNELS (one launch per stage per bounce, the profiling path) == the fused persistent kernel."""
    W, H = 256, 128
    blocks = host.make_blocks(W, H, 3, 17)
    want, ctr, _ = oracle.render_blocks(cbox_spheres, blocks, W, H)
    o = device.default_opts()
    o.flags = abi.RENDER_SPLIT_KERNELS | abi.RENDER_TIME_KERNELS
    got, st = render(gpu_renderer, cbox_spheres, W, H, blocks, o)
    assert_same(got, want, "split kernels")
    assert st["closest_rays"] == ctr["closest_calls"] and st["closest_launches"] == st["bounce_rounds"] > 10
    o.flags = abi.RENDER_TIME_KERNELS
    got, st = render(gpu_renderer, cbox_spheres, W, H, blocks, o)
    assert_same(got, want, "fused kernel")
    assert st["path_launches"] == 1 and st["path_ms"] > 0 and st["shadow_rays"] == ctr["shadow_calls"]


def test_linear_scan_mode(gpu_renderer, oracle, cbox_small):
    """USE_BVH == 0, the reference CLI's default (src/main.rs:1432-1434, scene.glsl:134-158)."""
    W, H = 128, 128
    blocks = host.make_blocks(W, H, 2, 8)
    o = device.default_opts()
    o.use_bvh = 0
    want, _, _ = oracle.render_blocks(cbox_small, blocks, W, H, opts=o)
    got, _ = render(gpu_renderer, cbox_small, W, H, blocks, o)
    assert_same(got, want, "linear scan")


def test_ragged_image_and_small_batches(gpu_renderer, oracle, cbox_small):
    """Image size not a multiple of 128 (edge blocks 72 x 8), and a batch size that cuts passes in pieces."""
    W, H = 200, 136
    blocks = host.make_blocks(W, H, 3, 2)
    want, _, _ = oracle.render_blocks(cbox_small, blocks, W, H)
    for batch in (0, 1, 5):
        o = device.default_opts()
        o.batch_blocks = batch
        got, st = render(gpu_renderer, cbox_small, W, H, blocks, o)
        assert_same(got, want, f"ragged batch={batch}")
        assert st["paths"] == W * H * 3


def test_bounce_limit_and_roulette_options(gpu_renderer, oracle, cbox_small):
    W = H = 128
    blocks = host.make_blocks(W, H, 2, 3)
    for mb, rr in ((1, 4), (3, 1), (6, 0)):
        o = device.default_opts()
        o.max_bounces, o.rr_start = mb, rr
        want, _, _ = oracle.render_blocks(cbox_small, blocks, W, H, opts=o)
        got, _ = render(gpu_renderer, cbox_small, W, H, blocks, o)
        assert_same(got, want, f"max_bounces={mb} rr_start={rr}")


def test_custom_blocks_any_order_and_overlap(gpu_renderer, oracle, cbox_small):
    """hj_render_blocks takes ANY block list (as Renderer::render does): odd sizes, overlaps, seed 61 (RNG fixed point)."""
    W, H = 256, 256
    mk = lambda i, seed, ox, oy, dx, dy, off: abi.ImageBlock(id=i, seed=seed, origin=(ox, oy), dimension=(dx, dy),
                                                            original_dimension=(W, H), sample_offset=off)
    blocks = (abi.ImageBlock * 5)(mk(0, 61, 0, 0, 128, 128, (0.5, 0.5)), mk(1, 7, 100, 60, 64, 100, (0.1, 0.9)),
                                  mk(2, 8, 100, 60, 64, 100, (0.7, 0.2)), mk(3, 9, 128, 128, 128, 128, (0.0, 0.0)),
                                  mk(4, 10, 250, 250, 6, 6, (0.99, 0.99)))
    want, _, _ = oracle.render_blocks(cbox_small, blocks, W, H)
    got, _ = render(gpu_renderer, cbox_small, W, H, blocks)
    assert_same(got, want, "custom blocks")


def test_traversal_probe_matches_oracle(gpu_renderer, oracle, cbox):
    g = np.load(os.path.join(GOLD, "cbox_rays.npz"))
    gpu_renderer.upload_scene(cbox)
    ids, t, u, v = gpu_renderer.trace(g["rays"])
    assert (ids == g["ids"]).all() and (bits(t) == bits(g["t"])).all()
    assert (bits(u) == bits(g["u"])).all() and (bits(v) == bits(g["v"])).all()
    # random rays, incl. shadow-style windows
    r = np.random.default_rng(5)
    n = 20000
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = np.stack([r.uniform(-0.95, 0.95, n), r.uniform(0.05, 1.5, n), r.uniform(-1.0, 0.95, n)], 1)
    d = r.normal(size=(n, 3))
    rays[:, 3:6] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 6], rays[:, 7] = 2e-4, r.uniform(0.1, 3.0, n)
    oi, ot, ou, ov = oracle.intersect(cbox, rays)
    gi, gt, gu, gv = gpu_renderer.trace(rays)
    assert (gi == oi).all() and (bits(gt) == bits(ot)).all() and (bits(gu) == bits(ou)).all() and (bits(gv) == bits(ov)).all()
    # any-hit (the shadow kernel's walk) is boolean-equivalent to the reference's closest-hit shadow query
    ai, *_ = gpu_renderer.trace(rays, any_hit=True)
    assert ((ai >= 0) == (oi >= 0)).all()


def test_uploaded_tree_with_inconsistent_boxes(gpu_renderer, oracle):
    """hj_scene_upload takes ANY skip-link tree.  Shrinking inner boxes so that their children stick out breaks the
    containment that the device re-layout (dropping redundant inner nodes) relies on: those nodes must be kept, and
    the walk must still be the reference's, box for box."""
    cs = host.Scene.synthetic(host.SYNTH_CBOX, mesh_triangles=1280).compile()
    nodes, f = cs.bvh, cs.bvh_f32
    inner = np.nonzero(nodes[:, 3] == 0xFFFFFFFF)[0]
    rng = np.random.default_rng(5)
    for i in rng.choice(inner[1:], size=len(inner) // 3, replace=False):
        c = 0.5 * (f[i, 0:3] + f[i, 4:7])
        f[i, 0:3] = c + (f[i, 0:3] - c) * 0.8
        f[i, 4:7] = c + (f[i, 4:7] - c) * 0.8
    W = H = 128
    blocks = host.make_blocks(W, H, 3, 11)
    want, _, _ = oracle.render_blocks(cs, blocks, W, H)
    got, _ = render(gpu_renderer, cs, W, H, blocks)
    assert_same(got, want, "shrunk inner boxes")


@pytest.fixture
def device_relayout(monkeypatch):
    """hj_scene_upload's re-layout on the device (api/scene_relayout.hip) forced for every tree, not only the large ones."""
    monkeypatch.setenv("HJ_UPLOAD_DEVICE", "1")


def test_device_relayout_matches_the_oracle(gpu_renderer, oracle, cbox, cbox_spheres, device_relayout):
    """The kernels' tree derived ON THE DEVICE (collapse by levels, pair nodes, hot-first order, sibling groups or pre-order):
    every scene kind bit for bit against the oracle, with and without pair nodes, both node orders, a tree with shrunk inner
    boxes (still a tree: device path, the collapse must keep those nodes); a skip-link array that is NOT a tree is refused."""
    W = H = 128
    blocks = host.make_blocks(W, H, 2, 21)
    for name, cs in (("cbox", cbox), ("spheres", cbox_spheres), ("rich", scenes.rich_scene()), ("random", scenes.random_scene(3)),
                     ("mesh 20k", host.Scene.synthetic(host.SYNTH_CBOX_MESH, mesh_triangles=20000).compile())):
        want, ctr, _ = oracle.render_blocks(cs, blocks, W, H)
        got, st = render(gpu_renderer, cs, W, H, blocks)
        assert_same(got, want, f"device re-layout, {name}")
        assert st["closest_rays"] == ctr["closest_calls"] and st["shadow_rays"] == ctr["shadow_calls"]
    want, _, _ = oracle.render_blocks(cbox, blocks, W, H)
    for env in ({"HJ_PAIR_LEAVES": "0"}, {"HJ_NODE_ORDER": "1"}, {"HJ_NODE_ORDER": "0", "HJ_COLLAPSE_PCT": "0"}, {"HJ_COLLAPSE_PCT": "1000"}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            got, _ = render(gpu_renderer, cbox, W, H, blocks)
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        assert_same(got, want, f"device re-layout with {env}")
    cs = host.Scene.synthetic(host.SYNTH_CBOX, mesh_triangles=1280).compile()
    f = cs.bvh_f32
    inner = np.nonzero(cs.bvh[:, 3] == 0xFFFFFFFF)[0]
    for i in np.random.default_rng(5).choice(inner[1:], size=len(inner) // 3, replace=False):
        c = 0.5 * (f[i, 0:3] + f[i, 4:7])
        f[i, 0:3], f[i, 4:7] = c + (f[i, 0:3] - c) * 0.8, c + (f[i, 4:7] - c) * 0.8
    want, _, _ = oracle.render_blocks(cs, blocks, W, H)
    assert_same(render(gpu_renderer, cs, W, H, blocks)[0], want, "device re-layout, shrunk inner boxes")
    # not a tree: hj_scene_upload refuses links that are not a pre-order skip-link tree (boxes may be anything, links may not:
    # camera packets, the collapse and the node order lean on "a subtree is left only through its root's exit")
    cs = host.Scene.synthetic(host.SYNTH_CBOX, mesh_triangles=320).compile()
    nodes = cs.bvh
    i = int(np.nonzero(nodes[:, 3] == 0xFFFFFFFF)[0][5])
    r = int(nodes[i + 1, 7])
    assert int(nodes[r, 7]) < len(nodes) - 1
    nodes[r, 7] = int(nodes[r, 7]) + 1
    with pytest.raises(abi.HijikiError, match="not a pre-order skip-link tree"):
        gpu_renderer.upload_scene(cs)
    gpu_renderer.upload_scene(cbox)


def _check_skip_link_tree(nodes, boxes_of_shapes):
    """The invariants of the reference's flattened tree (src/main.rs:203-231) on an (N, 8) uint32 array."""
    N = len(nodes)
    f = nodes.view(np.float32)
    inner = nodes[:, 3] == 0xFFFFFFFF
    shapes = nodes[~inner, 3]
    assert N == 2 * len(shapes) - 1 and sorted(shapes.tolist()) == list(range(len(shapes)))   # every shape in ONE leaf
    exits = nodes[:, 7].astype(np.int64)
    assert (exits > np.arange(N)).all()                                   # the walk moves forward
    assert exits[0] == max(1000000, N)                                    # root exit (src/main.rs:231)
    size = np.zeros(N, np.int64)
    for i in range(N - 1, -1, -1):                                        # pre-order: left child = next record
        if not inner[i]:
            size[i] = 1
            continue
        l = i + 1
        r = l + size[l]
        assert r < N and nodes[l, 7] == r                                 # exit of the left child = its sibling
        size[i] = 1 + size[l] + size[r]
        end = i + size[i]
        assert nodes[r, 7] == nodes[i, 7] and (exits[i] == end if end < N else exits[i] == exits[0])
        for c in (l, r):                                                  # a node's box = the bounds of its subtree
            assert (f[c, 0:3] >= f[i, 0:3]).all() and (f[c, 4:7] <= f[i, 4:7]).all()
        assert (np.minimum(f[l, 0:3], f[r, 0:3]) == f[i, 0:3]).all() and (np.maximum(f[l, 4:7], f[r, 4:7]) == f[i, 4:7]).all()
    assert size[0] == N
    lo, hi = boxes_of_shapes
    leaf = np.nonzero(~inner)[0]
    assert (f[leaf, 0:3] == lo[nodes[leaf, 3]]).all() and (f[leaf, 4:7] == hi[nodes[leaf, 3]]).all()


def _shape_boxes(cs):
    tri = cs.vertices[:, 0:3][cs.triangles]                               # (T, 3, 3)
    lo = [cs.spheres[:, 0:3] - cs.spheres[:, 3:4]] if len(cs.spheres) else []
    hi = [cs.spheres[:, 0:3] + cs.spheres[:, 3:4]] if len(cs.spheres) else []
    if len(cs.quads):
        o, e1, e2 = cs.quads[:, 0:3], cs.quads[:, 4:7], cs.quads[:, 8:11]
        corners = np.stack([o, o + e1, o + e2, (o + e1) + e2], axis=1)
        lo.append(corners.min(axis=1)); hi.append(corners.max(axis=1))
    lo.append(tri.min(axis=1)); hi.append(tri.max(axis=1))
    return np.concatenate(lo).astype(np.float32), np.concatenate(hi).astype(np.float32)


@pytest.mark.parametrize("kind", [host.SYNTH_CBOX, host.SYNTH_CBOX_SPHERES])
def test_device_built_bvh(gpu_renderer, oracle, kind):
    """hj_build_bvh_device: the LBVH comes back in the reference's flattened format (every invariant of
    src/main.rs:203-231 checked on the host), renders bit-identically on GPU and oracle, and gives the image of the
    host-built SAH tree up to epsilon-ties (the topology only decides between hits closer than 1e-4)."""
    cs = host.Scene.synthetic(kind, mesh_triangles=1280).compile()
    sah_nodes = cs.bvh.copy()
    nodes = gpu_renderer.build_bvh(cs)
    _check_skip_link_tree(nodes, _shape_boxes(cs))
    assert (gpu_renderer.build_bvh(cs) == nodes).all()                    # deterministic
    W, H = 128, 96
    blocks = host.make_blocks(W, H, 4, 17)
    ref, _, _ = oracle.render_blocks(cs, blocks, W, H)                    # SAH tree
    cs.set_bvh(nodes)
    assert (cs.bvh == nodes).all()
    want, ctr, _ = oracle.render_blocks(cs, blocks, W, H)
    got, _ = render(gpu_renderer, cs, W, H, blocks)
    assert_same(got, want, "device-built tree")
    rgb = lambda a: a[..., :3] / a[..., 3:4]
    differ = (np.abs(rgb(want) - rgb(ref)) > 1e-6).any(axis=-1).mean()
    assert differ < 0.02, differ                                          # a tie flips a whole path: rare pixels only
    # the build ends with the ray-voted child order (kernels/hj_vote.h); without it (HJ_LBVH_VOTE_PATHS=0): the same leaves and
    # boxes in the order by shape count, more nodes per ray
    os.environ["HJ_LBVH_VOTE_PATHS"] = "0"
    try:
        plain = gpu_renderer.build_bvh(cs)
    finally:
        del os.environ["HJ_LBVH_VOTE_PATHS"]
    _check_skip_link_tree(plain, _shape_boxes(cs))
    assert _record_multiset(plain) == _record_multiset(nodes)
    cs.set_bvh(plain)
    want0, ctr0, _ = oracle.render_blocks(cs, blocks, W, H)
    got0, _ = render(gpu_renderer, cs, W, H, blocks)
    assert_same(got0, want0, "device-built tree, no vote")
    assert ctr["nodes"] + ctr["shadow_nodes"] < ctr0["nodes"] + ctr0["shadow_nodes"], (ctr, ctr0)
    # ... and with compile's tree passes run on it (hjh_compiled_tune_bvh: reinsertion + the host's vote): still a valid tree over
    # the same shapes, cheaper to walk than the unvoted one, bit-identical on GPU and oracle
    cs.tune_bvh(reinsert_passes=1, vote_paths=20000)
    _check_skip_link_tree(cs.bvh, _shape_boxes(cs))
    want2, ctr2, _ = oracle.render_blocks(cs, blocks, W, H)
    got2, _ = render(gpu_renderer, cs, W, H, blocks)
    assert_same(got2, want2, "device-built tree, tuned")
    assert ctr2["nodes"] < ctr0["nodes"]
    cs.set_bvh(sah_nodes)


@pytest.mark.parametrize("kind,tris", [(host.SYNTH_CBOX_SPHERES, 1280), (host.SYNTH_CBOX_MESH, 150000)])
def test_device_tree_stays_on_the_device(oracle, kind, tris):
    """The device route without the host in the middle (VERDICT r5 task 6a): shapes compiled WITHOUT a tree, hj_build_bvh_device
    with out_nodes == NULL, hj_scene_upload with scene->bvh == NULL - the tree never visits the host.  The frame is the one of
    the same tree handed over through the host, bit for bit, and the oracle's on the read-back copy (hj_bvh_device_read); the
    small scene takes the grid's route back (the light-shaft grid is host code), the large one none at all.  Misuse is refused."""
    gpu_renderer = device.Renderer(0)                                        # (a context of its own: no tree of an earlier test on it)
    scene = host.Scene.synthetic(kind, mesh_triangles=tris)
    cs = scene.compile(with_tree=False)
    assert cs.desc.num_bvh_nodes == 0 and not cs.desc.bvh
    with pytest.raises(abi.HijikiError):                                     # no tree anywhere yet
        gpu_renderer.upload_scene(cs, device_tree=True)
    n = gpu_renderer.build_bvh(cs, keep_on_device=True)
    assert n == 2 * cs.num_shapes - 1
    nodes = gpu_renderer.read_device_bvh()
    _check_skip_link_tree(nodes, _shape_boxes(cs))
    other = host.Scene.synthetic(kind, mesh_triangles=tris // 2).compile(with_tree=False)
    with pytest.raises(abi.HijikiError):                                     # a tree over OTHER shape arrays
        gpu_renderer.upload_scene(other, device_tree=True)
    moved = host.Scene.synthetic(kind, mesh_triangles=tris, gen_seed=5).compile(with_tree=False)
    if moved.num_shapes == cs.num_shapes:                                    # ... also when only their CONTENT differs (sampled fingerprint)
        with pytest.raises(abi.HijikiError):
            gpu_renderer.upload_scene(moved, device_tree=True)
    W, H = 160, 96
    blocks = host.make_blocks(W, H, 3, 23)
    gpu_renderer.upload_scene(cs, device_tree=True)
    gpu_renderer.create_framebuffer(W, H)
    st = gpu_renderer.render_blocks(blocks)
    got = gpu_renderer.read().copy()
    with pytest.raises(abi.HijikiError):                                     # consumed: the next upload needs a new build
        gpu_renderer.upload_scene(cs, device_tree=True)
    cs.set_bvh(nodes)                                                        # the same tree through the host
    want, ctr, _ = oracle.render_blocks(cs, blocks, W, H)
    assert_same(got, want, "tree that stayed on the device")
    assert st["closest_rays"] == ctr["closest_calls"] and st["shadow_rays"] == ctr["shadow_calls"] and st["hits"] == ctr["hits"]
    through_host, st2 = render(gpu_renderer, cs, W, H, blocks)
    assert_same(through_host, got, "device route against the host route")
    assert st2["shadow_rays_proven_free"] == st["shadow_rays_proven_free"]    # (the same light grid, or none, on both routes)
    gpu_renderer.close()


def _record_multiset(nodes):
    """The records of a flattened tree without their links: (box, shape word), sorted."""
    n = np.asarray(nodes, np.uint32).reshape(-1, 8)
    return sorted(map(tuple, n[:, :7].tolist()))


@pytest.mark.parametrize("kind,tris", [(host.SYNTH_CBOX_SPHERES, 0), (host.SYNTH_CBOX_MESH, 20000)])
def test_device_vote_keeps_the_tree_and_the_image(gpu_renderer, oracle, kind, tris):
    """hj_tune_bvh_device: the child order of an installed tree voted by sampled rays on the device.  What comes back is the same
    tree - same boxes, same leaves, children exchanged - as a valid pre-order skip-link array; the HIP path and the oracle walk it
    to the same bits; it is as cheap to walk as the host's vote makes it (hjh_compiled_tune_bvh) and cheaper than before."""
    cs = host.Scene.synthetic(kind, mesh_triangles=tris).compile()
    sah_nodes = cs.bvh.copy()
    cs.tune_bvh(reinsert_passes=0, vote_paths=0)                          # (identity: the passes are off)
    assert (cs.bvh == sah_nodes).all()
    # an order the vote has something to say about: every node's children exchanged where the right one has more shapes
    worst = _order_children(sah_nodes, larger_first=True)
    _check_skip_link_tree(worst, _shape_boxes(cs))
    cs.set_bvh(worst)
    W, H = 128, 96
    blocks = host.make_blocks(W, H, 2, 5)
    _, ctr_w, _ = oracle.render_blocks(cs, blocks, W, H)
    assert (gpu_renderer.tune_bvh_device(cs, vote_paths=0) == worst).all()
    voted = gpu_renderer.tune_bvh_device(cs, vote_paths=30000)
    assert (gpu_renderer.tune_bvh_device(cs, vote_paths=30000) == voted).all()          # deterministic (integer votes)
    _check_skip_link_tree(voted, _shape_boxes(cs))
    assert _record_multiset(voted) == _record_multiset(worst)
    cs.set_bvh(voted)
    want, ctr_v, _ = oracle.render_blocks(cs, blocks, W, H)
    got, _ = render(gpu_renderer, cs, W, H, blocks)
    assert_same(got, want, "device-voted tree")
    cs.set_bvh(worst)
    cs.tune_bvh(reinsert_passes=0, vote_paths=30000)                      # the host's vote from the same starting point
    _, ctr_h, _ = oracle.render_blocks(cs, blocks, W, H)
    total = lambda c: c["nodes"] + c["shadow_nodes"]
    assert total(ctr_v) < 0.97 * total(ctr_w), (total(ctr_v), total(ctr_w))
    assert total(ctr_v) < 1.02 * total(ctr_h), (total(ctr_v), total(ctr_h))
    # links that are not a tree's are refused (the same check as hj_scene_upload's)
    cs.set_bvh(sah_nodes)
    nodes = cs.bvh                                                        # (a view of the installed array)
    i = int(np.nonzero(nodes[:, 3] == 0xFFFFFFFF)[0][5])
    r = int(nodes[i + 1, 7])
    assert int(nodes[r, 7]) < len(nodes) - 1
    nodes[r, 7] = int(nodes[r, 7]) + 1
    with pytest.raises(abi.HijikiError, match="not a pre-order skip-link tree"):
        gpu_renderer.tune_bvh_device(cs)
    cs.set_bvh(sah_nodes)


def test_device_vote_on_a_chain_deeper_than_its_level_loop(gpu_renderer, oracle):
    """A tree that is one long right spine (9000 spheres, every inner node = one leaf + the rest): deeper than the 8192 levels the
    exchange's top-down pass walks.  hj_tune_bvh_device says HJ_ERR_UNSUPPORTED, nothing hangs, the context stays usable and the
    chain itself uploads and renders like the oracle."""
    n = 9000
    s = host.Scene()
    s.set_camera_cbox()
    m, e = s.add_diffuse((0.6, 0.6, 0.6)), s.add_emissive((9, 9, 9))
    rng = np.random.default_rng(11)
    for k in range(n):
        s.add_sphere(tuple(rng.uniform(-0.9, 0.9, 3) + (0, 1, 0)), 0.01 + 0.02 * rng.random(), e if k % 600 == 0 else m)
    cs = s.compile()
    boxes = _shape_boxes(cs)                                              # (lo, hi) per shape
    N = 2 * n - 1
    chain = np.zeros((N, 8), np.uint32)
    f = chain.view(np.float32)
    lo, hi = boxes
    suf_lo, suf_hi = np.minimum.accumulate(lo[::-1])[::-1], np.maximum.accumulate(hi[::-1])[::-1]
    for k in range(n - 1):
        i = 2 * k
        f[i, 0:3], f[i, 4:7] = suf_lo[k], suf_hi[k]
        chain[i, 3], chain[i, 7] = 0xFFFFFFFF, max(N, abi.BVH_ROOT_EXIT)
        f[i + 1, 0:3], f[i + 1, 4:7] = lo[k], hi[k]
        chain[i + 1, 3], chain[i + 1, 7] = k, i + 2
    f[N - 1, 0:3], f[N - 1, 4:7] = lo[n - 1], hi[n - 1]
    chain[N - 1, 3], chain[N - 1, 7] = n - 1, max(N, abi.BVH_ROOT_EXIT)
    _check_skip_link_tree(chain, boxes)
    cs.set_bvh(chain)
    with pytest.raises(abi.HijikiError) as err:
        gpu_renderer.tune_bvh_device(cs, vote_paths=2000)
    assert err.value.status == abi.HJ_ERR_UNSUPPORTED and "deeper" in str(err.value)
    W, H = 64, 48
    blocks = host.make_blocks(W, H, 1, 2)
    want, _, _ = oracle.render_blocks(cs, blocks, W, H)
    got, _ = render(gpu_renderer, cs, W, H, blocks)
    assert_same(got, want, "chain tree")


def _sphere_cluster(n, radius_scale, seed=11):
    s = host.Scene()
    s.set_camera_cbox()
    m, e = s.add_diffuse((0.6, 0.6, 0.6)), s.add_emissive((9, 9, 9))
    rng = np.random.default_rng(seed)
    for _ in range(n):
        s.add_sphere(tuple(rng.uniform(-0.9, 0.9, 3) + (0, 1, 0)), radius_scale * (0.01 + 0.02 * rng.random()), m)
    s.add_quad((-0.3, 1.98, -0.3), (0.6, 0, 0), (0, 0, 0.6), e)
    return s.compile()


@pytest.mark.parametrize("n,scale,spp", [(20, 5.0, 16), (100, 5.0, 4), (600, 1.0, 2)])
def test_sphere_clusters_and_directions_that_are_not_unit_vectors(gpu_renderer, oracle, n, scale, spp):
    """Many small overlapping diffuse spheres.  The reference never re-normalises a path's direction: a hit point is o + t d, the
    next normal (p - c) / r, the next direction built on it, and among small spheres the error of one step is amplified at the next -
    a few bounces deep directions have length 1.3 or 3, for which sphere.glsl:18-41 (a = 1 assumed) no longer finds the line's
    intersection with the sphere.  A padded box in front of a sphere leaf (round 5's guard nodes) culled hits the reference
    accepts: 3 of 56 000 rays of the first scene.  Guards are for triangles and quads only now; here the frame, the counters and
    EVERY ray the oracle traced - replayed through the uploaded tree - must agree."""
    cs = _sphere_cluster(n, scale)
    W, H = 64, 48
    blocks = host.make_blocks(W, H, spp, 2)
    want, ctr, _ = oracle.render_blocks(cs, blocks, W, H)
    got, st = render(gpu_renderer, cs, W, H, blocks)
    assert_same(got, want, f"{n} spheres")
    assert st["closest_rays"] == ctr["closest_calls"] and st["shadow_rays"] == ctr["shadow_calls"] and st["hits"] == ctr["hits"]
    log = oracle.logged_rays(cs, blocks)
    assert len(log) == ctr["closest_calls"] + ctr["shadow_calls"]
    length = np.linalg.norm(log[:, 3:6].astype(np.float64), axis=1)
    if n == 20:
        assert (np.abs(length - 1.0) > 0.05).any()                        # (the premise: such rays exist in this frame)
    ids, t, _, _ = gpu_renderer.trace(np.ascontiguousarray(log[:, 0:8]))
    assert (ids == log[:, 9].astype(np.int32)).all(), np.nonzero(ids != log[:, 9].astype(np.int32))[0][:10]


@pytest.mark.parametrize("seed", [77, 3, 14])
def test_cluster_scenes_and_rays_not_in_general_position(gpu_renderer, oracle, seed):
    """tests/scenes.py random_cluster_scene: hundreds of small spheres of every material, small triangles, three lights - on the
    compiled tree and on the device-built one.  Seed 77 holds the ray that showed what guard nodes (and, in principle, the collapse)
    had overlooked: a direction with a component of exactly 0 (a cosine sample on an axis-aligned quad: sin(2 pi u) = 0).  The slab
    test then forms inf - inf and drops the NaN in its min / max; its answer depends on the SIGNS of a box's bounds - the wall's
    own box [0, 2] passes where the guard's [-1e-4, 2] fails - so nothing that replaces a box by another is exact for such rays.
    They walk a second copy of the tree, the reference's own (kernels/hj_intersect.h general_position, DeviceScene::root2)."""
    cs = scenes.random_cluster_scene(seed)
    W, H = 160, 96
    blocks = host.make_blocks(W, H, 3, seed)
    compiled = cs.bvh.copy()
    for tree in ("compiled", "device-built"):
        if tree == "device-built":
            cs.set_bvh(gpu_renderer.build_bvh(cs))
        want, ctr, _ = oracle.render_blocks(cs, blocks, W, H)
        got, st = render(gpu_renderer, cs, W, H, blocks)
        assert_same(got, want, f"cluster scene {seed}, {tree} tree")
        assert st["closest_rays"] == ctr["closest_calls"] and st["shadow_rays"] == ctr["shadow_calls"] and st["hits"] == ctr["hits"]
    cs.set_bvh(compiled)
    gpu_renderer.upload_scene(cs)
    if seed == 77:                                                        # the ray itself (bits as the oracle logged them)
        ray = np.array([[-2.6812655e-01, 3.6761138e-01, -1.1999997e+00, -6.9234103e-01, 0.0, 7.2157043e-01, 1.9999999e-04, np.inf]], np.float32)
        oi, ot, _, _ = oracle.intersect(cs, ray)
        gi, gt, _, _ = gpu_renderer.trace(ray)
        assert oi[0] >= 0 and gi[0] == oi[0] and bits(gt)[0] == bits(ot)[0]
    # rays with zero components (+0, -0, two at once), origins on and off the scene's axis-aligned planes, open and closed intervals
    rng = np.random.default_rng(seed)
    n = 200000
    rays = np.zeros((n, 8), np.float32)
    planes = np.array([-1.2, -1.0, 0.0, 0.02, 1.2, 1.99, 2.0], np.float32)
    rays[:, 0:3] = np.where(rng.random((n, 3)) < 0.3, rng.choice(planes, (n, 3)), rng.uniform(-1.2, 2.0, (n, 3)))
    d = rng.normal(0, 1, (n, 3)).astype(np.float32)
    z = rng.integers(0, 3, n)
    d[np.arange(n), z] = np.where(rng.random(n) < 0.5, 0.0, -0.0)
    two = rng.random(n) < 0.2
    d[two, (z[two] + 1) % 3] = 0.0
    rays[:, 3:6] = d
    rays[:, 6], rays[:, 7] = 2e-4, np.where(rng.random(n) < 0.5, np.inf, rng.uniform(0.2, 3.0, n))
    oi, ot, _, _ = oracle.intersect(cs, rays)
    gi, gt, _, _ = gpu_renderer.trace(rays)
    ai, *_ = gpu_renderer.trace(rays, any_hit=True)
    assert (oi >= 0).sum() > 1000
    assert (gi == oi).all() and (bits(gt)[oi >= 0] == bits(ot)[oi >= 0]).all() and ((ai >= 0) == (oi >= 0)).all()


def _order_children(nodes, larger_first):
    """The flattened tree `nodes` with the children of every inner node in the order of their record counts."""
    n = np.asarray(nodes, np.uint32).reshape(-1, 8)
    N = len(n)
    out = np.zeros_like(n)

    def end(i):
        return min(int(n[i, 7]), N)

    stack = [(0, 0, int(n[0, 7]))]                                        # (old index, new position, new exit)
    while stack:
        i, pos, ex = stack.pop()
        out[pos] = n[i]
        out[pos, 7] = ex
        if n[i, 3] != 0xFFFFFFFF:
            continue
        l = i + 1
        r = int(n[l, 7])
        sl, sr = r - l, end(i) - r
        first, second, sf = (l, r, sl)
        if (sr > sl) == larger_first and sr != sl:
            first, second, sf = (r, l, sr)
        stack.append((first, pos + 1, pos + 1 + sf))
        stack.append((second, pos + 1 + sf, ex))
    return out


def _sah_cost(nodes):
    """Expected box tests per random ray, up to a constant: sum of the records' surface areas over the root's."""
    n = np.asarray(nodes).view(np.float32).reshape(-1, 8)
    d = np.maximum(n[:, 4:7].astype(np.float64) - n[:, 0:3], 0.0)             # (min, shape, max, exit)
    area = d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0]
    return area.sum() / area[0]


def test_device_built_bvh_large_and_degenerate(gpu_renderer, oracle, monkeypatch):
    """200 k triangles (deep Morton prefixes, many equal codes) and a scene whose shapes all share one centroid.  The tree
    with the clusters re-split by SAH on the device is valid, nearly as cheap to walk as the host's SAH tree (surface-area
    cost within 15 %: the host's tree also gets rotation passes, which the device path does not have; without them
    the two are within 10 %) and cheaper than the plain Morton clusters, which stay available (HJ_LBVH_SAH=0)."""
    cs = host.Scene.synthetic(host.SYNTH_CBOX_MESH, mesh_triangles=200000).compile()
    host_cost = _sah_cost(cs.bvh)
    monkeypatch.setenv("HJ_LBVH_SAH", "0")
    morton = gpu_renderer.build_bvh(cs)
    _check_skip_link_tree(morton, _shape_boxes(cs))
    monkeypatch.delenv("HJ_LBVH_SAH")
    nodes = gpu_renderer.build_bvh(cs)
    _check_skip_link_tree(nodes, _shape_boxes(cs))
    assert _sah_cost(nodes) < 1.15 * host_cost, (_sah_cost(nodes), host_cost)
    assert _sah_cost(nodes) < 0.98 * _sah_cost(morton), (_sah_cost(nodes), _sah_cost(morton))   # leaf areas are common to both
    assert (gpu_renderer.build_bvh(cs) == nodes).all()                    # deterministic (LDS atomics on ordered ints, ballot partition)
    # both re-split kernels: clusters of up to 512 leaves by one WAVE each (the default), of up to 64 by one thread each
    monkeypatch.setenv("HJ_LBVH_CLUSTER", "64")
    small = gpu_renderer.build_bvh(cs)
    _check_skip_link_tree(small, _shape_boxes(cs))
    assert _sah_cost(nodes) < 1.02 * _sah_cost(small), (_sah_cost(nodes), _sah_cost(small))     # larger SAH domains are not worse
    monkeypatch.setenv("HJ_LBVH_CLUSTER", "300")                          # (a cluster size that is no power of two)
    _check_skip_link_tree(gpu_renderer.build_bvh(cs), _shape_boxes(cs))
    monkeypatch.delenv("HJ_LBVH_CLUSTER")
    cs.set_bvh(nodes)
    W = H = 128
    blocks = host.make_blocks(W, H, 1, 3)
    want, _, _ = oracle.render_blocks(cs, blocks, W, H)
    got, _ = render(gpu_renderer, cs, W, H, blocks)
    assert_same(got, want, "device-built tree, 200k mesh")
    s = host.Scene()
    s.set_camera_cbox()
    m, e = s.add_diffuse((0.5, 0.5, 0.5)), s.add_emissive((5, 5, 5))
    for r in (0.1, 0.2, 0.3, 0.4, 0.5):
        s.add_sphere((0.0, 0.8, 0.0), r, e if r == 0.1 else m)            # concentric: identical centroids
    cs = s.compile()
    nodes = gpu_renderer.build_bvh(cs)
    _check_skip_link_tree(nodes, _shape_boxes(cs))


def test_rendering_is_deterministic_and_additive(gpu_renderer, cbox):
    """Size-independent properties at a larger size: run-to-run bitwise determinism, and passes [0,a)+[a,b)
    accumulated by two calls == one call (the framebuffer is a running sum)."""
    W = H = 512
    r = gpu_renderer
    r.upload_scene(cbox)
    r.create_framebuffer(W, H)
    r.render_frame(16, 3)
    a = r.read()
    r.clear()
    r.render_frame(16, 3, pass_begin=0, pass_end=5)
    r.render_frame(16, 3, pass_begin=5, pass_end=16)
    b = r.read()
    assert (bits(a) == bits(b)).all()
    assert np.isfinite(a).all() and (a[..., 3] > 0).all()


def test_tile_sharding_sums_to_the_full_frame(gpu_renderer, cbox):
    """N 'virtual ranks' on one device.  Static deal (all passes of a block on one rank): the sum of the per-rank
    buffers equals the 1-GPU frame exactly away from block borders and within a few ulp on the 2-pixel aprons
    (different association of the same addends).  Default deal (moves one diagonal per pass): the same addends
    in another association everywhere, so a few ulp everywhere; every block is rendered exactly once either way."""
    W = H = 384
    r = gpu_renderer
    r.upload_scene(cbox)
    r.create_framebuffer(W, H)
    st_full = r.render_frame(6, 9)
    full = r.read()
    static = device.default_opts()
    static.flags = abi.RENDER_STATIC_DEAL
    for world in (2, 3):
        for opts in (static, None):
            parts, paths = [], 0
            for rank in range(world):
                r.clear()
                paths += r.render_frame(6, 9, rank=rank, world=world, opts=opts)["paths"]
                parts.append(r.read().astype(np.float64))
            assert paths == st_full["paths"]
            total = np.sum(parts, axis=0)
            if opts is static:
                interior = np.ones((H, W), bool)
                for e in (128, 256):
                    interior[e - 2:e + 2, :] = False
                    interior[:, e - 2:e + 2] = False
                assert (total[interior].astype(np.float32) == full[interior]).all()
            np.testing.assert_allclose(total, full, rtol=3e-6, atol=1e-6)


def test_config2_full_size_properties(gpu_renderer, cbox):
    """BASELINE.json configs[1] at FULL size (cbox 1024x1024, 512 spp = 537 M paths), through size-independent
    properties: run-to-run bitwise determinism, invariance to the wavefront batch size, additivity over pass
    ranges, and 8-way tile sharding summing to the 1-GPU frame."""
    W = H = 1024
    spp = 512
    r = gpu_renderer
    r.upload_scene(cbox)
    r.create_framebuffer(W, H)
    st = r.render_frame(spp, 1)
    a = r.read()
    assert st["paths"] == W * H * spp and st["closest_rays"] > 3 * st["paths"] and st["shadow_rays"] > st["paths"]
    assert np.isfinite(a).all() and (a[..., 3] > 0).all()
    rgb = a[..., :3] / a[..., 3:4]
    assert 0.05 < rgb.mean() < 1.0 and rgb.min() >= 0
    o = device.default_opts()
    o.batch_blocks = 320                      # cuts passes in pieces: 5 blocks of the next pass ride along
    r.clear()
    r.render_frame(spp, 1, opts=o)
    assert (bits(r.read()) == bits(a)).all()
    r.clear()
    r.render_frame(spp, 1, pass_begin=0, pass_end=200)
    r.render_frame(spp, 1, pass_begin=200, pass_end=512)
    assert (bits(r.read()) == bits(a)).all()
    static = device.default_opts()
    static.flags = abi.RENDER_STATIC_DEAL
    for opts in (static, None):               # None = the default deal that rotates with the pass
        total = np.zeros((H, W, 4), np.float64)
        paths = []
        for rank in range(8):
            r.clear()
            paths.append(r.render_frame(spp, 1, rank=rank, world=8, opts=opts)["paths"])
            total += r.read()
        assert sum(paths) == st["paths"] and max(paths) == min(paths)
        if opts is static:
            interior = np.ones((H, W), bool)
            for e in range(128, 1024, 128):
                interior[e - 2:e + 2, :] = False
                interior[:, e - 2:e + 2] = False
            assert (total[interior].astype(np.float32) == a[interior]).all()
        np.testing.assert_allclose(total, a, rtol=5e-5, atol=1e-5)


def _frame_properties(r, W, H, spp, seed, split_at, virtual_ranks=0, rtol=5e-5):
    """Size-independent properties of a whole frame at a BASELINE configuration's own size: sane statistics, bitwise
    run-to-run determinism, additivity over pass ranges (the framebuffer is a running sum in block order) and, when
    asked, `virtual_ranks`-way tile sharding (each rank's share rendered in turn on this GPU) summing to the frame."""
    r.create_framebuffer(W, H)
    st = r.render_frame(spp, seed)
    a = r.read()
    assert st["paths"] == W * H * spp
    assert st["closest_rays"] >= st["paths"] and st["hits"] <= st["closest_rays"]
    assert st["unoccluded_shadow_rays"] <= st["shadow_rays"]
    assert np.isfinite(a).all() and (a[..., 3] > 0).all()
    rgb = a[..., :3] / a[..., 3:4]
    assert 0.02 < rgb.mean() < 2.0 and rgb.min() >= 0
    r.clear()
    r.render_frame(spp, seed, pass_begin=0, pass_end=split_at)
    r.render_frame(spp, seed, pass_begin=split_at, pass_end=spp)
    assert (bits(r.read()) == bits(a)).all(), "pass ranges [0,a) + [a,spp) != one call"
    if virtual_ranks:
        total = np.zeros((H, W, 4), np.float64)
        paths = 0
        for rank in range(virtual_ranks):
            r.clear()
            paths += r.render_frame(spp, seed, rank=rank, world=virtual_ranks)["paths"]
            total += r.read()
        assert paths == st["paths"]
        np.testing.assert_allclose(total, a, rtol=rtol, atol=1e-5)
    return st, a


def test_config3_full_size_properties(gpu_renderer, cbox_spheres):
    """BASELINE.json configs[2] at FULL size: cbox + mirror sphere + dielectric sphere, 1024 x 1024, 1024 spp
    (1.07 G paths; the divergent-BSDF configuration)."""
    gpu_renderer.upload_scene(cbox_spheres)
    st, _ = _frame_properties(gpu_renderer, 1024, 1024, 1024, 1, split_at=300, virtual_ranks=8)
    assert st["closest_rays"] > 3.3 * st["paths"]          # specular chains: longer paths than the diffuse box (3.05)
    gpu_renderer.create_framebuffer(64, 64)


@pytest.fixture(scope="module")
def mesh_1m():
    """BASELINE.json configs[3]'s scene: 1 000 000 triangles -> 1 999 999 nodes (> the reference's hard-coded root exit
    of 1 000 000, src/main.rs:231)."""
    cs = host.Scene.synthetic(host.SYNTH_CBOX_MESH, mesh_triangles=1000000).compile()
    assert len(cs.bvh) > 1000000 and int(cs.bvh[0, 7]) == len(cs.bvh)
    return cs


def test_config4_million_triangles_bit_exact(gpu_renderer, oracle, mesh_1m):
    """configs[3]'s scene against the oracle, SAH tree and device-built LBVH: both have more than 10^6 nodes, the
    case where the root's exit index is the node count instead of the reference's 1 000 000."""
    cs = mesh_1m
    W = H = 128
    blocks = host.make_blocks(W, H, 1, 3)
    want, ctr, _ = oracle.render_blocks(cs, blocks, W, H)
    got, st = render(gpu_renderer, cs, W, H, blocks)
    assert_same(got, want, "1M mesh, SAH tree")
    assert st["closest_rays"] == ctr["closest_calls"] and st["shadow_rays"] == ctr["shadow_calls"]
    assert ctr["nodes"] / ctr["closest_calls"] > 35          # a deep tree (48.8 node visits per closest-hit ray with round 5's tree passes, 53 before)
    sah = cs.bvh.copy()
    nodes = gpu_renderer.build_bvh(cs)
    assert len(nodes) == len(sah) and int(nodes[0, 7]) == len(nodes)
    cs.set_bvh(nodes)
    try:
        want, _, _ = oracle.render_blocks(cs, blocks, W, H)
        got, _ = render(gpu_renderer, cs, W, H, blocks)
        assert_same(got, want, "1M mesh, device-built tree")
    finally:
        cs.set_bvh(sah)


def test_config4_scene_256x256x4_against_the_oracle(gpu_renderer, oracle, mesh_1m):
    """configs[3]'s 1 M-triangle scene, 262 k paths (bounce rays deep in the tree, not only camera rays), bit for bit."""
    W = H = 256
    blocks = host.make_blocks(W, H, 4, 9)
    want, ctr, _ = oracle.render_blocks(mesh_1m, blocks, W, H)
    got, st = render(gpu_renderer, mesh_1m, W, H, blocks)
    assert_same(got, want, "1M mesh 256x256x4")
    assert st["closest_rays"] == ctr["closest_calls"] and st["shadow_rays"] == ctr["shadow_calls"] and st["hits"] == ctr["hits"]


def test_config4_full_size_properties(gpu_renderer, mesh_1m):
    """configs[3] at FULL size: 1 M triangles, 2048 x 2048, 256 spp (1.07 G paths; deep BVH traversal)."""
    gpu_renderer.upload_scene(mesh_1m)
    st, _ = _frame_properties(gpu_renderer, 2048, 2048, 256, 1, split_at=100)
    gpu_renderer.create_framebuffer(64, 64)


def test_config5_frame_size_eight_virtual_ranks(gpu_renderer, cbox):
    """configs[4]'s frame (cbox 4096 x 4096, 1024 blocks per pass, 268 MB framebuffer) at 64 of its 4096 spp, the
    8-rank tile sharding rendered rank by rank on this GPU and summed.  (The 8-GPU run itself needs the 8-GPU node.)"""
    gpu_renderer.upload_scene(cbox)
    _frame_properties(gpu_renderer, 4096, 4096, 64, 2, split_at=24, virtual_ranks=8)
    gpu_renderer.create_framebuffer(64, 64)              # release the 268 MB buffer for the tests that follow


def test_config5_full_size(gpu_renderer, cbox):
    """configs[4] at its OWN size on one GPU: cbox 4096 x 4096, 4096 spp = 2^36 camera paths (the product that overflows
    the reference's u32 ray count, src/main.rs:1491), 4.19 M ImageBlocks generated chunk by chunk.  About 21 s.  Checked:
    the 64-bit path count, a finite frame with positive weight everywhere, the statistics' consistency, and additivity
    over pass ranges on the 512-pass prefix (two calls == one call, bit for bit)."""
    W = H = 4096
    spp = 4096
    r = gpu_renderer
    r.upload_scene(cbox)
    r.create_framebuffer(W, H)
    st = r.render_frame(spp, 1)
    assert st["paths"] == 2 ** 36 == W * H * spp
    assert st["closest_rays"] > 3 * st["paths"] and st["shadow_rays"] > st["paths"] and st["hits"] <= st["closest_rays"]
    assert 0 < st["unoccluded_shadow_rays"] <= st["shadow_rays"]
    a = r.read()
    assert np.isfinite(a).all() and (a[..., 3] > 0).all()
    rgb = a[..., :3] / a[..., 3:4]
    assert 0.05 < rgb.mean() < 1.0 and rgb.min() >= 0
    del a, rgb
    r.clear()
    r.render_frame(spp, 1, pass_begin=0, pass_end=512)
    prefix = r.read()
    r.clear()
    r.render_frame(spp, 1, pass_begin=0, pass_end=200)
    r.render_frame(spp, 1, pass_begin=200, pass_end=512)
    assert (bits(r.read()) == bits(prefix)).all()
    r.create_framebuffer(64, 64)


def test_large_mesh_and_large_frame(gpu_renderer, oracle):
    """configs[3]/[4] shapes at reduced sample counts: a 200 k-triangle mesh (deep tree) bit-exact against the oracle,
    and a 4096 x 4096 frame (1024 blocks per pass, 268 MB framebuffer) sharded 8 ways."""
    cs = host.Scene.synthetic(host.SYNTH_CBOX_MESH, mesh_triangles=200000).compile()
    W = H = 256
    blocks = host.make_blocks(W, H, 1, 3)
    want, ctr, _ = oracle.render_blocks(cs, blocks, W, H)
    got, _ = render(gpu_renderer, cs, W, H, blocks)
    assert_same(got, want, "200k mesh")
    assert ctr["nodes"] / ctr["closest_calls"] > 30
    r = gpu_renderer
    cbox = host.Scene.synthetic(host.SYNTH_CBOX).compile()
    r.upload_scene(cbox)
    r.create_framebuffer(4096, 4096)
    st = r.render_frame(1, 2)
    full = r.read()
    assert st["paths"] == 4096 * 4096
    total = np.zeros_like(full, dtype=np.float64)
    for rank in range(8):
        r.clear()
        r.render_frame(1, 2, rank=rank, world=8)
        total += r.read()
    np.testing.assert_allclose(total, full, rtol=2e-6, atol=1e-6)
    r.create_framebuffer(64, 64)              # release the 268 MB buffer for the tests that follow


def test_error_paths(gpu_renderer, cbox_small):
    r = device.Renderer(0)
    blocks = host.make_blocks(128, 128, 1, 1)
    with pytest.raises(abi.HijikiError) as e:
        r.render_blocks(blocks)
    assert e.value.status == abi.HJ_ERR_STATE                    # no scene yet
    r.upload_scene(cbox_small)
    with pytest.raises(abi.HijikiError) as e:
        r.render_blocks(blocks)
    assert e.value.status == abi.HJ_ERR_STATE                    # no framebuffer yet
    r.create_framebuffer(256, 128)
    with pytest.raises(abi.HijikiError) as e:
        r.render_blocks(blocks)                                  # original_dimension != framebuffer
    assert e.value.status == abi.HJ_ERR_INVALID
    o = device.default_opts()
    o.recon_radius = 3
    with pytest.raises(abi.HijikiError) as e:
        r.render_blocks(host.make_blocks(256, 128, 1, 1), o)
    assert e.value.status == abi.HJ_ERR_UNSUPPORTED
    # a scene whose BVH could loop forever is rejected at upload
    bad = cbox_small.bvh.copy()
    bad[5, 7] = 2
    desc = abi.SceneDesc.from_buffer_copy(bytes(cbox_small.desc))
    desc.bvh = bad.ctypes.data_as(C.POINTER(abi.BvhNode))
    with pytest.raises(abi.HijikiError) as e:
        r.upload_scene(desc)
    assert e.value.status == abi.HJ_ERR_INVALID and "forward" in str(e.value)
    r.close()


def test_external_framebuffer_torch_tensor(cbox_small, oracle):
    """The RCCL path accumulates straight into a torch CUDA tensor handed over as a raw device pointer."""
    import torch
    W = H = 128
    fb = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    blocks = host.make_blocks(W, H, 2, 4)
    with device.Renderer(0) as r:
        r.upload_scene(cbox_small)
        r.create_framebuffer(W, H, external_device_ptr=fb.data_ptr())
        r.render_blocks(blocks)
    want, _, _ = oracle.render_blocks(cbox_small, blocks, W, H)
    assert (bits(fb.cpu().numpy()) == bits(want)).all()


def test_pure_c_host_example(tmp_path):
    """examples/render_cbox.c: a C99 program drives scene build -> upload -> render -> EXR through the two C ABIs."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe, out = str(tmp_path / "render_cbox"), str(tmp_path / "o.exr")
    cmd = ["gcc", "-std=c99", "-Wall", "-pedantic", "-I" + os.path.join(root, "include"),
           os.path.join(root, "examples", "render_cbox.c"), "-L" + os.path.join(root, "hijiki_amd", "lib"),
           "-lhijiki_hip", "-lhijiki_host", "-Wl,-rpath," + os.path.join(root, "hijiki_amd", "lib"),
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-o", exe]
    assert subprocess.run(cmd, capture_output=True, text=True).returncode == 0
    r = subprocess.run([exe, out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "8388608 paths" in r.stdout and os.path.getsize(out) > 512 * 512 * 12


def test_pure_c_host_frames_back_to_back(tmp_path):
    """examples/render_frames.c: a C99 host renders five frames back to back (HJ_RENDER_NO_DRAIN, hj_framebuffer_bind,
    hj_pipeline_wait) into two hipMalloc'ed buffers and compares each, bit for bit, with a blocking call's frame."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "render_frames")
    cmd = ["gcc", "-std=c99", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(root, "include"), "-I/opt/rocm/include",
           os.path.join(root, "examples", "render_frames.c"), "-L" + os.path.join(root, "hijiki_amd", "lib"),
           "-lhijiki_hip", "-lhijiki_host", "-Wl,-rpath," + os.path.join(root, "hijiki_amd", "lib"),
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-o", exe]
    c = subprocess.run(cmd, capture_output=True, text=True)
    assert c.returncode == 0, c.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "5 frames back to back: 7864320 paths" in r.stdout and "bit-identical" in r.stdout


def test_edge_inputs(gpu_renderer, oracle):
    """Empty block list, 1x1 blocks, a scene without emitters, a camera that sees nothing."""
    r = gpu_renderer
    # no emitters: NEE draws its 3 numbers and contributes nothing (the reference reads emitters[0] out of bounds)
    s = host.Scene()
    s.set_camera_cbox()
    d = s.add_diffuse((0.8, 0.7, 0.6))
    s.add_quad((-1, 0, 1), (2, 0, 0), (0, 0, -2), d)
    s.add_quad((-1, 0, -1), (2, 0, 0), (0, 1.6, 0), d)
    cs = s.compile()
    W = H = 128
    r.upload_scene(cs)
    r.create_framebuffer(W, H)
    st = r.render_blocks((abi.ImageBlock * 0)())
    assert st["paths"] == 0 and (r.read() == 0).all()
    blocks = host.make_blocks(W, H, 2, 1)
    want, ctr, _ = oracle.render_blocks(cs, blocks, W, H)
    r.render_blocks(blocks)
    got = r.read()
    assert_same(got, want, "no emitters")
    assert ctr["shadow_calls"] == 0 and (got[..., :3] == 0).all() and (got[..., 3] > 0).all()
    # 1x1 and 3x2 blocks anywhere in the image
    mk = lambda i, ox, oy, dx, dy: abi.ImageBlock(id=i, seed=100 + i, origin=(ox, oy), dimension=(dx, dy),
                                                  original_dimension=(W, H), sample_offset=(0.25, 0.75))
    tiny = (abi.ImageBlock * 3)(mk(0, 0, 0, 1, 1), mk(1, 127, 127, 1, 1), mk(2, 60, 61, 3, 2))
    want, _, _ = oracle.render_blocks(cs, tiny, W, H)
    r.clear()
    r.render_blocks(tiny)
    assert_same(r.read(), want, "tiny blocks")
    # a camera looking away from everything: every path misses at bounce 0
    s2 = host.Scene()
    s2.set_camera((0, 0.9, 5.4), (0, 1, 0, 0), 27.7)      # rotated 180 deg about +y: looks along +z, away from the box
    e = s2.add_emissive((5, 5, 5))
    s2.add_quad((-1, 0, 1), (2, 0, 0), (0, 0, -2), e)
    s2.add_quad((-1, 0, -1), (2, 0, 0), (0, 1.6, 0), e)
    cs2 = s2.compile()
    want, ctr, _ = oracle.render_blocks(cs2, blocks, W, H)
    got, st = render(r, cs2, W, H, blocks)
    assert_same(got, want, "all miss")
    assert ctr["hits"] == 0 and st["closest_rays"] == st["paths"] and st["shadow_rays"] == 0


def test_in_process_reduce_entry_point(gpu_renderer, cbox_small):
    """hj_reduce_framebuffers: with one context it is a no-op that leaves the frame intact (several GPUs are not
    available to this test; argument checking is)."""
    r = gpu_renderer
    r.upload_scene(cbox_small)
    r.create_framebuffer(128, 128)
    r.render_frame(1, 1)
    before = r.read()
    device.reduce_framebuffers([r], root=0)
    assert (bits(r.read()) == bits(before)).all()
    with pytest.raises(abi.HijikiError) as e:
        device.reduce_framebuffers([r, r], root=0)            # two contexts on one GPU
    assert e.value.status == abi.HJ_ERR_INVALID


def test_async_frame_progress_and_comm_object(gpu_renderer, cbox_small):
    """hj_render_frame_async + hj_sync == hj_render_frame bit for bit; the progress callback counts completed blocks
    up to the total; an hj_comm over one context reduces to a no-op and waits for the frame in flight."""
    r = gpu_renderer
    r.upload_scene(cbox_small)
    W, H, spp = 384, 256, 40
    r.create_framebuffer(W, H)
    r.render_frame(spp, 3)
    want = r.read()
    seen = []
    r.set_progress(lambda done, total: seen.append((done, total)), interval_blocks=16)
    try:
        r.clear()
        r.render_frame_async(spp, 3)
        with pytest.raises(abi.HijikiError) as e:
            r.render_frame_async(spp, 3)                      # one frame in flight per context
        assert e.value.status == abi.HJ_ERR_STATE
        st = r.sync()
        assert st["paths"] == W * H * spp
        assert (bits(r.read()) == bits(want)).all()
        total = 6 * spp                                       # 3 x 2 blocks per pass
        assert seen and seen[-1] == (total, total)
        assert all(a[0] < b[0] for a, b in zip(seen, seen[1:])) and all(t == total for _, t in seen)
        comm = device.Comm([r])
        r.clear()
        r.render_frame_async(spp, 3)
        comm.reduce(0)                                        # joins the frame in flight
        assert (bits(r.read()) == bits(want)).all()
        assert r.sync()["paths"] in (0, W * H * spp)          # already joined by the reduce: nothing pending
        comm.close()
    finally:
        r.set_progress(None)


@pytest.mark.skipif(device.device_count() < 2 if os.path.exists(device.HIP_LIB_PATH) else True,
                    reason="needs two or more GPUs in this process (hipGetDeviceCount() >= 2)")
def test_in_process_multi_gpu_frame(cbox):
    """The C-ABI multi-GPU path, whenever the box has the GPUs: one context per GPU, every rank's share rendered
    concurrently (hj_render_frame_async), one RCCL reduce through a reused hj_comm, against the 1-GPU frame."""
    n = min(8, device.device_count())
    W = H = 512
    spp = 64
    rs = [device.Renderer(i) for i in range(n)]
    try:
        for r in rs:
            r.upload_scene(cbox)
            r.create_framebuffer(W, H)
        rs[0].render_frame(spp, 1)
        full = rs[0].read()
        comm = device.Comm(rs)
        for rep in range(2):                                  # the communicators are reused by the second frame
            for r in rs:
                r.clear()
            for i, r in enumerate(rs):
                r.render_frame_async(spp, 1, rank=i, world=n)
            comm.reduce(0)
            got = rs[0].read()
            np.testing.assert_allclose(got, full, rtol=5e-5, atol=1e-5)
        for r in rs:
            r.clear()
        for i, r in enumerate(rs):
            r.render_frame(spp, 1, rank=i, world=n)
        device.reduce_framebuffers(rs, root=0)                # the form without a communicator object (cached inside)
        np.testing.assert_allclose(rs[0].read(), full, rtol=5e-5, atol=1e-5)
        comm.close()
    finally:
        for r in rs:
            r.close()



def test_single_context_reduce_through_rccl(cbox_small, monkeypatch):
    """HJ_COMM_FORCE_RCCL test rig: a communicator over ONE context still loads librccl, creates a communicator with
    ncclCommInitAll and sends the framebuffer through ncclReduce (one rank, in place) on the context's stream - the calls
    a multi-GPU box makes, checked here for loading, signatures and stream handling.  A one-rank sum changes no bit."""
    monkeypatch.setenv("HJ_COMM_FORCE_RCCL", "1")
    r = device.Renderer(0)
    try:
        r.upload_scene(cbox_small)
        W, H, spp = 256, 256, 8
        r.create_framebuffer(W, H)
        r.render_frame(spp, 5)
        want = r.read()
        comm = device.Comm([r])
        for _ in range(2):                                    # the communicator is reused
            r.clear()
            r.render_frame_async(spp, 5)
            comm.reduce(0)
            assert (bits(r.read()) == bits(want)).all()
        comm.close()
    finally:
        r.close()


def test_rccl_reduce_at_config5_frame_size(cbox_small, monkeypatch):
    """BASELINE.json configs[4]'s framebuffer through the RCCL calls of the C ABI: 4096 x 4096 RGBA32F = 67 108 864 floats
    (268 MB) in ONE ncclReduce on the context's stream (one rank, in place - all a one-GPU box allows): count and
    stream handling at the size the 8-GPU frame reduces.  One pass of the frame is rendered first so that the buffer holds
    real sums; the reduce must return it bit for bit."""
    monkeypatch.setenv("HJ_COMM_FORCE_RCCL", "1")
    W = H = 4096
    with device.Renderer(0) as r:
        r.upload_scene(cbox_small)
        r.create_framebuffer(W, H)
        st = r.render_frame(1, 9)
        assert st["paths"] == W * H
        want = r.read()
        assert want[..., 3].min() > 0                                    # every pixel was written
        comm = device.Comm([r])
        comm.reduce(0)
        comm.reduce(0)
        got = r.read()
        comm.close()
    assert (bits(got) == bits(want)).all()


_TORCH_RCCL_SCRIPT = r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
from hijiki_amd import host, device, dist as hjdist
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1],
                  HSA_ENABLE_IPC_MODE_LEGACY="0")
os.environ["HIJIKI_DIST_FORCE"] = "1"                        # collectives also with one rank (hijiki_amd/dist.py)
hjdist._init_group("nccl", 0, 1, 0)                          # bench.py's own init: high-priority stream option, first collective
assert dist.get_backend() == "nccl"
cs = host.Scene.synthetic(host.SYNTH_CBOX).compile()
sr = hjdist.ShardedRenderer(cs, 256, 256)
# bench.py's timed loop through RCCL: frames back to back, frame k reduced (ncclReduce on torch's stream) while frame k + 1 renders
st = sr.render_frames(5, 8, 5, reduce=True)
assert st["paths"] == 5 * 256 * 256 * 8
hjdist.barrier(); assert hjdist.max_over_ranks(1.5, device=0) == 1.5
pipelined = sr.fb.clone()
sr.render_frame(8, 5, reduce=True)
assert torch.equal(sr.fb.view(torch.int32), pipelined.view(torch.int32))
sr.render_frame(8, 5, reduce=False)
before = sr.fb.clone()
dist.all_reduce(sr.fb, op=dist.ReduceOp.SUM)                 # RCCL on the buffer the C ABI rendered into
dist.reduce(sr.fb, dst=0, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
assert torch.equal(sr.fb.view(torch.int32), before.view(torch.int32))
plain = device.Renderer(0); plain.upload_scene(cs); plain.create_framebuffer(256, 256); plain.render_frame(8, 5)
assert (sr.fb.cpu().numpy().view(np.uint32) == plain.read().view(np.uint32)).all()
dist.destroy_process_group()
print("rccl-ok")
"""


def test_torch_rccl_reduce_of_the_external_framebuffer(tmp_path):
    """The path `bench.py --gpus N` takes, with N = 1 forced through RCCL: torch.distributed's nccl backend (= RCCL)
    all-reduces and reduces the torch tensor that the C ABI uses as its external framebuffer.  In a child process: the
    process group must not leak into the suite."""
    script = tmp_path / "rccl_one_rank.py"
    script.write_text(_TORCH_RCCL_SCRIPT)
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    p = subprocess.run([sys.executable, str(script), "29533"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "rccl-ok" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


@pytest.mark.parametrize("pairs", ["1", "0"])
def test_pair_nodes_forced_on_and_off(gpu_renderer, oracle, cbox, cbox_spheres, pairs):
    """Pair nodes (an inner node over two triangle leaves tested in one stop of the walk) are the default; forced on and
    forced off (plain leaves: the other instantiation of the walk) here: frames, raw hits and any-hit results stay
    bit-identical to the oracle's either way."""
    old = os.environ.get("HJ_PAIR_LEAVES")
    os.environ["HJ_PAIR_LEAVES"] = pairs
    try:
        for cs, name in ((cbox, "cbox"), (cbox_spheres, "spheres")):
            W, H = 192, 128
            blocks = host.make_blocks(W, H, 3, 23)
            want, ctr, _ = oracle.render_blocks(cs, blocks, W, H)
            got, st = render(gpu_renderer, cs, W, H, blocks)
            assert_same(got, want, f"pair nodes, {name}")
            assert st["closest_rays"] == ctr["closest_calls"] and st["shadow_rays"] == ctr["shadow_calls"]
        rs = np.random.RandomState(5)
        n = 20000
        o = np.stack([rs.uniform(-0.9, 0.9, n), rs.uniform(0.1, 1.5, n), rs.uniform(-0.9, 0.9, n)], 1)
        d = rs.normal(size=(n, 3))
        d /= np.linalg.norm(d, axis=1)[:, None]
        rays = np.concatenate([o, d, np.full((n, 1), 1e-4), np.full((n, 1), np.inf)], 1).astype(np.float32)
        gpu_renderer.upload_scene(cbox)
        ids, t, u, v = gpu_renderer.trace(rays)
        wi, wt, wu, wv = oracle.intersect(cbox, rays)
        assert (ids == wi).all() and (bits(t) == bits(wt)).all() and (bits(u) == bits(wu)).all() and (bits(v) == bits(wv)).all()
        occluded = gpu_renderer.trace(rays, any_hit=True)[0] >= 0
        assert (occluded == (wi >= 0)).all()
    finally:
        if old is None:
            del os.environ["HJ_PAIR_LEAVES"]
        else:
            os.environ["HJ_PAIR_LEAVES"] = old
        gpu_renderer.upload_scene(cbox)


def test_four_contexts_in_flight_on_one_gpu(cbox, monkeypatch):
    """The in-process multi-context path on a one-GPU box: four contexts (HJ_COMM_SHARED_GPU test rig: same GPU, the sum by
    a kernel instead of RCCL) render their shares of the frame AT THE SAME TIME - four worker threads, twelve batch
    streams - and the communicator's reduce joins them.  With the static deal the block interiors equal the 1-context
    frame bit for bit; the rotating deal agrees to the usual tolerance."""
    monkeypatch.setenv("HJ_COMM_SHARED_GPU", "1")
    n, W, H, spp = 4, 512, 384, 24
    rs = [device.Renderer(0) for _ in range(n)]
    try:
        for r in rs:
            r.upload_scene(cbox)
            r.create_framebuffer(W, H)
        rs[0].render_frame(spp, 7)
        full = rs[0].read()
        comm = device.Comm(rs)
        static = device.default_opts()
        static.flags = abi.RENDER_STATIC_DEAL
        for opts in (None, static):
            for r in rs:
                r.clear()
            for i, r in enumerate(rs):
                r.render_frame_async(spp, 7, rank=i, world=n, opts=opts)
            comm.reduce(0)
            stats = [r.sync() for r in rs]
            got = rs[0].read()
            np.testing.assert_allclose(got, full, rtol=5e-5, atol=1e-5)
            if opts is static:
                interior = np.ones((H, W), bool)
                for e in range(128, max(W, H), 128):
                    interior[max(0, e - 2):e + 2, :] = False
                    interior[:, max(0, e - 2):e + 2] = False
                assert (bits(got)[interior] == bits(full)[interior]).all()
        comm.close()
    finally:
        for r in rs:
            r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env", [
    {"HJ_INNER_BURST": "1"},                                        # a round of the walk loop = the merged step alone
    {"HJ_INNER_BURST": "16", "HJ_REFILL_MIN": "1"},                 # long bursts, refill as soon as one lane is free
    {"HJ_REFILL_MIN": "64", "HJ_POOL": "1024"},                     # refill only when the whole wave is idle, small pool
    {"HJ_STREAM_STATE": "1", "HJ_NODE_ORDER": "0"},                 # streamed path state on a small tree, pre-order nodes
    {"HJ_PAIR_LEAVES": "0", "HJ_STREAM_STATE": "1"},                # the plain-leaf instantiation with streamed state
    {"HJ_COLLAPSE_PCT": "0", "HJ_NODE_ORDER": "1"},                 # no collapse, sibling groups
    {"HJ_SLOTS": "1", "HJ_WG_PER_CU": "1", "batch_blocks": "1"},    # one batch slot, 256 workgroups, one ImageBlock per batch
], ids=lambda e: ",".join(f"{k.replace('HJ_', '')}={v}" for k, v in e.items()))
def test_tuning_switches_never_change_a_bit(oracle, cbox_spheres, monkeypatch, env):
    """DESIGN.md section 4's tuning switches (read by the library at context creation / scene upload) steer scheduling
    and layout only: at their extremes, too, the frame is the oracle's bit for bit (every walk instantiation: pair / plain
    leaves x streamed / cached path state)."""
    opts = device.default_opts()
    for k, v in env.items():
        if k == "batch_blocks":
            opts.batch_blocks = int(v)             # (hj_render_opts, not an environment switch)
        else:
            monkeypatch.setenv(k, v)
    W, H = 160, 96
    blocks = host.make_blocks(W, H, 3, 31)
    want, ctr, _ = oracle.render_blocks(cbox_spheres, blocks, W, H)
    with device.Renderer(0) as r:                  # a context of its own: some switches are read when it is created
        got, st = render(r, cbox_spheres, W, H, blocks, opts)
    assert_same(got, want, f"switches {env}")
    assert st["closest_rays"] == ctr["closest_calls"] and st["shadow_rays"] == ctr["shadow_calls"]


@pytest.mark.gpu
def test_small_pool_with_partial_block_rows(oracle, cbox_spheres, monkeypatch):
    """A pool of ONE 64-sample group per workgroup on a frame whose bottom block row is 3 pixels high: most top-ups of the
    workgroups that own that row append nothing (every sample of the group lies outside its block) and the round is skipped
    without a shade - the path through k_path_wavefront's `continue`, which must not find an older round's path count again
    (samples would be shaded twice).  Mirror + glass + emitter scene, exact."""
    monkeypatch.setenv("HJ_POOL", "64")
    monkeypatch.setenv("HJ_WG_PER_CU", "1")
    W, H = 200, 131
    blocks = host.make_blocks(W, H, 3, 17)
    want, ctr, _ = oracle.render_blocks(cbox_spheres, blocks, W, H)
    with device.Renderer(0) as r:
        got, st = render(r, cbox_spheres, W, H, blocks)
    assert_same(got, want, "pool 64, ragged rows")
    assert st["closest_rays"] == ctr["closest_calls"] and st["shadow_rays"] == ctr["shadow_calls"]


@pytest.mark.gpu
def test_reserve_changes_nothing_but_the_first_frame_s_time(cbox_small):
    """hj_reserve allocates the batch slots ahead of the first frame (set-up); the frame is the same bit for bit, and a second
    reserve (or one for a smaller call) is a no-op."""
    W = H = 256
    with device.Renderer(0) as r:
        r.upload_scene(cbox_small)
        r.create_framebuffer(W, H)
        r.render_frame(8, 5)
        want = r.read()
    with device.Renderer(0) as r:
        r.upload_scene(cbox_small)
        r.create_framebuffer(W, H)
        r.reserve(8 * host.blocks_per_pass(W, H))
        r.reserve(8 * host.blocks_per_pass(W, H))
        r.reserve(1)
        r.render_frame(8, 5)
        assert (bits(r.read()) == bits(want)).all()


@pytest.mark.gpu
def test_async_frame_state_and_statistics(cbox_small):
    """hj_render_frame_async: other entry points answer HJ_ERR_STATE while the frame is in flight (never a race with the
    worker thread); the frame's statistics stay retrievable after a reduce has already joined it; a second frame reuses
    the same worker thread; the frame equals the synchronous one bit for bit.  The probe that must be refused does not
    change anything when it is not (hj_framebuffer_read into a scratch buffer), and the frame is long enough (512 x 512 x
    64: tens of milliseconds) to be in flight for certain when the probe runs right behind the call."""
    import threading
    W = H = 512
    spp = 64
    with device.Renderer(0) as r:
        r.upload_scene(cbox_small)
        r.create_framebuffer(W, H)
        want_st = r.render_frame(spp, 5)
        want = r.read()
        L = device.lib()
        scratch = np.zeros((H, W, 4), np.float32)
        for _ in range(2):
            r.clear()
            before = threading.active_count()
            r.render_frame_async(spp, 5)
            rc = L.hj_framebuffer_read(r._h, scratch.ctypes.data_as(C.POINTER(C.c_float)))
            assert rc == abi.HJ_ERR_STATE and b"in flight" in L.hj_last_error(r._h)
            assert L.hj_framebuffer_clear(r._h) == abi.HJ_ERR_STATE          # (refused: the frame's sums are not wiped)
            assert threading.active_count() == before                        # (the worker is the library's own, not a Python thread)
            comm = device.Comm([r])
            comm.reduce(0)                            # joins the frame (hj_sync(ctx, NULL) inside)
            comm.close()
            st = r.sync()                             # ... and the statistics are still there, unconditionally
            assert st["paths"] == want_st["paths"] == W * H * spp and st["closest_rays"] == want_st["closest_rays"]
            assert st["shadow_rays"] == want_st["shadow_rays"] and st["hits"] == want_st["hits"]
            assert (bits(r.read()) == bits(want)).all()
            assert r.sync() == st                     # the result stays retrievable until the next asynchronous frame


@pytest.mark.gpu
def test_frames_back_to_back_without_draining(cbox_small):
    """hj_render_frame with HJ_RENDER_NO_DRAIN + hj_framebuffer_bind + hj_pipeline_wait: five frames (different seeds, two
    external framebuffers in turn) submitted back to back - the batch pipeline is never drained between them - are the
    blocking frames bit for bit, the statistics of the sequence are the sums, the entry points that need an idle context
    answer HJ_ERR_STATE while frames are in flight, and a blocking frame works again after the drain."""
    import torch
    W, H, spp = 384, 256, 12
    seeds = [3, 4, 5, 6, 7]
    with device.Renderer(0) as r:
        r.upload_scene(cbox_small)
        bufs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0") for _ in range(2)]
        r.create_framebuffer(W, H, external_device_ptr=bufs[0].data_ptr())
        want, want_st = [], []
        for s_ in seeds:
            bufs[0].zero_(); torch.cuda.synchronize()
            want_st.append(r.render_frame(spp, s_))
            want.append(bufs[0].cpu().numpy().copy())
        L = device.lib()
        got = []
        torch.cuda.synchronize()
        for k, s_ in enumerate(seeds):
            fb = bufs[k % 2]
            fb.zero_(); torch.cuda.synchronize()
            r.bind_framebuffer(fb.data_ptr())
            r.submit_frame(spp, s_)
            assert L.hj_framebuffer_clear(r._h) == abi.HJ_ERR_STATE and b"NO_DRAIN" in L.hj_last_error(r._h)
            with pytest.raises(abi.HijikiError):
                r.render_frame(spp, s_)                          # a blocking frame in the middle of a sequence: refused
            if k >= 1:
                assert r.pipeline_wait(keep=1) is None
                got.append(bufs[(k - 1) % 2].cpu().numpy().copy())
        st = r.pipeline_wait(keep=0)
        got.append(bufs[(len(seeds) - 1) % 2].cpu().numpy().copy())
        for k in range(len(seeds)):
            assert (bits(got[k]) == bits(want[k])).all(), f"frame {k}"
        for key in ("paths", "closest_rays", "shadow_rays", "hits", "unoccluded_shadow_rays", "batches"):
            assert st[key] == sum(w[key] for w in want_st), key
        assert r.pipeline_wait(keep=0)["paths"] == st["paths"]       # (nothing in flight: the last totals again)
        r.bind_framebuffer(bufs[0].data_ptr())
        bufs[0].zero_(); torch.cuda.synchronize()
        r.render_frame(spp, seeds[0])
        assert (bits(bufs[0].cpu().numpy()) == bits(want[0])).all()


@pytest.mark.gpu
def test_defaults_shrink_to_the_free_device_memory(cbox_small, monkeypatch):
    """A frame whose default pool and batches would take 21 GB of path state and samples, on a device that (HJ_MEM_LIMIT_MB: a test
    rig) has 3 GB free: the render call lowers the pool, then the batch, instead of failing with HJ_ERR_NOMEM, and the frame is the
    same bit for bit (pool and batch size steer scheduling only)."""
    W = H = 1024
    spp = 128
    with device.Renderer(0) as r:
        r.upload_scene(cbox_small)
        r.create_framebuffer(W, H)
        opts = device.default_opts()
        opts.batch_blocks = 2048                       # (an explicit batch: the reference frame itself stays small)
        st0 = r.render_frame(spp, 3, opts=opts)
        want = r.read()
    monkeypatch.setenv("HJ_MEM_LIMIT_MB", "3072")
    with device.Renderer(0) as r:
        r.upload_scene(cbox_small)
        r.create_framebuffer(W, H)
        st = r.render_frame(spp, 3)
        got = r.read()
    assert st["batches"] > st0["batches"]              # smaller batches than the default rule's 2048 blocks
    assert st["paths"] == st0["paths"] == W * H * spp and st["closest_rays"] == st0["closest_rays"]
    assert (bits(got) == bits(want)).all()


_ALLOC_LIMIT_SCRIPT = r"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from hijiki_amd import abi, device, host
cs = host.Scene.synthetic(host.SYNTH_CBOX, mesh_triangles=320).compile()
W = H = 1024; spp = 64
want = np.load(sys.argv[1])
a, b = device.Renderer(0), device.Renderer(0)
for r in (a, b):
    r.upload_scene(cs); r.create_framebuffer(W, H)
# run_begin's estimate sees a nearly empty 288 GB device; the allocations hit the (process-wide) limit
st = a.render_frame(spp, 3)
assert (a.read().view(np.uint32) == want.view(np.uint32)).all(), "context a"
assert st["batches"] >= 4
# the second context renders while the first one still holds its slots: less is left for it
st2 = b.render_frame(spp, 3)
assert (b.read().view(np.uint32) == want.view(np.uint32)).all(), "context b"
assert st2["batches"] >= st["batches"]
# both at once (worker threads), then hj_reserve of a whole large frame on top: shrinks, never crashes
a.clear(); b.clear()
a.render_frame_async(spp, 3); b.render_frame_async(spp, 3)
a.sync(); b.sync()
assert (a.read().view(np.uint32) == want.view(np.uint32)).all() and (b.read().view(np.uint32) == want.view(np.uint32)).all()
a.reserve(64 * 512)
# an explicit batch that cannot fit fails cleanly, with the allocation's message, and the context stays usable
o = device.default_opts(); o.batch_blocks = 8192
try:
    b.render_frame(spp * 4, 3, opts=o)
    raise SystemExit("an 8192-block batch (4.3 GB of samples) fitted a 3 GB limit?")
except abi.HijikiError as e:
    assert e.status == abi.HJ_ERR_NOMEM and "out of memory" in str(e), str(e)
b.clear(); b.render_frame(spp, 3)
assert (b.read().view(np.uint32) == want.view(np.uint32)).all()
print("alloc-limit-ok", st["batches"], st2["batches"])
"""


@pytest.mark.gpu
def test_out_of_memory_at_the_allocation_shrinks_and_retries(cbox_small, tmp_path):
    """run_begin fits the defaults to the free memory it SEES; two contexts on one GPU (or a host allocator) can
    still over-commit it between that check and the allocations.  HJ_ALLOC_LIMIT_MB (test rig: the process's contexts may
    hold 3 GB together) makes the allocations themselves fail: the render call gives its slots back, halves the pool, then
    the batch, retries, and the frame is the same bit for bit; a second context beside the first, both at once, hj_reserve;
    an explicit batch that cannot fit fails with HJ_ERR_NOMEM and leaves the context usable.  Child process: the limit is
    read once per process."""
    W = H = 1024
    with device.Renderer(0) as r:
        r.upload_scene(cbox_small)
        r.create_framebuffer(W, H)
        r.render_frame(64, 3)
        np.save(tmp_path / "want.npy", r.read())
    script = tmp_path / "alloc_limit.py"
    script.write_text(_ALLOC_LIMIT_SCRIPT)
    env = dict(os.environ, HJ_ALLOC_LIMIT_MB="3072", GPU_MAX_HW_QUEUES="8")
    p = subprocess.run([sys.executable, str(script), str(tmp_path / "want.npy")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "alloc-limit-ok" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


@pytest.mark.gpu
def test_degenerate_rays_inside_camera_packets(gpu_renderer, oracle):
    """An axis-aligned camera and a sample offset of exactly (0, 0): the rays of the pixel column x = W / 2 have d.x == 0 and those
    of the row y = H / 2 have d.y == -0 - the slab test's inf - inf (such rays miss every box that straddles the
    origin's plane, root included, as in the reference).  They sit in the middle of 64-ray packets whose other lanes walk on: the
    packet walk's sleep / wake bookkeeping must leave both kinds exactly the oracle's results.  Triangle pairs (floor, wall,
    light) switch the packet stage on."""
    s = host.Scene()
    s.set_camera((0.37, 0.21, 3.0), (0.0, 0.0, 0.0, 1.0), 50.0)      # (off the planes x = 0, y = 0: origin * inf = inf, not NaN)
    white, red = s.add_diffuse((0.75, 0.75, 0.75)), s.add_diffuse((0.7, 0.2, 0.15))
    mirror, lamp = s.add_mirror(), s.add_emissive((12, 11, 10))
    P = np.array([[-2, -1, 1], [2, -1, 1], [2, -1, -3], [-2, -1, -3],       # floor (does not straddle y = 0)
                  [-2, -1, -3], [2, -1, -3], [2, 2, -3], [-2, 2, -3],       # back wall (straddles x = 0 and y = 0)
                  [-0.5, 1.9, -1.5], [0.5, 1.9, -1.5], [0.5, 1.9, -0.5], [-0.5, 1.9, -0.5],   # light, facing down
                  [0.3, -1, -1], [1.3, -1, -1], [0.8, 0.4, -1.4]], np.float32)                # a triangle beside the axis
    N = np.zeros_like(P)
    N[0:4] = (0, 1, 0); N[4:8] = (0, 0, 1); N[8:12] = (0, -1, 0); N[12:15] = (0, 0.3, 1)
    N /= np.linalg.norm(N, axis=1, keepdims=True)
    b = s.add_vertices(P, N)
    for q, m in ((0, white), (4, red), (8, lamp)):
        s.add_triangle(b + q, b + q + 1, b + q + 2, m)
        s.add_triangle(b + q, b + q + 2, b + q + 3, m)
    s.add_triangle(b + 12, b + 13, b + 14, white)
    s.add_sphere((-0.8, -0.5, -1.2), 0.5, mirror)
    cs = s.compile()
    W = H = 128
    blocks = (abi.ImageBlock * 3)(*[abi.ImageBlock(id=i, seed=7 + 13 * i, origin=(0, 0), dimension=(W, H), original_dimension=(W, H),
                                                   sample_offset=(0.0, 0.0)) for i in range(3)])
    want, ctr, _ = oracle.render_blocks(cs, blocks, W, H)
    got, st = render(gpu_renderer, cs, W, H, blocks)
    assert_same(got, want, "degenerate rays in packets")
    assert st["closest_rays"] == ctr["closest_calls"] and st["shadow_rays"] == ctr["shadow_calls"]
    smp = gpu_renderer.samples(blocks[0])
    # the degenerate column / row really are special: no first hit there although the back wall fills the view
    assert (smp[:, W // 2, 7] == 0).all() and (smp[H // 2, :, 7] == 0).all() and (smp[:, W // 2 + 1, 7] > 0).any()
