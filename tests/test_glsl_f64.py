"""The C oracle against the second restatement (tests/golden/glsl_f64.py: numpy, float64, written from the GLSL only).

Both follow the same shader text but share no code, precision or control structure, so an error both make would have
to be made twice.  Per-function vectors agree to <= 1e-5 relative (binary32 rounding of the oracle); whole paths agree
except where a binary32 rounding flips a branch (a fraction of a per cent of the specular paths), and never in the mean.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import glsl_f64 as G  # noqa: E402
import scenes  # noqa: E402
from hijiki_amd import abi, host  # noqa: E402


@pytest.fixture(scope="module")
def rich():
    cs = scenes.rich_scene()
    return cs, G.Scene(cs)


def _rays(n, seed, lo=(-1.1, 0.05, -1.1), hi=(1.1, 1.9, 1.1)):
    rs = np.random.RandomState(seed)
    o = rs.uniform(lo, hi, (n, 3))
    d = rs.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    return np.concatenate([o, d, np.full((n, 1), 1e-4), np.full((n, 1), np.inf)], 1).astype(np.float32)


def test_rng_known_answers_appendix_d():
    """SURVEY.md Appendix D (u32 arithmetic of rand.glsl), on the numpy restatement."""
    kat = {0: (0xC0A9496A, [0xD90BC8A8, 0xA3CD8C47, 0x5AE9C9C5, 0x19FA5D8D]),
           1: (0x27922C9D, [0x22360E3D, 0x9DCA2765, 0xFDFB9536, 0x64FF4198]),
           12345: (0x0DDEEC13, [0xDBC0639D, 0x21C6A0C4, 0x4E151F4B, 0x527D3F15]),
           0xDEADBEEF: (0x572E7C2D, [0x8DD99F78, 0x78EECC03, 0x8CB16A34, 0x9F00E32F]),
           61: (0, [0, 0, 0, 0])}
    for seed, (state, draws) in kat.items():
        r = G.Rng(G.seed_rng(np.array([seed], np.uint32)))
        assert int(r.s[0]) == state
        assert [int(r.uint(np.array([0]))[0]) for _ in range(4)] == draws
    r = G.Rng(np.array([0xFFFFFF80 ^ 0], np.uint32))
    assert G.Rng(np.array([1], np.uint32)).uniform(np.array([0]))[0] < 1.0


def test_hits_and_populated_intersections(oracle, rich):
    """intersectScene (BVH walk + populate*) on 30 k random rays: same shape, t / p / n / uv / frame within 1e-5."""
    cs, sc = rich
    rays = _rays(30000, 1)
    ids, t, u, v, full = oracle.intersect(cs, rays, full=True)
    r = rays.astype(np.float64)
    its = G.intersect_scene(sc, r[:, 0:3], r[:, 3:6], r[:, 6], r[:, 7])
    assert (its.id != ids).mean() < 2e-4            # epsilon-ties only
    m = (its.id == ids) & (ids >= 0)
    assert m.sum() > 20000 and len(np.unique(ids[m])) > 20
    for got, want in ((its.t[m], t[m]), (its.p[m], full[m, 0:3]), (its.u[m], full[m, 6]), (its.v[m], full[m, 7])):
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-5)
    for got, want in ((its.n[m], full[m, 3:6]), (its.ft[m], full[m, 8:11]), (its.fb[m], full[m, 11:14])):
        np.testing.assert_allclose(got, want, atol=1e-4)      # unit vectors: binary32 normalisation of short sums
        assert np.abs(got - want).mean() < 2e-7
    lin = G.intersect_scene(sc, r[:2000, 0:3], r[:2000, 3:6], r[:2000, 6], r[:2000, 7], use_bvh=False)
    assert (lin.id != its.id[:2000]).mean() < 1e-3  # the brute-force branch (scene.glsl:134-158) finds the same hits


def test_shading_step_vectors(oracle, rich):
    """sampleEmitter (sphere, quad and triangle lights), evalBSDF, sampleBSDF (diffuse, checkerboard, mirror, dielectric with
    TIR and the inverted extinction flag, emissive) on 30 k hits: importance-weighted NEE term, shadow ray, outgoing
    direction, weight, extinction and the RNG state afterwards."""
    cs, sc = rich
    rays = _rays(30000, 2)
    rng0 = G.seed_rng(np.arange(len(rays), dtype=np.uint32) * 7 + 3)
    out, ids, rng_after = oracle.shade_probe(cs, rays, rng0)
    r = rays.astype(np.float64)
    its = G.intersect_scene(sc, r[:, 0:3], r[:, 3:6], r[:, 6], r[:, 7])
    same = (its.id == ids) & (ids >= 0)
    idx = np.nonzero(same)[0]
    rng = G.Rng(rng0)
    mat = sc.materials[its.id[idx]]
    tag, midx = mat >> G.TAG_SHIFT, mat & ((1 << G.TAG_SHIFT) - 1)
    assert set(np.unique(tag)) == {0, 1, 2, 3, 4}
    nee, sd, stm = np.zeros((len(rays), 3)), np.zeros((len(rays), 3)), np.zeros(len(rays))
    md = (tag == G.DIFFUSE) | (tag == G.CBOARD)
    k = idx[md]
    imp, sdir, stmax = G.sample_emitter(sc, its.p[k], rng, k)
    want = (np.sqrt((imp * imp).sum(1)) > G.EPS) & ((sdir * its.n[k]).sum(1) > 0)
    col = G.albedo(sc, tag[md], midx[md], its.u[k], its.v[k])
    nee[k[want]] = ((its.n[k] * sdir).sum(1)[:, None] * col / G.PI * imp)[want]
    sd[k], stm[k] = sdir, stmax
    w, wo, ext, alive = G.sample_bsdf(sc, tag, midx, r[idx, 3:6], its.n[idx], its.ft[idx], its.fb[idx], its.u[idx],
                                      its.v[idx], np.zeros((len(idx), 3)), rng, idx)
    flipped = rng.s[idx] != rng_after[idx]          # a draw compared with a threshold the other way round: < 1e-3 of the hits
    assert flipped.mean() < 1e-3
    g = ~flipped
    assert (nee[idx][g] != 0).any(axis=1).sum() > 5000
    a, b = nee[idx][g], out[idx, 1:4][g]
    np.testing.assert_allclose(a, b, rtol=5e-4, atol=1e-6)              # grazing lights: cos(theta) itself is ill-conditioned
    assert (np.abs(a - b) <= 1e-5 * np.abs(b) + 1e-7).mean() > 0.999    # everywhere else: 1e-5 relative
    np.testing.assert_allclose(sd[idx][g], out[idx, 4:7][g], atol=5e-6)
    np.testing.assert_allclose(stm[idx][g], out[idx, 7][g], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(wo[g], out[idx, 8:11][g], atol=1e-4)
    np.testing.assert_allclose(w[g], out[idx, 11:14][g], rtol=1e-6)
    assert (alive[g] == (out[idx, 14][g] > 0)).all()
    np.testing.assert_allclose(ext[g], out[idx, 16:19][g], rtol=1e-6)
    assert (ext[g] != 0).any(axis=1).sum() > 100    # the tinted dielectric switched its extinction on (inverted flag)
    tir = (tag == G.DIELECTRIC) & (rng.s[idx] == rng0[idx])    # dielectric hit without a draw = total internal reflection
    assert tir.sum() > 10


def test_whole_paths_and_reconstruction(oracle, rich):
    """integrateRay per path (render.glsl:81-147) and the reconstruction splat (reconstruction.glsl:22-66)."""
    cs, sc = rich
    blocks = host.make_blocks(96, 64, 2, 9)
    for b in blocks[:2]:
        s32, _ = oracle.integrate_block(cs, b)
        s64 = G.integrate_block(sc, b)
        close = (np.abs(s64 - s32) <= 1e-4 * np.maximum(1.0, np.abs(s32))).all(-1)
        assert close.mean() > 0.985, close.mean()
        assert abs(s64[..., :3].sum() - s32[..., :3].sum()) < 0.02 * s32[..., :3].sum()
        a32 = oracle.reconstruct_block(b, s32, np.zeros((64, 96, 4), np.float32))
        a64 = G.reconstruct_block(b, s32.astype(np.float64), np.zeros((64, 96, 4)))
        np.testing.assert_allclose(a64, a32, rtol=2e-5, atol=1e-6)
    # a ragged frame: blocks cut by the image border, aprons clipped
    cs2 = host.Scene.synthetic(host.SYNTH_CBOX, mesh_triangles=320).compile()
    sc2 = G.Scene(cs2)
    W, H = 150, 70
    blocks = host.make_blocks(W, H, 2, 4)
    a32, _, _ = oracle.render_blocks(cs2, blocks, W, H)
    a64 = G.render_blocks(sc2, blocks, W, H)
    close = (np.abs(a64 - a32) <= 1e-4 * np.maximum(1.0, np.abs(a32))).all(-1)
    assert close.mean() > 0.995
    np.testing.assert_allclose(a64[..., 3], a32[..., 3], rtol=1e-5)


def test_tap_weights_of_the_filter():
    """Gaussian part at |d| = 0, 0.5, 1, 1.5, 2 for sigma 0.5, radius 2."""
    b = abi.ImageBlock()
    b.dimension[0] = b.dimension[1] = 5
    b.sample_offset[0] = b.sample_offset[1] = 0.5
    smp = np.zeros((5, 5, 8))
    smp[2, 2, 0:4] = 1.0                          # one unit sample in the middle, all normals 0
    acc = G.reconstruct_block(b, smp, np.zeros((5, 5, 4)))
    c0 = np.exp(-8.0)
    for d, want in ((0, 1.0 - c0), (1, np.exp(-2.0) - c0), (2, 0.0)):
        assert abs(acc[2, 2 + d, 3] - want) < 1e-12 and abs(acc[2 + d, 2, 3] - want) < 1e-12


def test_white_furnace_cpu(oracle):
    """Energy check with a closed form (tests/scenes.py furnace_scene): radiance rho * L on the sphere, L on the walls."""
    cs = scenes.furnace_scene()
    W = H = 64
    disc, wall = scenes.furnace_masks(W, H)
    acc, _, _ = oracle.render_blocks(cs, host.make_blocks(W, H, 64, 3), W, H)
    img = oracle.resolve(acc)
    assert np.abs(img[wall] - scenes.FURNACE_L).max() < 1e-5
    assert abs(img[disc].mean() - scenes.FURNACE_RHO * scenes.FURNACE_L) < 0.004 * scenes.FURNACE_L
    a64 = G.render_blocks(G.Scene(cs), host.make_blocks(W, H, 8, 3), W, H)
    i64 = a64[..., :3] / a64[..., 3:4]
    assert np.abs(i64[wall] - scenes.FURNACE_L).max() < 1e-9
    assert abs(i64[disc].mean() - scenes.FURNACE_RHO * scenes.FURNACE_L) < 0.01 * scenes.FURNACE_L


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_scenes_oracle_vs_float64(oracle, seed):
    """The six randomised scenes of the GPU suite (tests/scenes.py random_scene: every shape and material kind, random
    camera), whole frames: the binary32 oracle and the float64 restatement agree pixel for pixel except where a rounding
    sent a path another way, and agree in the mean."""
    cs = scenes.random_scene(seed)
    W, H = 80, 48
    blocks = host.make_blocks(W, H, 2, seed)
    a32, _, _ = oracle.render_blocks(cs, blocks, W, H)
    a64 = G.render_blocks(G.Scene(cs), blocks, W, H)
    close = (np.abs(a64 - a32) <= 1e-4 * np.maximum(1.0, np.abs(a32))).all(-1)
    assert close.mean() > 0.95, close.mean()
    np.testing.assert_allclose(a64[..., 3], a32[..., 3], rtol=3e-4)              # filter weights: first-hit normals only, no random walk
    s32, s64 = a32[..., :3].sum(), a64[..., :3].sum()
    assert abs(s64 - s32) < 0.03 * s32
