"""Every BASELINE.json configuration at its OWN size against the ORACLE (VERDICT r4 "Next" #1).

The oracle rendered these frames in the build container (tests/golden/make_full_size.py: minutes to half an hour of CPU per
configuration) and committed, per configuration, the SHA-256 of the RGBA32F accumulation buffer, a CRC32 per 128 x 128 tile,
256 sampled pixels and its work counters.  Here the same frames go through hj_render_frame at the library's DEFAULTS
(8192-block batches, 32768 positions per workgroup, three batch slots: the batching only the full-size frames exercise) and
must reproduce digest and counters: 0 differing bits, as everywhere else (HJ-NUM-1).

The image depends on the BVH through epsilon ties; the fixture carries the SHA-256 of the tree it
was rendered with, and a scene compiler that builds another tree fails here with "regenerate", not with a pixel diff.
"""
import json
import os
import sys
import zlib

import numpy as np
import pytest

from hijiki_amd import host

GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLD)
import make_full_size as mfs  # noqa: E402   (frame_digest / build_scene / tree_digest: one text for maker and checker)

CONFIGS = ["c2", "c3", "c4", "c5p", "c2s2", "c2s3"]


def load(name):
    with open(os.path.join(GOLD, f"full_size_{name}.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", CONFIGS)
def test_fixture_is_well_formed(name):
    """CPU: the committed digest describes the configuration bench.py times (size, passes, seed) and is complete."""
    fx, cfg = load(name), mfs.FULL_SIZE[name]
    W = H = cfg["size"]
    assert (fx["width"], fx["height"], fx["spp"], fx["master_seed"]) == (W, H, cfg["spp"], cfg["seed"])
    assert (fx["pass_begin"], fx["pass_end"]) == tuple(cfg["passes"])
    assert len(fx["block_crc32"]) == (W // 128) * (H // 128) and len(fx["pixels"]) == 256 and len(fx["sha256"]) == 64
    c = fx["counters"]
    assert c["paths"] == W * H * (cfg["passes"][1] - cfg["passes"][0])
    assert c["closest_calls"] >= c["paths"] and c["hits"] <= c["closest_calls"] and c["shadow_calls"] <= c["nee_evals"]
    if name in ("c2", "c5p", "c2s2", "c2s3"):            # bench.py's headline configuration and its sizes
        import bench
        b = bench.CONFIGS["c5" if name == "c5p" else "c2"]
        assert (b["size"], b["spp"]) == (cfg["size"], cfg["spp"])


@pytest.mark.parametrize("name", ["c2", "c3"])
def test_fixture_tree_is_the_compilers_tree(name):
    """CPU: the scene compiler still builds the tree the digest was rendered with (small scenes only: the 1 M-triangle tree
    is checked on the GPU box, where it is built anyway)."""
    fx = load(name)
    cs = mfs.build_scene(mfs.FULL_SIZE[name])
    assert mfs.tree_digest(cs) == fx["tree_sha256"], "the host BVH builder changed: run tests/golden/make_full_size.py again"


def test_digest_function_localises_a_flipped_bit():
    a = np.zeros((256, 384, 4), np.float32)
    s0, c0, p0 = mfs.frame_digest(a)
    a[130, 300, 2] = np.float32(1e-30)
    s1, c1, p1 = mfs.frame_digest(a)
    assert s0 != s1 and [i for i in range(6) if c0[i] != c1[i]] == [1 * 3 + 2]
    assert c0[0] == zlib.crc32(bytes(128 * 128 * 16))


@pytest.mark.gpu
@pytest.mark.parametrize("name", CONFIGS)
def test_full_size_frame_against_the_oracle_digest(gpu_renderer, name):
    fx, cfg = load(name), mfs.FULL_SIZE[name]
    cs = mfs.build_scene(cfg)
    assert mfs.tree_digest(cs) == fx["tree_sha256"], "the host BVH builder changed: run tests/golden/make_full_size.py again"
    W = H = cfg["size"]
    r = gpu_renderer
    r.upload_scene(cs)
    r.create_framebuffer(W, H)
    try:
        st = r.render_frame(cfg["spp"], cfg["seed"], pass_begin=cfg["passes"][0], pass_end=cfg["passes"][1])
        got = r.read()
    finally:
        r.create_framebuffer(64, 64)
    c = fx["counters"]
    assert st["paths"] == c["paths"] and st["closest_rays"] == c["closest_calls"], (st, c)
    assert st["shadow_rays"] == c["shadow_calls"] and st["hits"] == c["hits"], (st, c)
    sha, crcs, px = mfs.frame_digest(got)
    if sha != fx["sha256"]:
        tiles = [i for i, (a, b) in enumerate(zip(crcs, fx["block_crc32"])) if a != b]
        per_row = W // 128
        where = [(i % per_row, i // per_row) for i in tiles[:12]]
        pbad = sum(1 for a, b in zip(px, fx["pixels"]) if a != b)
        raise AssertionError(f"{name}: frame differs from the oracle's in {len(tiles)} of {len(crcs)} tiles, first (tx, ty): {where}; "
                             f"{pbad} of 256 sampled pixels differ")
    assert crcs == fx["block_crc32"] and px == fx["pixels"]
