"""The oracle reproduces the committed golden fixtures bit for bit (guards the restatement against regressions)."""
import os

import numpy as np
import pytest

from hijiki_amd import host

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", ["cbox_64x64x4", "cbox_spheres_64x64x4", "cbox_cboard_96x40x3"])
def test_golden_images(oracle, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    W, H, spp, seed, kind = (int(g[k]) for k in ("width", "height", "spp", "seed", "kind"))
    cs = host.Scene.synthetic(kind).compile()
    blocks = host.make_blocks(W, H, spp, seed)
    acc, ctr, _ = oracle.render_blocks(cs, blocks, W, H, nthreads=3)
    assert (acc.view(np.uint32) == g["accum"].view(np.uint32)).all()
    assert list(ctr.values()) == g["counters"].tolist()
    smp, _ = oracle.integrate_block(cs, blocks[0])
    assert (smp.view(np.uint32) == g["samples0"].view(np.uint32)).all()


def test_thread_count_does_not_change_the_image(oracle, cbox):
    blocks = host.make_blocks(192, 160, 2, 4)
    a, _, _ = oracle.render_blocks(cbox, blocks, 192, 160, nthreads=1)
    b, _, _ = oracle.render_blocks(cbox, blocks, 192, 160, nthreads=7)
    assert (a.view(np.uint32) == b.view(np.uint32)).all()


def test_render_equals_integrate_plus_reconstruct_per_block(oracle, cbox):
    """hjo_render_blocks == the reference's serial loop: for block: integrate, then accumulate (main.rs:1316-1355)."""
    W, H = 192, 128
    blocks = host.make_blocks(W, H, 2, 6)
    full, _, _ = oracle.render_blocks(cbox, blocks, W, H, nthreads=4)
    acc = np.zeros((H, W, 4), np.float32)
    for b in blocks:
        smp, _ = oracle.integrate_block(cbox, b)
        oracle.reconstruct_block(b, smp, acc)
    assert (acc.view(np.uint32) == full.view(np.uint32)).all()


def test_image_is_plausible(oracle, cbox):
    g = np.load(os.path.join(GOLD, "cbox_64x64x4.npz"))
    img = oracle.resolve(g["accum"])
    assert not np.isnan(img).any() and img.min() >= 0
    assert 0.05 < img.mean() < 1.0
    # left wall red, right wall blue (Appendix E): compare mid-height columns near the image borders
    assert img[32, 4, 0] > 2 * img[32, 4, 2] and img[32, 59, 2] > 2 * img[32, 59, 0]
