"""The bug-finders in the driver-run suite (VERDICT r5 task 2).  Round 5's three wrong "exact by argument" shortcuts (sphere
guards, triangle / quad guards, the collapse for rays with a zero direction component) had each passed 10^9-path digests and
were found by tools that ran only by hand.  Here they run with every `pytest -m gpu`, on a fixed time budget (about a minute in
all) and on cases derived from a hash of the product's sources (tests/fuzz_cases.py source_seed: the GPU box has no .git), so
every commit that touches the path explores new cases.  A failure prints the seed; HJ_FUZZ_SEED=<seed> reproduces it.
"""
import os
import time

import numpy as np
import pytest

import fuzz_cases as F
import scenes
from hijiki_amd import host

pytestmark = pytest.mark.gpu

SEED = F.source_seed()


def test_fuzzed_render_frames_against_the_oracle(gpu_renderer):
    """400 cases of tools/fuzz_render.py (random scene, tree, size, pass range, rank, options) or as many as fit 30 s (200 take 14 s)."""
    first = SEED % 1_000_000_000
    t0, done, fails = time.time(), 0, []
    while done < 400 and (time.time() - t0 < 30.0 or done < 40):
        ok, line = F.fuzz_case(gpu_renderer, first + done)
        if not ok:
            fails.append(line)
        done += 1
    print(f"fuzz: seed {SEED} (cases {first} ... {first + done - 1}), {done} cases in {time.time() - t0:.1f} s, {len(fails)} failures")
    assert not fails, f"HJ_FUZZ_SEED={SEED}: " + " | ".join(fails[:5])


def test_oracle_rays_replayed_through_uploaded_trees(gpu_renderer, monkeypatch):
    """60 scenes of tools/replay_oracle_rays.py (20 take 1.5 s): every ray of the oracle's log plus 20 k crafted degenerate rays per scene through
    hj_debug_trace, closest-hit and any-hit - on the compiled tree, on the device-built tree and on the tree re-laid out on the
    device (api/scene_relayout.hip) - random, cluster and degenerate scenes."""
    rng = np.random.default_rng(SEED)
    W, H, spp = 96, 64, 2
    t0, total, scenes_done = time.time(), 0, 0
    for k in range(60):
        s = int(rng.integers(0, 1_000_000))
        gen = (scenes.random_scene, scenes.random_cluster_scene, scenes.nasty_scene, scenes.random_cluster_scene)[k % 4]
        cs = gen(s)
        tree = ("compiled", "device-built", "device re-layout")[k % 3]
        if tree == "device-built" and cs.num_shapes >= 2:
            cs.set_bvh(gpu_renderer.build_bvh(cs))
        if tree == "device re-layout":
            monkeypatch.setenv("HJ_UPLOAD_DEVICE", "1")
        else:
            monkeypatch.delenv("HJ_UPLOAD_DEVICE", raising=False)
        n, bad = F.replay_scene(gpu_renderer, cs, host.make_blocks(W, H, spp, s), F.degenerate_rays(rng, 20000))
        total += n
        scenes_done += 1
        assert bad == 0, f"HJ_FUZZ_SEED={SEED}: scene {k} ({gen.__name__}({s}), {tree} tree): {bad} of {n} rays differ"
        if time.time() - t0 > 15.0 and scenes_done >= 8:
            break
    print(f"replay: seed {SEED}, {scenes_done} scenes, {total} rays, 0 mismatches, {time.time() - t0:.1f} s")


def test_light_grid_never_frees_an_occluded_ray(gpu_renderer, cbox, cbox_spheres):
    """The light-shaft grid attacked on the GPU: c2's and c3's scenes at 1024 x 1024 (16 passes) and 200 random scenes (as many
    as fit the budget: 50 take 2 s), each rendered with the grid and with HJ_RENDER_NO_LIGHT_GRID - the proven-free rays walked after all.
    Disagreements (proven free, found occluded) must be 0 and the frames identical."""
    t0 = time.time()
    proven_total = shadow_total = 0
    for name, cs, spp in (("c2 scene", cbox, 16), ("c3 scene", cbox_spheres, 16)):
        proven, shadow, dis, diff = F.light_grid_disagreements(gpu_renderer, cs, 1024, 1024, spp, SEED % 1000 + 1)
        assert dis == 0 and diff == 0, f"HJ_FUZZ_SEED={SEED}: {name}: {dis} proven-free rays are occluded, {diff} pixels differ"
        assert proven > 0.5 * shadow                      # (the grid is in use: 78 % / 69 % of the shadow rays on these scenes)
        proven_total += proven
        shadow_total += shadow
    rng = np.random.default_rng(SEED + 1)
    done = 0
    while done < 200 and (time.time() - t0 < 12.0 or done < 10):
        s = int(rng.integers(0, 1_000_000))
        scale = float(rng.choice([1.0, 1.0, 0.1, 7.0]))
        cs = scenes.random_cluster_scene(s, scale=scale) if done % 2 else scenes.random_scene(s)
        proven, shadow, dis, diff = F.light_grid_disagreements(gpu_renderer, cs, 160, 96, 3, s)
        assert dis == 0 and diff == 0, f"HJ_FUZZ_SEED={SEED}: scene {done} (seed {s}, scale {scale}): {dis} proven-free rays are occluded, {diff} pixels differ"
        proven_total += proven
        shadow_total += shadow
        done += 1
    print(f"light grid: seed {SEED}, 2 + {done} scenes, {proven_total} of {shadow_total} shadow rays proven free, 0 disagreements, {time.time() - t0:.1f} s")


def test_the_shade_stage_checks_hit_points_like_its_cpu_restatement(gpu_renderer, cbox, cbox_spheres):
    """Cells on meshes and in corners are proven for hit points that lie ON their shape, and kernels/hj_shade.h hit_point_on_its_shape
    decides that per hit.  tests/test_light_grid.py Grid.on_its_shape restates the check in numpy; over the oracle's logged shadow
    rays of a frame it must count exactly the rays the GPU reports as proven free for the same blocks (the rays are the same bit
    for bit, so one hit judged differently shows) - and among them not one that the oracle found occluded."""
    import test_light_grid as TL
    from oracle import hj_oracle as O
    rng = np.random.default_rng(SEED + 2)
    cases = [("c2 scene", cbox), ("c3 scene", cbox_spheres)] + [(f"smooth-mesh scene {s}", scenes.smooth_mesh_scene(s)) for s in (int(x) for x in rng.integers(0, 100000, 6))]
    checked = 0
    for name, cs in cases:
        grid = TL.Grid(cs)
        if not grid.got:
            continue
        blocks = host.make_blocks(128, 96, 2, 5)
        sh, ids, d, u, v = TL.oracle_shadow_rays(cs, O, blocks)
        e = sh[:, 10].astype(np.int64)
        planar = grid.proven(sh[:, 0:3], e)
        proven = grid.proven(sh[:, 0:3], e, ids, d, u, v)
        assert not (proven & (sh[:, 9] >= 0)).any(), f"HJ_FUZZ_SEED={SEED}: {name}: a proven ray is occluded"
        gpu_renderer.upload_scene(cs)
        gpu_renderer.create_framebuffer(128, 96)
        st = gpu_renderer.render_blocks(blocks)
        assert st["shadow_rays"] == len(sh)
        assert st["shadow_rays_proven_free"] == int(proven.sum()), (f"HJ_FUZZ_SEED={SEED}: {name}: the GPU proved {st['shadow_rays_proven_free']} rays free, "
                                                                    f"the restatement {int(proven.sum())} ({int(planar.sum())} of them in planar cells)")
        checked += int((proven & ~planar).sum())
    assert checked > 1000                                                  # (rays that only a checked hit point proves)
