"""The bug-finders of round 5 as functions: one differential fuzz case of the render entry points, one ray-level replay of the
oracle's own rays through an uploaded tree, one attack on the light-shaft grid - shared by tools/fuzz_render.py,
tools/replay_oracle_rays.py and the driver-run suite (tests/test_gpu_fuzz.py).  Everything here needs the GPU (hijiki_amd.device)
and uses the oracle only as the checker."""
import hashlib
import os

import numpy as np

import scenes
from hijiki_amd import abi, device, host
from oracle import hj_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_seed():
    """A seed that changes whenever the product's sources change (the GPU box has no .git: a hash of the kernels, the C ABI's
    translation units, the host compiler and the oracle stands in for the commit hash), so every round - every commit that
    touches the path - explores new cases; HJ_FUZZ_SEED pins it to reproduce a failure (the failing test prints it)."""
    env = os.environ.get("HJ_FUZZ_SEED")
    if env:
        return int(env, 0)
    h = hashlib.sha256()
    for top in ("hijiki_amd/csrc", "oracle", "include"):
        for d, _, files in sorted(os.walk(os.path.join(ROOT, top))):
            if "_build" in d or "_ref" in d:
                continue
            for f in sorted(files):
                if f.endswith((".h", ".hpp", ".hip", ".cpp", ".c")):
                    h.update(f.encode())
                    h.update(open(os.path.join(d, f), "rb").read())
    return int.from_bytes(h.digest()[:6], "little")


def fuzz_scene(r, rng):
    kind = int(rng.integers(0, 5))
    if kind == 0:
        cs = scenes.random_scene(int(rng.integers(0, 10000)))
    elif kind == 1:
        cs = scenes.random_cluster_scene(int(rng.integers(0, 10000)), scale=float(rng.choice([1.0, 1.0, 0.1, 7.0])))
    elif kind == 2:
        cs = host.Scene.synthetic(host.SYNTH_CBOX_SPHERES, mesh_triangles=int(rng.choice([320, 1280]))).compile()
    elif kind == 3:
        cs = host.Scene.synthetic(host.SYNTH_CBOX_MESH, mesh_triangles=int(rng.choice([2000, 20000]))).compile()
    else:
        cs = scenes.nasty_scene(int(rng.integers(0, 10000)))
    on_device = False
    if rng.random() < 0.25 and cs.num_shapes >= 2:
        if rng.random() < 0.5:                      # the tree STAYS on the device: hj_scene_upload takes it over (scene->bvh == NULL) ...
            r.build_bvh(cs, keep_on_device=True)
            cs.set_bvh(r.read_device_bvh())         # ... and the oracle walks the copy hj_bvh_device_read hands out
            on_device = True
        else:
            cs.set_bvh(r.build_bvh(cs))
    return kind, cs, on_device


def fuzz_case(r, it, big=False):
    """Differential fuzz of hj_render_frame: random scene (random_scene / random_cluster_scene / nasty_scene - degenerate geometry
    on purpose - / the synthetic box with a small mesh), compiled or device-built tree (handed over through the host or left on
    the device), random image size (not multiples of the
    block size), samples per pixel, master seed, pass range, rank of a random world size, options (bounce limit, roulette start,
    batch size, light-shaft grid on / off, split kernels, static deal) - against the oracle's render of the same ImageBlocks with
    the same options, bit for bit, counters included.  Returns (ok, one line that reproduces and describes the case)."""
    rng = np.random.default_rng(90000 + it)
    kind, cs, on_device = fuzz_scene(r, rng)
    W, H = int(rng.integers(16, 1700 if big else 420)), int(rng.integers(16, 1200 if big else 300))
    spp = int(rng.integers(1, 10 if big else 6))
    seed = int(rng.integers(0, 2 ** 40))
    p0 = int(rng.integers(0, spp))
    p1 = int(rng.integers(p0 + 1, spp + 1))
    world = int(rng.choice([1, 1, 2, 3, 8]))
    rank = int(rng.integers(0, world))
    o = device.default_opts()
    o.max_bounces = int(rng.choice([1, 2, 3, 6, 1000]))
    o.rr_start = int(rng.choice([1, 2, 4, 9]))
    o.batch_blocks = int(rng.choice([0, 0, 1, 3, 64]))
    o.flags = int(rng.choice([0, 0, 16, 2, 4]))                    # NO_LIGHT_GRID, SPLIT_KERNELS, STATIC_DEAL
    r.upload_scene(cs, device_tree=on_device)
    r.create_framebuffer(W, H)
    st = r.render_frame(spp, seed, pass_begin=p0, pass_end=p1, rank=rank, world=world, opts=o)
    got = r.read()
    blocks_all = host.make_blocks(W, H, spp, seed, pass_begin=p0, pass_end=p1)
    per = host.blocks_per_pass(W, H)
    assert len(blocks_all) == per * (p1 - p0)
    L = host.lib()
    keep = []
    for k, b in enumerate(blocks_all):                              # (make_blocks: pass after pass, block j of a pass at k % per)
        p, j = p0 + k // per, k % per
        if world == 1 or L.hj_block_owner(W, H, 0 if (o.flags & 4) else p, j, world) == rank:
            keep.append(b)
    mine = (abi.ImageBlock * len(keep))(*keep)
    oo = abi.RenderOpts.default()
    oo.max_bounces, oo.rr_start = o.max_bounces, o.rr_start
    want, ctr, _ = O.render_blocks(cs, mine, W, H, opts=oo)
    bad = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=-1).sum())
    ok = bad == 0 and st["closest_rays"] == ctr["closest_calls"] and st["shadow_rays"] == ctr["shadow_calls"] \
        and st["hits"] == ctr["hits"] and st["paths"] == ctr["paths"]
    line = (f"{it}: {'ok ' if ok else 'FAIL'} kind {kind}{' (tree left on the device)' if on_device else ''} {W}x{H} spp {spp} passes [{p0},{p1}) rank {rank}/{world} bounces {o.max_bounces} "
            f"rr {o.rr_start} batch {o.batch_blocks} flags {o.flags} blocks {len(keep)} differing pixels {bad} paths {st['paths']} {ctr['paths']}")
    return ok, line


def degenerate_rays(rng, n, lo=-1.3, hi=2.1):
    """Rays NOT in general position: direction components of exactly +0 / -0 (one or two at once), origins on and off round
    coordinates, open and closed intervals - what the guard nodes and the collapse are not exact for (such
    rays walk the second copy of the tree)."""
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = rng.uniform(lo, hi, (n, 3))
    snap = rng.random((n, 3)) < 0.3
    rays[:, 0:3][snap] = np.round(rays[:, 0:3][snap] * 5) / 5          # origins ON planes like 0, 0.2, 1.0, 2.0
    d = rng.normal(size=(n, 3)).astype(np.float32)
    zero = rng.random((n, 3)) < 0.45
    zero[zero.all(axis=1), 0] = False
    d[zero] = 0.0
    neg = rng.random((n, 3)) < 0.5
    d[zero & neg] = -0.0
    rays[:, 3:6] = d
    rays[:, 6] = rng.choice(np.array([1e-4, 2e-4, 0.0], np.float32), n)
    rays[:, 7] = np.where(rng.random(n) < 0.5, np.inf, rng.uniform(0.1, 4.0, n)).astype(np.float32)
    return rays


def replay_scene(r, cs, blocks, extra_rays=None):
    """Ray-level parity: EVERY ray the oracle traces for `blocks` (closest-hit rays of every bounce and next-event shadow rays,
    with the directions the reference's arithmetic really produces - not always unit vectors) and `extra_rays` through the tree
    as hj_scene_upload re-laid it out: hj_debug_trace must find the same shape and the same t bits, the any-hit walk the same
    boolean.  Returns (rays, mismatches)."""
    log = O.logged_rays(cs, blocks)
    rays = np.ascontiguousarray(log[:, 0:8])
    if extra_rays is not None and len(extra_rays):
        rays = np.ascontiguousarray(np.concatenate([rays, extra_rays]))
    r.upload_scene(cs)
    oi, ot, _, _ = O.intersect(cs, rays)
    ids, t, _, _ = r.trace(rays)
    anyhit, *_ = r.trace(rays, any_hit=True)
    want_logged = log[:, 9].astype(np.int32)
    bad = int((ids != oi).sum()) + int((t.view(np.uint32) != ot.view(np.uint32))[oi >= 0].sum()) + int(((anyhit >= 0) != (oi >= 0)).sum())
    bad += int((oi[:len(log)] != want_logged).sum())                 # (the oracle's probe agrees with its own log)
    return len(rays), bad


def light_grid_disagreements(r, cs, W, H, spp, seed):
    """GPU-side attack on the light-shaft grid (api/light_grid.cpp).  The frame is rendered twice: with the grid, where every
    next-event sample whose cell is proven free is added without a walk and COUNTED as an unoccluded shadow ray, and with
    HJ_RENDER_NO_LIGHT_GRID, where the very same rays (same paths, same RNG draws) are walked.  Every other ray is the same in
    both runs, so unoccluded(grid) - unoccluded(no grid) IS the number of proven-free rays the walk finds occluded: it has to be
    0, and the frames have to agree bit for bit.  Returns (proven, shadow, disagreements, differing pixels)."""
    r.upload_scene(cs)
    r.create_framebuffer(W, H)
    o = device.default_opts()
    st1 = r.render_frame(spp, seed, opts=o)
    f1 = r.read().copy()
    r.clear()
    o.flags = abi.RENDER_NO_LIGHT_GRID
    st0 = r.render_frame(spp, seed, opts=o)
    f0 = r.read()
    assert st0["shadow_rays_proven_free"] == 0 and st0["shadow_rays"] == st1["shadow_rays"] and st0["closest_rays"] == st1["closest_rays"]
    diff = int((f1.view(np.uint32) != f0.view(np.uint32)).any(axis=-1).sum())
    return st1["shadow_rays_proven_free"], st1["shadow_rays"], st1["unoccluded_shadow_rays"] - st0["unoccluded_shadow_rays"], diff
