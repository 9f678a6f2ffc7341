"""Differential fuzz of the render entry points: random scene (random_scene / random_cluster_scene / nasty_scene - degenerate geometry on
purpose - / the synthetic box with a small mesh), random image size (not multiples of the block size), samples per pixel, master seed, pass range, rank of a random world size,
options (bounce limit, roulette start, batch size, light-shaft grid on / off, split kernels) - hj_render_frame on the GPU against the
oracle's render of the same ImageBlocks with the same options, bit for bit, counters included.  The suite tests each of these on its
own; this looks for what only their combinations do.

    python tools/fuzz_render.py [first_seed] [count]          (FUZZ_BIG=1: frames up to 1700 x 1200 x 9 spp)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import scenes
from hijiki_amd import abi, host, device
from oracle import hj_oracle as O

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
r = device.Renderer(0)
fails = 0
for it in range(first, first + count):
    rng = np.random.default_rng(90000 + it)
    kind = int(rng.integers(0, 5))
    if kind == 0: cs = scenes.random_scene(int(rng.integers(0, 10000)))
    elif kind == 1: cs = scenes.random_cluster_scene(int(rng.integers(0, 10000)), scale=float(rng.choice([1.0, 1.0, 0.1, 7.0])))
    elif kind == 2: cs = host.Scene.synthetic(host.SYNTH_CBOX_SPHERES, mesh_triangles=int(rng.choice([320, 1280]))).compile()
    elif kind == 3: cs = host.Scene.synthetic(host.SYNTH_CBOX_MESH, mesh_triangles=int(rng.choice([2000, 20000]))).compile()
    else: cs = scenes.nasty_scene(int(rng.integers(0, 10000)))
    if rng.random() < 0.25 and cs.num_shapes >= 2: cs.set_bvh(r.build_bvh(cs))
    big = os.environ.get("FUZZ_BIG") == "1"                      # frames of many batches: up to 1700 x 1200 x 9 spp (seconds of oracle time each)
    W, H = int(rng.integers(16, 1700 if big else 420)), int(rng.integers(16, 1200 if big else 300))
    spp = int(rng.integers(1, 10 if big else 6))
    seed = int(rng.integers(0, 2 ** 40))
    p0 = int(rng.integers(0, spp)); p1 = int(rng.integers(p0 + 1, spp + 1))
    world = int(rng.choice([1, 1, 2, 3, 8])); rank = int(rng.integers(0, world))
    o = device.default_opts()
    o.max_bounces = int(rng.choice([1, 2, 3, 6, 1000])); o.rr_start = int(rng.choice([1, 2, 4, 9]))
    o.batch_blocks = int(rng.choice([0, 0, 1, 3, 64]))
    o.flags = int(rng.choice([0, 0, 16, 2, 4]))                    # NO_LIGHT_GRID, SPLIT_KERNELS, STATIC_DEAL
    r.upload_scene(cs); r.create_framebuffer(W, H)
    st = r.render_frame(spp, seed, pass_begin=p0, pass_end=p1, rank=rank, world=world, opts=o)
    got = r.read()
    blocks_all = host.make_blocks(W, H, spp, seed, pass_begin=p0, pass_end=p1)
    per = host.blocks_per_pass(W, H)
    assert len(blocks_all) == per * (p1 - p0)
    L = host.lib()
    keep = []
    for k, b in enumerate(blocks_all):                              # (make_blocks: pass after pass, block j of a pass at k % per)
        p, j = p0 + k // per, k % per
        if world == 1 or L.hj_block_owner(W, H, 0 if (o.flags & 4) else p, j, world) == rank: keep.append(b)
    mine = (abi.ImageBlock * len(keep))(*keep)
    oo = abi.RenderOpts.default(); oo.max_bounces = o.max_bounces; oo.rr_start = o.rr_start
    want, ctr, _ = O.render_blocks(cs, mine, W, H, opts=oo)
    bad = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=-1).sum())
    ok = bad == 0 and st["closest_rays"] == ctr["closest_calls"] and st["shadow_rays"] == ctr["shadow_calls"] and st["hits"] == ctr["hits"] and st["paths"] == ctr["paths"]
    fails += 0 if ok else 1
    print(f"{it}: {'ok ' if ok else 'FAIL'} kind {kind} {W}x{H} spp {spp} passes [{p0},{p1}) rank {rank}/{world} bounces {o.max_bounces} rr {o.rr_start} "
          f"batch {o.batch_blocks} flags {o.flags} blocks {len(keep)} differing pixels {bad} paths {st['paths']} {ctr['paths']}", flush=True)
print(f"{count} cases, {fails} failures")
sys.exit(1 if fails else 0)
