"""Generates the golden fixtures in this directory from the CPU oracle.

The reference (mad-s/hijiki) cannot be built or run anywhere in this project
(Rust + wgpu + shaderc + Vulkan are absent) and has no tests or
golden images of its own, so these vectors pin the ORACLE's output — they
guard against regressions of the restatement and give the GPU tests fixed
expected values that do not depend on rebuilding the oracle on the GPU box.

    python tests/golden/make_golden.py      # rewrites *.npz next to this file
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from hijiki_amd import host  # noqa: E402
from oracle import hj_oracle as O  # noqa: E402


def rng_kat():
    L = O.lib()
    import ctypes as C
    seeds = np.array([0, 1, 2, 12345, 0xFFFFFFFF, 0xDEADBEEF, 61, 7, 1 << 31], np.uint32)
    states = np.array([L.hjo_rng_seed(int(s)) for s in seeds], np.uint32)
    draws = np.zeros((len(seeds), 4), np.uint32)
    floats = np.zeros((len(seeds), 2), np.float32)
    for i, st in enumerate(states):
        s = C.c_uint32(int(st))
        draws[i] = [L.hjo_rng_next(C.byref(s)) for _ in range(4)]
        s = C.c_uint32(int(st))
        floats[i] = [L.hjo_rng_float(C.byref(s)) for _ in range(2)]
    return dict(seeds=seeds, states=states, draws=draws, floats=floats)


def rays_fixture(cs, n=512, seed=7):
    r = np.random.default_rng(seed)
    o = np.stack([r.uniform(-0.95, 0.95, n), r.uniform(0.05, 1.5, n), r.uniform(-1.0, 0.95, n)], 1)
    d = r.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3], rays[:, 3:6], rays[:, 6], rays[:, 7] = o, d, 1e-4, np.inf
    ids, t, u, v, full = O.intersect(cs, rays, use_bvh=True, full=True)
    return dict(rays=rays, ids=ids, t=t, u=u, v=v, full=full)


def main():
    np.savez(os.path.join(HERE, "rng_kat.npz"), **rng_kat())
    for name, kind, W, H, spp, seed in (("cbox_64x64x4", host.SYNTH_CBOX, 64, 64, 4, 1),
                                        ("cbox_spheres_64x64x4", host.SYNTH_CBOX_SPHERES, 64, 64, 4, 2),
                                        ("cbox_cboard_96x40x3", host.SYNTH_CBOX_CBOARD, 96, 40, 3, 3)):
        cs = host.Scene.synthetic(kind).compile()
        blocks = host.make_blocks(W, H, spp, seed)
        acc, ctr, _ = O.render_blocks(cs, blocks, W, H, nthreads=4)
        smp, _ = O.integrate_block(cs, blocks[0])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), accum=acc, samples0=smp, width=W, height=H, spp=spp,
                            seed=seed, kind=kind, counters=np.array(list(ctr.values()), np.uint64),
                            counter_names=np.array(list(ctr.keys())))
        if kind == host.SYNTH_CBOX:
            np.savez_compressed(os.path.join(HERE, "cbox_rays.npz"), **rays_fixture(cs))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
