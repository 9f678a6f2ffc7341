"""Converged reference images made with the float64 numpy restatement of the GLSL (glsl_f64.py), NOT with the oracle.

    python tests/golden/make_converged.py [--spp 4096] [--procs 8]

For the synthetic Cornell box (BASELINE config 2's scene) and the box with the mirror and the dielectric sphere (config 3's
scene): 128 x 128, 4096 spp, master seed 1, the deterministic ImageBlock list of hijiki_amd.host.make_blocks (one
128 x 128 block per pass).  The passes are split into 16 groups; the fixture holds the resolved mean image and, from the
group-to-group scatter, the standard error of that mean per pixel and channel.  Consumers: tests/test_converged.py (the
oracle on CPU, the HIP path with -m gpu): SURVEY.md section 4's converged-image test.
"""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

W = H = 128
SEED = 1
GROUPS = 16


def _scene(kind):
    from hijiki_amd import host
    return host.Scene.synthetic({"cbox": host.SYNTH_CBOX, "spheres": host.SYNTH_CBOX_SPHERES}[kind]).compile()


def _work(args):
    kind, spp, p0, p1 = args
    import glsl_f64 as G
    from hijiki_amd import host
    cs = _scene(kind)
    sc = G.Scene(cs)
    return p0, G.render_blocks(sc, host.make_blocks(W, H, spp, SEED, pass_begin=p0, pass_end=p1), W, H, batch=32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=4096)
    ap.add_argument("--procs", type=int, default=8)
    a = ap.parse_args()
    per = a.spp // GROUPS
    for kind in ("cbox", "spheres"):
        t0 = time.time()
        jobs = []
        for g in range(GROUPS):                      # each group in 4 jobs for load balance
            for q in range(4):
                jobs.append((kind, a.spp, g * per + q * per // 4, g * per + (q + 1) * per // 4))
        with mp.Pool(a.procs) as pool:
            res = dict(pool.map(_work, jobs, chunksize=1))
        groups = []
        for g in range(GROUPS):
            acc = sum(res[g * per + q * per // 4] for q in range(4))
            groups.append(acc)
        total = sum(groups)
        mean = total[..., :3] / total[..., 3:4]
        gm = np.stack([x[..., :3] / x[..., 3:4] for x in groups])
        sem = gm.std(axis=0, ddof=1) / np.sqrt(GROUPS)
        out = os.path.join(HERE, f"converged_{kind}_{W}x{H}x{a.spp}.npz")
        np.savez_compressed(out, mean=mean.astype(np.float32), sem=sem.astype(np.float32), accum=total.astype(np.float32),
                            width=W, height=H, spp=a.spp, seed=SEED, groups=GROUPS, kind=kind,
                            generator="tests/golden/make_converged.py (tests/golden/glsl_f64.py, float64)")
        print(f"{kind}: {time.time() - t0:.0f} s, mean radiance {mean.mean():.5f}, median relative sem {np.median(sem / np.maximum(mean, 1e-6)):.4f} -> {out}", flush=True)


if __name__ == "__main__":
    main()
