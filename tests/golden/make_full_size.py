"""Oracle digests of BASELINE.json's configurations at their OWN sizes ("Next" #1).

The oracle (oracle/hj_oracle.c) renders the whole frame of a configuration in the build container (minutes of CPU per
configuration) and this script commits, per configuration, `full_size_<config>.json`:

  sha256        of the RGBA32F accumulation buffer (H x W x 4, row-major) = what hj_framebuffer_read returns
  block_crc32   CRC32 of every 128 x 128 tile of it, row-major over tiles (a mismatch localises)
  pixels        256 pixels (x, y, the four float bit patterns)
  counters      the oracle's work counters (paths, closest_calls, shadow_calls, hits, node / shape tests: SURVEY 8(d)'s
                B_path is computed from them)
  tree_sha256   of the flattened skip-link BVH the scene compiler produced: the image depends on the tree through
                epsilon ties, so a digest only binds a run that walks THIS tree; when the host
                builder changes, this script is run again

`-m gpu` tests (tests/test_full_size_oracle.py) render the same frames through hj_render_frame at the library's default
batching and compare digest + counters.  The reference itself cannot produce these values anywhere in this project
(SURVEY 8(c)): they pin the HIP path to the ORACLE at full size, nothing more.

    python tests/golden/make_full_size.py [c2 c3 c4 c5p c2s2 c2s3 ...]      # default: all
"""
import hashlib
import json
import os
import sys
import time
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from hijiki_amd import host  # noqa: E402

# (scene kind, mesh triangles, width = height, spp of the frame, pass range rendered, master seed): bench.py's CONFIGS
FULL_SIZE = {
    "c2": dict(kind="cbox", tris=0, size=1024, spp=512, passes=(0, 512), seed=1),
    "c3": dict(kind="spheres", tris=0, size=1024, spp=1024, passes=(0, 1024), seed=1),
    "c4": dict(kind="mesh", tris=1_000_000, size=2048, spp=256, passes=(0, 256), seed=1),
    # configs[4]'s frame: the pass prefix [0, 8) of its 4096 passes (the whole frame is 2^36 paths = hours of CPU); the GPU
    # test closes the rest through the bit-exact pass-range additivity it already checks
    "c5p": dict(kind="cbox", tris=0, size=4096, spp=4096, passes=(0, 8), seed=1),
    # the headline frame again under two other master seeds (other block seeds and sub-pixel offsets: other random streams through
    # the same kernels, light-shaft grid included)
    "c2s2": dict(kind="cbox", tris=0, size=1024, spp=512, passes=(0, 512), seed=2),
    "c2s3": dict(kind="cbox", tris=0, size=1024, spp=512, passes=(0, 512), seed=3),
}
BLOCK = 128


def build_scene(cfg):
    kind = {"cbox": host.SYNTH_CBOX, "spheres": host.SYNTH_CBOX_SPHERES, "mesh": host.SYNTH_CBOX_MESH}[cfg["kind"]]
    return host.Scene.synthetic(kind, mesh_triangles=cfg["tris"]).compile()


def tree_digest(cs):
    return hashlib.sha256(np.ascontiguousarray(cs.bvh).tobytes()).hexdigest()


def sample_positions(W, H, n=256, seed=20260):
    r = np.random.default_rng(seed)
    return np.stack([r.integers(0, W, n), r.integers(0, H, n)], 1).astype(np.int64)


def frame_digest(accum):
    """(sha256, per-tile CRC32 list, sampled pixels) of an (H, W, 4) float32 accumulation buffer."""
    a = np.ascontiguousarray(accum, np.float32)
    H, W = a.shape[:2]
    crcs = []
    for ty in range(0, H, BLOCK):
        for tx in range(0, W, BLOCK):
            crcs.append(zlib.crc32(np.ascontiguousarray(a[ty:ty + BLOCK, tx:tx + BLOCK]).tobytes()) & 0xFFFFFFFF)
    pos = sample_positions(W, H)
    px = [[int(x), int(y)] + [int(v) for v in a[y, x].view(np.uint32)] for x, y in pos]
    return hashlib.sha256(a.tobytes()).hexdigest(), crcs, px


def main(names):
    from oracle import hj_oracle as O
    for name in names:
        cfg = FULL_SIZE[name]
        t0 = time.time()
        cs = build_scene(cfg)
        W = H = cfg["size"]
        p0, p1 = cfg["passes"]
        accum = np.zeros((H, W, 4), np.float32)
        total = {}
        secs = 0.0
        step = max(1, (p1 - p0) // 16)          # in pieces: progress lines, and the block list of a piece stays small
        for a in range(p0, p1, step):
            b = min(p1, a + step)
            blocks = host.make_blocks(W, H, cfg["spp"], cfg["seed"], pass_begin=a, pass_end=b)
            _, ctr, s = O.render_blocks(cs, blocks, W, H, accum=accum)
            secs += s
            for k, v in ctr.items():
                total[k] = total.get(k, 0) + v
            print(f"[{name}] passes [{p0}, {b}) of [{p0}, {p1}): {secs:.0f} s", flush=True)
        sha, crcs, px = frame_digest(accum)
        out = dict(config=name, scene=cfg["kind"], mesh_triangles=cfg["tris"], width=W, height=H, spp=cfg["spp"],
                   pass_begin=p0, pass_end=p1, master_seed=cfg["seed"], block=BLOCK, tree_sha256=tree_digest(cs),
                   bvh_nodes=int(len(cs.bvh)), sha256=sha, block_crc32=crcs, pixels=px, counters=total,
                   oracle_seconds=round(secs, 1), oracle_threads=os.cpu_count(),
                   generated_by="tests/golden/make_full_size.py (oracle/hj_oracle.c, closest-hit shadow walks)")
        with open(os.path.join(HERE, f"full_size_{name}.json"), "w") as f:
            json.dump(out, f, separators=(",", ":"))
            f.write("\n")
        print(f"[{name}] {total['paths']} paths in {secs:.0f} s ({total['paths'] / secs / 1e6:.2f} Mrays/s), sha256 {sha[:16]}..., "
              f"total {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or list(FULL_SIZE))
