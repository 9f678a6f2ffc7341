"""Render N frames of cbox 1024x1024 at --spp for profiling (no oracle work)."""
import sys, os, time, argparse
import os as _os
if _os.environ.get("HJ_IMPORT_TORCH"):
    import torch  # noqa: F401  (use the HIP runtime bundled with the torch wheel, as bench.py does)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hijiki_amd import host, device, abi

ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=8)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--kind", type=int, default=host.SYNTH_CBOX)
ap.add_argument("--tris", type=int, default=0)
ap.add_argument("--time-kernels", type=int, default=1)
ap.add_argument("--split", type=int, default=0)
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--device-bvh", type=int, default=0, help="1: the tree of hj_build_bvh_device instead of the host's SAH tree")
a = ap.parse_args()
cs = host.Scene.synthetic(a.kind, mesh_triangles=a.tris).compile()
r = device.Renderer(0)
if a.device_bvh:
    cs.set_bvh(r.build_bvh(cs))
r.upload_scene(cs); r.create_framebuffer(a.size, a.size)
o = device.default_opts(); o.batch_blocks = a.batch; o.flags = (abi.RENDER_TIME_KERNELS if a.time_kernels else 0) | (abi.RENDER_SPLIT_KERNELS if a.split else 0)
if os.environ.get("HJ_PROBE_RESERVE", "1") == "1":
    r.reserve(a.spp * host.blocks_per_pass(a.size, a.size), o)      # set-up: the batch slots' device memory
for i in range(a.reps):
    r.clear(); t = time.time(); st = r.render_frame(a.spp, 1, opts=o); dt = time.time() - t
    print(f"[perf] {a.size}x{a.size}x{a.spp}: {dt*1e3:.2f} ms  {a.size*a.size*a.spp/dt/1e6:.1f} Mpaths/s",
          {k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items() if v}, flush=True)
