"""Quick HIP-vs-oracle parity + timing probe (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hijiki_amd import host, device, abi
from oracle import hj_oracle as O

def compare(name, cs, W, H, spp, seed=1, batch=0):
    blocks = host.make_blocks(W, H, spp, seed)
    ref, ctr, secs = O.render_blocks(cs, blocks, W, H)
    r = device.Renderer(0)
    r.upload_scene(cs); r.create_framebuffer(W, H)
    o = device.default_opts(); o.batch_blocks = batch; o.flags = abi.RENDER_TIME_KERNELS
    t = time.time(); st = r.render_blocks(blocks, o); dt = time.time() - t
    got = r.read()
    neq = (got.view(np.uint32) != ref.view(np.uint32)).any(axis=-1)
    a, b = O.resolve(ref), O.resolve(got)
    l2 = float(np.sqrt(np.nanmean((a - b) ** 2)))
    print(f"[{name}] {W}x{H}x{spp}: pixels differing bitwise {int(neq.sum())}/{W*H}  L2(resolved) {l2:.3e}  "
          f"max|d| {float(np.nanmax(np.abs(got-ref))):.3e}  cpu {secs:.3f}s ({W*H*spp/secs/1e6:.2f} Mpaths/s)  "
          f"gpu wall {dt:.3f}s ({W*H*spp/dt/1e6:.2f} Mpaths/s)")
    print("   stats", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()})
    print("   oracle ctr", ctr)
    r.close()
    return int(neq.sum())

if __name__ == "__main__":
    cs = host.Scene.synthetic(host.SYNTH_CBOX).compile()
    bad = compare("cbox C1", cs, 256, 256, 4)
    bad += compare("cbox 2 batches", cs, 256, 256, 4, batch=5)
    cs3 = host.Scene.synthetic(host.SYNTH_CBOX_SPHERES).compile()
    bad += compare("cbox+spheres", cs3, 256, 256, 8, seed=3)
    cs4 = host.Scene.synthetic(host.SYNTH_CBOX_CBOARD).compile()
    bad += compare("cbox+cboard", cs4, 256, 128, 4, seed=5)
    # throughput probe
    r = device.Renderer(0); r.upload_scene(cs); r.create_framebuffer(1024, 1024)
    o = device.default_opts(); o.flags = abi.RENDER_TIME_KERNELS
    for spp in (8, 32):
        r.clear(); t = time.time(); st = r.render_frame(spp, 1, opts=o); dt = time.time() - t
        print(f"[perf] 1024x1024x{spp}: {dt:.3f}s  {1024*1024*spp/dt/1e6:.1f} Mpaths/s", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items()})
    sys.exit(1 if bad else 0)
