"""Frames back to back (hj_render_frame with HJ_RENDER_NO_DRAIN) against one blocking frame after the other: rank 0's share of
the c2 frame at 1 / 2 / 4 / 8 ranks on one GPU, K frames each (no reduce: the rendering side only).

    python tools/frames_probe.py [K] [BATCH_DIV ...]     BATCH_DIV: the blocks of a frame per batch = share / BATCH_DIV (default rule: 4)
"""
import sys, os, time
import torch  # noqa: F401  (external framebuffers)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hijiki_amd import host, device
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
DIVS = [int(a) for a in sys.argv[2:]] or [0]
W = H = 1024; spp = 512
cs = host.Scene.synthetic(host.SYNTH_CBOX).compile()
bufs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0") for _ in range(2)]
r = device.Renderer(0); r.upload_scene(cs); r.create_framebuffer(W, H, external_device_ptr=bufs[0].data_ptr())
base = None
for world, div in [(w, d) for w in (1, 2, 4, 8) for d in DIVS]:
    share = spp * ((host.blocks_per_pass(W, H) + world - 1) // world)
    o = device.default_opts()
    if div:
        o.batch_blocks = min(32768, max(64, (share // div + 63) // 64 * 64))
    r.reserve(share, o)
    def blocking():
        t = time.time()
        for k in range(K):
            bufs[0].zero_(); torch.cuda.synchronize()
            r.render_frame(spp, 1, rank=0, world=world, opts=o)
        return (time.time() - t) / K
    def pipelined():
        torch.cuda.synchronize(); t = time.time()
        for k in range(K):
            fb = bufs[k % 2]; fb.zero_(); torch.cuda.current_stream().synchronize()
            r.bind_framebuffer(fb.data_ptr()); r.submit_frame(spp, 1, rank=0, world=world, opts=o)
            if k >= 1: r.pipeline_wait(keep=1)
        r.pipeline_wait(keep=0); r.bind_framebuffer(bufs[0].data_ptr())
        return (time.time() - t) / K
    blocking(); pipelined()
    tb = min(blocking(), blocking()); tp = min(pipelined(), pipelined())
    base = base or tp
    print(f"world {world} batch {o.batch_blocks or 'default'}: blocking {tb*1e3:.2f} ms/frame, back to back {tp*1e3:.2f} ms/frame ({tb/tp:.3f}x); "
          f"share of the one-rank back-to-back frame: ideal {base/world*1e3:.2f} ms -> efficiency {base/world/tp:.3f} (blocking: {base/world/tb:.3f})", flush=True)
