"""Frame time of rank 0's share of the C2 frame at 8 (and 4, 2) virtual ranks for several batch sizes (opts.batch_blocks):
the default rule min(8192, max(256, blocks / 8)) was tuned on the one-GPU frame (32768 blocks)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hijiki_amd import host, device
cs = host.Scene.synthetic(host.SYNTH_CBOX).compile()
r = device.Renderer(0); r.upload_scene(cs); r.create_framebuffer(1024, 1024)
for world in (8, 4, 2, 1):
    n = 32768 // world
    row = []
    for batch in (0, n // 16, n // 8, n // 6, n // 4, n // 3, n // 2, n):
        o = device.default_opts(); o.batch_blocks = batch
        best = 1e9
        for _ in range(3):
            r.clear(); t = time.time(); r.render_frame(512, 1, rank=0, world=world, opts=o); best = min(best, time.time() - t)
        row.append(f"{batch or 'default'}: {best*1e3:.1f}")
    print(f"world {world} ({n} blocks): " + "  ".join(row), flush=True)
