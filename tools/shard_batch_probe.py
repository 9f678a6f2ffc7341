"""Wall time of ONE rank's share of the C2 frame vs wavefront batch size (diagnostic for the adaptive batch rule)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hijiki_amd import host, device
cs = host.Scene.synthetic(host.SYNTH_CBOX).compile()
r = device.Renderer(0); r.upload_scene(cs); r.create_framebuffer(1024, 1024)
for world in (8, 4, 2, 1):
    line = []
    for bb in (0, 128, 192, 256, 384, 512, 768, 1024, 2048):
        o = device.default_opts(); o.batch_blocks = bb
        best = 1e9
        for _ in range(3):
            r.clear(); t = time.time(); r.render_frame(512, 1, rank=0, world=world, opts=o); best = min(best, time.time() - t)
        line.append(f"{bb}:{best*1e3:.1f}")
    print(f"world {world}: " + "  ".join(line), flush=True)
