"""Render rank 0's share of the C2 frame at a given world size a few times (for rocprofv3 --kernel-trace timelines)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hijiki_amd import host, device
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cs = host.Scene.synthetic(host.SYNTH_CBOX).compile()
r = device.Renderer(0); r.upload_scene(cs); r.create_framebuffer(1024, 1024)
for _ in range(3):
    r.clear(); t = time.time(); r.render_frame(512, 1, rank=0, world=world); print(f"{(time.time()-t)*1e3:.2f} ms", flush=True)
