"""Time every rank's share of the C2 frame for several world sizes (single GPU, ranks run one after another):
the slowest rank bounds the multi-GPU frame; ideal = full / world.

    python tools/shard_probe.py [WORLD ...]        default: 1 2 4 8
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hijiki_amd import host, device
worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
cs = host.Scene.synthetic(host.SYNTH_CBOX).compile()
r = device.Renderer(0); r.upload_scene(cs); r.create_framebuffer(1024, 1024)
o = device.default_opts()
base = None
for world in worlds:
    times = []; rays = []
    for rank in range(world):
        best = 1e9
        for _ in range(3 if world == 1 else 2):          # (the first frame after fresh allocations is slow)
            r.clear(); t = time.time(); st = r.render_frame(512, 1, rank=rank, world=world, opts=o); best = min(best, time.time() - t)
        times.append(best); rays.append((st["closest_rays"] + st["shadow_rays"]) / 1e6)
    base = base or times[0] * world
    print(f"world {world}: slowest {max(times)*1e3:.1f} ms fastest {min(times)*1e3:.1f} ms ideal {base/world*1e3:.1f} ms  "
          f"efficiency {base/world/max(times):.2f}  per-rank {[round(t*1e3,1) for t in times]} Mrays {[round(x,1) for x in rays]}", flush=True)
