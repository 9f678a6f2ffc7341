"""Per-rank kernel-time breakdown for one world size (diagnostic)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hijiki_amd import host, device
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cs = host.Scene.synthetic(host.SYNTH_CBOX).compile()
r = device.Renderer(0); r.upload_scene(cs); r.create_framebuffer(1024, 1024)
for flags in (0, device.abi.RENDER_TIME_KERNELS):
    o = device.default_opts(); o.flags = flags
    for rank in range(world):
        best = 1e9
        for _ in range(2):
            r.clear(); t = time.time(); st = r.render_frame(512, 1, rank=rank, world=world, opts=o); best = min(best, time.time() - t)
        print(f"flags {flags} rank {rank}: wall {best*1e3:.1f} ms path {st['path_ms']:.1f} recon {st['reconstruct_ms']:.1f} "
              f"batches {st['batches']} rounds {st['bounce_rounds']} rays {(st['closest_rays']+st['shadow_rays'])/1e6:.1f}M paths {st['paths']/1e6:.1f}M", flush=True)
