"""Frames in flight: K frames of a config rendered (a) one after the other on one context (hj_render_frame), (b) alternating
on two contexts with hj_render_frame_async, so that the tail of frame k overlaps the start of frame k + 1.
    python tools/pipeline_probe.py [--kind 0] [--size 1024] [--spp 512] [--frames 6] [--rank 0 --world 1]"""
import sys, os, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hijiki_amd import host, device
ap = argparse.ArgumentParser()
ap.add_argument("--kind", type=int, default=host.SYNTH_CBOX)
ap.add_argument("--tris", type=int, default=0)
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--spp", type=int, default=512)
ap.add_argument("--frames", type=int, default=6)
ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--world", type=int, default=1)
a = ap.parse_args()
cs = host.Scene.synthetic(a.kind, mesh_triangles=a.tris).compile()
rs = []
for i in range(2):
    r = device.Renderer(0); r.upload_scene(cs); r.create_framebuffer(a.size, a.size); rs.append(r)
paths = a.size * a.size * a.spp / a.world
rs[0].render_frame(a.spp, 1, rank=a.rank, world=a.world); rs[1].render_frame(a.spp, 1, rank=a.rank, world=a.world)   # warm-up
for rep in range(2):
    t = time.time()
    for k in range(a.frames):
        rs[0].clear(); rs[0].render_frame(a.spp, 1, rank=a.rank, world=a.world)
    dt = time.time() - t
    print(f"serial:    {a.frames} frames {dt*1e3:8.2f} ms  {paths*a.frames/dt/1e6:8.1f} Mpaths/s  {dt*1e3/a.frames:.2f} ms/frame", flush=True)
    t = time.time()
    inflight = [False, False]
    for k in range(a.frames):
        r = rs[k % 2]
        if inflight[k % 2]:
            r.sync()
        r.clear(); r.render_frame_async(a.spp, 1, rank=a.rank, world=a.world); inflight[k % 2] = True
    for i in range(2):
        if inflight[i]:
            rs[i].sync()
    dt = time.time() - t
    print(f"pipelined: {a.frames} frames {dt*1e3:8.2f} ms  {paths*a.frames/dt/1e6:8.1f} Mpaths/s  {dt*1e3/a.frames:.2f} ms/frame", flush=True)
