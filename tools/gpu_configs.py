"""BASELINE.json configs 3 and 4 on the GPU: parity at a small size + throughput at the named size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hijiki_amd import host, device, abi
from oracle import hj_oracle as O

def run(name, cs, size, spp, par_size=256, par_spp=2):
    r = device.Renderer(0); r.upload_scene(cs)
    blocks = host.make_blocks(par_size, par_size, par_spp, 7)
    want, ctr, secs = O.render_blocks(cs, blocks, par_size, par_size)
    r.create_framebuffer(par_size, par_size); r.render_blocks(blocks); got = r.read()
    bad = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=-1).sum())
    print(f"[{name}] parity {par_size}^2x{par_spp}: {bad} differing pixels; oracle {par_size*par_size*par_spp/secs/1e6:.2f} Mpaths/s; "
          f"closest nodes/ray {ctr['nodes']/ctr['closest_calls']:.1f} tri {ctr['tri_tests']/ctr['closest_calls']:.2f} rays/path {(ctr['closest_calls']+ctr['shadow_calls'])/ctr['paths']:.2f}", flush=True)
    r.create_framebuffer(size, size)
    o = device.default_opts(); o.flags = abi.RENDER_TIME_KERNELS
    for _ in range(2):
        r.clear(); t = time.time(); st = r.render_frame(spp, 1, opts=o); dt = time.time() - t
        print(f"[{name}] {size}x{size}x{spp}: {dt*1e3:.1f} ms  {size*size*spp/dt/1e6:.1f} Mpaths/s  path_ms {st['path_ms']:.1f} recon_ms {st['reconstruct_ms']:.1f} rays/path {(st['closest_rays']+st['shadow_rays'])/st['paths']:.2f}", flush=True)
    r.close()
    return bad

bad = 0
bad += run("C3 cbox+mirror+dielectric", host.Scene.synthetic(host.SYNTH_CBOX_SPHERES).compile(), 1024, 32)
t = time.time(); cs4 = host.Scene.synthetic(host.SYNTH_CBOX_MESH, mesh_triangles=1000000).compile(); print("C4 scene build", round(time.time()-t, 2), "s")
bad += run("C4 1M-triangle mesh", cs4, 2048, 8, par_size=256, par_spp=1)
sys.exit(1 if bad else 0)
