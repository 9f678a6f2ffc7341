"""Start-up costs at the 1 M-triangle scale, two routes:
  host    Scene::compile on the host (binned SAH + rotations + ray vote), hj_scene_upload (re-layout on the device)
  device  shapes only on the host, hj_build_bvh_device (tree stays on the device), hj_scene_upload with bvh == NULL
and the first frames of each.  HJ_UPLOAD_TIMING=1 / HJ_LBVH_TIMING=1 print the stages."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hijiki_amd import host, device
tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
t = time.time(); s = host.Scene.synthetic(host.SYNTH_CBOX_MESH, mesh_triangles=tris); t1 = time.time()
r = device.Renderer(0); t2 = time.time()
print(f"generate {t1-t:.2f} s, context {t2-t1:.2f} s")
for rep in range(2):
    a = time.time(); cs = s.compile(); b = time.time()
    r.upload_scene(cs); c = time.time()
    r.create_framebuffer(2048, 2048)
    d = time.time(); r.render_frame(4, 1); e = time.time(); st = r.render_frame(4, 1); f = time.time()
    print(f"host route   #{rep}: compile {1e3*(b-a):7.1f} ms, upload {1e3*(c-b):6.1f} ms = {1e3*(c-a):7.1f} ms; first 4-spp frame {1e3*(e-d):.0f} ms, second {1e3*(f-e):.0f} ms ({2048*2048*4/(f-e)/1e6:.0f} Mpaths/s)")
    a = time.time(); cs2 = s.compile(with_tree=False); b = time.time()
    n = r.build_bvh(cs2, keep_on_device=True); c = time.time()
    r.upload_scene(cs2, device_tree=True); d = time.time()
    r.create_framebuffer(2048, 2048)
    e = time.time(); r.render_frame(4, 1); f = time.time(); st = r.render_frame(4, 1); g = time.time()
    print(f"device route #{rep}: shapes {1e3*(b-a):7.1f} ms, build {1e3*(c-b):6.1f} ms ({n} records), upload {1e3*(d-c):6.1f} ms = {1e3*(d-b):7.1f} ms (build + upload); "
          f"first frame {1e3*(f-e):.0f} ms, second {1e3*(g-f):.0f} ms ({2048*2048*4/(g-f)/1e6:.0f} Mpaths/s)")
