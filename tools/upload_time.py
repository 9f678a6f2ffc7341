"""Start-up costs at the 1 M-triangle scale: host compile, hj_scene_upload (device tree re-layout + copies), first frame."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hijiki_amd import host, device
t = time.time(); s = host.Scene.synthetic(host.SYNTH_CBOX_MESH, mesh_triangles=1000000); t1 = time.time()
cs = s.compile(); t2 = time.time()
r = device.Renderer(0); t3 = time.time()
r.upload_scene(cs); t4 = time.time()
r.create_framebuffer(2048, 2048); t5 = time.time()
st = r.render_frame(4, 1); t6 = time.time()
st = r.render_frame(4, 1); t7 = time.time()
print(f"generate {t1-t:.2f} s, compile {t2-t1:.2f} s, context {t3-t2:.2f} s, upload {t4-t3:.2f} s, framebuffer {t5-t4:.3f} s, "
      f"first 4-spp frame {t6-t5:.2f} s, second {t7-t6:.2f} s")
