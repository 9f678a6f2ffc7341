"""Create / render / destroy contexts repeatedly and watch free device memory (leak check)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hijiki_amd import host, device
cs = host.Scene.synthetic(host.SYNTH_CBOX).compile()
hip = C.CDLL("libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value / 2**20
r0 = device.Renderer(0)          # keeps the runtime alive between iterations
base = None
for it in range(12):
    with device.Renderer(0) as r:
        r.upload_scene(cs); r.create_framebuffer(512, 512)
        r.render_frame(8, it)
        r.build_bvh(cs)
        r.upload_scene(cs)       # re-upload releases the previous scene buffers
        r.create_framebuffer(256, 256)
        r.render_frame(2, it)
    f = free_mb()
    if it == 1:
        base = f                 # (the runtime keeps some of the first iteration's memory in its own pool: compare from the second on)
    print(f"iteration {it}: free {f:.0f} MiB" + (f" (second: {base:.0f})" if base is not None else ""), flush=True)
assert abs(f - base) < 64, "device memory is leaking"
print("no leak")
