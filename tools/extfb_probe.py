"""External-framebuffer cost: torch tensor vs raw hipMalloc vs library-owned buffer (same frame, best of 4)."""
import sys, os, time, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hijiki_amd import host, device
cs = host.Scene.synthetic(host.SYNTH_CBOX).compile()
r = device.Renderer(0); r.upload_scene(cs)
hip = C.CDLL("libamdhip64.so.7")  # the copy torch already mapped
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
def run(label, ptr):
    r.create_framebuffer(1024, 1024, external_device_ptr=ptr)
    best = 1e9
    for _ in range(4):
        r.clear(); t = time.perf_counter(); r.render_frame(512, 1); best = min(best, time.perf_counter() - t)
    print(f"{label}: {best*1e3:.1f} ms  ptr {ptr if ptr else 0:#x}", flush=True)
run("own", None)
p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), 1024 * 1024 * 16) == 0
run("raw hipMalloc 16 MB", p.value)
t = torch.zeros((1024, 1024, 4), dtype=torch.float32, device="cuda:0")
run("torch tensor", t.data_ptr())
big = torch.zeros((64, 1024, 1024, 4), dtype=torch.float32, device="cuda:0")
run("slice 5 of a 1 GiB torch tensor", big[5].data_ptr())
run("own again", None)
def run_torch_clear(label, tensor):
    r.create_framebuffer(1024, 1024, external_device_ptr=tensor.data_ptr())
    best = 1e9
    for _ in range(4):
        tensor.zero_(); torch.cuda.synchronize(); t = time.perf_counter(); r.render_frame(512, 1); best = min(best, time.perf_counter() - t)
    print(f"{label}: {best*1e3:.1f} ms", flush=True)
run_torch_clear("torch tensor, cleared by torch zero_ + synchronize", t)
run("torch tensor, cleared by hj_framebuffer_clear", t.data_ptr())
r2 = device.Renderer(0); r2.upload_scene(cs)
r, r_old = r2, r
run("second context in the process, own fb", None)
run_torch_clear("second context, torch zero_", t)

