"""Why is bench.py's frame slower than perf_probe's?  Order of context creation vs torch's CUDA initialisation."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hijiki_amd import host, device, abi
mode = sys.argv[1]
cs = host.Scene.synthetic(host.SYNTH_CBOX).compile()
if mode == "torch_first":
    torch.cuda.set_device(0)
    fb = torch.zeros((1024, 1024, 4), dtype=torch.float32, device="cuda:0")
    r = device.Renderer(0)
elif mode == "lib_first":
    r = device.Renderer(0)
    torch.cuda.set_device(0)
    fb = torch.zeros((1024, 1024, 4), dtype=torch.float32, device="cuda:0")
elif mode == "torch_first_no_setdevice":
    fb = torch.zeros((1024, 1024, 4), dtype=torch.float32, device="cuda:0")
    r = device.Renderer(0)
r.upload_scene(cs); r.create_framebuffer(1024, 1024, external_device_ptr=fb.data_ptr())
best = 1e9
for rep in range(5):
    fb.zero_(); torch.cuda.synchronize(); t1 = time.perf_counter(); r.render_frame(512, 1); best = min(best, time.perf_counter() - t1)
print(f"{mode}: {best*1e3:.1f} ms", flush=True)
