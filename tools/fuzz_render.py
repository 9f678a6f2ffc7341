"""Differential fuzz of the render entry points: random scene (random_scene / random_cluster_scene / nasty_scene - degenerate geometry on
purpose - / the synthetic box with a small mesh), random image size (not multiples of the block size), samples per pixel, master seed, pass range, rank of a random world size,
options (bounce limit, roulette start, batch size, light-shaft grid on / off, split kernels) - hj_render_frame on the GPU against the
oracle's render of the same ImageBlocks with the same options, bit for bit, counters included.  The suite tests each of these on its
own; this looks for what only their combinations do.

    python tools/fuzz_render.py [first_seed] [count]          (FUZZ_BIG=1: frames up to 1700 x 1200 x 9 spp)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import fuzz_cases as F
from hijiki_amd import device

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
r = device.Renderer(0)
fails = 0
for it in range(first, first + count):
    ok, line = F.fuzz_case(r, it, big=os.environ.get("FUZZ_BIG") == "1")     # (tests/fuzz_cases.py: the driver-run suite runs the same cases)
    fails += 0 if ok else 1
    print(line, flush=True)
print(f"{count} cases, {fails} failures")
sys.exit(1 if fails else 0)
