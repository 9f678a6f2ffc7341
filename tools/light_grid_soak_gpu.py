"""GPU soak of the light-shaft grid's cells on meshes and in corners (api/light_grid.cpp, HJ_LIGHT_GRID_MESH): N scenes of
tests/scenes.py smooth_mesh_scene (tessellated smooth bodies - convex, concave, creased - in a box, random lights) rendered with
the grid and with HJ_RENDER_NO_LIGHT_GRID (tests/fuzz_cases.py light_grid_disagreements: the proven-free rays walked after all,
same paths, same draws).  Disagreements and differing pixels must be 0.  A third render with HJ_LIGHT_GRID_MESH=0 tells how many
of the proven rays only the bundle proofs prove.

    python tools/light_grid_soak_gpu.py [first_seed] [count] [width height spp]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch  # noqa: F401
import scenes
import fuzz_cases as F
from hijiki_amd import device

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
W, H, spp = (int(x) for x in sys.argv[3:6]) if len(sys.argv) > 5 else (256, 192, 4)
r = device.Renderer(0)
tot = np.zeros(5, np.int64)
t0 = t_last = time.time()
bad = 0
for s in range(first, first + count):
    cs = scenes.smooth_mesh_scene(s)
    os.environ["HJ_LIGHT_GRID_MESH"] = "1"
    proven, shadow, dis, diff = F.light_grid_disagreements(r, cs, W, H, spp, s + 1)
    os.environ["HJ_LIGHT_GRID_MESH"] = "0"
    r.upload_scene(cs)
    r.clear()
    planar = r.render_frame(spp, s + 1)["shadow_rays_proven_free"]
    os.environ["HJ_LIGHT_GRID_MESH"] = "1"
    tot += (proven, proven - planar, shadow, dis, diff)
    if dis or diff:
        bad += 1
        print(f"smooth-mesh scene {s}: {dis} proven-free rays are occluded, {diff} pixels differ   <-- WRONG", flush=True)
    if time.time() - t_last > 30:
        t_last = time.time()
        print(f"... {s - first + 1} scenes, {tot[0]} proven ({tot[1]} by bundle proofs) of {tot[2]} shadow rays, {tot[3]} disagreements, {tot[4]} differing pixels", flush=True)
print(f"seeds {first} ... {first + count - 1}, {W}x{H} x {spp} spp: {tot[0]} rays proven free ({tot[1]} of them by bundle proofs only) of {tot[2]} shadow rays; "
      f"{tot[3]} disagreements, {tot[4]} differing pixels, {bad} scenes wrong; {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
