"""What a direction-dependent child order would save (VERDICT r5 task 1, step 1; CPU only).

The reference visits the two children of a node in array order (shader/scene.glsl:97-133).  A "directional" tree keeps K link
orderings of the same boxes and leaves, one per direction class of the rays (include/hijiki_hip.h: hj_ray_direction_class), each
voted by the rays of its class.  This probe renders a small frame of a configuration's scene with the oracle - closest-hit rays
as the reference walks them, shadow rays any-hit as the kernels walk them - once on the compiled tree (today's voted static
order) and once per mode, and prints the oracle's node visits and shape tests per ray.

    python tools/dir_order_probe.py c2 [size spp paths_per_class]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hijiki_amd import host, abi
from oracle import hj_oracle as O

KIND = {"c2": (host.SYNTH_CBOX, 0), "c3": (host.SYNTH_CBOX_SPHERES, 0), "c4": (host.SYNTH_CBOX_MESH, 1_000_000),
        "m100k": (host.SYNTH_CBOX_MESH, 100_000)}
MODES = [("x", 1), ("y", 2), ("z", 4), ("xy", 3), ("xz", 5), ("yz", 6), ("octants", 7), ("major axis", 8)]


def measure(cs, blocks, size):
    _, c, _ = O.render_blocks(cs, blocks, size, size)
    cc, sc = max(1, c["closest_calls"]), max(1, c["shadow_calls"])
    return (c["nodes"] / cc, (c["tri_tests"] + c["sphere_tests"] + c["quad_tests"]) / cc, c["shadow_nodes"] / sc,
            (c["nodes"] + c["shadow_nodes"]) / (cc + sc), c["hits"], c["shadow_hits"])


def main():
    name = sys.argv[1]
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    ppc = int(sys.argv[4]) if len(sys.argv) > 4 else 60000
    kind, tris = KIND[name]
    cs = host.Scene.synthetic(kind, mesh_triangles=tris).compile()
    blocks = host.make_blocks(size, size, spp, 1)
    L = O.lib()
    L.hjo_set_shadow_anyhit(1)
    base = measure(cs, blocks, size)
    print(f"{name}: {int(cs.desc.num_bvh_nodes)} nodes, frame {size}x{size}x{spp}")
    print(f"{'order':34s} {'K':>2s} {'closest nodes/ray':>18s} {'shape tests':>12s} {'shadow nodes/ray':>17s} {'all rays':>9s}   closest vs static")
    print(f"{'static (voted by all rays)':34s} {1:2d} {base[0]:18.2f} {base[1]:12.3f} {base[2]:17.2f} {base[3]:9.2f}")
    L.hjo_set_directional_bvh(-1, None)
    m = measure(cs, blocks, size)
    L.hjo_set_directional_bvh(0, None)
    print(f"{'per ray: nearer box first (bound)':34s} {'-':>2s} {m[0]:18.2f} {m[1]:12.3f} {m[2]:17.2f} {m[3]:9.2f}   {100 * (m[0] / base[0] - 1):+6.1f} %", flush=True)
    for label, mode in MODES:
        k = abi.direction_classes(mode)
        for how, kw in (("voted", dict(vote_paths=ppc * k, fallback=0)), ("geometric near-first", dict(geometric_only=True))):
            t = time.time()
            arrays = cs.directional_bvh(mode, **kw)
            tb = time.time() - t
            L.hjo_set_directional_bvh(mode, arrays.ctypes.data)
            m = measure(cs, blocks, size)
            L.hjo_set_directional_bvh(0, None)
            assert m[4] == base[4] or abs(m[4] - base[4]) < 1e-4 * base[4]        # same hits but for epsilon ties
            print(f"{label + ': ' + how:34s} {k:2d} {m[0]:18.2f} {m[1]:12.3f} {m[2]:17.2f} {m[3]:9.2f}   {100 * (m[0] / base[0] - 1):+6.1f} %   ({tb:.1f} s)", flush=True)


if __name__ == "__main__":
    main()
