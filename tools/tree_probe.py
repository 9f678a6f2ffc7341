"""Tree-quality probe on the CPU: the oracle's walk counters (node visits and shape tests per ray, closest-hit and any-hit
shadow rays) on a small frame of a configuration's scene, for the host BVH builder as the environment configures it
(HJ_BVH_*: read once per process, so one process per variant).

    HJ_BVH_CHILD_ORDER=0 python tools/tree_probe.py c2 [size spp]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hijiki_amd import host
from oracle import hj_oracle as O

KIND = {"c2": (host.SYNTH_CBOX, 0), "c3": (host.SYNTH_CBOX_SPHERES, 0), "c4": (host.SYNTH_CBOX_MESH, 1_000_000),
        "m100k": (host.SYNTH_CBOX_MESH, 100_000)}


def main():
    name = sys.argv[1]
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    kind, tris = KIND[name]
    t = time.time()
    cs = host.Scene.synthetic(kind, mesh_triangles=tris).compile()
    tc = time.time() - t
    blocks = host.make_blocks(size, size, spp, 1)
    O.lib().hjo_set_shadow_anyhit(1)
    _, c, secs = O.render_blocks(cs, blocks, size, size)
    cc, sc = max(1, c["closest_calls"]), max(1, c["shadow_calls"])
    rays = cc + sc
    print(f"{name} compile {tc:.2f}s  closest: nodes/ray {c['nodes'] / cc:.2f} tri {c['tri_tests'] / cc:.3f} | shadow(any-hit): nodes/ray "
          f"{c['shadow_nodes'] / sc:.2f} tri {c['shadow_tri_tests'] / sc:.3f} | all rays: nodes {(c['nodes'] + c['shadow_nodes']) / rays:.2f} "
          f"tri {(c['tri_tests'] + c['shadow_tri_tests']) / rays:.3f} sphere {(c['sphere_tests'] + c['shadow_sphere_tests']) / rays:.3f}  occluded {c['shadow_hits'] / sc:.3f}")


if __name__ == "__main__":
    main()
