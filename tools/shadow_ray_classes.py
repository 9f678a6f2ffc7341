import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, ctypes as C
import torch
from hijiki_amd import host, device, abi
from oracle import hj_oracle as O
kind = {"c2": host.SYNTH_CBOX, "c3": host.SYNTH_CBOX_SPHERES}[sys.argv[1]]
cs = host.Scene.synthetic(kind).compile()
W = H = 192
blocks = host.make_blocks(W, H, 2, 1)
log = O.logged_rays(cs, blocks)        # (for the closest-hit rays' cost below)
# the grid, looked up as kernels/hj_shade.h does: planar bits for every hit point, bits of cells on meshes and in corners for hit
# points that pass the shade stage's check (tests/test_light_grid.py Grid)
import test_light_grid as TL
grid = TL.Grid(cs, 64)
print("grid res", grid.got, grid.stats)
sh, prev_hit, prev_d, prev_u, prev_v = TL.oracle_shadow_rays(cs, O, blocks)
prev_hit = prev_hit.astype(np.int64)
ns, nq = int(cs.desc.num_spheres), int(cs.desc.num_quads)
e = sh[:, 10].astype(np.int64)
planar = grid.proven(sh[:, 0:3], e)
proven = grid.proven(sh[:, 0:3], e, prev_hit, prev_d, prev_u, prev_v)
inside, cell = grid.cells(sh[:, 0:3])
mesh_bit = inside & (((grid.mesh[cell] >> np.clip(e, 0, 7)) & 1) != 0) & ~planar
print(f"proven in planar cells {planar.mean():.4f}; in cells on meshes and in corners {(proven & ~planar).mean():.4f} "
      f"({(proven & ~planar).sum() / max(1, mesh_bit.sum()):.3f} of the hit points in such cells pass the check)")
occl = sh[:, 9] >= 0
print(f"shadow rays {len(sh)}, proven {proven.mean():.3f}, occluded {occl.mean():.3f}, proven&occluded {int((proven & occl).sum())}")
# classify the origin: sphere / big triangle (wall) / small triangle (mesh)
verts = cs.vertices_np if hasattr(cs, 'vertices_np') else None
d = cs.desc
T = np.ctypeslib.as_array(C.cast(d.triangles, C.POINTER(C.c_uint32)), (int(d.num_triangles), 3))
V = np.ctypeslib.as_array(C.cast(d.vertices, C.POINTER(C.c_float)), (int(d.num_vertices), 8))
P = V[:, 0:3]
area = 0.5 * np.linalg.norm(np.cross(P[T[:, 1]] - P[T[:, 0]], P[T[:, 2]] - P[T[:, 0]]), axis=1)
is_wall_tri = area > 0.05
cls = np.where(prev_hit < ns, 0, np.where(is_wall_tri[np.clip(prev_hit - ns - nq, 0, len(T) - 1)], 1, 2))
names = ["sphere", "wall", "mesh"]
for c in range(3):
    m = cls == c
    if m.sum() == 0: continue
    print(f"  origin on {names[c]:6s}: {m.mean():.3f} of shadow rays; proven {proven[m].mean():.3f}; of the unproven: occluded {occl[m & ~proven].mean() if (m & ~proven).any() else 0:.3f}; unproven share of all shadow rays {(m & ~proven).mean():.3f} (unoccluded {(m & ~proven & ~occl).mean():.3f}, occluded {(m & ~proven & occl).mean():.3f})")
# cost (oracle node visits, closest-hit walk = the any-hit walk for unoccluded rays) per class
Lo = O.lib()
Lo.hjo_set_node_histogram.argtypes = [C.c_void_p]
def visits(rays):
    if len(rays) == 0: return 0.0
    hist = np.zeros(int(d.num_bvh_nodes), np.uint32)
    Lo.hjo_set_node_histogram(hist.ctypes.data)
    O.intersect(cs, np.ascontiguousarray(rays[:, 0:8]))
    Lo.hjo_set_node_histogram(None)
    return hist.sum() / len(rays)
cl = log[log[:, 8] == 0]
print(f"closest-hit rays: {visits(cl):.1f} node visits per ray")
for name, m in (("proven", proven), ("unproven wall unoccluded", (cls == 1) & ~proven & ~occl), ("unproven wall occluded (closest walk)", (cls == 1) & ~proven & occl),
                ("unproven mesh unoccluded", (cls == 2) & ~proven & ~occl), ("unproven sphere", (cls == 0) & ~proven)):
    print(f"  {name:40s} {m.mean():.3f} of shadow rays, {visits(sh[m]):.1f} node visits per ray")
