"""The light-shaft visibility grid (hijiki_amd/csrc/api/light_grid.cpp) against the ORACLE, on the CPU.

The grid claims, per (cell, emitter): every next-event shadow ray from a hit point in this cell to a sampled point of this
emitter is unoccluded - so the shade stage may add the sample without tracing the ray (same image, fewer rays).  Here the
claim is attacked with shadow rays built exactly as shader/scene.glsl:66-88 builds them (origin = a point of a planar shape,
moved off its plane by what float rounding can do to a computed hit point; target = a point of the emitter incl. its corners and
edges; tMin = 2 eps, tMax = distance - eps) and traced by the oracle's closest-hit walk: not one of them may hit anything.
hj_debug_light_grid is pure host code: no GPU is needed.
"""
import ctypes as C

import numpy as np
import pytest

import scenes
from hijiki_amd import abi, device, host

RES = 64


def build_grid(cs, res=RES):
    L = C.CDLL(device.HIP_LIB_PATH)
    L.hj_debug_light_grid.argtypes = [C.POINTER(abi.SceneDesc), C.c_uint32, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                      C.POINTER(C.c_uint64)]
    bits = np.zeros(res ** 3, np.uint8)
    lo, inv, st = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_uint64 * 3)()
    got = L.hj_debug_light_grid(C.byref(cs.desc), res, bits.ctypes.data, lo, inv, st)
    return got, bits, np.array(list(lo), np.float32), np.array(list(inv), np.float32), [int(x) for x in st]


class Grid:
    """The grid with its two kinds of proof apart (hj_debug_light_grid_planes): `planar` bits hold for every hit point of the cell,
    `mesh` bits (cells on meshes and in corners) for hit points that lie on their shape - which kernels/hj_shade.h
    shadow_ray_proven_free checks per hit, and `on_its_shape` below restates in numpy float32."""

    def __init__(self, cs, res=RES):
        L = C.CDLL(device.HIP_LIB_PATH)
        L.hj_debug_light_grid_planes.argtypes = [C.POINTER(abi.SceneDesc), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float),
                                                 C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
        d = cs.desc
        self.res, self.ns = res, int(d.num_spheres)
        self.planar, self.mesh = np.zeros(res ** 3, np.uint8), np.zeros(res ** 3, np.uint8)
        self.recs = np.zeros((int(d.num_quads) + int(d.num_triangles), 8), np.float32)
        lo, inv, lim, st = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 2)(), (C.c_uint64 * 3)()
        self.got = L.hj_debug_light_grid_planes(C.byref(d), res, self.planar.ctypes.data, self.mesh.ctypes.data, self.recs.ctypes.data, lim, lo, inv, st)
        self.lo, self.inv = np.array(list(lo), np.float32), np.array(list(inv), np.float32)
        self.sin_in, self.slide = np.float32(lim[0]), np.float32(lim[1])
        self.stats = [int(x) for x in st]

    def cells(self, p):
        """(inside the grid, cell index) of points p, in float32 as the kernel computes them."""
        f = ((np.asarray(p, np.float32) - self.lo) * self.inv).astype(np.float32)
        inside = (f >= 0).all(1) & (f < self.res).all(1)
        c = np.where(inside[:, None], f, 0).astype(np.uint32).astype(np.int64)
        return inside, (c[:, 2] * self.res + c[:, 1]) * self.res + c[:, 0]

    def on_its_shape(self, ids, p, d, u, v):
        """kernels/hj_shade.h: is hit point p (of the ray with direction d, which hit shape ids at (u, v)) where a bundle proof
        assumes it to be?"""
        ids = np.asarray(ids, np.int64)
        flat = ids >= self.ns
        r = self.recs[np.where(flat, ids - self.ns, 0)]
        n, delta, a, quad = r[:, 0:3], r[:, 3], r[:, 4:7], r[:, 7] != 0
        p, d, u, v = (np.asarray(x, np.float32) for x in (p, d, u, v))
        # kernels/hj_num.h dot3 = fmaf(z, z', fmaf(y, y', x x')): a float32 product is exact in float64, so is the sum before its one rounding
        fma = lambda x, y, z: (x.astype(np.float64) * y.astype(np.float64) + z.astype(np.float64)).astype(np.float32)
        dot = lambda x, y: fma(x[:, 2], y[:, 2], fma(x[:, 1], y[:, 1], (x[:, 0] * y[:, 0]).astype(np.float32)))
        pa = (p - a).astype(np.float32)
        l1 = ((np.abs(pa[:, 0]) + np.abs(pa[:, 1])).astype(np.float32) + np.abs(pa[:, 2])).astype(np.float32)
        f = (np.abs(dot(n, pa)) + (np.float32(3e-7) * l1).astype(np.float32)).astype(np.float32)
        dn, dd = dot(d, n), dot(d, d)
        one = np.float32(1)
        third = np.where(quad, np.minimum(one - u, one - v), (one - u) - v).astype(np.float32)
        inside = np.minimum(np.minimum(u, v), third)
        with np.errstate(all="ignore"):
            return flat & (inside >= delta) & (dn * dn >= (self.sin_in * self.sin_in) * dd) & (dd > 0) & ((f * f) * dd <= (self.slide * self.slide) * (dn * dn))

    def proven(self, p, e, ids=None, d=None, u=None, v=None):
        """shadow_ray_proven_free for hit points p and emitter indices e; without the hit (ids ... v): the planar bits alone."""
        e = np.asarray(e, np.int64)
        inside, cell = self.cells(p)
        ok = inside & (e >= 0) & (e < 8)
        es = np.where(ok, e, 0)
        out = ok & (((self.planar[cell] >> es) & 1) != 0)
        if ids is not None:
            out |= ok & (((self.mesh[cell] >> es) & 1) != 0) & self.on_its_shape(ids, p, d, u, v)
        return out


def oracle_shadow_rays(cs, O, blocks):
    """The oracle's next-event shadow rays for `blocks` with the hit each one starts from: (shadow rows of the log, id, d, u, v of
    the closest-hit ray logged right before each - re-traced for its (u, v), which the log does not hold)."""
    log = O.logged_rays(cs, blocks)
    idx = np.nonzero(log[:, 8] == 1)[0]
    idx = idx[idx > 0]
    prev = log[idx - 1]
    keep = prev[:, 8] == 0
    idx, prev = idx[keep], prev[keep]
    rays = np.ascontiguousarray(prev[:, 0:8])
    ids, _, u, v = O.intersect(cs, rays, use_bvh=True)
    return log[idx], ids, prev[:, 3:6], u, v


def crafted_hits(cs, grid, rng, per_shape):
    """Hit points as the shade stage can see them on every flat shape (uniform per shape): X = a + u e1 + v e2 on the shape, moved
    ALONG a random incoming direction d by up to 9e-6 either way (more than the check admits: it filters) - returns ids, p, d, u, v."""
    tri, pos, quad, _, ns, nq = scene_arrays(cs)
    ids, X, N, U, V = [], [], [], [], []
    if len(quad):
        k = np.repeat(np.arange(len(quad)), per_shape * 4)
        q = quad[k].astype(np.float64)
        u, v = rng.uniform(0, 1, (2, len(k)))
        ids.append(ns + k); X.append(q[:, 0:3] + q[:, 4:7] * u[:, None] + q[:, 8:11] * v[:, None]); N.append(np.cross(q[:, 4:7], q[:, 8:11])); U.append(u); V.append(v)
    if len(tri):
        k = np.repeat(np.arange(len(tri)), per_shape)
        a, b, c = (pos[tri[k, j]].astype(np.float64) for j in range(3))
        u, v = rng.uniform(0, 1, (2, len(k)))
        fl = u + v > 1
        u, v = np.where(fl, 1 - u, u), np.where(fl, 1 - v, v)
        ids.append(ns + nq + k); X.append(a + (b - a) * u[:, None] + (c - a) * v[:, None]); N.append(np.cross(b - a, c - a)); U.append(u); V.append(v)
    ids, X, N, U, V = (np.concatenate(x) for x in (ids, X, N, U, V))
    ln = np.linalg.norm(N, axis=1)
    keep = ln > 0
    ids, X, N, U, V = ids[keep], X[keep], N[keep] / ln[keep, None], U[keep], V[keep]
    d = rng.normal(size=X.shape)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d *= rng.uniform(0.5, 2.0, (len(d), 1))                                # (directions are not always unit vectors)
    p = X + d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(-9e-6, 9e-6, (len(X), 1))
    return ids, p.astype(np.float32), d.astype(np.float32), U.astype(np.float32), V.astype(np.float32)


def scene_arrays(cs):
    d = cs.desc
    tri = np.ctypeslib.as_array(d.triangles, shape=(d.num_triangles,)).view(np.uint32).reshape(-1, 3) if d.num_triangles else np.zeros((0, 3), np.uint32)
    vert = np.ctypeslib.as_array(d.vertices, shape=(d.num_vertices,)).view(np.float32).reshape(-1, 8) if d.num_vertices else np.zeros((0, 8), np.float32)
    quad = np.ctypeslib.as_array(d.quads, shape=(d.num_quads,)).view(np.float32).reshape(-1, 12) if d.num_quads else np.zeros((0, 12), np.float32)
    em = np.ctypeslib.as_array(d.emitters, shape=(d.num_emitters,)).view(np.uint32).reshape(-1, 4)[:, 0].astype(np.int64) if d.num_emitters else np.zeros(0, np.int64)
    return tri, vert[:, 0:3], quad, em, int(d.num_spheres), int(d.num_quads)


def planar_points(cs, rng, n):
    """n points on triangles (picked by area: the walls of a box are a dozen of thousands of triangles) and n / 4 on quads, with
    the shape's unit normal."""
    tri, pos, quad, _, ns, nq = scene_arrays(cs)
    pts, nrm = [], []
    if len(tri):
        area = np.linalg.norm(np.cross(pos[tri[:, 1]] - pos[tri[:, 0]], pos[tri[:, 2]] - pos[tri[:, 0]]), axis=1).astype(np.float64)
        t = tri[rng.choice(len(tri), n, p=area / area.sum())] if area.sum() > 0 else tri[rng.integers(0, len(tri), n)]
        a, b, c = pos[t[:, 0]], pos[t[:, 1]], pos[t[:, 2]]
        u, v = rng.uniform(size=(2, n))
        f = u + v > 1
        u, v = np.where(f, 1 - u, u), np.where(f, 1 - v, v)
        pts.append(a + (b - a) * u[:, None] + (c - a) * v[:, None])
        nrm.append(np.cross(b - a, c - a))
    if len(quad):
        q = quad[rng.integers(0, len(quad), max(1, n // 4))]
        o, e1, e2 = q[:, 0:3], q[:, 4:7], q[:, 8:11]
        u, v = rng.uniform(size=(2, len(q)))
        pts.append(o + e1 * u[:, None] + e2 * v[:, None])
        nrm.append(np.cross(e1, e2))
    p, m = np.concatenate(pts), np.concatenate(nrm)
    ln = np.linalg.norm(m, axis=1)
    keep = ln > 0
    return p[keep], m[keep] / ln[keep, None]


def emitter_points(cs, e_shape, rng, n):
    """Points of emitter shape `e_shape` (a triangle or quad): random ones, its corners, points on its edges."""
    tri, pos, quad, _, ns, nq = scene_arrays(cs)
    if e_shape < ns:
        return None
    if e_shape < ns + nq:
        q = quad[e_shape - ns]
        o, e1, e2 = q[0:3], q[4:7], q[8:11]
        u, v = rng.uniform(size=(2, n))
    else:
        t = tri[e_shape - ns - nq]
        o, e1, e2 = pos[t[0]], pos[t[1]] - pos[t[0]], pos[t[2]] - pos[t[0]]
        u, v = rng.uniform(size=(2, n))
        f = u + v > 1
        u, v = np.where(f, 1 - v, u), v            # the reference's randBarycentric: (1 - v, v) when u + v > 1 (rand.glsl:45-48)
    k = n // 8
    u[:k], v[:k] = rng.integers(0, 2, k), 0.0      # corners and edges
    v[k:2 * k] = 0.0
    if e_shape >= ns + nq:
        u[:k] = np.minimum(u[:k], 1 - v[:k])
    return (o + e1 * u[:, None] + e2 * v[:, None]).astype(np.float32)


def shape_points(cs, rng, per_shape):
    """per_shape points on EVERY flat shape (uniform per shape, not by area: the small triangles of a mesh count as much as the
    walls), a hair across the edges (a hit point is accepted by a float test), with the shape's unit normal."""
    tri, pos, quad, _, ns, nq = scene_arrays(cs)
    pts, nrm = [], []
    if len(tri):
        t = np.repeat(tri, per_shape, axis=0)
        a, b, c = pos[t[:, 0]], pos[t[:, 1]], pos[t[:, 2]]
        u, v = rng.uniform(-2e-6, 1 + 2e-6, (2, len(t)))
        f = u + v > 1
        u, v = np.where(f, 1 - u, u), np.where(f, 1 - v, v)
        pts.append(a + (b - a) * u[:, None] + (c - a) * v[:, None])
        nrm.append(np.cross(b - a, c - a))
    if len(quad):
        q = np.repeat(quad, per_shape * 4, axis=0)
        o, e1, e2 = q[:, 0:3], q[:, 4:7], q[:, 8:11]
        u, v = rng.uniform(-2e-6, 1 + 2e-6, (2, len(q)))
        pts.append(o + e1 * u[:, None] + e2 * v[:, None])
        nrm.append(np.cross(e1, e2))
    p, m = np.concatenate(pts), np.concatenate(nrm)
    ln = np.linalg.norm(m, axis=1)
    keep = ln > 0
    return p[keep], m[keep] / ln[keep, None]


def attack(cs, oracle, rng, n=60000, res=RES, per_shape=0):
    """Number of shadow rays tried from cells whose bit is set, and how many the oracle found occluded (must be 0)."""
    grid = Grid(cs, res)
    st = grid.stats
    if grid.got == 0:
        return 0, 0, st
    _, _, _, em, _, _ = scene_arrays(cs)
    # planar bits: any point of the cell within tol of the plane - points on the shapes, a hair across their edges, moved by up to
    # +-1e-5 along the normal (the grid allows 2e-6 x scale); bits of cells on meshes and in corners: hit points that pass the shade
    # stage's check (crafted_hits: on the shape, slid along a random incoming direction)
    p, nrm = shape_points(cs, rng, per_shape) if per_shape else planar_points(cs, rng, n)
    p = (p + nrm * rng.uniform(-1e-5, 1e-5, (len(p), 1))).astype(np.float32)
    hits = crafted_hits(cs, grid, rng, per_shape) if per_shape else None
    tried = bad = 0
    for e, shape in enumerate(em[:8]):
        sel = grid.proven(p, np.full(len(p), e))
        o = p[sel]
        if hits is not None:
            hsel = grid.proven(hits[1], np.full(len(hits[1]), e), *(hits[k] for k in (0, 2, 3, 4)))
            o = np.concatenate([o, hits[1][hsel]])
        if not len(o):
            continue
        y = emitter_points(cs, int(shape), rng, len(o))
        assert y is not None, "a bit is set for a sphere emitter"
        d = y - o
        dist = np.sqrt((d.astype(np.float32) ** 2).sum(1, dtype=np.float32)).astype(np.float32)
        d = (d / dist[:, None]).astype(np.float32)
        rays = np.zeros((len(o), 8), np.float32)
        rays[:, 0:3], rays[:, 3:6], rays[:, 6], rays[:, 7] = o, d, np.float32(2e-4), dist - np.float32(1e-4)     # scene.glsl:85
        ids, _, _, _ = oracle.intersect(cs, rays, use_bvh=True)
        tried += len(rays)
        bad += int((ids >= 0).sum())
    return tried, bad, st


def test_cbox_most_wall_cells_are_proven_and_no_proven_ray_is_occluded(oracle, cbox):
    rng = np.random.default_rng(1)
    tried, bad, st = attack(cbox, oracle, rng, n=200000)
    surface, planar, clear = st
    assert planar > 0.6 * surface and clear > 0.5 * 2 * planar        # two emitter triangles; measured: 0.79, 0.66
    assert tried > 50000 and bad == 0


def test_spheres_mesh_and_rich_scenes(oracle, cbox_spheres):
    rng = np.random.default_rng(2)
    for cs in (cbox_spheres, host.Scene.synthetic(host.SYNTH_CBOX_MESH, mesh_triangles=20000).compile(), scenes.rich_scene()):
        tried, bad, st = attack(cs, oracle, rng)
        assert bad == 0, (tried, bad, st)


@pytest.mark.parametrize("seed", range(4))
def test_random_scenes(oracle, seed):
    rng = np.random.default_rng(10 + seed)
    tried, bad, st = attack(scenes.random_scene(seed), oracle, rng, n=40000)
    assert bad == 0, (tried, bad, st)


def test_an_occluder_between_floor_and_light_clears_the_cells_below_it():
    """A diffuse quad floating under a quad light: floor cells in its shadow and penumbra must not be proven; floor cells far
    from it are; cells on the occluder's lit side are."""
    s = host.Scene()
    s.set_camera((0, 1, 4), (0, 0, 0, 1), 40.0)
    white, lamp = s.add_diffuse((0.7, 0.7, 0.7)), s.add_emissive((10, 10, 10))
    s.add_quad((-2, 0, 2), (4, 0, 0), (0, 0, -4), white)                       # floor y = 0
    s.add_quad((-0.25, 2, -0.25), (0.5, 0, 0), (0, 0, 0.5), lamp)              # light y = 2
    s.add_quad((-0.5, 1, 0.5), (1, 0, 0), (0, 0, -1), white)                   # occluder y = 1, above the floor's centre
    cs = s.compile()
    got, bits, lo, inv, st = build_grid(cs)
    assert got == RES

    def bit(p):
        f = ((np.array(p, np.float32) - lo) * inv).astype(int)
        return int(bits[(f[2] * RES + f[1]) * RES + f[0]]) & 1

    assert bit((0.0, 0.0, 0.0)) == 0 and bit((0.3, 0.0, -0.3)) == 0 and bit((0.7, 0.0, 0.0)) == 0     # umbra, penumbra
    assert bit((1.8, 0.0, 1.8)) == 1 and bit((-1.7, 0.0, 0.4)) == 1                                   # far from the occluder
    assert bit((0.0, 1.0, 0.0)) == 1                                                                   # on top of the occluder


def test_no_grid_without_planar_emitters_or_at_huge_scale():
    s = host.Scene()
    s.set_camera((0, 0, 5), (0, 0, 0, 1), 40.0)
    white, lamp = s.add_diffuse((0.7, 0.7, 0.7)), s.add_emissive((5, 5, 5))
    s.add_quad((-2, -1, 2), (4, 0, 0), (0, 0, -4), white)
    s.add_sphere((0, 1, 0), 0.3, lamp)                                         # a sphere light: nothing can be proven
    assert build_grid(s.compile())[0] == 0
    s = host.Scene()
    s.set_camera((0, 1e4, 4e4), (0, 0, 0, 1), 40.0)
    white, lamp = s.add_diffuse((0.7, 0.7, 0.7)), s.add_emissive((5, 5, 5))
    s.add_quad((-2e4, 0, 2e4), (4e4, 0, 0), (0, 0, -4e4), white)               # coordinates where eps = 1e-4 is below float resolution
    s.add_quad((-2e3, 2e4, -2e3), (4e3, 0, 0), (0, 0, 4e3), lamp)
    assert build_grid(s.compile())[0] == 0


@pytest.mark.parametrize("name", ["cbox", "cbox + spheres", "cluster 77", "cluster 3", "cluster 14 x 0.01", "random 5", "random 9"])
def test_grid_against_the_oracles_own_shadow_rays(name):
    """The attack above builds its own shadow rays; this one takes the ORACLE's: every next-event ray of a frame as the reference's
    arithmetic produced it (oracle.logged_rays: origin = the computed hit point, target = the sampled point, the emitter's index),
    looked up in the grid exactly as kernels/hj_shade.h shadow_ray_proven_free does.  A set bit and an occluded ray in the same log
    would be a wrong pixel."""
    from oracle import hj_oracle as O
    if name == "cbox":
        cs = host.Scene.synthetic(host.SYNTH_CBOX).compile()
    elif name == "cbox + spheres":
        cs = host.Scene.synthetic(host.SYNTH_CBOX_SPHERES).compile()
    elif name.startswith("cluster"):
        parts = name.split()
        cs = scenes.random_cluster_scene(int(parts[1]), scale=float(parts[3]) if len(parts) > 3 else 1.0)
    else:
        cs = scenes.random_scene(int(name.split()[1]))
    grid = Grid(cs)
    if not grid.got:
        pytest.skip("no grid for this scene")
    sh, ids, d, u, v = oracle_shadow_rays(cs, O, host.make_blocks(128, 96, 3, 4))
    assert len(sh) > 1000
    proven = grid.proven(sh[:, 0:3], sh[:, 10].astype(np.int64), ids, d, u, v)
    occluded = sh[:, 9] >= 0
    assert not (proven & occluded).any(), (name, int((proven & occluded).sum()), sh[proven & occluded][:3])
    if name in ("cbox", "cbox + spheres"):
        assert proven.mean() > 0.5                                        # (not vacuous: most of the box scenes' shadow rays are proven)


@pytest.mark.parametrize("kind,tris", [(host.SYNTH_CBOX, 1280), (host.SYNTH_CBOX, 6320), (host.SYNTH_CBOX_SPHERES, 6320), (host.SYNTH_CBOX_MESH, 6000)])
def test_cells_on_meshes_and_in_corners(oracle, kind, tris, monkeypatch):
    """Round 6: a cell whose shapes are flat but NOT coplanar - the facets of a mesh, the corner of two walls - is proven when the
    whole bundle of rays from every shape of the cell to the emitter misses every flat shape that touches the shaft
    (api/light_grid.cpp bundle_misses).  More (cell, emitter) pairs than without (HJ_LIGHT_GRID_MESH=0), and not one crafted ray -
    points on EVERY shape, uniform per shape, off their planes by +-1e-5, towards corners, edges and random points of the light -
    that the oracle finds occluded."""
    cs = host.Scene.synthetic(kind, mesh_triangles=tris).compile()
    monkeypatch.setenv("HJ_LIGHT_GRID_MESH", "0")
    _, bits0, _, _, st0 = build_grid(cs)
    monkeypatch.setenv("HJ_LIGHT_GRID_MESH", "1")
    got, bits1, _, _, st1 = build_grid(cs)
    assert got == RES and st1[2] > st0[2] and ((bits1 & bits0) == bits0).all()          # nothing that was proven is lost
    rng = np.random.default_rng(tris)
    tried, bad, _ = attack(cs, oracle, rng, per_shape=12 if tris > 2000 else 60)
    assert tried > (50 if kind == host.SYNTH_CBOX_MESH else 1000) and bad == 0, (tried, bad)      # (the dense mesh's own cells are rarely provable)


def test_a_mesh_cell_under_an_occluder_is_not_proven(oracle):
    """A shallow faceted dome on the floor under the light (a 6 x 6 height field, neighbouring facets a few degrees apart): the cells
    on it are proven - also those that hold a vertex where six facets meet; with a small quad hung between dome and light the cells
    under it are not (the quad touches their shafts and the bundle's crossing points reach it).  Creases sharper than the
    tolerances allow (tens of degrees) stay unproven by design."""
    def scene(with_occluder):
        s = host.Scene()
        s.set_camera_cbox()
        white, lamp = s.add_diffuse((0.7, 0.7, 0.7)), s.add_emissive((10, 10, 10))
        s.add_quad((-1, 0, 1), (2, 0, 0), (0, 0, -2), white)
        s.add_quad((-0.25, 1.58, 0.2), (0.5, 0, 0), (0, 0, -0.4), lamp)
        n = 7
        xs = np.linspace(-0.3, 0.3, n)
        pos = np.array([(x, 0.2 - 0.25 * (x * x + z * z), z) for z in xs for x in xs], np.float32)
        v0 = s.add_vertices(pos, np.tile(np.array([[0, 1, 0]], np.float32), (n * n, 1)), np.zeros((n * n, 2), np.float32))
        for j in range(n - 1):
            for i in range(n - 1):
                a, b, c, d = v0 + j * n + i, v0 + j * n + i + 1, v0 + (j + 1) * n + i, v0 + (j + 1) * n + i + 1
                s.add_triangle(a, c, b, white)
                s.add_triangle(b, c, d, white)
        if with_occluder:
            s.add_quad((-0.06, 0.8, 0.06), (0.12, 0, 0), (0, 0, -0.12), white)
        return s.compile()
    free, blocked = scene(False), scene(True)
    got, bits, lo, inv, st = build_grid(free)
    got2, bits2, lo2, inv2, st2 = build_grid(blocked)
    assert got == RES and got2 == RES
    def cell_of(p, lo_, inv_):
        c = ((np.array(p, np.float32) - lo_) * inv_).astype(np.int64)
        return (c[2] * RES + c[1]) * RES + c[0]
    apex = (0.0, 0.2 - 1e-5, 0.0)                                   # the middle vertex: six facets in one cell
    assert bits[cell_of(apex, lo, inv)] & 1
    assert not (bits2[cell_of(apex, lo2, inv2)] & 1)
    assert st2[2] < st[2]
    rng = np.random.default_rng(5)
    for cs in (free, blocked):
        tried, bad, _ = attack(cs, oracle, rng, per_shape=300)
        assert tried > 1000 and bad == 0
