"""Scenes shared by the tests (built through the host Scene API, i.e. the reference's Shape / Material model)."""
import numpy as np

from hijiki_amd import host


def rich_scene(seed=7):
    """Every shape kind and every material kind: quad enclosure (one quad light), spheres (one of them a light, a mirror,
    a clear and a TINTED dielectric, a checkerboard), a triangle soup with random shading normals (one triangle light)."""
    rng = np.random.default_rng(seed)
    s = host.Scene()
    s.set_camera((0.05, 0.9, 3.3), (-0.02, 0.01, 0.0, 0.9997), 38.0)
    white, red, blue = s.add_diffuse((0.7, 0.7, 0.7)), s.add_diffuse((0.6, 0.1, 0.1)), s.add_diffuse((0.1, 0.2, 0.6))
    cb = s.add_diffuse_cboard((0.9, 0.9, 0.2), 0.13, (0.1, 0.2, 0.8), 0.21)
    mirror, glass = s.add_mirror(), s.add_dielectric(1.5)
    tinted = s.add_dielectric(1.33, extinction=(0.5, 1.5, 3.0))
    lq, ls, lt = s.add_emissive((20, 18, 15)), s.add_emissive((9, 12, 14)), s.add_emissive((14, 6, 6))
    s.add_quad((-1.2, 0, 1.2), (2.4, 0, 0), (0, 0, -2.4), white)
    s.add_quad((-1.2, 0, -1.2), (2.4, 0, 0), (0, 2.0, 0), cb)
    s.add_quad((-1.2, 0, 1.2), (0, 0, -2.4), (0, 2.0, 0), red)
    s.add_quad((1.2, 0, -1.2), (0, 0, 2.4), (0, 2.0, 0), blue)
    s.add_quad((-0.4, 1.99, -0.4), (0.8, 0, 0), (0, 0, 0.8), lq)
    s.add_sphere((0.55, 1.45, 0.3), 0.12, ls)
    s.add_sphere((-0.55, 0.35, 0.1), 0.35, mirror)
    s.add_sphere((0.45, 0.3, 0.45), 0.3, glass)
    s.add_sphere((0.0, 0.95, -0.3), 0.25, tinted)
    s.add_sphere((0.6, 0.25, -0.5), 0.25, cb)
    nv = 24
    pos = rng.uniform([-0.9, 0.05, -0.9], [0.9, 1.5, 0.9], (nv, 3)).astype(np.float32)
    nrm = rng.normal(size=(nv, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    base = s.add_vertices(pos, nrm, rng.uniform(0, 1, (nv, 2)).astype(np.float32))
    mats = [white, red, cb, mirror, glass, tinted, lt]
    for i in range(14):
        a, b, c = (int(x) for x in rng.choice(nv, 3, replace=False))
        s.add_triangle(base + a, base + b, base + c, mats[i % len(mats)])
    return s.compile()


FURNACE_L = 2.0          # radiance of the enclosure
FURNACE_RHO = 0.6        # albedo of the sphere inside


def furnace_scene():
    """A diffuse sphere (albedo rho) inside a closed box of six emissive quads of radiance L facing inward, camera
    inside the box.  Incident radiance on the sphere is L from every direction, so its outgoing radiance is exactly
    rho * L everywhere (the sphere is convex: it never sees itself), estimated by next-event estimation over uniformly
    sampled quads (sampleQuad is unbiased; triangle lights would bring in randBarycentric's degeneracy).  The bounce ray
    ends on a wall and adds nothing (wasDiscrete is false after a diffuse bounce): expected image = rho * L on the
    sphere, L where the camera sees the walls."""
    s = host.Scene()
    s.set_camera((0.0, 0.0, 2.5), (0.0, 0.0, 0.0, 1.0), 40.0)
    lamp = s.add_emissive((FURNACE_L, FURNACE_L, FURNACE_L))
    a = 3.0
    # cross(edge1, edge2) points INTO the box for every wall (sampleEmitter is one-sided: scene.glsl:80-83)
    s.add_quad((-a, -a, -a), (0, 0, 2 * a), (2 * a, 0, 0), lamp)      # floor y = -a, normal +y
    s.add_quad((-a, a, -a), (2 * a, 0, 0), (0, 0, 2 * a), lamp)       # ceiling, normal -y
    s.add_quad((-a, -a, -a), (2 * a, 0, 0), (0, 2 * a, 0), lamp)      # back z = -a, normal +z
    s.add_quad((-a, -a, a), (0, 2 * a, 0), (2 * a, 0, 0), lamp)       # front z = +a, normal -z
    s.add_quad((-a, -a, -a), (0, 2 * a, 0), (0, 0, 2 * a), lamp)      # left x = -a, normal +x
    s.add_quad((a, -a, -a), (0, 0, 2 * a), (0, 2 * a, 0), lamp)       # right, normal -x
    s.add_sphere((0.0, 0.0, 0.0), 0.6, s.add_diffuse((FURNACE_RHO,) * 3))
    return s.compile()


def furnace_masks(W, H):
    """Pixels safely inside the sphere's disc / safely on the walls for the furnace camera (fov 40 deg, distance 2.5)."""
    ys, xs = np.mgrid[0:H, 0:W]
    s = np.tan(np.radians(20.0)) / (0.5 * W)
    x, y = (xs + 0.5 - 0.5 * W) * s, (ys + 0.5 - 0.5 * H) * s
    r = np.sqrt(x * x + y * y)
    r_disc = 0.6 / np.sqrt(2.5 ** 2 - 0.6 ** 2)             # tan of the sphere's angular radius
    return r < 0.8 * r_disc, r > 1.25 * r_disc


def random_scene(seed):
    """Randomised scene through the whole Scene API: random mixes of triangles (a small random soup with random shading
    normals), spheres and quads, all five material kinds (dielectrics with and without extinction, several lights of
    different shape kinds), random camera."""
    rng = np.random.default_rng(1000 + seed)
    s = host.Scene()
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    q = q * 0.15 + np.array([0, 0, 0, 1.0])          # a mild random rotation around the cbox view
    q /= np.linalg.norm(q)
    s.set_camera((float(rng.uniform(-0.2, 0.2)), float(rng.uniform(0.7, 1.0)), float(rng.uniform(3.0, 3.6))),
                 tuple(float(x) for x in q), float(rng.uniform(25, 45)))
    mats = [s.add_diffuse(tuple(rng.uniform(0.1, 0.9, 3))) for _ in range(3)]
    mats.append(s.add_diffuse_cboard(tuple(rng.uniform(0.2, 0.9, 3)), float(rng.uniform(0.05, 0.3)),
                                     tuple(rng.uniform(0.1, 0.8, 3)), float(rng.uniform(0.05, 0.3))))
    mats.append(s.add_mirror())
    mats.append(s.add_dielectric(float(rng.uniform(1.2, 1.8))))
    mats.append(s.add_dielectric(1.5, extinction=tuple(rng.uniform(0.0, 2.0, 3))))
    lights = [s.add_emissive(tuple(rng.uniform(5, 25, 3))) for _ in range(2)]
    # an enclosure of quads so that paths bounce, one of them a light
    s.add_quad((-1.2, 0, 1.2), (2.4, 0, 0), (0, 0, -2.4), mats[0])
    s.add_quad((-1.2, 0, -1.2), (2.4, 0, 0), (0, 2.0, 0), mats[1])
    s.add_quad((-1.2, 0, 1.2), (0, 0, -2.4), (0, 2.0, 0), mats[3])
    s.add_quad((1.2, 0, -1.2), (0, 0, 2.4), (0, 2.0, 0), mats[2])
    s.add_quad((-0.4, 1.99, -0.4), (0.8, 0, 0), (0, 0, 0.8), lights[0])
    for _ in range(int(rng.integers(2, 6))):
        s.add_sphere(tuple(rng.uniform([-0.8, 0.2, -0.8], [0.8, 1.2, 0.8])), float(rng.uniform(0.1, 0.35)),
                     int(rng.choice(mats + lights[1:])))
    nv = int(rng.integers(12, 40))
    pos = rng.uniform([-0.9, 0.05, -0.9], [0.9, 1.5, 0.9], (nv, 3)).astype(np.float32)
    nrm = rng.normal(size=(nv, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    base = s.add_vertices(pos, nrm, rng.uniform(0, 1, (nv, 2)).astype(np.float32))
    for _ in range(int(rng.integers(8, 30))):
        a, b, c = (int(x) for x in rng.choice(nv, 3, replace=False))
        s.add_triangle(base + a, base + b, base + c, int(rng.choice(mats + lights[1:])))
    return s.compile()


def random_cluster_scene(seed, scale=1.0):
    """What random_scene lacks (round 5's sphere-guard bug lived there): many SMALL shapes.  An enclosure with a quad light and a
    triangle light, 30 ... 400 small spheres of all materials (some emissive, some overlapping, radii 0.005 ... 0.08) and a strip of
    small triangles.  Paths among small spheres leave the reference with directions far from unit length (it never re-normalises),
    which is what every exact shortcut of the HIP path has to survive.  `scale`: the whole scene (camera included) magnified - the
    reference's epsilons (1e-4) stay what they are, so 100 and 0.01 put them at the other ends of float rounding."""
    S = float(scale)
    rng = np.random.default_rng(5000 + seed)
    s = host.Scene()
    s.set_camera((S * float(rng.uniform(-0.2, 0.2)), S * float(rng.uniform(0.8, 1.1)), S * float(rng.uniform(3.2, 3.8))), (0.0, 0.0, 0.0, 1.0),
                 float(rng.uniform(28, 42)))
    mats = [s.add_diffuse(tuple(rng.uniform(0.2, 0.9, 3))) for _ in range(3)]
    mats += [s.add_mirror(), s.add_dielectric(float(rng.uniform(1.3, 1.7))), s.add_dielectric(1.5, extinction=tuple(rng.uniform(0.0, 1.5, 3)))]
    lights = [s.add_emissive(tuple(rng.uniform(8, 30, 3))) for _ in range(3)]
    sc3 = lambda v: tuple(S * float(x) for x in v)
    s.add_quad(sc3((-1.2, 0, 1.2)), sc3((2.4, 0, 0)), sc3((0, 0, -2.4)), mats[0])
    s.add_quad(sc3((-1.2, 0, -1.2)), sc3((2.4, 0, 0)), sc3((0, 2.0, 0)), mats[1])
    s.add_quad(sc3((-1.2, 0, 1.2)), sc3((0, 0, -2.4)), sc3((0, 2.0, 0)), mats[2])
    s.add_quad(sc3((1.2, 0, -1.2)), sc3((0, 0, 2.4)), sc3((0, 2.0, 0)), mats[0])
    s.add_quad(sc3((-0.35, 1.99, -0.35)), sc3((0.7, 0, 0)), sc3((0, 0, 0.7)), lights[0])
    pos = np.float32(S) * np.array([[0.6, 1.7, -0.9], [1.0, 1.7, -0.9], [0.8, 1.95, -0.6]], np.float32)      # a triangle light near a wall
    nrm = np.tile(np.array([[0, -0.6, 0.8]], np.float32), (3, 1))
    b = s.add_vertices(pos, nrm, np.zeros((3, 2), np.float32))
    s.add_triangle(b, b + 1, b + 2, lights[1])
    n = int(rng.integers(30, 400))
    rmax = float(rng.choice([0.02, 0.05, 0.08]))
    for k in range(n):
        mat = lights[2] if k % 97 == 5 else int(rng.choice(mats))
        s.add_sphere(sc3(rng.uniform([-1.0, 0.1, -1.0], [1.0, 1.6, 1.0])), S * float(rng.uniform(0.005, rmax)), mat)
    m = int(rng.integers(4, 24))                                                               # a strip of small triangles on the floor
    xs = np.linspace(-0.9, 0.9, m + 1).astype(np.float32)
    vp = np.float32(S) * np.array([[x, 0.02 + 0.03 * rng.random(), z] for x in xs for z in (-0.2, 0.0)], np.float32)
    vn = np.tile(np.array([[0, 1, 0]], np.float32), (len(vp), 1))
    b = s.add_vertices(vp, vn, rng.uniform(0, 1, (len(vp), 2)).astype(np.float32))
    for i in range(m):
        s.add_triangle(b + 2 * i, b + 2 * i + 1, b + 2 * i + 2, int(rng.choice(mats[:3])))
        s.add_triangle(b + 2 * i + 1, b + 2 * i + 3, b + 2 * i + 2, int(rng.choice(mats[:3])))
    return s.compile()


def nasty_scene(seed):
    """Degenerate geometry on purpose: zero-area and collinear triangles, duplicated shapes, a sphere of radius 0 and one with a
    negative radius, a quad with a zero edge, shapes far outside the rest, coincident coplanar quads (exact t ties), a light inside
    a wall.  Nothing here has a "right" image; the oracle's arithmetic defines it and the HIP path has to reproduce it."""
    rng = np.random.default_rng(7000 + seed)
    s = host.Scene()
    s.set_camera_cbox()
    mats = [s.add_diffuse(tuple(rng.uniform(0.2, 0.9, 3))) for _ in range(2)] + [s.add_mirror(), s.add_dielectric(1.5)]
    light = s.add_emissive((12, 11, 10))
    s.add_quad((-1.2, 0, 1.2), (2.4, 0, 0), (0, 0, -2.4), mats[0])
    s.add_quad((-1.2, 0, -1.2), (2.4, 0, 0), (0, 2.0, 0), mats[1])
    s.add_quad((-1.2, 0, -1.2), (2.4, 0, 0), (0, 2.0, 0), mats[0])                  # the same quad again: exact ties
    s.add_quad((-0.4, 1.99, -0.4), (0.8, 0, 0), (0, 0, 0.8), light)
    s.add_quad((-0.4, 2.0, -0.4), (0.8, 0, 0), (0, 0, 0.0), mats[0])                # a zero edge
    s.add_quad((-1.2, 0.5, -1.2), (0.3, 0, 0), (0, 0.3, 0), light)                  # a light in the back wall's plane
    s.add_sphere((0.3, 0.4, 0.2), 0.0, mats[0])
    s.add_sphere((-0.4, 0.5, 0.1), -0.3, mats[2])
    s.add_sphere((0.5, 0.35, -0.3), 0.35, mats[3])
    s.add_sphere((0.5, 0.35, -0.3), 0.35, mats[1])                                   # coincident spheres
    s.add_sphere((300.0, 200.0, -5000.0), 1.0, mats[0])                              # far away: the scene's box explodes
    nv = 24
    pos = rng.uniform([-0.9, 0.05, -0.9], [0.9, 1.5, 0.9], (nv, 3)).astype(np.float32)
    pos[3] = pos[2]                                                                   # duplicate vertices
    pos[7] = 0.5 * (pos[5] + pos[6])                                                  # collinear
    nrm = rng.normal(size=(nv, 3)).astype(np.float32)
    nrm[4] = 0                                                                        # a zero shading normal
    nrm[np.arange(nv) != 4] /= np.linalg.norm(nrm[np.arange(nv) != 4], axis=1, keepdims=True)
    b = s.add_vertices(pos, nrm, rng.uniform(0, 1, (nv, 2)).astype(np.float32))
    s.add_triangle(b + 2, b + 3, b + 9, mats[0])                                      # zero area (two equal vertices)
    s.add_triangle(b + 1, b + 1, b + 1, mats[1])                                      # a point
    s.add_triangle(b + 5, b + 6, b + 7, mats[0])                                      # collinear
    s.add_triangle(b + 4, b + 10, b + 11, mats[1])                                    # zero normal at a vertex
    for _ in range(int(rng.integers(6, 20))):
        a, c, d = (int(x) for x in rng.choice(nv, 3, replace=False))
        s.add_triangle(b + a, b + c, b + d, int(rng.choice(mats + [light])))
        if rng.random() < 0.3: s.add_triangle(b + a, b + c, b + d, int(rng.choice(mats)))   # duplicates
    return s.compile()


def _rotation(rng, amount=1.0):
    """A random rotation matrix (amount 0: identity ... 1: anything)."""
    q = rng.normal(size=4) * amount + np.array([0, 0, 0, 1.0]) * (1.0 - amount) * 4
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _grid_surface(fn, nu, nv, wrap_u, wrap_v):
    """Vertices of the parametric surface fn(u, v) -> (n, 3) on an nu x nv grid over [0, 1]^2 and its two triangles per cell."""
    us = np.arange(nu) / (nu if wrap_u else nu - 1)
    vs = np.arange(nv) / (nv if wrap_v else nv - 1)
    U, V = np.meshgrid(us, vs, indexing="ij")
    P = fn(U.ravel(), V.ravel())
    tri = []
    for i in range(nu if wrap_u else nu - 1):
        for j in range(nv if wrap_v else nv - 1):
            a, b = i * nv + j, ((i + 1) % nu) * nv + j
            c, d = i * nv + (j + 1) % nv, ((i + 1) % nu) * nv + (j + 1) % nv
            tri += [(a, b, d), (a, d, c)]
    return P, np.array(tri, np.int64)


def smooth_mesh_scene(seed):
    """Tessellated smooth surfaces in a box - what the light-shaft grid's cells on meshes are meant for, and what could break them:
    convex bodies (ellipsoids, capsules), bodies with concave parts (tori, the INSIDE of bowls, wavy height fields, saddles),
    bodies with creases (cylinders with caps, cones, boxes), at random resolution (coarse facets ... facets far smaller than a
    grid cell), rotation, size and winding, with smooth, faceted or random shading normals; one to three lights - quads or
    triangles, on the ceiling, on a side wall or floating and tilted - and walls made of quads or of triangles."""
    rng = np.random.default_rng(91000 + seed)
    s = host.Scene()
    q = rng.normal(size=4) * 0.1 + np.array([0, 0, 0, 1.0])
    q /= np.linalg.norm(q)
    s.set_camera((float(rng.uniform(-0.3, 0.3)), float(rng.uniform(0.6, 1.3)), float(rng.uniform(3.0, 3.6))), tuple(float(x) for x in q),
                 float(rng.uniform(28, 45)))
    diffuse = [s.add_diffuse(tuple(rng.uniform(0.2, 0.9, 3))) for _ in range(4)]
    others = [s.add_mirror(), s.add_dielectric(1.5), s.add_diffuse_cboard((0.8, 0.8, 0.3), 0.1, (0.2, 0.3, 0.8), 0.17)]
    lights = [s.add_emissive(tuple(rng.uniform(6, 25, 3))) for _ in range(3)]
    sc = float(rng.choice([1.0, 1.0, 1.0, 0.05, 12.0]))                  # the whole scene magnified (the reference's epsilons are not)

    def quad(o, e1, e2, m, as_triangles=False):
        o, e1, e2 = np.array(o, float) * sc, np.array(e1, float) * sc, np.array(e2, float) * sc
        if as_triangles:
            n = np.cross(e1, e2)
            n /= np.linalg.norm(n)
            b = s.add_vertices(np.array([o, o + e1, o + e1 + e2, o + e2], np.float32), np.tile(n, (4, 1)).astype(np.float32))
            s.add_triangles(np.array([[b, b + 1, b + 2], [b, b + 2, b + 3]]), m)
        else:
            s.add_quad(tuple(o), tuple(e1), tuple(e2), m)

    wt = rng.random() < 0.4                                               # walls of triangles (as cbox.obj has them)
    quad((-1.2, 0, 1.2), (2.4, 0, 0), (0, 0, -2.4), diffuse[0], wt)       # floor
    quad((-1.2, 2.0, -1.2), (2.4, 0, 0), (0, 0, 2.4), diffuse[0], wt)     # ceiling
    quad((-1.2, 0, -1.2), (2.4, 0, 0), (0, 2.0, 0), diffuse[1], wt)       # back
    quad((-1.2, 0, 1.2), (0, 0, -2.4), (0, 2.0, 0), diffuse[2], wt)       # left
    quad((1.2, 0, -1.2), (0, 0, 2.4), (0, 2.0, 0), diffuse[3], wt)        # right
    for li in range(int(rng.integers(1, 4))):
        kind = rng.integers(0, 4)
        w, d = rng.uniform(0.15, 0.9, 2)
        if kind == 0:                                                      # under the ceiling, facing down
            o = (rng.uniform(-1.1, 1.1 - w), 2.0 - rng.choice([0.01, 0.002, 0.1]), rng.uniform(-1.1, 1.1 - d))
            e1, e2 = (w, 0, 0), (0, 0, d)
        elif kind == 1:                                                    # on the left wall, facing +x
            o = (-1.2 + rng.choice([0.01, 0.003]), rng.uniform(0.2, 1.8 - w), rng.uniform(-1.0, 1.0 - d))
            e1, e2 = (0, w, 0), (0, 0, d)
        elif kind == 2:                                                    # on the back wall, facing +z
            o = (rng.uniform(-1.0, 1.0 - w), rng.uniform(0.2, 1.8 - d), -1.2 + 0.01)
            e1, e2 = (w, 0, 0), (0, d, 0)
        else:                                                              # floating and tilted
            R = _rotation(rng, 0.5)
            o = tuple(rng.uniform([-0.6, 1.0, -0.6], [0.4, 1.7, 0.4]))
            e1, e2 = tuple(R @ np.array([w * 0.6, 0, 0])), tuple(R @ np.array([0, 0, d * 0.6]))
        quad(o, e1, e2, lights[li], rng.random() < 0.5)

    for _ in range(int(rng.integers(1, 4))):
        kind = int(rng.integers(0, 8))
        n1, n2 = int(rng.choice([5, 8, 12, 20, 32, 56])), int(rng.choice([4, 7, 12, 20, 32, 56]))
        rx, ry, rz = rng.uniform(0.12, 0.45, 3)
        wrap = (True, False)
        if kind == 0:                                                      # ellipsoid
            fn = lambda u, v: np.stack([rx * np.cos(2 * np.pi * u) * np.sin(np.pi * v), ry * np.cos(np.pi * v), rz * np.sin(2 * np.pi * u) * np.sin(np.pi * v)], 1)
        elif kind == 1:                                                    # torus
            r2 = rng.uniform(0.2, 0.6) * rx
            wrap = (True, True)
            fn = lambda u, v: np.stack([(rx + r2 * np.cos(2 * np.pi * v)) * np.cos(2 * np.pi * u), r2 * np.sin(2 * np.pi * v), (rx + r2 * np.cos(2 * np.pi * v)) * np.sin(2 * np.pi * u)], 1)
        elif kind == 2:                                                    # a bowl: half an ellipsoid, seen from inside and outside
            fn = lambda u, v: np.stack([rx * np.cos(2 * np.pi * u) * np.sin(0.5 * np.pi * v), -ry * np.cos(0.5 * np.pi * v), rz * np.sin(2 * np.pi * u) * np.sin(0.5 * np.pi * v)], 1)
        elif kind == 3:                                                    # a wavy sheet
            wrap = (False, False)
            k1, k2, amp = rng.uniform(1, 5), rng.uniform(1, 5), rng.uniform(0.01, 0.15)
            fn = lambda u, v: np.stack([2 * rx * (u - 0.5) * 2, amp * np.sin(k1 * 2 * np.pi * u) * np.cos(k2 * 2 * np.pi * v), 2 * rz * (v - 0.5) * 2], 1)
        elif kind == 4:                                                    # a saddle
            wrap = (False, False)
            fn = lambda u, v: np.stack([2 * rx * (u - 0.5), ry * ((2 * u - 1) ** 2 - (2 * v - 1) ** 2), 2 * rz * (v - 0.5)], 1)
        elif kind == 5:                                                    # a cylinder with flat caps (creases at the rims)
            def fn(u, v):
                rr = np.where((v < 0.2) | (v > 0.8), np.minimum(v, 1 - v) / 0.2, 1.0)
                y = np.clip((v - 0.2) / 0.6, 0, 1) - 0.5
                return np.stack([rx * rr * np.cos(2 * np.pi * u), 2 * ry * y, rx * rr * np.sin(2 * np.pi * u)], 1)
        elif kind == 6:                                                    # a cone
            fn = lambda u, v: np.stack([rx * v * np.cos(2 * np.pi * u), ry * (1 - 2 * v), rx * v * np.sin(2 * np.pi * u)], 1)
        else:                                                              # a superellipsoid: from a rounded box to almost a box
            e = rng.uniform(0.15, 1.0)
            sp = lambda x: np.sign(x) * np.abs(x) ** e
            fn = lambda u, v: np.stack([rx * sp(np.cos(2 * np.pi * u)) * sp(np.sin(np.pi * v)), ry * sp(np.cos(np.pi * v)), rz * sp(np.sin(2 * np.pi * u)) * sp(np.sin(np.pi * v))], 1)
        P, tri = _grid_surface(fn, n1, n2, *wrap)
        R = _rotation(rng, float(rng.choice([0.0, 0.3, 1.0])))
        c = rng.uniform([-0.7, 0.35, -0.7], [0.7, 1.3, 0.7])
        if rng.random() < 0.3: c[1] = float(np.abs(P @ R.T)[:, 1].max()) + rng.choice([0.0, 0.001, 0.05])       # standing on the floor
        P = (P @ R.T + c) * sc
        if rng.random() < 0.3: tri = tri[:, ::-1]                         # the other winding
        # shading normals: smooth (area-weighted facet normals), the facet's own (vertices not shared), or random
        fnrm = np.cross(P[tri[:, 1]] - P[tri[:, 0]], P[tri[:, 2]] - P[tri[:, 0]])
        mode = rng.integers(0, 4)
        if mode == 1:
            P2 = P[tri].reshape(-1, 3)
            N2 = np.repeat(fnrm, 3, axis=0)
            tri = np.arange(len(P2)).reshape(-1, 3)
            P, N = P2, N2
        elif mode == 2:
            N = rng.normal(size=P.shape)
        else:
            N = np.zeros_like(P)
            for k in range(3): np.add.at(N, tri[:, k], fnrm)
        ln = np.linalg.norm(N, axis=1, keepdims=True)
        N = np.where(ln > 0, N / np.where(ln > 0, ln, 1), np.array([0.0, 1.0, 0.0]))
        b = s.add_vertices(P.astype(np.float32), N.astype(np.float32))
        mat = int(rng.choice(diffuse + diffuse + others + lights[2:]))
        s.add_triangles(tri + b, mat)
    if rng.random() < 0.3:
        s.add_sphere(tuple(rng.uniform([-0.8, 0.2, -0.8], [0.8, 1.2, 0.8]) * sc), float(rng.uniform(0.1, 0.3) * sc), int(rng.choice(diffuse + others)))
    return s.compile()
