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
