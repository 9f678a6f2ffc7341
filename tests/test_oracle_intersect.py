"""Analytic intersection cases and BVH == linear-scan equivalence (shader/scene.glsl:97-175, shapes/*.glsl)."""
import numpy as np
import pytest

from hijiki_amd import host


def one_scene(build):
    s = host.Scene()
    build(s)
    return s.compile()


def ray(o, d, tmin=1e-4, tmax=np.inf):
    # NOTE: directions get a tiny tilt.  With an exactly zero component the reference's fused slab test
    # `bound * invDir + (-origin * invDir)` (scene.glsl:100-101,121-122) evaluates inf - inf = NaN and rejects
    # the root box; that degenerate behaviour is reproduced (see test_axis_aligned_ray_misses_through_the_bvh).
    d = np.asarray(d, np.float64) + np.array([1e-4, 2e-4, 3e-4])
    d = d / np.linalg.norm(d)
    return np.array([[*o, *d, tmin, tmax]], np.float32)


@pytest.fixture(scope="module")
def unit_scene():
    def build(s):
        m = s.add_diffuse((0.5, 0.5, 0.5))
        v0 = s.add_vertices([[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 0, 1]] * 3, [[0, 0], [1, 0], [0, 1]])
        s.add_triangle(v0, v0 + 1, v0 + 2, m)
        s.add_sphere((3, 0, 0), 1.0, m)
        s.add_quad((-3, -1, 0), (2, 0, 0), (0, 2, 0), m)
    return one_scene(build)


def test_triangle_hit_and_barycentrics(oracle, unit_scene):
    # global index space [spheres | quads | triangles] (scene.glsl:44-52): sphere 0, quad 1, triangle 2
    ids, t, u, v, full = oracle.intersect(unit_scene, ray((0.25, 0.5, 2), (0, 0, -1)), full=True)
    assert ids[0] == 2 and abs(t[0] - 2) < 1e-5
    assert abs(u[0] - 0.25) < 1e-3 and abs(v[0] - 0.5) < 1e-3            # raw barycentrics
    # populate overwrites uv with the interpolated texture coordinates (triangle.glsl:65-68) = (u, v) here
    assert abs(full[0, 6] - 0.25) < 1e-3 and abs(full[0, 7] - 0.5) < 1e-3
    np.testing.assert_allclose(full[0, 0:3], [0.25, 0.5, 0], atol=1e-3)
    np.testing.assert_allclose(full[0, 3:6], [0, 0, 1], atol=1e-6)       # shading normal
    # two-sided, no culling (triangle.glsl has no sign test)
    ids, t, *_ = oracle.intersect(unit_scene, ray((0.25, 0.5, -2), (0, 0, 1)))
    assert ids[0] == 2 and abs(t[0] - 2) < 1e-5
    # outside: u + v > 1
    ids, *_ = oracle.intersect(unit_scene, ray((0.75, 0.75, 2), (0, 0, -1)))
    assert ids[0] == -1


def test_tmin_tmax_window(oracle, unit_scene):
    o, d = (0.25, 0.25, 2), (0, 0, -1)
    assert oracle.intersect(unit_scene, ray(o, d, 1e-4, 1.9))[0][0] == -1
    assert oracle.intersect(unit_scene, ray(o, d, 2.1, 10))[0][0] == -1
    t = float(oracle.intersect(unit_scene, ray(o, d))[1][0])
    # inclusive on both ends (triangle.glsl:34); linear scan, because the strict slab test rejects the flat box
    assert oracle.intersect(unit_scene, ray(o, d, t, t), use_bvh=False)[0][0] == 2


def test_axis_aligned_ray_misses_through_the_bvh(oracle, unit_scene):
    r = np.array([[0.25, 0.5, 2, 0, 0, -1, 1e-4, np.inf]], np.float32)
    assert oracle.intersect(unit_scene, r, use_bvh=True)[0][0] == -1      # NaN slab -> root rejected
    assert oracle.intersect(unit_scene, r, use_bvh=False)[0][0] == 2      # the linear scan has no slab test


def test_parallel_ray_is_a_miss(oracle, unit_scene):
    # d . n == 0 -> 1/0 = inf, u/v become NaN or inf: all comparisons reject
    assert oracle.intersect(unit_scene, ray((0.2, 0.2, 1), (1, 0, 0)))[0][0] == -1


def test_sphere_near_then_far_root(oracle, unit_scene):
    ids, t, u, v, full = oracle.intersect(unit_scene, ray((3, 0, 5), (0, 0, -1)), full=True)
    assert ids[0] == 0 and abs(t[0] - 4) < 1e-4
    np.testing.assert_allclose(full[0, 3:6], [0, 0, 1], atol=2e-3)
    # from inside the near root is negative: far root is taken (sphere.glsl:34-38)
    # lat-long uv (sphere.glsl:47): n = (0,0,1) -> atan2(1,0)/2pi + .5 = .75 ; asin(0)/pi + .5 = .5
    assert abs(full[0, 6] - 0.75) < 1e-3 and abs(full[0, 7] - 0.5) < 1e-3
    ids, t, *_ = oracle.intersect(unit_scene, ray((3, 0, 0), (0, 0, -1)))
    assert ids[0] == 0 and abs(t[0] - 1) < 1e-4


def test_quad_bounds(oracle, unit_scene):
    ids, t, u, v = oracle.intersect(unit_scene, ray((-2.5, 0.5, 3), (0, 0, -1)))
    assert ids[0] == 1 and abs(t[0] - 3) < 1e-5 and abs(u[0] - 0.25) < 1e-3 and abs(v[0] - 0.75) < 1e-3
    assert oracle.intersect(unit_scene, ray((-0.9, 0.5, 3), (0, 0, -1)))[0][0] == -1     # u > 1


def test_bvh_equals_linear_scan_on_random_rays(oracle, cbox_small):
    r = np.random.default_rng(11)
    n = 3000
    o = np.stack([r.uniform(-0.95, 0.95, n), r.uniform(0.05, 1.5, n), r.uniform(-1.0, 0.95, n)], 1)
    d = r.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3], rays[:, 3:6], rays[:, 6], rays[:, 7] = o, d, 1e-4, np.inf
    ib, tb, *_ = oracle.intersect(cbox_small, rays, use_bvh=True)
    il, tl, *_ = oracle.intersect(cbox_small, rays, use_bvh=False)
    # hit/miss must agree; the winner may differ only among hits within M_EPS of each other
    # (tMax = t - eps after every hit: first VISITED wins)
    assert ((ib >= 0) == (il >= 0)).all()
    differ = ib != il
    assert differ.mean() < 0.01
    assert (np.abs(tb - tl)[differ] <= 2.1e-4).all()
    same = ~differ & (ib >= 0)
    assert (tb[same] == tl[same]).all()


def test_linear_scan_failsafe_over_100_spheres(oracle):
    def build(s):
        m = s.add_diffuse((1, 1, 1))
        for i in range(101):
            s.add_sphere((i * 3.0, 0, 0), 1.0, m)
    cs = one_scene(build)
    r = ray((0, 0, 5), (0, 0, -1))
    assert oracle.intersect(cs, r, use_bvh=True)[0][0] == 0
    assert oracle.intersect(cs, r, use_bvh=False)[0][0] == -1     # scene.glsl:135-138 "failsafe"


def test_golden_rays(oracle, cbox):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cbox_rays.npz"))
    ids, t, u, v, full = oracle.intersect(cbox, g["rays"], full=True)
    assert (ids == g["ids"]).all()
    assert (t.view(np.uint32) == g["t"].view(np.uint32)).all()
    assert (full.view(np.uint32) == g["full"].view(np.uint32)).all()


def test_camera_ray_through_image_centre(oracle, cbox):
    cam = cbox.desc.camera
    rays = oracle.camera_rays(cam, 256, 256, [[128.0, 128.0], [0.0, 128.0], [256.0, 128.0]])
    np.testing.assert_allclose(rays[0, :3], [0, 0.91, 5.41], atol=1e-6)
    # rotation about +x by -1.45 deg applied to (0,0,-1) (src/main.rs:417-418)
    a = np.radians(-1.45)
    np.testing.assert_allclose(rays[0, 3:], [0, np.sin(a), -np.cos(a)], atol=1e-6)
    # horizontal fov 27.7 deg: the left/right image edges are +-13.85 deg off axis (render.glsl:28)
    ang = np.degrees(np.arccos(np.clip(rays[1, 3:] @ rays[0, 3:], -1, 1)))
    assert abs(ang - 13.85) < 1e-3
    assert abs(rays[1, 3] + rays[2, 3]) < 1e-6 and rays[1, 3] < 0 < rays[2, 3]
