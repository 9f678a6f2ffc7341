"""Second, independent restatement of the hot path: numpy, float64, vectorised over paths.

TEST INFRASTRUCTURE.  Written from the reference's GLSL only (`/root/reference/shader/*.glsl`, cited per function) and
sharing no code with `oracle/hj_oracle.c`: different language, different precision (every real is a float64; only the
u32 -> float conversion of `randUniformFloat` is rounded to binary32, because `float(randUint())` IS a binary32 value),
different control structure (all paths of a batch advance together; the BVH walk is a masked loop over the whole ray set).
Its job is to catch a transliteration error that the C oracle and the HIP kernels could share: the tests compare
per-function vectors of the oracle with it (<= 1e-5 relative), and the committed converged images made with it
(`make_converged.py`) with what the GPU renders.

Inputs are the reference's own data contract: the compiled scene arrays (SURVEY.md Appendix A layouts) and ImageBlocks.
"""
import ctypes as C

import numpy as np

EPS = 1e-4                                      # math.glsl:2
PI = 3.1415926535897932384626433832795          # math.glsl:1
TAG_SHIFT = 24                                  # src/main.rs:776
DIFFUSE, CBOARD, MIRROR, DIELECTRIC, EMISSIVE = range(5)   # src/main.rs:34-45 (strum discriminants)
INNER = 0xFFFFFFFF                              # scene.glsl:105 (`shapeIndex != -1`)


def _view(ptr, count, dtype, cols):
    if count == 0:
        return np.zeros((0, cols) if cols else (0,), dtype)
    n = count * max(cols, 1)
    a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), shape=(n,)).view(dtype)
    return a.reshape(count, cols) if cols else a


class Scene:
    """float64 copies of the scene buffers the shaders bind (scene.glsl:1-42, triangle.glsl:1-12, material.glsl:1-15)."""

    def __init__(self, cs):
        d = cs.desc
        f = lambda p, n, k: _view(p, n, np.float32, k).astype(np.float64)
        bu = _view(d.bvh, d.num_bvh_nodes, np.uint32, 8)
        bf = _view(d.bvh, d.num_bvh_nodes, np.float32, 8)
        self.bmin, self.bmax = bf[:, 0:3].astype(np.float64), bf[:, 4:7].astype(np.float64)
        self.shape, self.exit = bu[:, 3].astype(np.int64), bu[:, 7].astype(np.int64)
        self.leaf = bu[:, 3] != INNER
        self.spheres = f(d.spheres, d.num_spheres, 4)
        q = f(d.quads, d.num_quads, 12)
        self.q_o, self.q_e1, self.q_e2 = q[:, 0:3], q[:, 4:7], q[:, 8:11]
        self.tri = _view(d.triangles, d.num_triangles, np.uint32, 3).astype(np.int64)
        v = f(d.vertices, d.num_vertices, 8)
        self.v_pos, self.v_u, self.v_nrm, self.v_v = v[:, 0:3], v[:, 3], v[:, 4:7], v[:, 7]
        self.materials = _view(d.materials, d.num_materials, np.uint32, 0).astype(np.int64)
        e_u = _view(d.emitters, d.num_emitters, np.uint32, 4)
        self.em_shape = e_u[:, 0].astype(np.int64)
        self.em_pdf = e_u[:, 1].copy().view(np.float32).astype(np.float64)
        self.diffuse = f(d.diffuse, d.num_diffuse, 4)
        self.diffusecb = f(d.diffusecb, d.num_diffusecb, 8)
        self.dielectric = f(d.dielectric, d.num_dielectric, 4)
        self.emissive = f(d.emissive, d.num_emissive, 4)
        self.ns, self.nq, self.nt = len(self.spheres), len(q), len(self.tri)
        self.cam_pos = np.array(d.camera.position[:3], np.float64)
        self.cam_rot = np.array(d.camera.rotation[:4], np.float64)
        self.cam_fov = float(d.camera.fov)


# ------------------------------------------------------------------ rand.glsl

def seed_rng(seed):
    """rand.glsl:9-16 (Wang hash), u32 arithmetic."""
    s = np.asarray(seed, np.uint32).copy()
    s = (s ^ np.uint32(61)) ^ (s >> np.uint32(16))
    s = s * np.uint32(9)
    s = s ^ (s >> np.uint32(4))
    s = s * np.uint32(0x27d4eb2d)
    s = s ^ (s >> np.uint32(15))
    return s


class Rng:
    def __init__(self, state):
        self.s = np.asarray(state, np.uint32).copy()

    def uint(self, idx):
        """rand.glsl:2-7 on the paths `idx`."""
        x = self.s[idx]
        x ^= x << np.uint32(13)
        x ^= x >> np.uint32(17)
        x ^= x << np.uint32(5)
        self.s[idx] = x
        return x

    def uniform(self, idx):
        """rand.glsl:18-20: float(randUint()) * 2^-32; the conversion rounds to binary32."""
        return self.uint(idx).astype(np.float32).astype(np.float64) * (1.0 / 4294967296.0)


def cos_hemisphere(rng, idx):
    u, v = rng.uniform(idx), rng.uniform(idx)                       # rand.glsl:22-30
    r, th = np.sqrt(u), 2 * PI * v
    return np.stack([r * np.cos(th), r * np.sin(th), np.sqrt(np.maximum(0.0, 1 - u))], 1)


def uniform_sphere(rng, idx):
    u, v = rng.uniform(idx), rng.uniform(idx)                       # rand.glsl:32-40
    z, th = 2.0 * u - 1.0, 2 * PI * v
    r = np.sqrt(1 - z * z)
    return np.stack([r * np.cos(th), r * np.sin(th), z], 1)


def barycentric(rng, idx):
    u, v = rng.uniform(idx), rng.uniform(idx)                       # rand.glsl:42-50
    flip = u + v > 1
    u = np.where(flip, 1 - v, u)
    v = np.where(flip, 1 - u, v)                                    # sic: uses the NEW u, so v comes back unchanged
    return np.stack([u, v, 1 - u - v], 1)


# ------------------------------------------------------------------ shapes

def _dot(a, b):
    return (a * b).sum(1)


def _normalize(a):
    return a / np.sqrt(_dot(a, a))[:, None]


def _tri_vertices(sc, ix):
    return sc.tri[ix, 0], sc.tri[ix, 1], sc.tri[ix, 2]


def intersect_triangle(sc, o, d, tmin, tmax, ix):
    """triangle.glsl:15-52 -> (accepted, t, u, v)."""
    ia, ib, ic = _tri_vertices(sc, ix)
    a = sc.v_pos[ia]
    ab, ac = sc.v_pos[ib] - a, sc.v_pos[ic] - a
    n = np.cross(ab, ac)
    ro = o - a
    q = np.cross(ro, d)
    with np.errstate(divide="ignore", invalid="ignore"):
        dd = 1.0 / _dot(d, n)
        u, v = dd * _dot(-q, ac), dd * _dot(q, ab)
        t = dd * _dot(-n, ro)
        ok = ~((u < 0) | (v < 0) | (u + v > 1)) & (tmin <= t) & (t <= tmax)
    return ok, t, u, v


def intersect_sphere(sc, o, d, tmin, tmax, ix):
    """sphere.glsl:18-41 (3-argument overload)."""
    pos, r = sc.spheres[ix, 0:3], sc.spheres[ix, 3]
    l = o - pos
    b = 2 * _dot(d, l)
    c = _dot(l, l) - r * r
    disc = b * b - 4 * c
    ok0 = disc >= 0
    sq = np.sqrt(np.where(ok0, disc, 0.0))
    t0, t1 = -0.5 * (b + sq), -0.5 * (b - sq)
    in0 = (tmin <= t0) & (t0 <= tmax)
    in1 = (tmin <= t1) & (t1 <= tmax)
    return ok0 & (in0 | in1), np.where(in0, t0, t1), np.zeros_like(t0), np.zeros_like(t0)


def intersect_quad(sc, o, d, tmin, tmax, ix):
    """quad.glsl:7-25."""
    e1, e2 = sc.q_e1[ix], sc.q_e2[ix]
    n = np.cross(e1, e2)
    ro = o - sc.q_o[ix]
    q = np.cross(ro, d)
    with np.errstate(divide="ignore", invalid="ignore"):
        dd = 1.0 / _dot(d, n)
        u, v = dd * _dot(-q, e2), dd * _dot(q, e1)
        t = dd * _dot(-n, ro)
        ok = ~((u < 0) | (u > 1) | (v < 0) | (v > 1)) & (tmin <= t) & (t <= tmax)
    return ok, t, u, v


def intersect_shape(sc, o, d, tmin, tmax, shape):
    """The leaf dispatch of scene.glsl:106-114 for a vector of (ray, global shape index) pairs."""
    n = len(shape)
    ok, t, u, v = np.zeros(n, bool), np.zeros(n), np.zeros(n), np.zeros(n)
    for m, fn, base in ((shape < sc.ns, intersect_sphere, 0),
                        ((shape >= sc.ns) & (shape < sc.ns + sc.nq), intersect_quad, sc.ns),
                        (shape >= sc.ns + sc.nq, intersect_triangle, sc.ns + sc.nq)):
        if m.any():
            ok[m], t[m], u[m], v[m] = fn(sc, o[m], d[m], tmin[m], tmax[m], shape[m] - base)
    return ok, t, u, v


class Its:
    """struct Intersection (render.glsl:38-46) for n rays."""

    def __init__(self, n):
        self.id = np.full(n, -1, np.int64)
        self.t, self.u, self.v = np.zeros(n), np.zeros(n), np.zeros(n)
        self.p, self.n = np.zeros((n, 3)), np.zeros((n, 3))
        self.ft, self.fb = np.zeros((n, 3)), np.zeros((n, 3))   # frame = mat3(ft, fb, n)


def intersect_scene(sc, o, d, tmin, tmax, use_bvh=True, populate=True):
    """scene.glsl:97-175.  BVH branch: stackless pre-order walk with exit links, every ray of the set in lock step."""
    n = len(o)
    its = Its(n)
    tmax = np.array(tmax, np.float64, copy=True)
    if use_bvh:
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / d                                                    # :100
            off = -o * inv                                                   # :101
        cur = np.zeros(n, np.int64)
        count = len(sc.shape)
        act = np.arange(n)
        while act.size:
            c = cur[act]
            lf = sc.leaf[c]
            il, cl = act[lf], c[lf]
            if il.size:                                                      # :105-119
                shp = sc.shape[cl]
                ok, t, u, v = intersect_shape(sc, o[il], d[il], tmin[il], tmax[il], shp)
                h = il[ok]
                tmax[h] = t[ok] - EPS
                its.id[h], its.t[h], its.u[h], its.v[h] = shp[ok], t[ok], u[ok], v[ok]
                cur[il] = sc.exit[cl]
            ii, ci = act[~lf], c[~lf]
            if ii.size:                                                      # :120-131
                with np.errstate(invalid="ignore"):
                    tn = sc.bmin[ci] * inv[ii] + off[ii]
                    tp = sc.bmax[ci] * inv[ii] + off[ii]
                    t0 = np.fmax(np.fmax(np.fmin(tn[:, 0], tp[:, 0]), np.fmin(tn[:, 1], tp[:, 1])), np.fmin(tn[:, 2], tp[:, 2]))
                    t1 = np.fmin(np.fmin(np.fmax(tn[:, 0], tp[:, 0]), np.fmax(tn[:, 1], tp[:, 1])), np.fmax(tn[:, 2], tp[:, 2]))
                    enter = (t0 < t1 + EPS) & (t0 < tmax[ii]) & (t1 > tmin[ii])
                cur[ii] = np.where(enter, ci + 1, sc.exit[ci])
            act = act[cur[act] < count]
    else:                                                                    # :134-158
        if sc.ns > 100 or sc.nq > 100:
            return its
        for s in range(sc.ns + sc.nq + sc.nt):
            ok, t, u, v = intersect_shape(sc, o, d, tmin, tmax, np.full(n, s, np.int64))
            tmax[ok] = t[ok] - EPS
            its.id[ok], its.t[ok], its.u[ok], its.v[ok] = s, t[ok], u[ok], v[ok]
    if populate:
        hit = its.id >= 0
        its.p[hit] = o[hit] + its.t[hit][:, None] * d[hit]                   # :164
        _populate(sc, its, np.nonzero(hit)[0])
    return its


def _populate(sc, its, idx):
    ids = its.id[idx]
    ms = ids < sc.ns
    if ms.any():                                                             # sphere.glsl:43-52
        k = idx[ms]
        sp = sc.spheres[ids[ms]]
        nn = (its.p[k] - sp[:, 0:3]) / sp[:, 3:4]
        with np.errstate(invalid="ignore", divide="ignore"):
            t = _normalize(np.stack([-nn[:, 2], np.zeros(len(k)), nn[:, 0]], 1))
        its.n[k], its.ft[k], its.fb[k] = nn, t, np.cross(nn, t)
        ux = 0.5 + np.arctan2(nn[:, 2], nn[:, 0]) / (2 * PI)
        its.u[k] = np.where(np.isnan(ux), 0.0, ux)
        its.v[k] = 0.5 + np.arcsin(np.clip(nn[:, 1], -1, 1)) / PI
    mq = (ids >= sc.ns) & (ids < sc.ns + sc.nq)
    if mq.any():                                                             # quad.glsl:27-32 (uv stay the hit's)
        k = idx[mq]
        q = ids[mq] - sc.ns
        t, b = _normalize(sc.q_e1[q]), _normalize(sc.q_e2[q])
        its.ft[k], its.fb[k], its.n[k] = t, b, np.cross(t, b)
    mt = ids >= sc.ns + sc.nq
    if mt.any():                                                             # triangle.glsl:54-78
        k = idx[mt]
        ia, ib, ic = _tri_vertices(sc, ids[mt] - sc.ns - sc.nq)
        l0, l1, l2 = 1.0 - its.u[k] - its.v[k], its.u[k], its.v[k]
        nn = _normalize(sc.v_nrm[ia] * l0[:, None] + sc.v_nrm[ib] * l1[:, None] + sc.v_nrm[ic] * l2[:, None])
        its.u[k] = sc.v_u[ia] * l0 + sc.v_u[ib] * l1 + sc.v_u[ic] * l2
        its.v[k] = sc.v_v[ia] * l0 + sc.v_v[ib] * l1 + sc.v_v[ic] * l2
        bt = np.where((np.abs(nn[:, 0]) > np.abs(nn[:, 1]))[:, None], np.array([0.0, 1.0, 0.0]), np.array([1.0, 0.0, 0.0]))
        t = _normalize(np.cross(nn, bt))
        its.n[k], its.ft[k], its.fb[k] = nn, t, np.cross(nn, t)


# ------------------------------------------------------------------ emitters, BSDFs

def sample_shape(sc, shape, rng, idx):
    """scene.glsl:44-52 + the three sample* functions -> (p, n, pdf) for the paths `idx` (global indices into rng)."""
    n = len(shape)
    p, nn, pdf = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros(n)
    ms = shape < sc.ns
    mq = (shape >= sc.ns) & (shape < sc.ns + sc.nq)
    mt = shape >= sc.ns + sc.nq
    if ms.any():                                                             # sphere.glsl:54-58
        sp = sc.spheres[shape[ms]]
        nn[ms] = uniform_sphere(rng, idx[ms])
        p[ms] = sp[:, 0:3] + sp[:, 3:4] * nn[ms]
        pdf[ms] = 1.0 / (sp[:, 3] * sp[:, 3] * 4 * PI)
    if mq.any():                                                             # quad.glsl:34-45
        q = shape[mq] - sc.ns
        nrm = np.cross(sc.q_e1[q], sc.q_e2[q])
        area = np.sqrt(_dot(nrm, nrm))
        nn[mq] = nrm / area[:, None]
        u, v = rng.uniform(idx[mq]), rng.uniform(idx[mq])
        p[mq] = sc.q_o[q] + u[:, None] * sc.q_e1[q] + v[:, None] * sc.q_e2[q]
        pdf[mq] = 1.0 / area
    if mt.any():                                                             # triangle.glsl:81-102
        ia, ib, ic = _tri_vertices(sc, shape[mt] - sc.ns - sc.nq)
        a, b, c = sc.v_pos[ia], sc.v_pos[ib], sc.v_pos[ic]
        nrm = np.cross(b - a, c - a)
        area = np.sqrt(_dot(nrm, nrm)) / 2.0
        lam = barycentric(rng, idx[mt])
        nn[mt] = _normalize(sc.v_nrm[ia] * lam[:, 0:1] + sc.v_nrm[ib] * lam[:, 1:2] + sc.v_nrm[ic] * lam[:, 2:3])
        p[mt] = a * lam[:, 0:1] + b * lam[:, 1:2] + c * lam[:, 2:3]
        pdf[mt] = 1.0 / area
    return p, nn, pdf


def sample_emitter(sc, ref, rng, idx):
    """scene.glsl:54-89 -> (importance, shadow direction, shadow tMax); shadow tMin is 2 eps."""
    n = len(idx)
    xi = rng.uniform(idx)
    if len(sc.em_pdf) == 0:            # the reference reads emitters[0] out of bounds; defined as "no light" 
        rng.uint(idx); rng.uint(idx)
        return np.zeros((n, 3)), np.zeros((n, 3)), np.zeros(n)
    em = np.zeros(n, np.int64)
    done = np.zeros(n, bool)
    for i in range(len(sc.em_pdf)):                                          # :58-64
        xi = np.where(done, xi, xi - sc.em_pdf[i])
        take = ~done & (xi < 0)
        em[take] = i
        done |= take
    shape = sc.em_shape[em]
    p, nn, spdf = sample_shape(sc, shape, rng, idx)
    power = sc.emissive[sc.materials[shape] & ((1 << TAG_SHIFT) - 1), 0:3]   # :68-69
    dirv = p - ref
    dist = np.sqrt(_dot(dirv, dirv))
    with np.errstate(invalid="ignore", divide="ignore"):
        dirv = dirv / dist[:, None]
        cos_t = -_dot(dirv, nn)
        pdf = sc.em_pdf[em] * spdf * dist * dist / cos_t                     # :86
        imp = np.where((cos_t < 0)[:, None], 0.0, power / pdf[:, None])
    return imp, dirv, dist - EPS


def checkerboard(sc, idx, u, v):
    """materials/diffusecb.glsl:6-13."""
    m = sc.diffusecb[idx]
    fu, fv = 0.5 * u / m[:, 3], 0.5 * v / m[:, 7]
    fu, fv = fu - np.floor(fu), fv - np.floor(fv)
    return np.where(((fu < 0.5) ^ (fv < 0.5))[:, None], m[:, 4:7], m[:, 0:3])


def albedo(sc, tag, midx, u, v):
    col = np.zeros((len(tag), 3))
    md = tag == DIFFUSE
    col[md] = sc.diffuse[midx[md], 0:3]
    mc = tag == CBOARD
    if mc.any():
        col[mc] = checkerboard(sc, midx[mc], u[mc], v[mc])
    return col


def sample_bsdf(sc, tag, midx, wi, its_n, its_ft, its_fb, its_u, its_v, ext, rng, idx):
    """material.glsl:33-91 -> (weight, wo, new extinction, alive).  `idx` = global path indices (for the RNG)."""
    n = len(tag)
    w, wo = np.zeros((n, 3)), np.zeros((n, 3))
    ext = ext.copy()
    alive = np.ones(n, bool)
    md = (tag == DIFFUSE) | (tag == CBOARD)
    if md.any():                                                             # :37-46
        l = cos_hemisphere(rng, idx[md])
        wo[md] = its_ft[md] * l[:, 0:1] + its_fb[md] * l[:, 1:2] + its_n[md] * l[:, 2:3]
        w[md] = albedo(sc, tag[md], midx[md], its_u[md], its_v[md])
    mm = tag == MIRROR
    if mm.any():                                                             # :47-49  reflect(I, N) = I - 2 dot(N, I) N
        wo[mm] = wi[mm] - 2 * _dot(its_n[mm], wi[mm])[:, None] * its_n[mm]
        w[mm] = 1.0
    mg = tag == DIELECTRIC
    if mg.any():                                                             # :50-87
        k_ = np.nonzero(mg)[0]
        m = sc.dielectric[midx[mg]]
        eta = m[:, 3].copy()
        eta_inv = 1.0 / eta
        nrm = its_n[mg].copy()
        d = wi[mg]
        cos_i = -_dot(nrm, d)
        inside = cos_i > 0                                                   # sic
        fl = cos_i < 0
        eta = np.where(fl, eta_inv, eta)
        eta_inv = np.where(fl, 1.0 / eta, eta_inv)
        nrm[fl] = -nrm[fl]
        cos_i = np.where(fl, -cos_i, cos_i)
        kk = 1.0 - eta_inv * eta_inv * (1 - cos_i * cos_i)
        refl = d - 2 * _dot(nrm, d)[:, None] * nrm
        out = refl.copy()
        tr = kk > 0
        if tr.any():
            cos_o = np.sqrt(kk[tr])
            e, ci = eta[tr], cos_i[tr]
            rpar = (e * ci - cos_o) / (e * ci + cos_o)
            rorth = (ci - e * cos_o) / (ci + e * cos_o)
            fr = 0.5 * (rpar * rpar + rorth * rorth)
            xi = rng.uniform(idx[k_[tr]])
            refract = ~(xi < fr)
            par = d[tr] - _dot(d[tr], nrm[tr])[:, None] * nrm[tr]
            t_dir = eta_inv[tr][:, None] * par - cos_o[:, None] * nrm[tr]
            sel = np.nonzero(tr)[0][refract]
            out[sel] = t_dir[refract]
            inside[sel] = ~inside[sel]
        wo[mg] = out
        w[mg] = 1.0
        e_ = ext[mg]
        e_[inside] = m[inside, 0:3]
        ext[mg] = e_
    me = tag == EMISSIVE
    alive[me] = False                                                        # :88-89: weight 0, wo unwritten
    return w, wo, ext, alive


# ------------------------------------------------------------------ render.glsl

def camera_rays(sc, px, py, W, H):
    """render.glsl:26-36 + quaternion.glsl:1-19; (px, py) = pixel + sample offset."""
    s = np.tan(np.radians(0.5 * sc.cam_fov)) / (0.5 * W)
    x, y = (px - 0.5 * W) * s, (py - 0.5 * H) * s
    v = np.stack([x, -y, -np.ones_like(x)], 1)
    q = sc.cam_rot

    def qmul(a_xyz, a_w, b_xyz, b_w):
        return np.cross(a_xyz, b_xyz) + a_xyz * b_w + b_xyz * a_w, a_w * b_w - (a_xyz * b_xyz).sum(-1, keepdims=True)
    qx = np.broadcast_to(q[0:3], v.shape)
    qw = np.full((len(v), 1), q[3])
    t_xyz, t_w = qmul(qx, qw, v, np.zeros((len(v), 1)))
    r_xyz, _ = qmul(t_xyz, t_w, -qx, qw)
    return np.broadcast_to(sc.cam_pos, v.shape).copy(), _normalize(r_xyz)


def integrate(sc, o, d, rng_state, max_bounces=1000, rr_start=4, use_bvh=True):
    """render.glsl:81-147 for n camera rays -> (total rgb, first-hit normal, first-hit depth, rng state)."""
    n = len(o)
    rng = Rng(rng_state)
    total, normal, depth = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros(n)
    T, ext = np.ones((n, 3)), np.zeros((n, 3))
    o, d = o.copy(), d.copy()
    discrete = np.ones(n, bool)
    tmin = np.full(n, EPS)
    act = np.arange(n)
    for bounce in range(max_bounces):
        if act.size == 0:
            break
        its = intersect_scene(sc, o[act], d[act], tmin[act], np.full(act.size, np.inf), use_bvh)
        hit = its.id >= 0
        a = act[hit]                                                         # :94-96
        if a.size == 0:
            break
        iid, ip, inn = its.id[hit], its.p[hit], its.n[hit]
        ift, ifb, iu, iv = its.ft[hit], its.fb[hit], its.u[hit], its.v[hit]
        if bounce == 0:
            depth[a], normal[a] = its.t[hit], inn                            # :102-105
        mat = sc.materials[iid]
        tag, midx = mat >> TAG_SHIFT, mat & ((1 << TAG_SHIFT) - 1)
        dist = np.sqrt(_dot(o[a] - ip, o[a] - ip))
        T[a] *= np.exp(-ext[a] * dist[:, None])                              # :111-112
        me = (tag == EMISSIVE) & discrete[a]
        total[a[me]] += T[a[me]] * sc.emissive[midx[me], 0:3]                # :114-116
        mdif = (tag == DIFFUSE) | (tag == CBOARD)
        if mdif.any():                                                       # :117-126
            k = a[mdif]
            imp, sdir, stmax = sample_emitter(sc, ip[mdif], rng, k)
            want = (np.sqrt(_dot(imp, imp)) > EPS) & (_dot(sdir, inn[mdif]) > 0)
            if want.any():
                kk = k[want]
                sh = intersect_scene(sc, ip[mdif][want], sdir[want], np.full(kk.size, 2 * EPS), stmax[want], use_bvh, populate=False)
                free = sh.id < 0
                col = albedo(sc, tag[mdif][want], midx[mdif][want], iu[mdif][want], iv[mdif][want])
                f = _dot(inn[mdif][want], sdir[want])[:, None] * col / PI   # material.glsl:18-30
                total[kk[free]] += (T[kk] * f * imp[want])[free]
        w, wo, ext_new, alive = sample_bsdf(sc, tag, midx, d[a], inn, ift, ifb, iu, iv, ext[a], rng, a)
        T[a] *= w                                                            # :129
        ext[a] = ext_new
        d[a], o[a] = wo, ip                                                  # :130-133
        tmin[a] = 2 * EPS
        discrete[a] = ~mdif                                                  # :135
        a = a[alive]
        if bounce >= rr_start and a.size:                                    # :137-144 (`bounce > 3`)
            q = np.minimum(0.99, T[a].max(1))
            xi = rng.uniform(a)
            keep = ~(xi > q)
            T[a[keep]] /= q[keep][:, None]
            a = a[keep]
        act = a
    return total, normal, depth, rng.s


def integrate_blocks(sc, blocks, max_bounces=1000, rr_start=4, use_bvh=True):
    """render.glsl:149-175 for a list of ImageBlocks at once (their paths advance together) -> one sample array
    (dim_y, dim_x, 8) = (rgb, 1, normal, depth) per block."""
    parts, o_all, d_all, seed_all = [], [], [], []
    for b in blocks:
        dx, dy = int(b.dimension[0]), int(b.dimension[1])
        W, H = int(b.original_dimension[0]), int(b.original_dimension[1])
        ly, lx = np.mgrid[0:dy, 0:dx]
        lx, ly = lx.ravel(), ly.ravel()
        ok = (lx < W) & (ly < H)                                             # :152
        seed = (np.uint32(b.seed) + lx.astype(np.uint32) + ly.astype(np.uint32) * np.uint32(dx))   # :156
        o, d = camera_rays(sc, lx + int(b.origin[0]) + float(b.sample_offset[0]),
                           ly + int(b.origin[1]) + float(b.sample_offset[1]), W, H)
        parts.append((dy, dx, ok))
        o_all.append(o[ok]); d_all.append(d[ok]); seed_all.append(seed[ok])
    tot, nrm, dep, _ = integrate(sc, np.concatenate(o_all), np.concatenate(d_all), seed_rng(np.concatenate(seed_all)),
                                 max_bounces, rr_start, use_bvh)
    res, at = [], 0
    for dy, dx, ok in parts:
        k = int(ok.sum())
        out = np.zeros((dy * dx, 8))
        out[ok, 0:3], out[ok, 3], out[ok, 4:7], out[ok, 7] = tot[at:at + k], 1.0, nrm[at:at + k], dep[at:at + k]
        res.append(out.reshape(dy, dx, 8))
        at += k
    return res


def integrate_block(sc, b, max_bounces=1000, rr_start=4, use_bvh=True):
    return integrate_blocks(sc, [b], max_bounces, rr_start, use_bvh)[0]


# ------------------------------------------------------------------ reconstruction.glsl

def reconstruct_block(b, samples, accum, radius=2, stddev=0.5):
    """reconstruction.glsl:22-66: splat one block's samples (dim_y, dim_x, 8) into accum (H, W, 4), float64."""
    dy, dx = samples.shape[:2]
    H, W = accum.shape[:2]
    ox, oy = int(b.origin[0]), int(b.origin[1])
    g = -1.0 / (2 * stddev * stddev)
    c0 = np.exp(g * radius * radius)
    pad = np.zeros((dy + 4 * radius, dx + 4 * radius, 8))
    inb = np.zeros((dy + 4 * radius, dx + 4 * radius), bool)
    pad[2 * radius:2 * radius + dy, 2 * radius:2 * radius + dx] = samples
    inb[2 * radius:2 * radius + dy, 2 * radius:2 * radius + dx] = True
    ys, xs = slice(radius, radius + dy + 2 * radius), slice(radius, radius + dx + 2 * radius)   # local in [-r, D + r)
    n_c = pad[ys, xs, 4:7]                       # 0 outside the block (out-of-range imageLoad)
    out = np.zeros((dy + 2 * radius, dx + 2 * radius, 4))
    for ddx in range(-radius, radius + 1):       # :39-62, dx outer, dy inner
        for ddy in range(-radius, radius + 1):
            sx, sy = ddx + float(b.sample_offset[0]) - 0.5, ddy + float(b.sample_offset[1]) - 0.5
            wgt = np.exp(g * (sx * sx + sy * sy)) - c0
            if wgt < 0:
                continue
            ty = slice(radius + ddy, radius + ddy + dy + 2 * radius)
            tx = slice(radius + ddx, radius + ddx + dx + 2 * radius)
            tap, valid = pad[ty, tx], inb[ty, tx]
            dn = tap[..., 4:7] - n_c
            ww = wgt * np.exp(-((dn * dn).sum(-1) * 2))                     # albedo layer is always 0 (render.glsl:174)
            val = ww[..., None] * tap[..., 0:4]
            val[~valid | np.isnan(val).any(-1)] = 0.0
            out += val
    gy0, gx0 = oy - radius, ox - radius
    y0, y1, x0, x1 = max(gy0, 0), min(gy0 + out.shape[0], H), max(gx0, 0), min(gx0 + out.shape[1], W)
    accum[y0:y1, x0:x1] += out[y0 - gy0:y1 - gy0, x0 - gx0:x1 - gx0]
    return accum


def render_blocks(sc, blocks, W, H, max_bounces=1000, rr_start=4, use_bvh=True, batch=16):
    """The dispatch loop of src/main.rs:1316-1355 -> accumulation image (H, W, 4), float64.  `batch` blocks are
    integrated together (pure vectorisation: every path is independent); they are splatted one by one, in order."""
    accum = np.zeros((H, W, 4))
    blocks = list(blocks)
    for i in range(0, len(blocks), batch):
        chunk = blocks[i:i + batch]
        for b, smp in zip(chunk, integrate_blocks(sc, chunk, max_bounces, rr_start, use_bvh)):
            reconstruct_block(b, smp, accum)
    return accum
