"""ctypes loader for the CPU oracle (oracle/hj_oracle.c).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libhj_oracle.so")
_LIB = None


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("paths", "closest_calls", "shadow_calls", "nodes", "tri_tests",
                                           "sphere_tests", "quad_tests", "hits", "nee_evals", "shadow_nodes",
                                           "shadow_tri_tests", "shadow_sphere_tests", "shadow_quad_tests",
                                           "shadow_hits")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build():
    subprocess.check_call(["make", "-s", "oracle"], cwd=os.path.dirname(_HERE))


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            build()
        from hijiki_amd import abi
        L = C.CDLL(LIB_PATH)
        fp, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        L.hjo_render_blocks.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(abi.ImageBlock), C.c_size_t,
                                        C.POINTER(abi.RenderOpts), C.c_uint32, C.c_uint32, fp, C.c_int,
                                        C.POINTER(Counters), C.POINTER(C.c_double)]
        L.hjo_block_seed.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.hjo_block_seed.restype = C.c_uint32
        L.hjo_pass_offset.argtypes = [C.c_uint64, C.c_uint32, fp]
        L.hjo_pass_offset.restype = None
        L.hjo_make_blocks.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32,
                                      C.POINTER(abi.ImageBlock), C.c_size_t]
        L.hjo_make_blocks.restype = C.c_size_t
        L.hjo_rng_seed.argtypes = [C.c_uint32]
        L.hjo_rng_seed.restype = C.c_uint32
        L.hjo_rng_next.argtypes = [u32p]
        L.hjo_rng_next.restype = C.c_uint32
        L.hjo_rng_float.argtypes = [u32p]
        L.hjo_rng_float.restype = C.c_float
        L.hjo_exp.argtypes = [C.c_float]
        L.hjo_exp.restype = C.c_float
        L.hjo_sincos2pi.argtypes = [C.c_float, fp]
        L.hjo_sincos2pi.restype = None
        L.hjo_atan2.argtypes = [C.c_float, C.c_float]
        L.hjo_atan2.restype = C.c_float
        L.hjo_asin.argtypes = [C.c_float]
        L.hjo_asin.restype = C.c_float
        for name in ("hjo_cos_hemisphere", "hjo_barycentric", "hjo_uniform_sphere"):
            getattr(L, name).argtypes = [u32p, fp]
            getattr(L, name).restype = None
        L.hjo_intersect.argtypes = [C.POINTER(abi.SceneDesc), C.c_int, fp, C.c_size_t, fp, fp]
        L.hjo_camera_rays.argtypes = [C.POINTER(abi.Camera), C.c_uint32, C.c_uint32, fp, C.c_size_t, fp]
        L.hjo_camera_rays.restype = None
        L.hjo_integrate_block.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(abi.ImageBlock),
                                          C.POINTER(abi.RenderOpts), fp, C.POINTER(Counters)]
        L.hjo_reconstruct_block.argtypes = [C.POINTER(abi.ImageBlock), C.POINTER(abi.RenderOpts), fp, fp, C.c_uint32,
                                            C.c_uint32]
        L.hjo_recon_gauss.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int]
        L.hjo_recon_gauss.restype = C.c_float
        L.hjo_dielectric_probe.argtypes = [C.c_float, fp, fp, u32p, fp]
        L.hjo_dielectric_probe.restype = None
        L.hjo_shade_probe.argtypes = [C.POINTER(abi.SceneDesc), fp, u32p, C.c_size_t, fp]
        L.hjo_set_directional_bvh.argtypes = [C.c_int, C.c_void_p]
        L.hjo_set_directional_bvh.restype = None
        L.hjo_sizeof_counters.restype = C.c_size_t
        assert L.hjo_sizeof_counters() == C.sizeof(Counters)
        _LIB = L
    return _LIB


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def render_blocks(compiled, blocks, width, height, opts=None, nthreads=None, accum=None):
    """Render `blocks` (ctypes array of ImageBlock) in order.  Returns (accum[H,W,4], counters dict, seconds)."""
    from hijiki_amd import abi
    L = lib()
    opts = opts or abi.RenderOpts.default()
    nthreads = nthreads or os.cpu_count() or 1
    if accum is None:
        accum = np.zeros((height, width, 4), np.float32)
    ctr, secs = Counters(), C.c_double(0)
    rc = L.hjo_render_blocks(C.byref(compiled.desc), blocks, len(blocks), C.byref(opts), width, height, _fp(accum),
                             nthreads, C.byref(ctr), C.byref(secs))
    if rc != 0:
        raise RuntimeError(f"hjo_render_blocks failed: {abi.STATUS_NAMES.get(rc, rc)}")
    return accum, ctr.as_dict(), secs.value


def make_blocks(width, height, spp, master_seed, pass_begin=0, pass_end=None, block_size=128):
    from hijiki_amd import abi
    L = lib()
    pass_end = spp if pass_end is None else pass_end
    n = L.hjo_make_blocks(width, height, block_size, master_seed, pass_begin, pass_end, None, 0)
    arr = (abi.ImageBlock * n)()
    L.hjo_make_blocks(width, height, block_size, master_seed, pass_begin, pass_end, arr, n)
    return arr


def resolve(accum):
    """rgb / w (src/main.rs:1399)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return accum[..., :3] / accum[..., 3:4]


def intersect(compiled, rays, use_bvh=True, full=False):
    """rays (n,8) f32 -> ids (n,) int32, t,u,v (n,) [, full (n,16)]."""
    rays = np.ascontiguousarray(rays, np.float32)
    n = len(rays)
    hits = np.zeros((n, 4), np.float32)
    fullbuf = np.zeros((n, 16), np.float32) if full else None
    lib().hjo_intersect(C.byref(compiled.desc), int(use_bvh), _fp(rays), n, _fp(hits),
                        _fp(fullbuf) if full else None)
    ids = hits[:, 0].copy().view(np.int32)
    return (ids, hits[:, 1], hits[:, 2], hits[:, 3]) + ((fullbuf,) if full else ())


def shade_probe(compiled, rays, rng_states):
    """One shading step per ray (hjo_shade_probe): returns the (n, 20) float32 record array, ids (int32), RNG after (uint32)."""
    rays = np.ascontiguousarray(rays, np.float32)
    rng = np.ascontiguousarray(rng_states, np.uint32)
    out = np.zeros((len(rays), 20), np.float32)
    lib().hjo_shade_probe(C.byref(compiled.desc), _fp(rays), rng.ctypes.data_as(C.POINTER(C.c_uint32)), len(rays), _fp(out))
    return out, out[:, 0].copy().view(np.int32), out[:, 15].copy().view(np.uint32)


def camera_rays(camera, width, height, pix_xy):
    pix_xy = np.ascontiguousarray(pix_xy, np.float32).reshape(-1, 2)
    out = np.zeros((len(pix_xy), 6), np.float32)
    lib().hjo_camera_rays(C.byref(camera), width, height, _fp(pix_xy), len(pix_xy), _fp(out))
    return out


def integrate_block(compiled, block, opts=None):
    """Per-path samples of one block: (dim_y, dim_x, 8) = (rgb, w, normal, depth); counters."""
    from hijiki_amd import abi
    opts = opts or abi.RenderOpts.default()
    out = np.zeros((block.dimension[1], block.dimension[0], 8), np.float32)
    ctr = Counters()
    lib().hjo_integrate_block(C.byref(compiled.desc), C.byref(block), C.byref(opts), _fp(out), C.byref(ctr))
    return out, ctr.as_dict()


def logged_rays(compiled, blocks, opts=None):
    """Every ray the oracle traces for `blocks` (hjo_set_ray_log around hjo_integrate_block, single-threaded): (n, 11) float32 =
    o, d, tMin, tMax, kind (0 closest-hit, 1 shadow), id of the shape hit or -1, index of the emitter a shadow ray aims at (else -1).  Directions are whatever the reference's
    arithmetic made them - not always unit vectors."""
    import tempfile
    L = lib()
    L.hjo_set_ray_log.argtypes = [C.c_char_p]
    L.hjo_set_ray_log.restype = None
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "rays.bin")
        L.hjo_set_ray_log(path.encode())
        try:
            for b in blocks:
                integrate_block(compiled, b, opts)
        finally:
            L.hjo_set_ray_log(None)
        return np.fromfile(path, np.float32).reshape(-1, 11)


def reconstruct_block(block, samples, accum, opts=None):
    from hijiki_amd import abi
    opts = opts or abi.RenderOpts.default()
    samples = np.ascontiguousarray(samples, np.float32)
    h, w = accum.shape[:2]
    lib().hjo_reconstruct_block(C.byref(block), C.byref(opts), _fp(samples), _fp(accum), w, h)
    return accum
