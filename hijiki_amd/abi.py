"""ctypes mirror of include/hijiki_hip.h (POD records + option/stat structs).

Pure declarations; nothing here computes.  Layouts are asserted against the
byte sizes of SURVEY.md Appendix A (= reference std430 / #[repr(C)] layouts).
"""
import ctypes as C

HJ_OK, HJ_ERR_INVALID, HJ_ERR_DEVICE, HJ_ERR_NOMEM, HJ_ERR_STATE, HJ_ERR_UNSUPPORTED = range(6)
STATUS_NAMES = {0: "HJ_OK", 1: "HJ_ERR_INVALID", 2: "HJ_ERR_DEVICE", 3: "HJ_ERR_NOMEM", 4: "HJ_ERR_STATE",
                5: "HJ_ERR_UNSUPPORTED"}

# MaterialType discriminants (reference src/main.rs:34-45)
MAT_DIFFUSE, MAT_DIFFUSECBOARD, MAT_MIRROR, MAT_DIELECTRIC, MAT_EMISSIVE = range(5)
MATERIAL_TAG_SHIFT = 24
MATERIAL_INDEX_MASK = 0x00FFFFFF
BVH_INNER = 0xFFFFFFFF
BVH_ROOT_EXIT = 1000000
BLOCK_SIZE = 128
RENDER_TIME_KERNELS = 1
RENDER_SPLIT_KERNELS = 2
RENDER_STATIC_DEAL = 4
RENDER_NO_DRAIN = 8
RENDER_NO_LIGHT_GRID = 16

f32, u32, u64 = C.c_float, C.c_uint32, C.c_uint64


class Camera(C.Structure):
    _fields_ = [("position", f32 * 4), ("rotation", f32 * 4), ("fov", f32), ("_pad", f32 * 3)]


class SceneInfo(C.Structure):
    _fields_ = [("camera", Camera), ("num_spheres", u32), ("num_quads", u32), ("num_triangles", u32),
                ("num_emitters", u32)]


class BvhNode(C.Structure):
    _fields_ = [("aabb_min", f32 * 3), ("shape_index", u32), ("aabb_max", f32 * 3), ("exit_index", u32)]


def direction_classes(mode):
    """hj_direction_classes (include/hijiki_hip.h): link orderings of a directional tree."""
    if mode <= 0 or mode > 8:
        return 1
    return 6 if mode == 8 else 1 << bin(mode & 7).count("1")


def ray_direction_class(mode, d):
    """hj_ray_direction_class for an (n, 3) float32 array of directions."""
    import numpy as np
    b = np.ascontiguousarray(d, np.float32).view(np.uint32).reshape(-1, 3)
    if mode <= 0 or mode > 8:
        return np.zeros(len(b), np.int64)
    if mode == 8:
        major = np.argmax(b & 0x7FFFFFFF, axis=1)          # (first of equals, as the C text)
        return 2 * major + (b[np.arange(len(b)), major] >> 31).astype(np.int64)
    cls, k = np.zeros(len(b), np.int64), 0
    for a in range(3):
        if mode & (1 << a):
            cls |= (b[:, a] >> 31).astype(np.int64) << k
            k += 1
    return cls


class Sphere(C.Structure):
    _fields_ = [("center", f32 * 3), ("radius", f32)]


class Quad(C.Structure):
    _fields_ = [("origin", f32 * 3), ("_pad1", f32), ("edge1", f32 * 3), ("_pad2", f32), ("edge2", f32 * 3),
                ("_pad3", f32)]


class Triangle(C.Structure):
    _fields_ = [("v", u32 * 3)]


class Vertex(C.Structure):
    _fields_ = [("pos", f32 * 3), ("u", f32), ("normal", f32 * 3), ("v", f32)]


class Emitter(C.Structure):
    _fields_ = [("shape", u32), ("pdf", f32), ("cdf", f32), ("_pad", f32)]


class Diffuse(C.Structure):
    _fields_ = [("color", f32 * 3), ("_pad", f32)]


class DiffuseCB(C.Structure):
    _fields_ = [("color_a", f32 * 3), ("scale_u", f32), ("color_b", f32 * 3), ("scale_v", f32)]


class Dielectric(C.Structure):
    _fields_ = [("extinction", f32 * 3), ("eta", f32)]


class Emissive(C.Structure):
    _fields_ = [("power", f32 * 3), ("_pad", f32)]


class ImageBlock(C.Structure):
    _fields_ = [("id", u32), ("seed", u32), ("origin", u32 * 2), ("dimension", u32 * 2),
                ("original_dimension", u32 * 2), ("sample_offset", f32 * 2)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("camera", Camera),
        ("bvh", C.POINTER(BvhNode)), ("num_bvh_nodes", C.c_size_t),
        ("spheres", C.POINTER(Sphere)), ("num_spheres", C.c_size_t),
        ("quads", C.POINTER(Quad)), ("num_quads", C.c_size_t),
        ("triangles", C.POINTER(Triangle)), ("num_triangles", C.c_size_t),
        ("vertices", C.POINTER(Vertex)), ("num_vertices", C.c_size_t),
        ("materials", C.POINTER(u32)), ("num_materials", C.c_size_t),
        ("emitters", C.POINTER(Emitter)), ("num_emitters", C.c_size_t),
        ("diffuse", C.POINTER(Diffuse)), ("num_diffuse", C.c_size_t),
        ("diffusecb", C.POINTER(DiffuseCB)), ("num_diffusecb", C.c_size_t),
        ("dielectric", C.POINTER(Dielectric)), ("num_dielectric", C.c_size_t),
        ("emissive", C.POINTER(Emissive)), ("num_emissive", C.c_size_t),
    ]


class RenderOpts(C.Structure):
    _fields_ = [("use_bvh", u32), ("recon_radius", u32), ("recon_stddev", f32), ("max_bounces", u32),
                ("rr_start", u32), ("batch_blocks", u32), ("flags", u32), ("_reserved", u32)]

    @staticmethod
    def default():
        return RenderOpts(use_bvh=1, recon_radius=2, recon_stddev=0.5, max_bounces=1000, rr_start=4, batch_blocks=0)


class RenderStats(C.Structure):
    _fields_ = [("paths", u64), ("closest_rays", u64), ("shadow_rays", u64), ("batches", u64),
                ("bounce_rounds", u64), ("trace_closest_ms", C.c_double), ("trace_shadow_ms", C.c_double),
                ("shade_ms", C.c_double), ("reconstruct_ms", C.c_double), ("total_ms", C.c_double),
                ("closest_launches", u64), ("path_ms", C.c_double), ("path_launches", u64),
                ("hits", u64), ("unoccluded_shadow_rays", u64), ("path_busy_ms", C.c_double),
                ("shadow_rays_proven_free", u64)]


# byte sizes of SURVEY.md Appendix A
_SIZES = {Camera: 48, SceneInfo: 64, BvhNode: 32, Sphere: 16, Quad: 48, Triangle: 12, Vertex: 32, Emitter: 16,
          Diffuse: 16, DiffuseCB: 32, Dielectric: 16, Emissive: 16, ImageBlock: 40}
for _t, _n in _SIZES.items():
    assert C.sizeof(_t) == _n, (_t.__name__, C.sizeof(_t), _n)


class HijikiError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status
