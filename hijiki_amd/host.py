"""Host-side mirror of Hijiki's Scene / Shape / Material API and scene compiler.

Thin ctypes binding over libhijiki_host.so (C++; include/hijiki_host.h), which
mirrors `Scene`, `Scene::compile` and `ImageBlockGenerator` of the reference
(src/main.rs:162-357, 619-682).  No GPU needed.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_LIB = None
_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "lib", "libhijiki_host.so")

SYNTH_CBOX, SYNTH_CBOX_SPHERES, SYNTH_CBOX_MESH, SYNTH_CBOX_CBOARD = range(4)


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise ImportError(f"{HOST_LIB_PATH} missing: run `make host` (or __graft_entry__.build())")
        L = C.CDLL(HOST_LIB_PATH)
        vp, f3 = C.c_void_p, C.POINTER(C.c_float)
        L.hjh_last_error.restype = C.c_char_p
        L.hjh_scene_create.argtypes = [C.POINTER(vp)]
        L.hjh_scene_destroy.argtypes = [vp]
        L.hjh_scene_destroy.restype = None
        L.hjh_scene_set_camera.argtypes = [vp, f3, f3, C.c_float]
        L.hjh_scene_set_camera_cbox.argtypes = [vp]
        L.hjh_scene_add_diffuse.argtypes = [vp, f3]
        L.hjh_scene_add_diffuse_cboard.argtypes = [vp, f3, C.c_float, f3, C.c_float]
        L.hjh_scene_add_mirror.argtypes = [vp]
        L.hjh_scene_add_dielectric.argtypes = [vp, f3, C.c_float]
        L.hjh_scene_add_emissive.argtypes = [vp, f3]
        L.hjh_scene_add_vertices.argtypes = [vp, C.POINTER(abi.Vertex), C.c_size_t]
        L.hjh_scene_add_vertices.restype = C.c_long
        L.hjh_scene_add_sphere.argtypes = [vp, f3, C.c_float, C.c_int]
        L.hjh_scene_add_quad.argtypes = [vp, f3, f3, f3, C.c_int]
        L.hjh_scene_add_triangle.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        L.hjh_scene_add_triangles.argtypes = [vp, C.POINTER(C.c_uint32), C.c_size_t, C.c_int]
        L.hjh_scene_num_shapes.argtypes = [vp]
        L.hjh_scene_num_shapes.restype = C.c_size_t
        L.hjh_scene_compile.argtypes = [vp, C.POINTER(vp)]
        L.hjh_scene_compile_shapes.argtypes = [vp, C.POINTER(vp)]
        L.hjh_compiled_destroy.argtypes = [vp]
        L.hjh_compiled_destroy.restype = None
        L.hjh_compiled_desc.argtypes = [vp, C.POINTER(abi.SceneDesc)]
        L.hjh_compiled_packed_size.argtypes = [vp]
        L.hjh_compiled_packed_size.restype = C.c_size_t
        L.hjh_compiled_pack.argtypes = [vp, vp, C.c_size_t]
        L.hjh_compiled_set_bvh.argtypes = [vp, C.POINTER(abi.BvhNode), C.c_size_t]
        L.hjh_compiled_tune_bvh.argtypes = [vp, C.c_int, C.c_size_t]
        L.hjh_compiled_directional_bvh.argtypes = [vp, C.c_int, C.c_size_t, C.c_int, C.c_int, C.POINTER(abi.BvhNode), C.c_size_t]
        L.hjh_num_blocks_per_pass.argtypes = [C.c_uint32] * 3
        L.hjh_num_blocks_per_pass.restype = C.c_size_t
        L.hjh_make_blocks.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32,
                                      C.POINTER(abi.ImageBlock), C.c_size_t]
        L.hjh_make_blocks.restype = C.c_size_t
        L.hjh_scene_from_obj.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.hjh_scene_put_cbox_spheres.argtypes = [vp]
        L.hjh_write_exr.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, f3]
        L.hjh_write_pfm.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, f3]
        L.hjh_write_png.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, f3]
        L.hjh_scene_make_synthetic.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.POINTER(vp)]
        L.hj_block_seed.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.hj_block_seed.restype = C.c_uint32
        L.hj_pass_offset.argtypes = [C.c_uint64, C.c_uint32, C.POINTER(C.c_float)]
        L.hj_pass_offset.restype = None
        L.hj_block_owner.argtypes = [C.c_uint32] * 5
        L.hj_block_owner.restype = C.c_uint32
        _LIB = L
    return _LIB


def _f3(v, n=3):
    a = (C.c_float * n)(*[float(x) for x in v])
    return a


def _check(rc):
    if rc != abi.HJ_OK:
        raise abi.HijikiError(rc, lib().hjh_last_error().decode())


def _index(rc):
    if rc < 0:
        raise abi.HijikiError(-rc, lib().hjh_last_error().decode())
    return rc


class Scene:
    """`struct Scene` (src/main.rs:162-170): camera, (shape, material) objects, vertices, materials."""

    def __init__(self, _handle=None):
        self._h = C.c_void_p()
        self._destroy = lib().hjh_scene_destroy      # bound now: module globals may be gone at interpreter exit
        if _handle is not None:
            self._h = _handle
        else:
            _check(lib().hjh_scene_create(C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    @staticmethod
    def synthetic(kind=SYNTH_CBOX, mesh_triangles=0, gen_seed=1):
        h = C.c_void_p()
        _check(lib().hjh_scene_make_synthetic(kind, mesh_triangles, gen_seed, C.byref(h)))
        return Scene(_handle=h)

    @staticmethod
    def from_obj(path):
        """`Scene::from_obj` (src/main.rs:414-530)."""
        h = C.c_void_p()
        _check(lib().hjh_scene_from_obj(os.fsencode(path), C.byref(h)))
        return Scene(_handle=h)

    def put_cbox_spheres(self):
        """`--put-cbox-spheres` (src/main.rs:1463-1483)."""
        _check(lib().hjh_scene_put_cbox_spheres(self._h))

    def set_camera(self, position, rotation_xyzw, fov_deg):
        _check(lib().hjh_scene_set_camera(self._h, _f3(position), _f3(rotation_xyzw, 4), fov_deg))

    def set_camera_cbox(self):
        _check(lib().hjh_scene_set_camera_cbox(self._h))

    # `enum Material` (src/main.rs:38-44)
    def add_diffuse(self, color):
        return _index(lib().hjh_scene_add_diffuse(self._h, _f3(color)))

    def add_diffuse_cboard(self, color1, scale_u, color2, scale_v):
        return _index(lib().hjh_scene_add_diffuse_cboard(self._h, _f3(color1), scale_u, _f3(color2), scale_v))

    def add_mirror(self):
        return _index(lib().hjh_scene_add_mirror(self._h))

    def add_dielectric(self, eta_ratio, extinction=(0.0, 0.0, 0.0)):
        return _index(lib().hjh_scene_add_dielectric(self._h, _f3(extinction), eta_ratio))

    def add_emissive(self, power):
        return _index(lib().hjh_scene_add_emissive(self._h, _f3(power)))

    def add_vertices(self, pos, normal, uv=None):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        normal = np.ascontiguousarray(normal, np.float32).reshape(-1, 3)
        n = len(pos)
        uv = np.zeros((n, 2), np.float32) if uv is None else np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        rec = np.zeros((n, 8), np.float32)
        rec[:, 0:3], rec[:, 3], rec[:, 4:7], rec[:, 7] = pos, uv[:, 0], normal, uv[:, 1]
        first = lib().hjh_scene_add_vertices(self._h, rec.ctypes.data_as(C.POINTER(abi.Vertex)), n)
        if first < 0:
            raise abi.HijikiError(-first, lib().hjh_last_error().decode())
        return first

    # `enum Shape` (src/main.rs:47-52)
    def add_sphere(self, center, radius, material):
        _check(lib().hjh_scene_add_sphere(self._h, _f3(center), radius, material))

    def add_quad(self, origin, edge1, edge2, material):
        _check(lib().hjh_scene_add_quad(self._h, _f3(origin), _f3(edge1), _f3(edge2), material))

    def add_triangle(self, a, b, c, material):
        _check(lib().hjh_scene_add_triangle(self._h, a, b, c, material))

    def add_triangles(self, abc, material):
        abc = np.ascontiguousarray(abc, np.uint32).reshape(-1, 3)
        _check(lib().hjh_scene_add_triangles(self._h, abc.ctypes.data_as(C.POINTER(C.c_uint32)), len(abc), material))

    @property
    def num_shapes(self):
        return lib().hjh_scene_num_shapes(self._h)

    def compile(self, with_tree=True):
        """`Scene::compile` (src/main.rs:173-357).  with_tree=False: the arrays without a tree (hjh_scene_compile_shapes), for
        `device.Renderer.build_bvh` to build it on the device."""
        h = C.c_void_p()
        _check((lib().hjh_scene_compile if with_tree else lib().hjh_scene_compile_shapes)(self._h, C.byref(h)))
        return CompiledScene(h)


def _as_np(ptr, count, dtype, cols):
    if count == 0:
        return np.zeros((0, cols) if cols else (0,), dtype)
    n = count * (cols or 1)
    arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(n,))
    return arr.reshape(count, cols) if cols else arr


class CompiledScene:
    """`struct CompiledScene` (src/main.rs:376-397); `.desc` is the hj_scene_desc the device library uploads."""

    def __init__(self, handle):
        self._h = handle
        self._destroy = lib().hjh_compiled_destroy
        self.desc = abi.SceneDesc()
        _check(lib().hjh_compiled_desc(self._h, C.byref(self.desc)))

    def __del__(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    # numpy views (borrowed; valid while self lives)
    @property
    def bvh(self):
        """(N, 8) uint32 view: [min.xyz bits, shape_index, max.xyz bits, exit_index]."""
        return _as_np(self.desc.bvh, self.desc.num_bvh_nodes, np.uint32, 8)

    @property
    def bvh_f32(self):
        return _as_np(self.desc.bvh, self.desc.num_bvh_nodes, np.float32, 8)

    @property
    def triangles(self):
        return _as_np(self.desc.triangles, self.desc.num_triangles, np.uint32, 3)

    @property
    def vertices(self):
        return _as_np(self.desc.vertices, self.desc.num_vertices, np.float32, 8)

    @property
    def spheres(self):
        return _as_np(self.desc.spheres, self.desc.num_spheres, np.float32, 4)

    @property
    def quads(self):
        return _as_np(self.desc.quads, self.desc.num_quads, np.float32, 12)

    @property
    def materials(self):
        return _as_np(self.desc.materials, self.desc.num_materials, np.uint32, 0)

    @property
    def emitters(self):
        return _as_np(self.desc.emitters, self.desc.num_emitters, np.uint32, 4)

    @property
    def num_shapes(self):
        return self.desc.num_materials

    def set_bvh(self, nodes):
        """Replace the tree by `nodes` ((2 * shapes - 1, 8) uint32, the reference's record layout), e.g. the result
        of `device.Renderer.build_bvh`."""
        nodes = np.ascontiguousarray(nodes, np.uint32).reshape(-1, 8)
        _check(lib().hjh_compiled_set_bvh(self._h, nodes.ctypes.data_as(C.POINTER(abi.BvhNode)), len(nodes)))
        _check(lib().hjh_compiled_desc(self._h, C.byref(self.desc)))

    def tune_bvh(self, reinsert_passes=0, vote_paths=60000):
        """hjh_compiled_tune_bvh: compile()'s tree passes (insertion-based optimisation, ray-voted child order) on the installed
        tree, e.g. after `set_bvh(renderer.build_bvh(self))`."""
        _check(lib().hjh_compiled_tune_bvh(self._h, int(reinsert_passes), int(vote_paths)))
        _check(lib().hjh_compiled_desc(self._h, C.byref(self.desc)))

    def directional_bvh(self, mode, vote_paths=60000, fallback=0, geometric_only=False):
        """hjh_compiled_directional_bvh: (K, num_nodes) link orderings of the installed tree, one per direction class of the rays."""
        n = int(self.desc.num_bvh_nodes)
        k = abi.direction_classes(mode)
        out = np.zeros((k, n, 8), np.uint32)         # the reference's record layout, as set_bvh takes it
        _check(lib().hjh_compiled_directional_bvh(self._h, int(mode), int(vote_paths), int(fallback), int(bool(geometric_only)),
                                                  out.ctypes.data_as(C.POINTER(abi.BvhNode)), k * n))
        return out

    def packed(self):
        """The reference's packed scene buffer image (src/main.rs:561-605)."""
        n = lib().hjh_compiled_packed_size(self._h)
        buf = np.zeros(n, np.uint8)
        _check(lib().hjh_compiled_pack(self._h, buf.ctypes.data, n))
        return buf


def write_image(path, rgb):
    """`Renderer::save_image` tail (src/main.rs:1402-1419): (H, W, 3) float32 -> .exr (3 x FLOAT), .pfm, or .png
    (8-bit sRGB, the preview window's image)."""
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w = rgb.shape[:2]
    ext = str(path).lower().rsplit(".", 1)[-1]
    fn = {"pfm": lib().hjh_write_pfm, "png": lib().hjh_write_png}.get(ext, lib().hjh_write_exr)
    _check(fn(os.fsencode(path), w, h, rgb.ctypes.data_as(C.POINTER(C.c_float))))


def blocks_per_pass(width, height, block_size=abi.BLOCK_SIZE):
    return lib().hjh_num_blocks_per_pass(width, height, block_size)


def make_blocks(width, height, spp, master_seed, pass_begin=0, pass_end=None, block_size=abi.BLOCK_SIZE):
    """Deterministic `ImageBlockGenerator` (src/main.rs:619-682): ctypes array of hj_image_block."""
    pass_end = spp if pass_end is None else pass_end
    n = lib().hjh_make_blocks(width, height, block_size, master_seed, pass_begin, pass_end, None, 0)
    if n == 0 and pass_end > pass_begin:
        raise abi.HijikiError(abi.HJ_ERR_INVALID, lib().hjh_last_error().decode())
    arr = (abi.ImageBlock * n)()
    lib().hjh_make_blocks(width, height, block_size, master_seed, pass_begin, pass_end, arr, n)
    return arr
