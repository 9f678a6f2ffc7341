"""ctypes mirror of the device C ABI (include/hijiki_hip.h -> libhijiki_hip.so).

`Renderer` plays the role of the reference's `Renderer` (src/main.rs:1143-1424)
for the hot path only: scene upload, the per-block render loop, read-back.
There is no CPU fallback: without the HIP library or a GPU every call raises.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.environ.get("HIJIKI_HIP_LIB", os.path.join(_HERE, "lib", "libhijiki_hip.so"))
_LIB = None

# every symbol include/hijiki_hip.h declares
EXPORTS = ("hj_context_create", "hj_context_destroy", "hj_last_error", "hj_version", "hj_default_render_opts",
           "hj_scene_upload", "hj_framebuffer_create", "hj_framebuffer_clear", "hj_framebuffer_device_ptr",
           "hj_framebuffer_read", "hj_framebuffer_resolve", "hj_render_blocks", "hj_render_frame", "hj_block_seed",
           "hj_pass_offset", "hj_block_owner", "hj_debug_trace", "hj_debug_samples", "hj_reduce_framebuffers",
           "hj_build_bvh_device", "hj_render_frame_async", "hj_sync", "hj_set_progress_callback", "hj_device_count",
           "hj_comm_create", "hj_comm_destroy", "hj_comm_reduce_framebuffers", "hj_reserve", "hj_framebuffer_bind",
           "hj_pipeline_wait", "hj_debug_light_grid", "hj_debug_light_grid_planes", "hj_tune_bvh_device", "hj_bvh_device_read")

PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_uint64)


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise ImportError(f"{HIP_LIB_PATH} missing: run `make hip` (or __graft_entry__.build()); "
                              "the HIP path has no fallback")
        L = C.CDLL(HIP_LIB_PATH)
        vp = C.c_void_p
        L.hj_context_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.hj_context_destroy.argtypes = [vp]
        L.hj_context_destroy.restype = None
        L.hj_last_error.argtypes = [vp]
        L.hj_last_error.restype = C.c_char_p
        L.hj_version.restype = C.c_uint32
        L.hj_default_render_opts.argtypes = [C.POINTER(abi.RenderOpts)]
        L.hj_default_render_opts.restype = None
        L.hj_scene_upload.argtypes = [vp, C.POINTER(abi.SceneDesc)]
        L.hj_framebuffer_create.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
        L.hj_framebuffer_clear.argtypes = [vp]
        L.hj_framebuffer_device_ptr.argtypes = [vp]
        L.hj_framebuffer_device_ptr.restype = vp
        L.hj_framebuffer_read.argtypes = [vp, C.POINTER(C.c_float)]
        L.hj_framebuffer_resolve.argtypes = [vp, C.POINTER(C.c_float)]
        L.hj_render_blocks.argtypes = [vp, C.POINTER(abi.ImageBlock), C.c_size_t, C.POINTER(abi.RenderOpts),
                                       C.POINTER(abi.RenderStats)]
        L.hj_render_frame.argtypes = [vp, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.POINTER(abi.RenderOpts), C.POINTER(abi.RenderStats)]
        L.hj_reserve.argtypes = [vp, C.c_size_t, C.POINTER(abi.RenderOpts)]
        L.hj_framebuffer_bind.argtypes = [vp, vp]
        L.hj_pipeline_wait.argtypes = [vp, C.c_uint32, C.POINTER(abi.RenderStats)]
        L.hj_reduce_framebuffers.argtypes = [C.POINTER(vp), C.c_int, C.c_int]
        L.hj_render_frame_async.argtypes = [vp, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                            C.POINTER(abi.RenderOpts)]
        L.hj_sync.argtypes = [vp, C.POINTER(abi.RenderStats)]
        L.hj_set_progress_callback.argtypes = [vp, PROGRESS_FN, vp, C.c_uint32]
        L.hj_set_progress_callback.restype = None
        L.hj_device_count.restype = C.c_int
        L.hj_comm_create.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(vp)]
        L.hj_comm_destroy.argtypes = [vp]
        L.hj_comm_destroy.restype = None
        L.hj_comm_reduce_framebuffers.argtypes = [vp, C.c_int]
        L.hj_debug_trace.argtypes = [vp, C.POINTER(C.c_float), C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
        L.hj_debug_samples.argtypes = [vp, C.POINTER(abi.ImageBlock), C.POINTER(abi.RenderOpts), C.POINTER(C.c_float)]
        L.hj_build_bvh_device.argtypes = [vp, C.POINTER(abi.SceneDesc), C.POINTER(abi.BvhNode), C.c_size_t, C.POINTER(C.c_size_t)]
        L.hj_tune_bvh_device.argtypes = [vp, C.POINTER(abi.SceneDesc), C.POINTER(abi.BvhNode), C.c_size_t, C.c_size_t]
        L.hj_bvh_device_read.argtypes = [vp, C.POINTER(abi.BvhNode), C.c_size_t, C.POINTER(C.c_size_t)]
        L.hj_block_seed.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.hj_block_seed.restype = C.c_uint32
        L.hj_pass_offset.argtypes = [C.c_uint64, C.c_uint32, C.POINTER(C.c_float)]
        L.hj_pass_offset.restype = None
        L.hj_block_owner.argtypes = [C.c_uint32] * 5
        L.hj_block_owner.restype = C.c_uint32
        _LIB = L
    return _LIB


def reduce_framebuffers(renderers, root=0):
    """hj_reduce_framebuffers: RCCL sum of the framebuffers of several in-process renderers (one per GPU) into `root`."""
    arr = (C.c_void_p * len(renderers))(*[r._h for r in renderers])
    rc = lib().hj_reduce_framebuffers(arr, len(renderers), root)
    if rc != abi.HJ_OK:
        raise abi.HijikiError(rc, lib().hj_last_error(renderers[root]._h).decode())


def device_count():
    return lib().hj_device_count()


class Comm:
    """hj_comm: the in-process contexts (one per GPU) + their RCCL communicators, made once and reused by every reduce."""

    def __init__(self, renderers):
        self.renderers = list(renderers)
        self._h = C.c_void_p()
        self._destroy = lib().hj_comm_destroy
        arr = (C.c_void_p * len(self.renderers))(*[r._h for r in self.renderers])
        rc = lib().hj_comm_create(arr, len(self.renderers), C.byref(self._h))
        if rc != abi.HJ_OK:
            raise abi.HijikiError(rc, lib().hj_last_error(self.renderers[0]._h).decode())

    def reduce(self, root=0):
        rc = lib().hj_comm_reduce_framebuffers(self._h, root)
        if rc != abi.HJ_OK:
            raise abi.HijikiError(rc, lib().hj_last_error(self.renderers[root]._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


def default_opts():
    o = abi.RenderOpts()
    lib().hj_default_render_opts(C.byref(o))
    return o


def stats_dict(st):
    return {n: getattr(st, n) for n, _ in abi.RenderStats._fields_}


class Renderer:
    """One GPU context: scene + framebuffer + render calls."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        self._destroy = lib().hj_context_destroy     # bound now: module globals may be gone at interpreter exit
        rc = lib().hj_context_create(device, C.byref(self._h))
        if rc != abi.HJ_OK:
            raise abi.HijikiError(rc, lib().hj_last_error(None).decode())
        self.width = self.height = 0

    def close(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != abi.HJ_OK:
            raise abi.HijikiError(rc, lib().hj_last_error(self._h).decode())

    def upload_scene(self, compiled, device_tree=False):
        """hj_scene_upload.  device_tree: scene->bvh = NULL - the tree `build_bvh(compiled, keep_on_device=True)` left on the device."""
        desc = compiled.desc if hasattr(compiled, "desc") else compiled
        if device_tree:
            d2 = abi.SceneDesc()
            C.memmove(C.byref(d2), C.byref(desc), C.sizeof(abi.SceneDesc))
            d2.bvh = None
            d2.num_bvh_nodes = 0
            desc = d2
        self._check(lib().hj_scene_upload(self._h, C.byref(desc)))

    def create_framebuffer(self, width, height, external_device_ptr=None):
        self._check(lib().hj_framebuffer_create(self._h, width, height, external_device_ptr))
        self.width, self.height = width, height

    def clear(self):
        self._check(lib().hj_framebuffer_clear(self._h))

    @property
    def framebuffer_ptr(self):
        return lib().hj_framebuffer_device_ptr(self._h)

    def render_blocks(self, blocks, opts=None):
        st = abi.RenderStats()
        n = len(blocks)
        self._check(lib().hj_render_blocks(self._h, blocks, n, C.byref(opts) if opts is not None else None,
                                           C.byref(st)))
        return stats_dict(st)

    def render_frame(self, spp, master_seed, pass_begin=0, pass_end=None, rank=0, world=1, opts=None):
        st = abi.RenderStats()
        pass_end = spp if pass_end is None else pass_end
        self._check(lib().hj_render_frame(self._h, spp, master_seed, pass_begin, pass_end, rank, world,
                                          C.byref(opts) if opts is not None else None, C.byref(st)))
        return stats_dict(st)

    def submit_frame(self, spp, master_seed, pass_begin=0, pass_end=None, rank=0, world=1, opts=None):
        """hj_render_frame with HJ_RENDER_NO_DRAIN: returns when the frame's batches are enqueued; `pipeline_wait` joins.
        The frame accumulates into the framebuffer bound at the time of the call (`bind_framebuffer`)."""
        o = abi.RenderOpts()
        C.memmove(C.byref(o), C.byref(opts if opts is not None else default_opts()), C.sizeof(abi.RenderOpts))
        o.flags |= abi.RENDER_NO_DRAIN
        pass_end = spp if pass_end is None else pass_end
        self._check(lib().hj_render_frame(self._h, spp, master_seed, pass_begin, pass_end, rank, world, C.byref(o), None))

    def bind_framebuffer(self, device_ptr):
        """hj_framebuffer_bind: frames submitted from now on accumulate into this caller-owned buffer (no synchronisation)."""
        self._check(lib().hj_framebuffer_bind(self._h, device_ptr))

    def pipeline_wait(self, keep=0):
        """hj_pipeline_wait: until at most `keep` submitted frames are in flight; keep = 0 drains and returns the statistics of
        all frames since the last drain."""
        st = abi.RenderStats()
        self._check(lib().hj_pipeline_wait(self._h, keep, C.byref(st) if keep == 0 else None))
        return stats_dict(st) if keep == 0 else None

    def reserve(self, total_blocks, opts=None):
        """hj_reserve: allocate now what a render call of `total_blocks` ImageBlocks will use (set-up, not rendering)."""
        self._check(lib().hj_reserve(self._h, int(total_blocks), C.byref(opts) if opts is not None else None))

    def render_frame_async(self, spp, master_seed, pass_begin=0, pass_end=None, rank=0, world=1, opts=None):
        """hj_render_frame on the context's worker thread; `sync()` waits for it and returns the statistics."""
        pass_end = spp if pass_end is None else pass_end
        self._check(lib().hj_render_frame_async(self._h, spp, master_seed, pass_begin, pass_end, rank, world,
                                                C.byref(opts) if opts is not None else None))

    def sync(self):
        st = abi.RenderStats()
        self._check(lib().hj_sync(self._h, C.byref(st)))
        return stats_dict(st)

    def set_progress(self, fn, interval_blocks=128):
        """fn(blocks_done, blocks_total) whenever `interval_blocks` more ImageBlocks have completed; None = off."""
        self._progress = PROGRESS_FN(lambda _u, d, t: fn(d, t)) if fn else PROGRESS_FN()
        lib().hj_set_progress_callback(self._h, self._progress, None, interval_blocks)

    def build_bvh(self, compiled, keep_on_device=False):
        """LBVH over the shapes of `compiled`, built on the device (hj_build_bvh_device): (2 * shapes - 1, 8) uint32
        records in the reference's layout.  `compiled.set_bvh(nodes)` installs it.  keep_on_device: nothing comes back to the host
        (returns the record count): `upload_scene(compiled, device_tree=True)` takes the tree over, `read_device_bvh()` copies it out."""
        if keep_on_device:
            got = C.c_size_t(0)
            self._check(lib().hj_build_bvh_device(self._h, C.byref(compiled.desc), None, 0, C.byref(got)))
            return got.value
        n = 2 * compiled.num_shapes - 1
        nodes = np.zeros((max(n, 1), 8), np.uint32)
        got = C.c_size_t(0)
        self._check(lib().hj_build_bvh_device(self._h, C.byref(compiled.desc), nodes.ctypes.data_as(C.POINTER(abi.BvhNode)),
                                              len(nodes), C.byref(got)))
        return nodes[:got.value]

    def read_device_bvh(self):
        """hj_bvh_device_read: the tree the last build left on the device, (nodes, 8) uint32."""
        got = C.c_size_t(0)
        self._check(lib().hj_bvh_device_read(self._h, None, 0, C.byref(got)))
        nodes = np.zeros((max(got.value, 1), 8), np.uint32)
        self._check(lib().hj_bvh_device_read(self._h, nodes.ctypes.data_as(C.POINTER(abi.BvhNode)), len(nodes), C.byref(got)))
        return nodes[:got.value]

    def tune_bvh_device(self, compiled, vote_paths=60000):
        """The tree of `compiled` with its child order voted by `vote_paths` sampled camera paths on the device
        (hj_tune_bvh_device): (nodes, 8) uint32 records, same boxes and leaves.  `compiled.set_bvh(nodes)` installs it."""
        nodes = np.zeros((max(len(compiled.bvh), 1), 8), np.uint32)
        self._check(lib().hj_tune_bvh_device(self._h, C.byref(compiled.desc), nodes.ctypes.data_as(C.POINTER(abi.BvhNode)),
                                             len(nodes), int(vote_paths)))
        return nodes[:len(compiled.bvh)]

    def trace(self, rays, use_bvh=True, any_hit=False):
        """intersectScene for (n,8) rays -> ids (n,) int32, t, u, v (n,) float32 (raw hit, before populate)."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        hits = np.zeros((len(rays), 4), np.float32)
        fp = C.POINTER(C.c_float)
        self._check(lib().hj_debug_trace(self._h, rays.ctypes.data_as(fp), len(rays), int(use_bvh), int(any_hit),
                                         hits.ctypes.data_as(fp)))
        return hits[:, 0].copy().view(np.int32), hits[:, 1], hits[:, 2], hits[:, 3]

    def samples(self, block, opts=None):
        """Intermediate image of one block: (dim_y, dim_x, 8) = (rgb, 1, normal, depth)."""
        out = np.zeros((block.dimension[1], block.dimension[0], 8), np.float32)
        self._check(lib().hj_debug_samples(self._h, C.byref(block), C.byref(opts) if opts is not None else None,
                                           out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def read(self):
        """(H, W, 4) float32 accumulation image (sum w*rgb, sum w)."""
        out = np.zeros((self.height, self.width, 4), np.float32)
        self._check(lib().hj_framebuffer_read(self._h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def resolve(self):
        """(H, W, 3) float32 rgb / w (src/main.rs:1399)."""
        out = np.zeros((self.height, self.width, 3), np.float32)
        self._check(lib().hj_framebuffer_resolve(self._h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out
