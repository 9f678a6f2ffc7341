"""Multi-GPU tile sharding: one process per GPU, one RCCL reduce of the framebuffer.

The path shards by ImageBlock: block j (column bx, row by) of pass p belongs to
rank (bx + by + p) mod world (hj_block_owner), every rank accumulates its blocks (and their 2-pixel aprons)
into a private full-frame RGBA32F buffer that starts at zero, and one
sum-reduce over xGMI (`torch.distributed`, backend "nccl" == RCCL) produces the
frame on rank 0.  The reference has no multi-GPU path; this is the exchange
step BASELINE.json defines.  torch is plumbing here (device memory for the
shared framebuffer + the collective); the rendering is the C ABI's.
"""
import os

import numpy as np
import torch  # noqa: F401  (before libhijiki_hip.so is loaded: one HIP runtime per process)


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def owned_blocks(width, height, rank, world, pass_index=0):
    """Block indices (within pass `pass_index`) that `rank` renders — the rule hj_render_frame applies
    (hj_block_owner; with HJ_RENDER_STATIC_DEAL every pass uses pass_index 0)."""
    from . import host
    L = host.lib()
    per = host.blocks_per_pass(width, height)
    return [j for j in range(per) if L.hj_block_owner(width, height, pass_index, j, world) == rank]


def _init_group(backend, rank, world, local):
    """dist.init_process_group for this package's use + the first collective (set-up, not rendering)."""
    import torch
    import torch.distributed as dist
    done = False
    if backend == "nccl":
        torch.cuda.set_device(local)
        # the collective's kernels on a high-priority stream: they run beside the NEXT frame's persistent workgroups
        # (render_frames) and should get the next free wave slots, as the reconstructions do
        try:
            pg = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, pg_options=pg)
            done = True
        except (AttributeError, TypeError):
            done = False
    if not done:
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    # RCCL builds its communicator (rings over xGMI) at the FIRST collective: do that here, as set-up, so that the first
    # frame's reduce - inside a timed region when a caller asks for no warm-up - is an ordinary one
    t = torch.zeros(1, device=f"cuda:{local}") if backend == "nccl" else torch.zeros(1)
    dist.all_reduce(t)


def init_process_group(backend=None):
    """Idempotent init from the torchrun environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    import torch
    import torch.distributed as dist
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:   # HIJIKI_DIST_BACKEND=gloo lets several ranks share one GPU (test rigs); RCCL needs one GPU per rank
            backend = os.environ.get("HIJIKI_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        _init_group(backend, rank, world, local)
    return rank, world, local


def _active():
    """Collectives run when there is more than one rank (HIJIKI_DIST_FORCE=1, a test rig: also with one, so that the RCCL calls
    of the multi-GPU path execute on a one-GPU box)."""
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("HIJIKI_DIST_FORCE") == "1")


def _through_host(t):
    """gloo (the shared-GPU / CPU test rigs) moves host memory: a device tensor goes through a host copy there.  RCCL
    ("nccl") reduces the device tensor in place over xGMI."""
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend() == "gloo"


def reduce_framebuffer(fb, root=0, all_ranks=False):
    """Sum the per-rank accumulation buffers (torch tensor, in place).  No-op for world == 1."""
    import torch.distributed as dist
    if not _active():
        return fb
    buf = fb.cpu() if _through_host(fb) else fb
    if all_ranks:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    else:
        dist.reduce(buf, dst=root, op=dist.ReduceOp.SUM)
    if buf is not fb and (all_ranks or dist.get_rank() == root):
        fb.copy_(buf)
    return fb


def barrier(device=None):
    """All ranks have arrived AND this rank's GPU is idle (bench.py brackets its timed region with this)."""
    import torch
    import torch.distributed as dist
    if _active():
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize(device)


def max_over_ranks(value, device=None):
    """MAX of a host scalar over the ranks (the wall time of the slowest rank), returned on every rank."""
    import torch
    import torch.distributed as dist
    if not _active():
        return float(value)
    on_gpu = dist.get_backend() == "nccl"
    t = torch.tensor([float(value)], dtype=torch.float64, device=(f"cuda:{device}" if on_gpu else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(values, device=None):
    """Every rank's list of host scalars, as a (world, len(values)) list of lists on every rank (one all_gather)."""
    import torch
    import torch.distributed as dist
    if not _active():
        return [[float(v) for v in values]]
    on_gpu = dist.get_backend() == "nccl"
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=(f"cuda:{device}" if on_gpu else "cpu"))
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.cpu().tolist() for o in out]


class ShardedRenderer:
    """A `device.Renderer` whose framebuffer is a torch CUDA tensor, so RCCL can reduce it in place."""

    def __init__(self, compiled, width, height, local_rank=None):
        import torch
        from . import device
        rank, world, local = env_rank_world()
        self.rank, self.world = rank, world
        self.local = local if local_rank is None else local_rank
        self.local %= max(1, torch.cuda.device_count())      # ranks beyond the GPU count share devices (gloo test rigs only)
        torch.cuda.set_device(self.local)
        self.fb = torch.zeros((height, width, 4), dtype=torch.float32, device=f"cuda:{self.local}")
        self.renderer = device.Renderer(self.local)
        self.renderer.upload_scene(compiled)
        self.renderer.create_framebuffer(width, height, external_device_ptr=self.fb.data_ptr())
        self.width, self.height = width, height
        self.reset_timing()

    def reset_timing(self):
        """Host wall time this rank spent (a) inside the render calls - submitting frames and waiting for them - and (b) inside the
        framebuffer reduces (the collective is made to finish before the clock stops: a rank that waits for a slower peer sees it
        here), since the last reset.  bench.py gathers them per rank: the first multi-GPU run has to explain itself."""
        self.timing = {"render_s": 0.0, "reduce_s": 0.0, "frames": 0}

    def _timed_reduce(self, fb):
        import time
        import torch
        t = time.perf_counter()
        reduce_framebuffer(fb, root=0)
        if _active() and fb.is_cuda:
            torch.cuda.current_stream(self.local).synchronize()      # (torch's nccl work has been joined to this stream)
        self.timing["reduce_s"] += time.perf_counter() - t

    def reserve(self, spp, opts=None):
        """Allocate the batch slots this rank's share of an spp-pass frame needs (set-up: the first frame then renders at speed)."""
        from . import host
        per_pass = host.blocks_per_pass(self.width, self.height)
        self.renderer.reserve(spp * ((per_pass + self.world - 1) // self.world), opts)

    def render_frame(self, spp, master_seed, opts=None, reduce=True):
        """Zero the buffer, render this rank's blocks of all passes, reduce to rank 0.  Returns the stats dict."""
        import time
        import torch
        self.fb.zero_()
        torch.cuda.synchronize(self.local)      # the C ABI renders on its own stream
        t = time.perf_counter()
        stats = self.renderer.render_frame(spp, master_seed, rank=self.rank, world=self.world, opts=opts)
        self.timing["render_s"] += time.perf_counter() - t
        self.timing["frames"] += 1
        if reduce:
            self._timed_reduce(self.fb)
        return stats

    def render_frames(self, steps, spp, master_seed, opts=None, reduce=True):
        """`steps` whole frames BACK TO BACK without draining the batch pipeline between them (hj_render_frame with
        HJ_RENDER_NO_DRAIN): frame k + 1 is submitted - into the next of three framebuffers - before frame k is waited for and
        reduced, so the path-depth tail of a frame's last batches runs beside the first batches of the next one, and the
        collective beside the rendering.  Every frame is complete and reduced when this returns; the LAST frame is in
        `self.fb`.  Returns the statistics summed over the frames.  Frames are the blocking call's, bit for bit."""
        import time
        import torch
        # THREE framebuffers in turn, not two: frame k + 1 may only be submitted into a buffer whose last reduce has completed, and
        # the collective's kernels have to find room on a GPU that frame k's persistent workgroups fill - with two buffers that
        # reduce (frame k - 1's) was enqueued moments ago, with three it is frame k - 2's and has had a whole frame to run.
        if not hasattr(self, "_fb_ring"):
            self._fb_ring = [self.fb, torch.zeros_like(self.fb), torch.zeros_like(self.fb)]
        n = len(self._fb_ring)
        shift = (-(steps - 1)) % n                      # so that frame steps - 1 lands in self.fb (= ring[0])
        bufs = [self._fb_ring[(k + shift) % n] for k in range(steps)]
        torch.cuda.synchronize(self.local)
        for k in range(steps):
            fb = bufs[k]
            fb.zero_()                                 # (behind the reduce of the frame that last used this buffer, on torch's stream)
            torch.cuda.current_stream(self.local).synchronize()
            t = time.perf_counter()
            self.renderer.bind_framebuffer(fb.data_ptr())
            self.renderer.submit_frame(spp, master_seed, rank=self.rank, world=self.world, opts=opts)
            if k >= 1:
                self.renderer.pipeline_wait(keep=1)    # frame k - 1 is complete (frame k renders on)
            self.timing["render_s"] += time.perf_counter() - t
            if k >= 1 and reduce:
                self._timed_reduce(bufs[k - 1])
        t = time.perf_counter()
        stats = self.renderer.pipeline_wait(keep=0)
        self.timing["render_s"] += time.perf_counter() - t
        self.timing["frames"] += steps
        if reduce:
            self._timed_reduce(bufs[steps - 1])
        self.renderer.bind_framebuffer(self.fb.data_ptr())
        return stats

    def image(self):
        """Resolved rgb/w on the host (valid on rank 0 after a reduce)."""
        acc = self.fb.cpu().numpy()
        with np.errstate(divide="ignore", invalid="ignore"):
            return acc[..., :3] / acc[..., 3:4]

    def close(self):
        self.renderer.close()
