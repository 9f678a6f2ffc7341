#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (camera paths/s, the reference's own numerator, src/main.rs:1491-1492) of one whole
frame of a BASELINE.json configuration on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W [--config c2|c3|c4] [--no-secondary]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

  c2 (default, the configuration the metric is quoted on)  cbox 1024 x 1024, 512 spp, diffuse + emissive
  c3                                                       cbox + mirror sphere + dielectric sphere, 1024 x 1024, 1024 spp
  c4                                                       1 M-triangle mesh in the box, 2048 x 2048, 256 spp

A step = one whole frame: every rank renders its ImageBlocks (rotating diagonal deal, hj_block_owner) of all passes into
a private full-frame RGBA32F buffer and one RCCL sum-reduce brings the frame to rank 0 (strong scaling: the frame is
fixed).  Scene, BVH and block lists are resident/derived before the timed region; nothing is read back inside it.

The default run (c2 on one GPU) also times 3 frames each of c3 and c4 after the headline (`secondary`: value, ms_per_step,
roofline per configuration; the CPU baseline is timed for the headline only).

Prints ONE JSON line on rank 0 with the contract fields plus
  roofline     - HBM roofline of the dominant kernel (k_path_wavefront).  `achieved` = the bytes the IMPLEMENTED
                 algorithm has to stream through HBM (path records, hit records, shadow records, sample buffer; for c4
                 also the BVH nodes and triangles that are not LDS-resident) per launch, divided by the kernel's
                 EXCLUSIVE time per launch (union of the launches' HIP-event intervals / launches: launches of the three
                 batch slots overlap).  It can not exceed the peak, and bytes/step / ms_per_step is printed beside it
                 (`achieved_wall`).  `traffic` = HBM bytes per launch from the rocprofv3 PMC passes under profiles/
                 (tools/roofline_inputs.py regenerates that file from the CSVs; FETCH_SIZE corrected as calibrated in
                 profiles/r02_fetch_size_calibration.txt: gathers exact, wide coalesced reads counted at half).  `achieved` never
                 exceeds `traffic`.  `limited_by`, `valu_issue_frac` (share of the chip's VALU issue slots in use) and
                 `lane_fill` (active lanes per VALU instruction / 64) say what binds the kernel when it is not HBM.
  cpu_baseline - the CPU oracle ("Nori-style" port) timed on this host on a bounded sample of the same workload
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# before torch initialises HIP (hijiki_amd/__init__.py): seven library streams (3 batch slots + their 3 high-priority
# reconstruction streams + 1) want hardware queues of their own; with more than one rank torch's stream and RCCL's come on top
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12" if int(os.environ.get("WORLD_SIZE", "1")) > 1 else "8")

# the roofline block (bytes models, roofs, limited_by, profile stamps): hijiki_amd/roofline.py
from hijiki_amd.roofline import (HBM_PEAK_GBS, build_stamp, coalesced_read_bytes, implemented_bytes, limited_by_counters,  # noqa: E402,F401
                                 reference_bytes_per_path, roofline_block, roofline_inputs, roofs_block, survey_8d_counters, valu_probe)

CONFIGS = {
    "c2": dict(kind="cbox", short="cbox", tris=0, size=1024, spp=512,
               name="cbox-synth {W}x{H} {spp}spp diffuse+emissive, 6332 triangles"),
    "c3": dict(kind="spheres", short="cbox+mirror+dielectric spheres", tris=0, size=1024, spp=1024,
               name="cbox-synth + mirror sphere + dielectric sphere {W}x{H} {spp}spp, 6332 triangles + 2 spheres"),
    "c4": dict(kind="mesh", short="1M-triangle mesh", tris=1_000_000, size=2048, spp=256,
               name="synthetic 1M-triangle mesh in the box {W}x{H} {spp}spp"),
    # BASELINE.json configs[4], the 8-GPU config (68.7 G paths: about 28 s per step on ONE GPU, 3.5 s per rank on eight;
    # `--gpus 8 --config c5 --steps 1 --warmup 0` is the intended use)
    "c5": dict(kind="cbox", short="cbox", tris=0, size=4096, spp=4096,
               name="cbox-synth {W}x{H} {spp}spp diffuse+emissive, 6332 triangles"),
}


def host_cores():
    """Threads the CPU baseline may really use: the cgroup CPU quota when there is one, else the CPU count."""
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(cs, width, height, total_spp, seed, label, budget_s=12.0):
    """Oracle timed on the host cores of this box on whole passes of the same workload (cost is linear in passes)."""
    from hijiki_amd import host
    from oracle import hj_oracle
    cores = host_cores()
    # shadow rays stop at their first hit, as a production CPU renderer's do ("Nori-style"; the image
    # is the same as with the reference's closest-hit shadow walks, which the parity runs keep)
    hj_oracle.lib().hjo_set_shadow_anyhit(1)
    try:
        _, _, secs = hj_oracle.render_blocks(cs, host.make_blocks(width, height, 1, seed), width, height, nthreads=cores)
        rate1 = width * height / max(secs, 1e-9)
        spp = int(max(1, min(64, total_spp, budget_s * rate1 / (width * height))))
        _, _, secs = hj_oracle.render_blocks(cs, host.make_blocks(width, height, spp, seed), width, height, nthreads=cores)
    finally:
        hj_oracle.lib().hjo_set_shadow_anyhit(0)
    # the reference algorithm's own work counters (closest-hit shadow walks: SURVEY 8(d)'s B_path) from one pass
    _, ctr, _ = hj_oracle.render_blocks(cs, host.make_blocks(width, height, 1, seed), width, height, nthreads=cores)
    return {"value": round(width * height * spp / secs / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{spp} of the {total_spp} passes of {label} {width}x{height} (oracle/hj_oracle.c, any-hit shadow rays, "
                      f"{cores} threads, {secs:.2f} s)"}, ctr


def main_inproc(args, cfg):
    """The multi-GPU frame without torch: one context per GPU in this process, frames in flight on all of them at once
    (each context's worker thread), one RCCL reduce through a communicator object that is created once."""
    from hijiki_amd import abi, device, host
    n = args.gpus
    ndev = device.device_count()
    shared = os.environ.get("HJ_COMM_SHARED_GPU") == "1"       # test rig: several contexts per GPU, kernel sum instead of RCCL
    if ndev < n and not (shared and ndev >= 1):
        raise SystemExit(f"--inproc --gpus {n}: only {ndev} GPU(s) visible (HJ_COMM_SHARED_GPU=1 lets contexts share one)")
    kind = {"cbox": host.SYNTH_CBOX, "spheres": host.SYNTH_CBOX_SPHERES, "mesh": host.SYNTH_CBOX_MESH}[cfg["kind"]]
    cs = host.Scene.synthetic(kind, mesh_triangles=cfg["tris"]).compile()
    W, H, spp = args.width or cfg["size"], args.height or cfg["size"], args.spp or cfg["spp"]
    rs = [device.Renderer(i % ndev) for i in range(n)]
    for r in rs:
        r.upload_scene(cs)
        r.create_framebuffer(W, H)
    comm = device.Comm(rs)
    opts = device.default_opts()
    opts.flags = abi.RENDER_TIME_KERNELS
    for r in rs:
        r.reserve(spp * ((host.blocks_per_pass(W, H) + n - 1) // n), opts)      # set-up: device memory of the batch slots

    def step():
        for r in rs:
            r.clear()
        for i, r in enumerate(rs):
            r.render_frame_async(spp, args.seed, rank=i, world=n, opts=opts)
        comm.reduce(0)                                   # joins every frame, then the RCCL sum into GPU 0
        return [r.sync() for r in rs]

    for _ in range(args.warmup):
        step()
    t0 = time.perf_counter()
    per = [[0.0, 0.0, 0, 0] for _ in rs]                # per context: host time of its frames, exclusive GPU time of its path kernels, paths, rays
    for _ in range(args.steps):
        for i, st in enumerate(step()):
            per[i][0] += st["total_ms"]; per[i][1] += st["path_busy_ms"]; per[i][2] += st["paths"]; per[i][3] += st["closest_rays"] + st["shadow_rays"]
    elapsed = time.perf_counter() - t0                   # comm.reduce returns after every stream has been synchronised
    paths = W * H * spp * args.steps
    out = {"metric": f"Mrays/s (camera paths/s) at {spp}spp, {cfg['short']} {W}x{H}", "value": round(paths / elapsed / 1e6, 3),
           "unit": "Mrays/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"{args.config}: {cfg['name'].format(W=W, H=H, spp=spp)}, BVH, block 128, seed {args.seed}",
                      "partition": f"ImageBlock (bx, by) of pass p -> GPU (bx + by + p) mod {n}; one process, hj_render_frame_async per "
                                   f"context, hj_comm_reduce_framebuffers (RCCL sum of the {W}x{H} RGBA32F framebuffers)"}}
    # the same self-explaining fields as the one-process-per-GPU line (per step): a context's frame time as its worker thread saw it, its
    # path kernels' exclusive GPU time, what it was dealt; reduce_ms = what a step takes beyond the slowest context's frame (join + RCCL sum)
    k = 1.0 / args.steps
    busy = [p[1] * k for p in per]
    out["deal"] = "rotating"
    out["per_rank"] = {"render_ms": [round(p[0] * k, 3) for p in per], "kernel_busy_ms": [round(b, 3) for b in busy],
                       "paths": [int(p[2]) for p in per], "rays": [int(p[3]) for p in per], "unit": "ms per step (frame)"}
    out["imbalance"] = None if sum(busy) <= 0 else round(max(busy) / (sum(busy) / len(busy)), 4)
    out["reduce_ms"] = round(max(0.0, 1e3 * elapsed * k - max(p[0] for p in per) * k), 3)
    print(json.dumps(out), flush=True)
    comm.close()
    for r in rs:
        r.close()


def build_scene(cfg):
    from hijiki_amd import host
    kind = {"cbox": host.SYNTH_CBOX, "spheres": host.SYNTH_CBOX_SPHERES, "mesh": host.SYNTH_CBOX_MESH}[cfg["kind"]]
    return host.Scene.synthetic(kind, mesh_triangles=cfg["tris"]).compile()


def run_config(name, cfg, args, steps, warmup, hj, barrier):
    """W warm-up frames, then exactly `steps` timed frames of one configuration between barriers; MAX over ranks."""
    from hijiki_amd import abi, device
    hjdist, rank, world, local = hj
    cs = build_scene(cfg)
    if name != args.config:
        W, H, spp = cfg["size"], cfg["size"], cfg["spp"]
    else:
        W, H, spp = args.width or cfg["size"], args.height or cfg["size"], args.spp or cfg["spp"]
    sr = hjdist.ShardedRenderer(cs, W, H, local_rank=local)
    opts = device.default_opts()
    opts.flags = abi.RENDER_TIME_KERNELS      # HIP events around every kernel class, on the library's own streams
    if getattr(args, "static_deal", False):
        opts.flags |= abi.RENDER_STATIC_DEAL   # all passes of a block on one rank (SURVEY 8(e)'s form): block interiors bit-identical to one GPU's
    sr.reserve(spp, opts)                     # set-up: the batch slots' device memory (50 GB at the defaults) is allocated here, not in a frame
    # Frames back to back: the batch pipeline is not drained between two frames (hj_render_frame with HJ_RENDER_NO_DRAIN,
    # ShardedRenderer.render_frames): frame k + 1's first batches run beside the path-depth tail of frame k's last ones, and
    # frame k's reduce beside frame k + 1's rendering.  Every one of the K frames is rendered, reduced and complete inside the
    # timed region.  --no-pipeline: one blocking frame after the other (rounds 1-3).
    pipelined = not getattr(args, "no_pipeline", False)

    def frames(n):
        if n <= 0:
            return None
        if pipelined:
            return sr.render_frames(n, spp, args.seed, opts=opts, reduce=True)
        agg_ = None
        for _ in range(n):
            st = sr.render_frame(spp, args.seed, opts=opts, reduce=True)
            agg_ = st if agg_ is None else {k: agg_[k] + v for k, v in st.items()}
        return agg_

    frames(warmup)
    barrier()
    sr.reset_timing()
    t0 = time.perf_counter()
    agg = frames(steps)
    mine = time.perf_counter() - t0              # this rank's own wall time, before it waits for the others
    barrier()
    elapsed = hjdist.max_over_ranks(time.perf_counter() - t0, device=sr.local)     # the slowest rank's wall time
    # per rank, gathered to every rank (rank 0 prints them): wall time, time inside the render calls and inside the reduces, the
    # path kernels' exclusive GPU time, and the work it was dealt (paths, rays)
    per_rank = hjdist.gather_over_ranks([mine, sr.timing["render_s"], sr.timing["reduce_s"], agg["path_busy_ms"] * 1e-3, agg["paths"],
                                         agg["closest_rays"] + agg["shadow_rays"]], device=sr.local)
    # one BLOCKING frame after the timed region (not part of `value`): what a single frame takes from submission to the
    # reduced result when nothing overlaps its end - the latency figure beside the back-to-back throughput
    latency_ms = None
    if pipelined and not getattr(args, "no_latency_frame", False):
        barrier()
        t1 = time.perf_counter()
        sr.render_frame(spp, args.seed, opts=opts, reduce=True)
        barrier()
        latency_ms = hjdist.max_over_ranks(1e3 * (time.perf_counter() - t1), device=sr.local)
    if getattr(args, "dump_frame", None) and name == args.config and rank == 0:
        import numpy as np
        np.save(args.dump_frame, sr.fb.cpu().numpy())          # the reduced frame of the last timed step (tests compare it)
    sr.close()
    return dict(cs=cs, W=W, H=H, spp=spp, agg=agg, elapsed=elapsed, steps=steps, pipelined=pipelined, latency_ms=latency_ms, per_rank=per_rank,
                standard=(W, H, spp) == (cfg["size"], cfg["size"], cfg["spp"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="only the headline configuration (the default c2 run on one GPU also times 3 frames each of c3 and c4)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="drain the batch pipeline after every frame (one blocking hj_render_frame per step, as in rounds 1-3)")
    ap.add_argument("--no-latency-frame", action="store_true",
                    help="skip the one blocking frame after the timed region (`blocking_frame_ms`): profiling runs that count frames")
    ap.add_argument("--dump-frame", default=None, metavar="FILE.npy",
                    help="rank 0 saves the reduced RGBA32F accumulation buffer of the last timed frame (after the timed region)")
    ap.add_argument("--static-deal", action="store_true",
                    help="N > 1: all passes of an ImageBlock on one rank (HJ_RENDER_STATIC_DEAL, SURVEY 8(e)'s partition) instead of the "
                         "deal that moves one diagonal per pass")
    ap.add_argument("--inproc", action="store_true",
                    help="ONE process drives all --gpus GPUs through the C ABI alone (hj_render_frame_async per context, "
                         "hj_comm_reduce_framebuffers): no torch, no torchrun")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    if args.inproc:
        return main_inproc(args, cfg)

    import torch
    import torch.distributed as dist
    from hijiki_amd import dist as hjdist

    rank, world, local = hjdist.init_process_group()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")

    def barrier():
        hjdist.barrier()

    hj = (hjdist, rank, world, local)
    res = run_config(args.config, cfg, args, args.steps, args.warmup, hj, barrier)
    W, H, spp, agg, elapsed = res["W"], res["H"], res["spp"], res["agg"], res["elapsed"]
    # the secondary configurations run on every rank (they contain collectives) but only in the default one-GPU run
    secondary = {}
    if args.config == "c2" and res["standard"] and world == 1 and not args.no_secondary:
        for name in ("c3", "c4"):
            try:                                    # (the headline line is printed whatever happens here)
                secondary[name] = run_config(name, CONFIGS[name], args, 3, 1, hj, barrier)
            except Exception as e:                  # noqa: BLE001
                secondary[name] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        paths = W * H * spp * args.steps
        label = cfg["name"].format(W=W, H=H, spp=spp)
        out = {
            "metric": f"Mrays/s (camera paths/s) at {spp}spp, {cfg['short']} {W}x{H}",
            "value": round(paths / elapsed / 1e6, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: {label}, BVH, block 128, seed {args.seed}",
                       "partition": (f"ImageBlock (bx, by) of every pass -> rank (bx + by) mod {world} (static deal)" if args.static_deal else
                                     f"ImageBlock (bx, by) of pass p -> rank (bx + by + p) mod {world}") +
                                    f", RCCL sum-reduce of the {W}x{H} RGBA32F framebuffer"},
            "deal": "static" if args.static_deal else "rotating",
            # how many ranks the collective of the timed frames really spanned (torch.distributed's nccl backend = RCCL)
            "rccl_ranks": dist.get_world_size() if world > 1 and dist.is_initialized() else 1,
            "rccl_backend": dist.get_backend() if world > 1 and dist.is_initialized() else None,
            # steps overlap at their seams (the next frame's first batches beside this frame's last): all K frames lie inside the timed region
            "frames_back_to_back": bool(res["pipelined"]),
            "blocking_frame_ms": None if res["latency_ms"] is None else round(res["latency_ms"], 3),   # one frame alone, after the timed region
            "value_blocking": None if not res["latency_ms"] else round(W * H * spp / (res["latency_ms"] * 1e-3) / 1e6, 3),   # Mrays/s of that frame
        }
        if world > 1:
            # the multi-GPU run explains itself: per rank (index = rank) over the K timed frames.  render_ms = host time inside the
            # render calls (submission + waiting for the rank's own frames), reduce_ms = host time inside the framebuffer reduces
            # (a rank that finished early waits for the slowest one HERE), kernel_busy_ms = exclusive GPU time of the path kernels;
            # imbalance = max / mean of the ranks' kernel_busy_ms (1.0 = perfectly dealt work)
            pr = res["per_rank"]
            k = 1e3 / args.steps
            busy = [r[3] for r in pr]
            out["per_rank"] = {"wall_ms": [round(r[0] * k, 3) for r in pr], "render_ms": [round(r[1] * k, 3) for r in pr],
                               "reduce_ms": [round(r[2] * k, 3) for r in pr], "kernel_busy_ms": [round(b * k, 3) for b in busy],
                               "paths": [int(r[4]) for r in pr], "rays": [int(r[5]) for r in pr], "unit": "ms per step (frame)"}
            mean_busy = sum(busy) / len(busy)
            out["imbalance"] = None if mean_busy <= 0 else round(max(busy) / mean_busy, 4)
            mean_render = sum(r[1] for r in pr) / len(pr)
            out["imbalance_render"] = None if mean_render <= 0 else round(max(r[1] for r in pr) / mean_render, 4)
            out["reduce_ms"] = round(max(r[2] for r in pr) * k, 3)          # the slowest rank's share of a step spent in the collective
        oracle_counters = None
        if not args.no_cpu_baseline and world == 1:        # the CPU baseline is a one-GPU-run item (rank 0, N = 1 only)
            out["cpu_baseline"], oracle_counters = cpu_baseline(res["cs"], W, H, spp, args.seed, cfg["short"])
        out["roofline"] = roofline_block(args.config, agg, elapsed, args.steps, world, res["standard"], oracle_counters)
        busy_ms = agg["path_busy_ms"] or (1e3 * elapsed)
        out["kernel_ms_per_step"] = {"path_exclusive_ms": round(busy_ms / args.steps, 3),
                                     "path_overlapped_sum_ms": round(agg["path_ms"] / args.steps, 3),
                                     "reconstruct_sum_ms": round(agg["reconstruct_ms"] / args.steps, 3),
                                     "total_ms": round(agg["total_ms"] / args.steps, 3)}
        out["rays_per_path"] = round((agg["closest_rays"] + agg["shadow_rays"]) / max(1, agg["paths"]), 3)
        # next-event shadow rays the light-shaft grid proved unoccluded: counted in rays_per_path (they are intersectScene(shadowRay)
        # calls of the reference), never traced
        out["shadow_rays_proven_free_share"] = round(agg.get("shadow_rays_proven_free", 0) / max(1, agg["shadow_rays"]), 4)
        out["rays_walked_per_path"] = round((agg["closest_rays"] + agg["shadow_rays"] - agg.get("shadow_rays_proven_free", 0)) / max(1, agg["paths"]), 3)
        if secondary:
            out["secondary"] = {}
            for name, r in secondary.items():
                if "error" in r:
                    out["secondary"][name] = r
                    continue
                c2 = CONFIGS[name]
                p2 = r["W"] * r["H"] * r["spp"] * r["steps"]
                out["secondary"][name] = {
                    "workload": f"{name}: {c2['name'].format(W=r['W'], H=r['H'], spp=r['spp'])}, BVH, block 128, seed {args.seed}",
                    "value": round(p2 / r["elapsed"] / 1e6, 3), "unit": "Mrays/s", "steps": r["steps"], "warmup": 1,
                    "ms_per_step": round(1e3 * r["elapsed"] / r["steps"], 3),
                    "blocking_frame_ms": None if r["latency_ms"] is None else round(r["latency_ms"], 3),
                    "value_blocking": None if not r["latency_ms"] else round(r["W"] * r["H"] * r["spp"] / (r["latency_ms"] * 1e-3) / 1e6, 3),
                    "rays_per_path": round((r["agg"]["closest_rays"] + r["agg"]["shadow_rays"]) / max(1, r["agg"]["paths"]), 3),
                    "shadow_rays_proven_free_share": round(r["agg"].get("shadow_rays_proven_free", 0) / max(1, r["agg"]["shadow_rays"]), 4),
                    "roofline": roofline_block(name, r["agg"], r["elapsed"], r["steps"], world, True)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
