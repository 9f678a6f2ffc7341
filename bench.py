#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (camera paths/s, the reference's own numerator, src/main.rs:1491-1492) on the
synthetic Cornell box at 1024 x 1024, 512 spp (BASELINE.json configs[1]) on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one whole frame: every rank renders its ImageBlocks (rotating diagonal deal, hj_block_owner) of all 512 passes into a private
full-frame RGBA32F buffer and one RCCL sum-reduce brings the frame to rank 0 (strong scaling: the frame is fixed).
Scene, BVH and block lists are resident/derived before the timed region; nothing is read back inside it.

Prints ONE JSON line on rank 0 with the contract fields plus
  roofline     — dominant kernel (k_path_wavefront): algorithmic bytes per launch / HIP-event duration
  cpu_baseline — the CPU oracle ("Nori-style" port) timed on this host on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before torch initialises HIP: see hijiki_amd/__init__.py

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes_per_path(ctr):
    """SURVEY.md §8(d), B_path: bytes the REFERENCE algorithm touches per camera path =
    S*(32*N_n + 108*N_t + 16*N_s [+48*N_q]) over all intersectScene calls (closest AND shadow: the reference walks
    the full closest-hit query for shadow rays, scene.glsl:92-96) + H*(108+4+16) populate re-fetch + material word
    + record + E*(16+108+4+16) emitter record + light triangle + its material + 128 for the sample write,
    the reconstruction read and the accumulation read-modify-write.  Counters come from the CPU oracle on the same
    scene/seed (properties of algorithm + tree, not of the GPU)."""
    walk = 32.0 * (ctr["nodes"] + ctr["shadow_nodes"]) + 108.0 * (ctr["tri_tests"] + ctr["shadow_tri_tests"]) \
        + 16.0 * (ctr["sphere_tests"] + ctr["shadow_sphere_tests"]) + 48.0 * (ctr["quad_tests"] + ctr["shadow_quad_tests"])
    return (walk + 128.0 * ctr["hits"] + 144.0 * ctr["nee_evals"]) / max(1, ctr["paths"]) + 128.0


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


def cpu_baseline(cs, width, height, seed, budget_s=12.0):
    """Oracle timed on the host cores of this box on whole passes of the same workload (cost is linear in passes)."""
    from hijiki_amd import host
    from oracle import hj_oracle
    cores = host_cores()
    _, ctr, secs = hj_oracle.render_blocks(cs, host.make_blocks(width, height, 1, seed), width, height, nthreads=cores)
    rate1 = width * height / max(secs, 1e-9)
    spp = int(max(1, min(64, budget_s * rate1 / (width * height))))
    _, ctr, secs = hj_oracle.render_blocks(cs, host.make_blocks(width, height, spp, seed), width, height, nthreads=cores)
    return {"value": round(width * height * spp / secs / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{spp} of the 512 passes of cbox {width}x{height} (oracle/hj_oracle.c, {cores} threads, {secs:.2f} s)"}, ctr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=512)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from hijiki_amd import abi, device, host
    from hijiki_amd import dist as hjdist

    rank, world, local = hjdist.init_process_group()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")

    cs = host.Scene.synthetic(host.SYNTH_CBOX).compile()
    W, H, spp = args.width, args.height, args.spp
    sr = hjdist.ShardedRenderer(cs, W, H, local_rank=local)
    opts = device.default_opts()
    opts.flags = abi.RENDER_TIME_KERNELS      # HIP events around every kernel class, on the library's own stream

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        return sr.render_frame(spp, args.seed, opts=opts, reduce=True)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    agg = None
    for _ in range(args.steps):
        st = step()
        agg = st if agg is None else {k: agg[k] + v for k, v in st.items()}
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{sr.local}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        paths = W * H * spp * args.steps
        out = {
            "metric": f"Mrays/s (camera paths/s) at {spp}spp, cbox {W}x{H}",
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
            "config": {"workload": f"cbox-synth {W}x{H} {spp}spp diffuse+emissive, 6332 triangles, BVH, block 128, seed {args.seed}",
                       "partition": f"ImageBlock (bx, by) of pass p -> rank (bx + by + p) mod {world}, RCCL sum-reduce of the {W}x{H} RGBA32F framebuffer"},
        }
        base = None
        ctr = None
        if not args.no_cpu_baseline:
            base, ctr = cpu_baseline(cs, W, H, args.seed)
            out["cpu_baseline"] = base
        else:
            from oracle import hj_oracle
            _, ctr, _ = hj_oracle.render_blocks(cs, host.make_blocks(W, H, 1, args.seed), W, H)
        # dominant kernel: k_path_wavefront (the whole wavefront loop of a batch is ONE persistent launch).
        # achieved = algorithmic bytes per launch / average launch duration, both over the timed region of THIS
        # run (rank 0's launches, HIP events on the launch streams).  Two batches are in flight on two streams, so
        # launches overlap in time: `achieved_wall` divides the same bytes by the wall time of the region instead.
        bpp = algorithmic_bytes_per_path(ctr)
        launches = max(1, agg["path_launches"])
        bytes_per_launch = bpp * agg["paths"] / launches
        avg_ms = agg["path_ms"] / launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE and
        # WRITE_SIZE in separate runs of this command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes); only
        # quoted for the workload that was profiled (cbox 1024x1024, 512 spp, 1 GPU).
        traffic = traffic_bytes = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_v7_pmc_hbm_traffic.json")))
            if (W, H, spp) == (1024, 1024, 512) and world == 1:
                traffic_bytes = round(pmc["traffic_bytes_per_path_corrected"] * agg["paths"] / launches)
                traffic = round(traffic_bytes / (avg_ms * 1e-3) / 1e9, 1)
        except (OSError, ValueError, KeyError):
            pass
        out["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                           "traffic_bytes_per_launch": traffic_bytes, "algorithmic_bytes_per_launch": round(bytes_per_launch),
                           "kernel": "k_path_wavefront", "bytes_per_path": round(bpp, 1),
                           "avg_launch_ms": round(avg_ms, 4), "launches": int(launches),
                           "achieved_wall": round(bpp * agg["paths"] / elapsed / 1e9 * world, 1),
                           "note": "algorithmic bytes of the reference algorithm (SURVEY 8d B_path); the 0.6 MB scene is "
                                   "L1/L2-resident, so physical HBM traffic is far lower: the kernel is bound by "
                                   "dependent-fetch latency and lane utilisation (DESIGN.md 6, profiles/)"}
        out["kernel_ms_per_step"] = {k: round(agg[k] / args.steps, 3) for k in ("path_ms", "reconstruct_ms", "total_ms")}
        out["rays_per_path"] = round((agg["closest_rays"] + agg["shadow_rays"]) / max(1, agg["paths"]), 3)
        print(json.dumps(out), flush=True)
    sr.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
