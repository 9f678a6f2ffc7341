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

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (about 6.3 TB/s achievable)
# the other roofs the walk's bytes meet (same guide): every CU streaming ds_read_b128 (section LDS, "Aggregate with every CU
# streaming"), rows gathered from an XCD's L2 (section "Indexed rows: gather into LDS": 16.8-18.8 TB/s), and the VALU lanes
# (256 CUs x 64 lanes per clock; the clock the chip holds comes out of GRBM_GUI_ACTIVE, section "DVFS give-back")
LDS_PEAK_GBS = 150000.0
L2_GATHER_PEAK_GBS = 17800.0

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


def implemented_bytes(st):
    """HBM bytes the implemented wavefront algorithm has to move for the work counted in the statistics `st`
    (DESIGN.md section 4 lists every term).  Path records are far larger than any cache (pool x 2048 workgroups x
    ~200 B), so every one of these accesses is compulsory traffic; scene data (0.6 MB on cbox: LDS/L1/L2-resident)
    is NOT counted here."""
    P, C = st["paths"], st["closest_rays"]
    D = st.get("shadow_rays_proven_free", 0)          # next-event samples the light-shaft grid answered: one sample update, no records
    S = st["shadow_rays"] - D                         # shadow rays that were queued and walked
    Hh, U = st["hits"], st["unoccluded_shadow_rays"] - D
    A = C - P                      # continuing paths written by shade (every closest ray that is not a camera ray)
    first_hits = P * (Hh / C) if C else 0.0
    b = 0.0
    # camera paths have NO records (round 4): the packet stage builds the ray from the sample index, shade rebuilds the path
    b += P * (2 * 16)              # packet stage: sample init (smp_rgb, smp_nd)
    b += A * (2 * 16)              # walk: fetch ray_o, ray_d of a continuing path
    b += C * 16                    # walk / packet stage: the hit record
    b += C * (16 + 1 + 1)          # compaction: first pass reads the hit records and leaves a tag byte per ray for the second
    b += Hh * (4 + 4)              # hit queue: write + read of the position
    b += Hh * 16                   # shade: the hit record
    b += max(0.0, Hh - first_hits) * (3 * 16)   # shade: ray_o, ray_d, thr of a continuing path that hit
    b += A * (3 * 16)              # shade: record of the continuing path (ray_o, ray_d, thr)
    b += S * (3 * 16)              # shade: shadow record (origin, direction + tMax, contribution + sample)
    b += S * (3 * 16)              # walk: fetch shadow origin, direction, contribution + sample index (carried in registers)
    b += U * (2 * 16)              # unoccluded: read-modify-write of the sample
    b += D * (2 * 16)              # proven free: the same update, from the shade stage
    b += first_hits * 16           # first-hit normal + depth
    b += P * (2 * 16 * (20 * 20) / (16 * 16))   # reconstruction: both sample layers, 20x20 staged per 16x16 tile
    return b


def coalesced_read_bytes(st):
    """The part of implemented_bytes() that is READ as wide coalesced 16-byte-per-lane streams (records in queue order):
    rocprofv3's FETCH_SIZE counts these at half their size on gfx950 (MI355X_MICROARCH.md, section HBM), while it counts
    the 64-byte sectors of 16/32/48-byte gathers exactly (profiles/r02_fetch_size_calibration.txt)."""
    P, C, Hh = st["paths"], st["closest_rays"], st["hits"]
    S = st["shadow_rays"] - st.get("shadow_rays_proven_free", 0)
    first_hits = P * (Hh / C) if C else 0.0
    return (C - P) * (2 * 16) + C * (16 + 1) + Hh * 16 + max(0.0, Hh - first_hits) * (3 * 16) + S * (3 * 16) \
        + P * (2 * 16 * (20 * 20) / (16 * 16))


def valu_probe(config):
    """Newest profiles/rNN_<config>_valu_probe.json: a same-box A/B of the shipped kernel against a build with extra VALU
    instructions in every box step (tools/valu_probe.sh)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{config}_valu_probe.json")))
    try:
        return json.load(open(files[-1])) if files else None
    except (OSError, ValueError):
        return None


def roofline_inputs(config):
    """Newest profiles/rNN_<config>_roofline_inputs.json (written by tools/roofline_inputs.py from rocprofv3 CSVs)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{config}_roofline_inputs.json")))
    if not files:
        return None, None
    try:
        return json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)
    except (OSError, ValueError):
        return None, None


def build_stamp(config):
    """hijiki_amd/lib/build_stamp.json (tools/build_stamp.py, written by the build in the container, where .git is): the commit of
    this build and, for `config`, of the profile whose counters the roofline block replays, whether the path kernel's text is
    still the profiled one, and how many commits lie between the two."""
    try:
        st = json.load(open(os.path.join(ROOT, "hijiki_amd", "lib", "build_stamp.json")))
    except (OSError, ValueError):
        return None, None
    return st, (st.get("profiles") or {}).get(config)


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


def reference_bytes_per_path(c):
    """SURVEY.md 8(d)'s B_path: the bytes the REFERENCE algorithm touches per camera path (32 B per node visit, 108 B per
    triangle test - indices + three vertices -, 16 / 48 B per sphere / quad test, closest-hit shadow walks, 128 B per hit
    for populate + material, 144 B per next-event evaluation, 128 B of sample traffic), from the oracle's counters on the
    same scene.  Informational: on cbox all of it is cache-resident, and the kernels here move other bytes."""
    P = max(1, c["paths"])
    walk = 32 * (c["nodes"] + c["shadow_nodes"]) + 108 * (c["tri_tests"] + c["shadow_tri_tests"]) \
        + 16 * (c["sphere_tests"] + c["shadow_sphere_tests"]) + 48 * (c["quad_tests"] + c["shadow_quad_tests"])
    return round((walk + 128 * c["hits"] + 144 * c["nee_evals"]) / P + 128, 1)


def survey_8d_counters(config):
    """The oracle's work counters of the WHOLE frame of `config` (tests/golden/full_size_<config>.json: data written by
    tests/golden/make_full_size.py in the build container; c5 has its 8-pass prefix).  None when the fixture is missing."""
    name = {"c5": "c5p"}.get(config, config)
    try:
        with open(os.path.join(ROOT, "tests", "golden", f"full_size_{name}.json")) as f:
            return json.load(f)["counters"]
    except (OSError, ValueError, KeyError):
        return None


def roofs_block(inputs, agg, busy_s, traffic_gbs):
    """The roofs the kernel's bytes and instructions actually meet, each as achieved / peak / frac (VERDICT r4 #2b).  Per-ray
    figures come from the walk statistics of the profiled kernel (profiles/: `walk`), scaled by THIS run's ray count and
    exclusive kernel time; the VALU fractions come from the PMC counters alone (instructions and lane-instructions over
    GRBM_GUI_ACTIVE cycles: no clock assumed)."""
    if not inputs:
        return None
    rays = agg["closest_rays"] + agg["shadow_rays"]
    w = inputs.get("walk") or {}
    c = inputs.get("counters") or {}
    lim = inputs.get("limiter") or {}
    out = {"hbm": None if traffic_gbs is None else {"achieved": traffic_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                     "frac": round(traffic_gbs / HBM_PEAK_GBS, 4), "what": "PMC traffic (FETCH_SIZE corrected + WRITE_SIZE)"}}
    if w.get("box_lane_steps_per_ray") is not None and busy_s > 0:
        hot = max(0.0, w["box_lane_steps_per_ray"] - w.get("cold_node_steps_per_ray", 0.0))
        if w.get("merged_box_lane_steps") and w.get("rays"):
            # per LANE in the merged walk (a 32-byte read each), per WAVE-step in the camera packets (one broadcast read)
            hot = max(0.0, (w["merged_box_lane_steps"] + w.get("packet_wave_steps", 0) - w.get("packet_cold_wave_steps", 0)) / w["rays"]
                      - w.get("cold_node_steps_per_ray", 0.0))
        lds = 32.0 * hot * rays / busy_s / 1e9
        out["lds"] = {"achieved": round(lds, 1), "peak": LDS_PEAK_GBS, "unit": "GB/s", "frac": round(lds / LDS_PEAK_GBS, 4),
                      "what": f"32 B x {hot:.2f} box steps per ray on the LDS copy of the 512 hottest nodes"}
        gather = 32.0 * w.get("cold_node_steps_per_ray", 0.0) + 48.0 * w.get("triangle_records_per_ray", 0.0)
        l2 = gather * rays / busy_s / 1e9
        out["l2_gather"] = {"achieved": round(l2, 1), "peak": L2_GATHER_PEAK_GBS, "unit": "GB/s", "frac": round(l2 / L2_GATHER_PEAK_GBS, 4),
                            "what": f"{gather:.0f} B per ray of node and shape records gathered through the L2 (hit rate {lim.get('l2_hit_rate')})"}
    if c.get("SQ_THREAD_CYCLES_VALU") and c.get("GRBM_GUI_ACTIVE") and inputs.get("paths_per_frame"):
        lane_ops = c["SQ_THREAD_CYCLES_VALU"] * agg["paths"] / inputs["paths_per_frame"]          # scaled to this run's frames
        clock = None
        if inputs.get("pmc_kernel_seconds"):
            clock = c["GRBM_GUI_ACTIVE"] / 8.0 / inputs["pmc_kernel_seconds"]
        frac = c["SQ_THREAD_CYCLES_VALU"] / (256.0 * 64.0 * c["GRBM_GUI_ACTIVE"] / 8.0)
        out["valu_lanes"] = {"achieved": round(lane_ops / busy_s / 1e12, 2) if busy_s > 0 else None,
                             "peak": None if clock is None else round(256 * 64 * clock / 1e12, 2), "unit": "T lane-instructions/s",
                             "frac": round(frac, 4), "effective_clock_ghz": None if clock is None else round(clock / 1e9, 3),
                             "what": "SQ_THREAD_CYCLES_VALU over 256 CUs x 64 lanes x GRBM_GUI_ACTIVE / 8 (= VALU issue share x lane fill)"}
    return out


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
    for _ in range(args.steps):
        step()
    elapsed = time.perf_counter() - t0                   # comm.reduce returns after every stream has been synchronised
    paths = W * H * spp * args.steps
    out = {"metric": f"Mrays/s (camera paths/s) at {spp}spp, {cfg['short']} {W}x{H}", "value": round(paths / elapsed / 1e6, 3),
           "unit": "Mrays/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"{args.config}: {cfg['name'].format(W=W, H=H, spp=spp)}, BVH, block 128, seed {args.seed}",
                      "partition": f"ImageBlock (bx, by) of pass p -> GPU (bx + by + p) mod {n}; one process, hj_render_frame_async per "
                                   f"context, hj_comm_reduce_framebuffers (RCCL sum of the {W}x{H} RGBA32F framebuffers)"}}
    print(json.dumps(out), flush=True)
    comm.close()
    for r in rs:
        r.close()


def build_scene(cfg):
    from hijiki_amd import host
    kind = {"cbox": host.SYNTH_CBOX, "spheres": host.SYNTH_CBOX_SPHERES, "mesh": host.SYNTH_CBOX_MESH}[cfg["kind"]]
    return host.Scene.synthetic(kind, mesh_triangles=cfg["tris"]).compile()


def roofline_block(config, agg, elapsed, steps, world, standard, oracle_counters=None):
    """HBM roofline of the dominant kernel from THIS run's device counters and HIP events (rank 0's launches)."""
    inputs, src = roofline_inputs(config) if standard and world == 1 else (None, None)
    lim = (inputs or {}).get("limiter") or {}
    launches = max(1, agg["path_launches"])
    rays = agg["closest_rays"] + agg["shadow_rays"]
    state_bytes = implemented_bytes(agg)
    # Scene data counts only where it is not LDS/cache-resident (c4: nodes beyond the LDS copy + triangle records), and
    # only the part of it the L2 does NOT serve: the PMC passes see 61 % of those fetches hit the L2 on the 1 M-triangle
    # scene, and a numerator above the measured traffic is not an HBM figure (0.66 quoted, 0.42 measured).
    scene_bytes = 0.0
    if inputs and inputs.get("scene_bytes_per_ray"):
        scene_bytes = inputs["scene_bytes_per_ray"] * rays * (1.0 - lim.get("l2_hit_rate", 0.0))
    alg = state_bytes + scene_bytes
    busy_ms = agg["path_busy_ms"] or (1e3 * elapsed)             # exclusive GPU time of the path kernels, rank 0
    excl_ms = busy_ms / launches
    # HBM traffic from the PMC passes: WRITE_SIZE is exact; FETCH_SIZE is exact for this kernel's gathers and counts
    # wide coalesced reads at half their size, so the other half of the coalesced reads is added back (never more
    # than the counter itself).  The raw and the fully doubled figures stay in the inputs file.
    traffic = traffic_per_launch = None
    if inputs and inputs.get("fetch_bytes_per_path_raw") is not None:
        fetch = inputs["fetch_bytes_per_path_raw"] * agg["paths"]
        fetch += min(fetch, 0.5 * coalesced_read_bytes(agg))
        traffic_per_launch = (fetch + inputs["write_bytes_per_path"] * agg["paths"]) / launches
        traffic = round(traffic_per_launch / (excl_ms * 1e-3) / 1e9, 1)
    capped = False
    if traffic_per_launch is not None and alg / launches > traffic_per_launch:
        alg, capped = traffic_per_launch * launches, True          # never quote more bytes than the counters saw
    achieved = alg / launches / (excl_ms * 1e-3) / 1e9
    # SURVEY 8(d)'s split: the path / hit / shadow record traffic of the wavefront design is IMPLEMENTATION OVERHEAD; what the
    # algorithm itself has to move through HBM is the sample (48 B written, 48 B read by the reconstruction, 32 B accumulated =
    # 128 B per path) plus the scene bytes no cache serves.  `frac` is therefore an HBM *utilisation* figure of a kernel whose
    # bytes are mostly overhead: it rises when the kernel moves more.  `overhead_ratio` = measured traffic / compulsory bytes.
    paths = max(1, agg["paths"])
    compulsory = 128.0 + scene_bytes / paths
    traffic_per_path = None if traffic_per_launch is None else traffic_per_launch * launches / paths
    probe = valu_probe(config) if inputs else None
    limited_by, shares = limited_by_counters(lim, None if traffic is None else traffic / HBM_PEAK_GBS, probe)
    # SURVEY 8(d) literally: B_path of the REFERENCE algorithm (oracle counters of the whole frame) x paths/s over the HBM peak.
    # Above 1 on every configuration: those bytes are node and triangle fetches that the LDS copy, the scalar cache and the
    # L1 / L2 serve - HBM is not this kernel's roof (`roofs` has the ones the bytes do meet).
    sc_ = oracle_counters or survey_8d_counters(config)
    b8d = None if sc_ is None else reference_bytes_per_path(sc_)
    frac_8d = None if b8d is None else round(b8d * agg["paths"] * world / elapsed / 1e9 / HBM_PEAK_GBS, 4)
    roofs = roofs_block(inputs, agg, busy_ms * 1e-3, traffic)
    # ONE figure for `frac` (VERDICT r5 task 4): the HBM traffic the counters saw over the peak - where counters exist (the replayed
    # profile); the bytes of the implemented-algorithm MODEL over the same kernel time stay beside it under their own name.
    model_gbs, model_frac = round(achieved, 1), round(achieved / HBM_PEAK_GBS, 4)
    frac_source = "implemented-bytes model (no counters for this run)"
    if traffic is not None:
        achieved, frac_source = traffic, "pmc_traffic"
    # the highest of the roofs the kernel's bytes and instructions meet
    top = None
    for name, r_ in (roofs or {}).items():
        if r_ and r_.get("frac") is not None and (top is None or r_["frac"] > top["frac"]):
            top = {"name": name, "frac": r_["frac"]}
    stamp, pstamp = build_stamp(config)
    return {
        "frac_source": frac_source,
        "top_roof": top,
        "model_implemented_bytes": {"achieved": model_gbs, "frac": model_frac, "unit": "GB/s",
                                    "what": "bytes the implemented wavefront algorithm streams (path / hit / shadow records, samples) over the kernel's exclusive time"},
        # which kernel text the replayed counters describe (tools/build_stamp.py): commit of the profile, commits since, and whether
        # kernels/*.h + api/render.hip are still what was profiled
        "profile_commit": None if not (inputs and pstamp) else pstamp.get("commit"),
        "profile_age_commits": None if not (inputs and pstamp) else pstamp.get("age_commits"),
        "profile_kernels_match": None if not (inputs and pstamp) else pstamp.get("kernels_match"),
        "build_commit": None if not stamp else stamp.get("commit"),
        "frac_survey_8d": frac_8d,
        "survey_8d_note": "SURVEY 8(d)'s algorithmic bytes per path (reference_algorithm_bytes_per_path) x paths/s / 8 TB/s; above 1 = served "
                          "by LDS / scalar cache / L1 / L2, not by HBM",
        "roofs": roofs,
        "valu_probe": probe,
        # `bound` names the roof `frac` is measured against (the contract's vocabulary: this path has no MFMA work, its
        # roof is HBM); `limited_by` names what the counters say actually binds the kernel today.
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
        "limited_by": limited_by, "limited_by_shares": shares,
        "valu_issue_frac": lim.get("valu_issue_frac"), "lane_fill": lim.get("lane_fill"),
        # traffic / limiter / valu_issue_frac / lane_fill come from the rocprofv3 PMC passes committed under profiles/
        # (`traffic_source`), scaled by THIS run's path count and kernel time: counters cannot be read inside an un-profiled run
        "replayed_from_profile": bool(inputs),
        "compulsory_bytes_per_path": round(compulsory, 1),
        "traffic_bytes_per_path": None if traffic_per_path is None else round(traffic_per_path, 1),
        "overhead_ratio": None if traffic_per_path is None else round(traffic_per_path / compulsory, 2),
        "kernel": "k_path_wavefront", "launches": int(launches),
        "algorithmic_bytes_per_launch": round(alg / launches),
        "algorithmic_bytes_per_path": round(alg / max(1, agg["paths"]), 1),
        "scene_bytes_per_path": round(scene_bytes / max(1, agg["paths"]), 1),
        "achieved_capped_at_traffic": capped,
        "traffic_bytes_per_launch": None if traffic_per_launch is None else round(traffic_per_launch),
        "traffic_source": src,
        "exclusive_ms_per_launch": round(excl_ms, 4),
        "overlapped_ms_per_launch": round(agg["path_ms"] / launches, 4),
        "achieved_wall": round(alg * world / elapsed / 1e9, 1),
        "reference_algorithm_bytes_per_path": b8d,
        "limiter": lim or None,
        "note": "`achieved` / `frac` = HBM traffic of the kernel per launch (rocprofv3 FETCH_SIZE corrected + WRITE_SIZE, replayed from "
                "`traffic_source` and scaled by this run's path count) over its exclusive time, against the 8 TB/s peak; the same as "
                "`traffic` and roofs.hbm.  Most of those bytes are queue traffic of the wavefront design (`model_implemented_bytes`); the "
                "algorithm's compulsory HBM bytes are `compulsory_bytes_per_path`, `overhead_ratio` = traffic / compulsory.  HBM is not what "
                "binds this kernel: `top_roof` is the highest of the roofs it meets, `limited_by` what the counters and the VALU probe say"}


def limited_by_counters(lim, hbm_frac, probe=None):
    """What binds the kernel.  "hbm" when the measured traffic is above 0.6 of the peak.  "valu" needs more than busy issue
    slots: the slots are 0.6-0.7 busy on the box scenes, yet extra VALU instructions in every box step cost next to nothing
    (profiles/NOTES.md), so "valu" is only said when a probe run of the shipped kernel (`probe`: tools/valu_probe.sh,
    profiles/rNN_<config>_valu_probe.json) shows a slope of at least 0.3 % of frame time per 1 % more VALU instructions; without
    a probe the issue share alone decides at 0.85.  Otherwise "latency": waves waiting on dependent fetches at partial lane
    fill (`waiting`).  None without counters."""
    if not lim:
        return None, None
    slope = None if not probe else probe.get("slope_time_pct_per_valu_pct")
    shares = {"hbm": None if hbm_frac is None else round(hbm_frac, 4), "valu": lim.get("valu_issue_frac"),
              "waiting": lim.get("waiting_share_of_wave_cycles"), "valu_probe_slope": slope}
    if hbm_frac is not None and hbm_frac >= 0.6:
        return "hbm", shares
    valu = lim.get("valu_issue_frac")
    if valu is not None and ((slope is not None and slope >= 0.3 and valu >= 0.6) or (slope is None and valu >= 0.85)):
        return "valu", shares
    return "latency", shares


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
