"""The roofline block of bench.py's JSON line: what the path kernel moved and issued against the roofs it meets.

Pure host-side arithmetic on a run's device counters (hj_render_stats), its HIP-event kernel times and the rocprofv3 counters committed
under profiles/ (tools/profile_config.sh, tools/roofline_inputs.py) - no GPU call in here; bench.py imports it, tests/test_roofline_inputs.py
checks it on the CPU.  DESIGN.md section 4 "The roofline block" explains every field.
"""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (about 6.3 TB/s achievable)
# the other roofs the walk's bytes meet (same guide): every CU streaming ds_read_b128 (section LDS, "Aggregate with every CU
# streaming"), rows gathered from an XCD's L2 (section "Indexed rows: gather into LDS": 16.8-18.8 TB/s), and the VALU lanes
# (256 CUs x 64 lanes per clock; the clock the chip holds comes out of GRBM_GUI_ACTIVE, section "DVFS give-back")
LDS_PEAK_GBS = 150000.0
L2_GATHER_PEAK_GBS = 17800.0


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
