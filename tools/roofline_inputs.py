"""Turns rocprofv3 CSVs into the numbers bench.py's roofline block quotes (and back-checks them).

    roofline_inputs.py trace KERNEL_TRACE.csv STEPS     k_path_wavefront launches: average, and the average + exclusive
                                                        (union of intervals) time over the last STEPS frames
    roofline_inputs.py pmc COUNTER_COLLECTION.csv       per-kernel sums of every counter in the file (CSV on stdout)
    roofline_inputs.py build DIR CONFIG [PREFIX]        DIR/[PREFIX]pmc_pass*.csv (+ kernel_trace_summary.txt, walk stats JSON)
                                                        -> the roofline_inputs JSON on stdout;
                                                        `build profiles c2 r02_c2_` regenerates profiles/r02_c2_roofline_inputs.json
                                                        from the committed CSVs (tests/test_roofline_inputs.py checks exactly that)

`build` is what tools/profile_config.sh runs last; the JSON it prints is committed as profiles/<TAG>_<CONFIG>_roofline_inputs.json
and read by bench.py.  HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE come
from separate passes; on gfx950 FETCH_SIZE tallies a 128-byte request as 64 bytes, so wide reads are doubled - the
gathers of this kernel are another width; tools/fetch_calib.sh calibrated them (profiles/r02_fetch_size_calibration.txt):
random 16- / 32-byte gathers are counted as the 64-byte sectors they fetch, i.e. exactly.  Both the raw and the fully
doubled figure are kept here; bench.py adds back only the under-counted half of the reads it knows to be coalesced.
"""
import collections
import csv
import glob
import json
import os
import sys

PATHS = {"c2": 1024 * 1024 * 512, "c3": 1024 * 1024 * 1024, "c4": 2048 * 2048 * 256}


def cmd_trace(path, steps):
    rows = [r for r in csv.DictReader(open(path)) if "k_path_wavefront" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    d = [(b - a) / 1e6 for a, b in iv]
    per_frame = len(d) // (steps + 1) if steps else len(d)      # the run has 1 warm-up frame + STEPS timed frames
    n = per_frame * steps if steps else len(d)
    tail = iv[-n:]
    busy, ca, cb = 0, None, None
    for a, b in tail:
        if cb is None or a > cb:
            if cb is not None:
                busy += cb - ca
            ca, cb = a, b
        else:
            cb = max(cb, b)
    busy += cb - ca
    print(f"k_path_wavefront launches {len(d)}; all: avg {sum(d)/len(d):.4f} ms; last {n} (timed region): avg {sum(d[-n:])/n:.4f} ms "
          f"min {min(d[-n:]):.3f} max {max(d[-n:]):.3f}; exclusive (union of intervals) {busy/1e6:.3f} ms = {busy/1e6/n:.4f} ms per launch")


def cmd_pmc(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    names = []
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        c = r["Counter_Name"]
        if c not in names:
            names.append(c)
        agg[k][c] += float(r["Counter_Value"])
        cnt[(k, c)] += 1
        # the dispatch's own duration (counter passes serialise the launches): once per dispatch, as a pseudo-counter in ns
        if r.get("Start_Timestamp") and r.get("End_Timestamp") and (k, r.get("Dispatch_Id")) not in seen:
            seen.add((k, r.get("Dispatch_Id")))
            agg[k]["PASS_KERNEL_NS"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    if seen:
        names.append("PASS_KERNEL_NS")
    w = csv.writer(sys.stdout, lineterminator="\n")             # kernel names hold commas ("k<true, false>"): quoted
    w.writerow(["kernel", "launches"] + names)
    for k in sorted(agg):
        w.writerow([k, max(cnt[(k, c)] for c in names if c != "PASS_KERNEL_NS")] + [f"{agg[k][c]:.0f}" for c in names])


def read_passes(files):
    """{counter: sum over the k_path_wavefront launches}, launches - from the per-pass CSVs `cmd_pmc` writes, or from the
    file profiles/<TAG>_<CONFIG>_pmc_passes.csv that holds them one after the other (a header line before each)."""
    c, launches = {}, 0
    for f in files:
        header = None
        for row in csv.reader(open(f)):
            if not row:
                continue
            if row[0] == "kernel":
                header = row
            elif header and "k_path_wavefront" in row[0]:
                extra = len(row) - len(header)                     # an unquoted template-argument comma in the name
                if extra < 0:
                    continue
                row = [",".join(row[:extra + 1])] + row[extra + 1:]
                launches = max(launches, int(row[1]))
                for k, v in zip(header[2:], row[2:]):
                    if k == "PASS_KERNEL_NS":                     # per pass: keep the one of the pass that holds GRBM_GUI_ACTIVE
                        if "GRBM_GUI_ACTIVE" in header:
                            c["GRBM_PASS_KERNEL_NS"] = float(v)
                        continue
                    c[k] = float(v)
    return c, launches


def cmd_build(d, cfg, prefix=""):
    """`d` = a directory of tools/profile_config.sh, or (with `prefix` = "<TAG>_<CONFIG>_") profiles/ itself."""
    files = sorted(glob.glob(os.path.join(d, prefix + "pmc_pass*.csv")))
    c, launches = read_passes(files)
    paths = PATHS[cfg]
    out = {"config": cfg, "kernel": "k_path_wavefront", "paths_per_frame": paths, "launches_per_frame": launches,
           "command": f"tools/profile_config.sh: rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --config {cfg} --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-latency-frame, one pass per group, GPU_MAX_HW_QUEUES=8",
           "counters": {k: v for k, v in sorted(c.items())}}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:      # rocprofv3 reports both in KiB
        fetch, write = c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0
        out["fetch_bytes_per_path_raw"] = fetch / paths
        out["write_bytes_per_path"] = write / paths
        out["hbm_bytes_per_path_raw"] = (fetch + write) / paths
        out["hbm_bytes_per_path"] = (2 * fetch + write) / paths
    if c.get("GRBM_PASS_KERNEL_NS"):        # kernel time of the pass that counted GRBM_GUI_ACTIVE: the clock the chip held = cycles / 8 / time
        out["pmc_kernel_seconds"] = c["GRBM_PASS_KERNEL_NS"] * 1e-9
        out["effective_clock_ghz"] = round(c["GRBM_GUI_ACTIVE"] / 8.0 / c["GRBM_PASS_KERNEL_NS"], 3)
    lim = {}
    if c.get("SQ_INSTS_VALU"):
        lim["valu_lanes_per_instruction"] = round(c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"], 2)
        lim["lane_fill"] = round(c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"] / 64.0, 4)
    if c.get("SQ_INSTS_VALU") and c.get("GRBM_GUI_ACTIVE"):
        # share of the chip's VALU issue slots the kernel uses: a VALU instruction occupies its SIMD for 4 cycles (64 lanes on a
        # 16-lane unit); 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        lim["valu_issue_frac"] = round(c["SQ_INSTS_VALU"] * 4.0 / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0), 4)
    if c.get("SQ_WAVE_CYCLES"):
        for k, n in (("SQ_ACTIVE_INST_VALU", "valu_issue_share_of_wave_cycles"), ("SQ_WAIT_ANY", "waiting_share_of_wave_cycles"),
                     ("SQ_WAIT_INST_ANY", "issue_stall_share_of_wave_cycles"), ("SQ_ACTIVE_INST_VMEM", "vmem_issue_share_of_wave_cycles")):
            if k in c:
                lim[n] = round(c[k] / c["SQ_WAVE_CYCLES"], 4)
    if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
        lim["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    ws = os.path.join(d, prefix + "walk_stats.json")
    if os.path.exists(ws):
        w = json.load(open(ws))
        lim["walk_lanes_per_box_step"] = round(w["lanes_per_box_step"], 1)
        lim["walk_lanes_per_leaf_phase"] = round(w["lanes_per_leaf_phase"], 1)
        out["walk"] = w
        if cfg == "c4":   # scene data of the 1 M-triangle mesh (96 MB) is not cache-resident: nodes outside the LDS copy + triangle records
            out["scene_bytes_per_ray"] = 32.0 * w["cold_node_steps_per_ray"] + 48.0 * w.get("triangle_records_per_ray", w["leaf_tests_per_ray"])
    lim["name"] = ("dependent fetch latency of the BVH walk at partially filled waves (VALU instructions execute with "
                   f"{lim.get('valu_lanes_per_instruction', '?')} of 64 lanes; waves wait {100 * lim.get('waiting_share_of_wave_cycles', 0):.0f} % of their cycles)")
    out["limiter"] = lim
    t = os.path.join(d, prefix + "kernel_trace_summary.txt")
    if os.path.exists(t):
        out["kernel_trace_summary"] = open(t).read().strip()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "trace":
        cmd_trace(sys.argv[2], int(sys.argv[3]))
    elif sys.argv[1] == "pmc":
        cmd_pmc(sys.argv[2])
    elif len(sys.argv) > 4:
        cmd_build(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        cmd_build(sys.argv[2], sys.argv[3])
