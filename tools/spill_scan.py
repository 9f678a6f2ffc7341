"""Where the fused kernel's register spills landed: scratch instructions of each k_path_wavefront instantiation and of the
stage functions it CALLS (stage_camera_packets_call, compact_hits_call, stage_shade_call, stage_gen_camera_call: each is
register-allocated on its own), and the ones inside their hot loops.  A scratch reload inside the walk loop - or inside the
node loop of the packet stage - is a memory round trip per iteration; cold-stage edits move them around (the allocator is
global per function), so run this after touching any stage.

    python tools/spill_scan.py [extra hipcc flags]       # compiles hijiki_amd/csrc/api/render.hip (the unit with the path kernels) to gfx950 assembly in $TMPDIR

Per function: total scratch instructions; scratch instructions in loops of depth >= HOT (kernel: 2 = trace_persistent's
for(;;), its box-step loop is depth 3; packet stage: 2 = the node loop inside the chunk loop); for the kernel also the older,
stricter figure: scratch instructions anywhere between the two s_barrier that bracket the walk's node fetch.
`scan()` returns the same as a dict (tests/test_abi.py asserts the hot-loop figures are 0).  For the kernels it also accounts for what
-Rpass-analysis=kernel-resource-usage reports (hijiki_amd/lib/resource_usage.txt: "SGPRs Spill: 42, VGPRs Spill: 1, ScratchSize 88"):
the SGPR spills are lanes of ONE VGPR (v_writelane / v_readlane, counted per loop depth: depth 2 = once per round of the walk's
outer loop, depth 3 = the box-step loop), the one spilled VGPR is that lane register saved around the calls, and the 88 bytes per
lane are the frames of the CALLED stage functions (stage_shade_call's 34 scratch instructions), not the walk's.
"""
import os, re, subprocess, sys, tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# function-name pattern -> (label, loop depth from which a scratch instruction counts as "hot")
FUNCS = [
    (r"_ZN2hj16k_path_wavefrontILb([01])ELb([01])ELb([01])E", "k_path_wavefront<USE_BVH={0}, PAIRS={1}, NT={2}>", 2),
    (r"_ZN2hj25stage_camera_packets_callILb([01])E", "stage_camera_packets_call<NT={0}>", 2),
    (r"_ZN2hj17compact_hits_callILb([01])ELj(\d+)E", "compact_hits_call<NT={0}, R={1}>", 1),
    (r"_ZN2hj16stage_shade_callILb([01])E", "stage_shade_call<NT={0}>", 99),
    (r"_ZN2hj21stage_gen_camera_callILb([01])E", "stage_gen_camera_call<NT={0}>", 99),
]


def compile_asm(extra=()):
    asm = os.path.join(tempfile.gettempdir(), "hj_spill_scan.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fno-fast-math",
           "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wno-unused-function", "--cuda-device-only", "-S", *extra, "-o", asm,
           os.path.join(root, "hijiki_amd/csrc/api/render.hip")]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return open(asm).read().split("\n")


def loop_depths(body):
    """Loop depth of every line: the assembly printer annotates each basic block (".LBBn_m:" or "; %bb.m:") with its
    innermost loop and depth."""
    depth, out = 0, []
    for l in body:
        if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", l):
            m2 = re.search(r"Depth=(\d+)", l)
            depth = int(m2.group(1)) if m2 else 0
        elif l.lstrip().startswith("; =>") or l.lstrip().startswith(";   "):
            m2 = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", l)
            if m2:
                depth = int(m2.group(1))
        out.append(depth)
    return out


def scan(extra=()):
    s = compile_asm(extra)
    res = []
    for i0, line in enumerate(s):
        for pat, label, hot in FUNCS:
            m = re.match(r"^(" + pat + r"\S*): ;", line)
            if m:
                break
        else:
            continue
        end = next(i for i in range(i0, len(s)) if s[i].startswith(".Lfunc_end"))
        body = s[i0:end]
        depth = loop_depths(body)
        scratch = [(k, l.strip().split(";")[0].strip()) for k, l in enumerate(body) if "scratch_" in l]
        in_hot = [(k, l) for k, l in scratch if depth[k] >= hot]
        # SGPR spills do not go to memory: the allocator parks them in lanes of a VGPR (v_writelane_b32 / v_readlane_b32); by loop depth
        lanes = {}
        for k, l in enumerate(body):
            t = l.strip().split(" ")[0] if l.strip() else ""
            if t in ("v_writelane_b32", "v_readlane_b32"):
                lanes.setdefault(t, {}).setdefault(depth[k], 0)
                lanes[t][depth[k]] += 1
        calls = sum(1 for l in body if l.strip().startswith("s_swappc_b64"))
        entry = {"name": label.format(*m.groups()[1:]), "lines": len(body), "scratch": len(scratch), "hot_depth": hot,
                 "sgpr_spill_lanes": lanes, "calls": calls, "scratch_list": [(k, l, depth[k]) for k, l in scratch],
                 "scratch_in_hot_loops": len(in_hot), "hot_list": in_hot, "between_barriers": None,
                 "max_loop_depth": max(depth) if depth else 0}
        if label.startswith("k_path_wavefront"):
            fetch = [k for k, l in enumerate(body) if "flat_load_dwordx4" in l]
            if fetch:
                bars = [k for k, l in enumerate(body) if re.search(r"\bs_barrier\b", l)]
                lo = max([b for b in bars if b < fetch[0]], default=0)
                hi = min([b for b in bars if b > fetch[-1]], default=len(body))
                entry["between_barriers"] = sum(1 for k, _ in scratch if lo <= k <= hi)
                entry["node_fetch_line"] = fetch[0]
        res.append(entry)
    return res


if __name__ == "__main__":
    for e in scan(sys.argv[1:]):
        print(f"{e['name']}: {e['lines']} lines, {e['scratch']} scratch instructions, loops nest {e['max_loop_depth']} deep")
        if e["hot_depth"] < 99:
            extra = "" if e["between_barriers"] is None else f"; between the barriers around the walk: {e['between_barriers']}"
            print(f"  scratch instructions in loops of depth >= {e['hot_depth']}: {e['scratch_in_hot_loops']}{extra}")
            for k, l in e["hot_list"]:
                print(f"    {k:6d}  {l}")
        if e["name"].startswith("k_path_wavefront"):
            wl, rl = e["sgpr_spill_lanes"].get("v_writelane_b32", {}), e["sgpr_spill_lanes"].get("v_readlane_b32", {})
            fmt = lambda d: ", ".join(f"depth {k}: {v}" for k, v in sorted(d.items())) or "none"
            print(f"  SGPR spills (lanes of a VGPR, no memory): v_writelane_b32 {fmt(wl)}; v_readlane_b32 {fmt(rl)}; {e['calls']} calls of stage functions")
            for k, l, d in e["scratch_list"]:
                print(f"    scratch at line {k} (loop depth {d}): {l}")
