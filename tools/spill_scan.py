"""Where the fused kernel's register spills landed: scratch instructions of each k_path_wavefront instantiation, and the ones
within reach of the walk (the FLAT node fetch and the leaf phase behind it).  A scratch reload there is a memory round trip
per outer iteration of the walk; cold-stage edits move them around (the allocator is global per kernel), so run this after
touching any stage.

    python tools/spill_scan.py [extra hipcc flags]       # compiles hijiki_amd/csrc/hj_api.hip to gfx950 assembly in $TMPDIR
"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm = os.path.join(tempfile.gettempdir(), "hj_spill_scan.s")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fno-fast-math",
       "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wno-unused-function", "--cuda-device-only", "-S", *sys.argv[1:], "-o", asm,
       os.path.join(root, "hijiki_amd/csrc/hj_api.hip")]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
s = open(asm).read().split("\n")
for i0, line in enumerate(s):
    m = re.match(r"^(_ZN2hj16k_path_wavefrontILb([01])ELb([01])ELb([01])E\S*): ;", line)
    if not m:
        continue
    end = next(i for i in range(i0, len(s)) if s[i].startswith(".Lfunc_end"))
    body = s[i0:end]
    fetch = [k for k, l in enumerate(body) if "flat_load_dwordx4" in l]
    scratch = [(k, l.strip().split(";")[0].strip()) for k, l in enumerate(body) if "scratch_" in l]
    print(f"k_path_wavefront<USE_BVH={m.group(2)}, PAIRS={m.group(3)}, NT={m.group(4)}>: {len(body)} lines, {len(scratch)} scratch instructions")
    if fetch:
        # The walk = the loops nested inside the round loop of the kernel (depth 1): trace_persistent's for(;;) is depth 2, its
        # box-step loop depth 3; top-up, hit compaction and shade are calls and have no loops here.  The assembly printer
        # annotates every basic block (".LBBn_m:" or "; %bb.m:") with its innermost loop and depth.
        depth, in_walk = 0, []
        for k, l in enumerate(body):
            if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", l):
                m2 = re.search(r"Depth=(\d+)", l)
                depth = int(m2.group(1)) if m2 else 0
            elif l.lstrip().startswith("; =>") or l.lstrip().startswith(";   "):
                m2 = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", l)
                if m2:
                    depth = int(m2.group(1))
            if "scratch_" in l and depth >= 2:
                in_walk.append((k, l.strip().split(";")[0].strip()))
        fetch_depth_ok = True
        print(f"  node fetch at line {fetch[0]}; scratch instructions inside the walk (loops of depth >= 2 of the kernel): {len(in_walk)}")
        for k, l in in_walk:
            print(f"    {k:6d}  {l}")
