"""Is a change to the kernels' SOURCE a change to their machine code?  Compiles hijiki_amd/csrc/api/render.hip (the unit with the
path kernels) to gfx950 assembly twice - at a git revision and in the working tree - and compares every function's instruction
stream (labels, comments and debug directives dropped).  The hygiene work of round 6 (probe hooks instead of #ifdef blocks in
the walk, dead build alternatives removed, one text per shape test) was done under this check: 24 of 24 functions identical.

    python tools/isa_diff.py [REV] [extra hipcc flags]        # REV defaults to HEAD; exit code 1 when a function differs
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-Wno-unused-function", "--cuda-device-only", "-S"]


def functions(path):
    out, cur = {}, None
    for l in open(path).read().split("\n"):
        m = re.match(r"^(_Z\S+):\s*; @", l)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if l.startswith(".Lfunc_end"):
            cur = None
            continue
        if cur is not None:
            t = l.strip()
            if not t or t.startswith(";") or t.startswith(".loc") or t.startswith(".cfi"):
                continue
            out[cur].append(re.sub(r";.*$", "", t).rstrip())
    return out


def compile_tree(root, out, extra):
    subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-o", out, os.path.join(root, "hijiki_amd/csrc/api/render.hip")],
                   check=True, stderr=subprocess.DEVNULL)
    return functions(out)


def main():
    args = sys.argv[1:]
    rev = args.pop(0) if args and not args[0].startswith("-") else "HEAD"
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run(f"git -C {ROOT} archive {rev} hijiki_amd/csrc include | tar -x -C {tmp}", shell=True, check=True)
        old = compile_tree(tmp, os.path.join(tmp, "old.s"), args)
        new = compile_tree(ROOT, os.path.join(tmp, "new.s"), args)
    names = sorted(set(old) | set(new))
    differing = [n for n in names if old.get(n) != new.get(n)]
    for n in names:
        a, b = old.get(n), new.get(n)
        state = "same" if a == b else ("only in " + ("the working tree" if a is None else rev) if a is None or b is None else f"DIFFERS ({len(a)} -> {len(b)} instructions)")
        print(f"{state:40s} {n[:100]}")
    print(f"{len(names)} functions, {len(differing)} differ from {rev}")
    return 1 if differing else 0


if __name__ == "__main__":
    sys.exit(main())
