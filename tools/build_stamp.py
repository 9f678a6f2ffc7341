"""Which sources a build / a profile belongs to.

    python tools/build_stamp.py            writes hijiki_amd/lib/build_stamp.json (run by `make hip` and __graft_entry__.build())
    python tools/build_stamp.py --profile TAG CONFIG
                                           copies the current stamp to profiles/<TAG>_<CONFIG>_profile_stamp.json
                                           (tools/collect_profiles.sh: the counters under profiles/ describe THAT kernel text)

build_stamp.json = {"commit": HEAD (null where there is no .git: the GPU box), "dirty": uncommitted changes under hijiki_amd/csrc,
"kernels_sha256": hash of the path kernel's device code (kernels/*.h + api/render.hip), "profiles": per configuration the newest
profiles/rNN_<c>_roofline_inputs.json with its stamp, whether its kernel hash is the current one and how many commits lie between
its commit and HEAD}.  bench.py quotes these beside the replayed counters (VERDICT r5 task 4); tests/test_roofline_inputs.py fails
when the kernels changed after the newest profile was taken.
"""
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAMP = os.path.join(ROOT, "hijiki_amd", "lib", "build_stamp.json")


def kernels_sha256():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "hijiki_amd/csrc/kernels/*.h"))) + [os.path.join(ROOT, "hijiki_amd/csrc/api/render.hip")]
    for f in files:
        if os.path.basename(f) in ("hj_lbvh.h", "hj_vote.h"):       # (the BVH build and the vote are other kernels)
            continue
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def git(*args):
    try:
        return subprocess.check_output(["git", "-C", ROOT, *args], stderr=subprocess.DEVNULL, text=True).strip()
    except (OSError, subprocess.CalledProcessError):
        return None


def profile_stamps(now_sha):
    out = {}
    for cfg in ("c2", "c3", "c4"):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{cfg}_roofline_inputs.json")))
        if not files:
            continue
        newest = files[-1]
        stamp_file = newest[:-len("roofline_inputs.json")] + "profile_stamp.json"
        st = None
        try:
            st = json.load(open(stamp_file))
        except (OSError, ValueError):
            pass
        e = {"inputs": os.path.relpath(newest, ROOT), "commit": None, "kernels_match": None, "age_commits": None}
        if st:
            e["commit"] = st.get("commit")
            e["dirty"] = st.get("dirty")
            e["kernels_match"] = st.get("kernels_sha256") == now_sha
            if st.get("commit"):
                n = git("rev-list", "--count", f"{st['commit']}..HEAD")
                e["age_commits"] = int(n) if n is not None and n.isdigit() else None
        out[cfg] = e
    return out


def current():
    sha = kernels_sha256()
    commit = git("rev-parse", "HEAD")
    dirty = None if commit is None else bool(git("status", "--porcelain", "--", "hijiki_amd/csrc"))
    return {"commit": commit, "dirty": dirty, "kernels_sha256": sha, "profiles": profile_stamps(sha)}


def read():
    """The stamp the build left (bench.py on the GPU box, where there is no .git); None when there is none."""
    try:
        return json.load(open(STAMP))
    except (OSError, ValueError):
        return None


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--profile":
        st = read() or current()
        st = {k: st[k] for k in ("commit", "dirty", "kernels_sha256")}
        st["kernels_sha256"] = kernels_sha256()                 # (of the tree the profile really ran on)
        path = os.path.join(ROOT, "profiles", f"{sys.argv[2]}_{sys.argv[3]}_profile_stamp.json")
        json.dump(st, open(path, "w"), indent=1)
        print(path)
        return
    os.makedirs(os.path.dirname(STAMP), exist_ok=True)
    st = current()
    if st["commit"] is None:                                   # no .git (the GPU box): keep the commit the build container stamped
        old = read()
        if old and old.get("kernels_sha256") == st["kernels_sha256"]:
            return
    json.dump(st, open(STAMP, "w"), indent=1)


if __name__ == "__main__":
    main()
