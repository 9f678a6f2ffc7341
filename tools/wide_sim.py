"""Design tool (CPU only): memory trips per ray of the device walk as it is (binary skip-link tree with collapse and pair
nodes, hj_scene_upload) against a K-wide node walk that keeps the reference's visiting order ("wide nodes").

    python tools/wide_sim.py [KIND] [--tris N] [--rays N] [--width K]

Rays: camera rays of the scene, their diffuse / specular bounce rays and NEE shadow rays over four bounces (oracle's
hjo_shade_probe), i.e. the population the fused kernel walks.  Both walkers are vectorised over the rays (one masked step
per iteration) and count, per ray, node fetches ("box trips") and leaf stops.  Float64 arithmetic: this counts trips, it
is not a parity check.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hijiki_amd import host  # noqa: E402
from oracle import hj_oracle  # noqa: E402

INNER = 0xFFFFFFFF
EPS = 1e-4


def make_rays(cs, n_cam, seed=3):
    rng = np.random.default_rng(seed)
    cam = cs.desc.camera
    W = H = 1024
    pix = rng.uniform(0, W, (n_cam, 2)).astype(np.float32)
    od = hj_oracle.camera_rays(cam, W, H, pix)
    rays = np.zeros((n_cam, 8), np.float32)
    rays[:, :6] = od
    rays[:, 6] = 1e-4
    rays[:, 7] = np.inf
    closest, shadow = [rays], []
    cur = rays
    for bounce in range(4):
        ids, t, _, _ = hj_oracle.intersect(cs, cur)
        out, ids2, _ = hj_oracle.shade_probe(cs, cur, rng.integers(1, 2**32 - 1, len(cur), dtype=np.uint32))
        hit = ids >= 0
        p = cur[:, :3] + t[:, None] * cur[:, 3:6]
        want = hit & (np.abs(out[:, 1:4]).sum(1) > 0)
        sh = np.zeros((int(want.sum()), 8), np.float32)
        sh[:, :3] = p[want]; sh[:, 3:6] = out[want, 4:7]; sh[:, 6] = 2e-4; sh[:, 7] = out[want, 7]
        shadow.append(sh)
        alive = hit & (out[:, 14] > 0)
        if bounce >= 3:
            alive &= rng.uniform(size=len(cur)) < 0.7          # roulette, roughly
        nxt = np.zeros((int(alive.sum()), 8), np.float32)
        nxt[:, :3] = p[alive]; nxt[:, 3:6] = out[alive, 8:11]; nxt[:, 6] = 2e-4; nxt[:, 7] = np.inf
        if len(nxt) == 0:
            break
        closest.append(nxt)
        cur = nxt
    return np.concatenate(closest), np.concatenate(shadow)


class Shapes:
    def __init__(self, cs):
        self.ns, self.nq = len(cs.spheres), len(cs.quads)
        tri, v = cs.triangles, cs.vertices[:, :3].astype(np.float64)
        self.a = v[tri[:, 0]]; self.ab = v[tri[:, 1]] - self.a; self.ac = v[tri[:, 2]] - self.a
        self.sph = cs.spheres.astype(np.float64)

    def test(self, shape, o, d, tmin, tmax):
        """vectorised: shape (n,) global indices; returns (hit mask, t)"""
        n = len(shape)
        hit = np.zeros(n, bool); tt = np.zeros(n)
        issph = shape < self.ns
        if issph.any():
            sp = self.sph[shape[issph]]
            l = o[issph] - sp[:, :3]
            b = 2 * (d[issph] * l).sum(1); c = (l * l).sum(1) - sp[:, 3] ** 2
            disc = b * b - 4 * c
            ok = disc >= 0
            sq = np.sqrt(np.where(ok, disc, 0))
            t0 = -0.5 * (b + sq); t1 = -0.5 * (b - sq)
            h0 = ok & (tmin[issph] <= t0) & (t0 <= tmax[issph]); h1 = ok & ~h0 & (tmin[issph] <= t1) & (t1 <= tmax[issph])
            hit[issph] = h0 | h1; tt[issph] = np.where(h0, t0, t1)
        tri = ~issph
        if tri.any():
            ix = shape[tri] - self.ns - self.nq
            a, ab, ac = self.a[ix], self.ab[ix], self.ac[ix]
            nn = np.cross(ab, ac); ro = o[tri] - a; q = np.cross(ro, d[tri])
            with np.errstate(divide="ignore", invalid="ignore"):
                inv = 1.0 / (d[tri] * nn).sum(1)
                u = inv * -(q * ac).sum(1); v = inv * (q * ab).sum(1); t = inv * -(nn * ro).sum(1)
            h = (u >= 0) & (v >= 0) & (u + v <= 1) & (tmin[tri] <= t) & (t <= tmax[tri])
            hit[tri] = h; tt[tri] = t
        return hit, tt


def slab(lo, hi, inv, off, tmin, tmax):
    with np.errstate(invalid="ignore"):
        tn = lo * inv + off; tp = hi * inv + off
        t0 = np.fmax(np.fmax(np.fmin(tn[:, 0], tp[:, 0]), np.fmin(tn[:, 1], tp[:, 1])), np.fmin(tn[:, 2], tp[:, 2]))
        t1 = np.fmin(np.fmin(np.fmax(tn[:, 0], tp[:, 0]), np.fmax(tn[:, 1], tp[:, 1])), np.fmax(tn[:, 2], tp[:, 2]))
        return (t0 < t1 + EPS) & (t0 < tmax) & (t1 > tmin)


def area(lo, hi):
    d = np.maximum(hi - lo, 0)
    return d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0]


class Binary:
    """The device tree of hj_scene_upload: collapse at `pct` %, pair nodes; explicit links."""

    def __init__(self, cs, pct=50, pairs=True):
        b, bf = cs.bvh, cs.bvh_f32
        N = len(b)
        self.N = N
        self.lo = bf[:, 0:3].astype(np.float64); self.hi = bf[:, 4:7].astype(np.float64)
        self.shape = b[:, 3].astype(np.int64); self.exit = b[:, 7].astype(np.int64)
        sa = area(self.lo, self.hi)
        inner = self.shape == INNER
        dele = np.zeros(N, bool); anc = np.zeros(N)
        first_tri = len(cs.spheres) + len(cs.quads)
        for i in range(N):
            if not inner[i] or i + 1 >= N:
                continue
            l = i + 1; r = self.exit[l]
            if r >= N:
                continue
            if i != 0 and inner[l] and inner[r] and anc[i] > 0 and sa[i] > pct / 100.0 * anc[i]:
                dele[i] = True
            anc[l] = anc[r] = anc[i] if dele[i] else sa[i]
        self.pair = np.full(N, -1, np.int64); self.pair_r = np.full(N, -1, np.int64)
        if pairs:
            for i in range(N - 2):
                if not inner[i]:
                    continue
                l = i + 1; r = self.exit[l]
                if r != l + 1 or r >= N or inner[l] or inner[r] or self.shape[l] < first_tri or self.shape[r] < first_tri:
                    continue
                self.pair[i] = self.shape[l]; self.pair_r[i] = self.shape[r]
                dele[l] = dele[r] = True
        self.dele = dele
        nxt = np.arange(N + 1)
        for i in range(N - 1, -1, -1):
            nxt[i] = nxt[i + 1] if dele[i] else i
        self.resolve = nxt                                    # first kept node at or after i (N = end)
        self.left = np.where(inner, self.resolve[np.minimum(np.arange(N) + 1, N)], -1)
        self.ex = self.resolve[np.minimum(self.exit, N)]
        self.kept = int((~dele).sum())

    def walk(self, rays, shapes, anyhit):
        n = len(rays)
        o = rays[:, :3].astype(np.float64); d = rays[:, 3:6].astype(np.float64)
        tmin = rays[:, 6].astype(np.float64); tmax = rays[:, 7].astype(np.float64)
        with np.errstate(divide="ignore"):
            inv = 1.0 / d
        off = -o * inv
        cur = np.full(n, self.resolve[0]); steps = np.zeros(n, int); leaves = np.zeros(n, int); tests = np.zeros(n, int)
        done = np.zeros(n, bool)
        N = self.N
        while True:
            act = np.nonzero(~done & (cur < N))[0]
            if len(act) == 0:
                break
            c = cur[act]
            steps[act] += 1
            ent = slab(self.lo[c], self.hi[c], inv[act], off[act], tmin[act], tmax[act])
            leaf = self.shape[c] != INNER
            ispair = self.pair[c] >= 0
            stop = leaf | (ispair & ent)
            nxt = np.where(ent, self.left[c], self.ex[c])
            cur[act[~stop]] = nxt[~stop]
            s = act[stop]
            if len(s):
                leaves[s] += 1
                cs_ = cur[s]
                for which in (0, 1):
                    pr = self.pair[cs_] >= 0
                    sel = s if which == 0 else s[pr]
                    if len(sel) == 0:
                        continue
                    cc = cur[sel]
                    shp = np.where(self.pair[cc] >= 0, self.pair[cc] if which == 0 else self.pair_r[cc], self.shape[cc])
                    sel2 = sel[~done[sel]]
                    shp = shp[~done[sel]]
                    if len(sel2) == 0:
                        continue
                    tests[sel2] += 1
                    h, t = shapes.test(shp, o[sel2], d[sel2], tmin[sel2], tmax[sel2])
                    hs = sel2[h]
                    if anyhit:
                        done[hs] = True
                    else:
                        tmax[hs] = t[h] - EPS
                cur[s] = self.ex[cs_]
        return steps, leaves, tests


class Wide:
    """K-wide nodes derived from the reference's binary tree; slots in the reference's left-to-right order.
    slot kinds: 0 inner (box, link = wide node), 1 pair (box, two triangles), 2 leaf guarded by a box (its dropped
    parent's), 3 leaf without a box (its parent is the node this wide node stands for)."""

    def __init__(self, cs, K=4, pairs=True):
        b, bf = cs.bvh, cs.bvh_f32
        N = len(b)
        lo = bf[:, 0:3].astype(np.float64); hi = bf[:, 4:7].astype(np.float64)
        shape = b[:, 3].astype(np.int64); ex = b[:, 7].astype(np.int64)
        inner = shape == INNER
        sa = area(lo, hi)
        first_tri = len(cs.spheres) + len(cs.quads)

        def kids(i):
            l = i + 1
            return l, ex[l]

        def is_pair(i):
            if not pairs or not inner[i]:
                return False
            l, r = kids(i)
            return r < N and not inner[l] and not inner[r] and shape[l] >= first_tri and shape[r] >= first_tri

        def expandable(i):                 # an inner, non-pair node whose children can stand in its parent's wide node
            if not inner[i] or is_pair(i):
                return False
            l, r = kids(i)
            if r >= N:
                return False
            return inner[r]                 # a leaf as SECOND child has no box test of its own at the right time

        self.K = K
        nodes = []                          # per wide node: list of (kind, box node, link/shape, shape2)
        self.kind = []; self.box_lo = []; self.box_hi = []; self.link = []; self.link2 = []; self.count = []; self.ret_node = []; self.ret_slot = []
        # build recursively (explicit stack): wide node for binary inner node P
        order = []
        stack = [(0, -1, 0)]                # (binary node, parent wide index, slot in parent)
        wid_of = {}
        while stack:
            P, pw, ps = stack.pop()
            w = len(self.kind)
            wid_of[P] = w
            l, r = kids(P)
            ents = [("n", l, None), ("n", r, None)]      # ("n", node, guard) : a binary node standing with its own box; leaves: guard box node
            # first child leaf of P itself: no box (kind 3); second child leaf of P: no box either (P was entered)
            while len(ents) < K:
                best, bi = -1.0, -1
                for k, (tag, nd, g) in enumerate(ents):
                    if tag == "n" and expandable(nd) and sa[nd] > best:
                        best, bi = sa[nd], k
                if bi < 0:
                    break
                E = ents[bi][1]
                el, er = kids(E)
                newl = ("n", el, None) if inner[el] else ("g", el, E)     # a leaf first child is guarded by E's box
                ents[bi:bi + 1] = [newl, ("n", er, None)]
            kind, blo, bhi, link, link2 = [], [], [], [], []
            for s, (tag, nd, g) in enumerate(ents):
                if tag == "g":
                    kind.append(2); blo.append(lo[g]); bhi.append(hi[g]); link.append(shape[nd]); link2.append(-1)
                elif not inner[nd]:
                    kind.append(3); blo.append(lo[nd]); bhi.append(hi[nd]); link.append(shape[nd]); link2.append(-1)
                elif is_pair(nd):
                    pl, pr_ = kids(nd)
                    kind.append(1); blo.append(lo[nd]); bhi.append(hi[nd]); link.append(shape[pl]); link2.append(shape[pr_])
                else:
                    kind.append(0); blo.append(lo[nd]); bhi.append(hi[nd]); link.append(-2 - nd); link2.append(-1)   # patched below
                    stack.append((nd, w, s))
            self.kind.append(kind); self.box_lo.append(blo); self.box_hi.append(bhi); self.link.append(link); self.link2.append(link2)
            self.count.append(len(ents))
            self.ret_node.append(pw); self.ret_slot.append(ps + 1)
        M = len(self.kind)
        self.M = M
        # static return links: (parent, slot + 1) or, when that was the parent's last slot, the parent's own return
        rn, rs = np.array(self.ret_node), np.array(self.ret_slot)
        cnt = np.array(self.count)
        for w in range(M):                  # parents come before children (creation order): resolve top-down
            p, s = rn[w], rs[w]
            if p >= 0 and s >= cnt[p]:
                rn[w], rs[w] = rn[p], rs[p]
        self.rn, self.rs = rn, rs
        pad = lambda rows, fill: np.array([r + [fill] * (K - len(r)) for r in rows])
        self.kind_a = pad(self.kind, -1)
        self.link_a = pad(self.link, -1); self.link2_a = pad(self.link2, -1)
        z = np.zeros(3)
        self.lo_a = np.array([r + [z] * (K - len(r)) for r in self.box_lo]); self.hi_a = np.array([r + [z] * (K - len(r)) for r in self.box_hi])
        for w in range(M):
            for s in range(cnt[w]):
                if self.kind_a[w, s] == 0:
                    self.link_a[w, s] = wid_of[-2 - self.link_a[w, s]]
        self.cnt = cnt

    def walk(self, rays, shapes, anyhit):
        n = len(rays); K = self.K
        o = rays[:, :3].astype(np.float64); d = rays[:, 3:6].astype(np.float64)
        tmin = rays[:, 6].astype(np.float64); tmax = rays[:, 7].astype(np.float64)
        with np.errstate(divide="ignore"):
            inv = 1.0 / d
        off = -o * inv
        cur = np.zeros(n, int); s0 = np.zeros(n, int)
        steps = np.zeros(n, int); leaves = np.zeros(n, int); tests = np.zeros(n, int); revisit = np.zeros(n, int)
        done = np.zeros(n, bool)
        while True:
            act = np.nonzero(~done & (cur >= 0))[0]
            if len(act) == 0:
                break
            c = cur[act]
            steps[act] += 1
            revisit[act] += s0[act] > 0
            passed = np.zeros((len(act), K), bool)
            for s in range(K):
                k = self.kind_a[c, s]
                bx = slab(self.lo_a[c, s], self.hi_a[c, s], inv[act], off[act], tmin[act], tmax[act])
                passed[:, s] = (k >= 0) & (s >= s0[act]) & (bx | (k == 3))
            anyp = passed.any(1)
            f = np.argmax(passed, 1)
            kf = self.kind_a[c, f]
            # nothing passes: static return
            none = ~anyp
            cur[act[none]] = self.rn[c[none]]; s0[act[none]] = self.rs[c[none]]
            inn = anyp & (kf == 0)
            cur[act[inn]] = self.link_a[c[inn], f[inn]]; s0[act[inn]] = 0
            lf = anyp & (kf > 0)
            s = act[lf]
            if len(s):
                leaves[s] += 1
                cw, fs = c[lf], f[lf]
                for which in (0, 1):
                    ispair = self.kind_a[cw, fs] == 1
                    if which == 1:
                        sel = s[ispair]; shp = self.link2_a[cw[ispair], fs[ispair]]
                    else:
                        sel = s; shp = self.link_a[cw, fs]
                    live = ~done[sel]
                    sel, shp = sel[live], shp[live]
                    if len(sel) == 0:
                        continue
                    tests[sel] += 1
                    h, t = shapes.test(shp, o[sel], d[sel], tmin[sel], tmax[sel])
                    hs = sel[h]
                    if anyhit:
                        done[hs] = True
                    else:
                        tmax[hs] = t[h] - EPS
                last = fs + 1 >= self.cnt[cw]
                cur[s] = np.where(last, self.rn[cw], cw); s0[s] = np.where(last, self.rs[cw], fs + 1)
        return steps, leaves, tests, revisit


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kind", type=int, nargs="?", default=host.SYNTH_CBOX)
    ap.add_argument("--tris", type=int, default=0)
    ap.add_argument("--rays", type=int, default=4000)
    ap.add_argument("--width", type=int, default=4)
    a = ap.parse_args()
    cs = host.Scene.synthetic(a.kind, mesh_triangles=a.tris).compile()
    closest, shadow = make_rays(cs, a.rays)
    shapes = Shapes(cs)
    print(f"scene kind {a.kind}: {len(cs.bvh)} reference nodes; {len(closest)} closest-hit rays, {len(shadow)} shadow rays")
    bt = Binary(cs)
    wt = Wide(cs, a.width)
    print(f"binary device tree: {bt.kept} records; wide tree: {wt.M} nodes of up to {a.width} slots, mean fill {wt.cnt.mean():.2f}")
    tot = {}
    for name, rays, anyhit in (("closest", closest, False), ("shadow", shadow, True)):
        bs, bl, btst = bt.walk(rays, shapes, anyhit)
        ws, wl, wtst, wr = wt.walk(rays, shapes, anyhit)
        assert btst.sum() == wtst.sum(), (btst.sum(), wtst.sum())     # the same shape tests (counted; order is by construction)
        print(f"{name:8s} binary: {bs.mean():6.2f} box trips + {bl.mean():5.2f} leaf stops per ray ({btst.mean():.2f} shape tests) | "
              f"wide: {ws.mean():6.2f} node trips ({wr.mean():.2f} of them returns) + {wl.mean():5.2f} leaf stops")
        tot[name] = (bs.sum(), bl.sum(), ws.sum(), wl.sum(), len(rays))
    B = sum(v[0] for v in tot.values()); BL = sum(v[1] for v in tot.values())
    Wd = sum(v[2] for v in tot.values()); WL = sum(v[3] for v in tot.values()); n = sum(v[4] for v in tot.values())
    print(f"all rays: binary {B / n:.2f} + {BL / n:.2f}; wide {Wd / n:.2f} + {WL / n:.2f} trips per ray "
          f"(total trips {(Wd + WL) / (B + BL):.2f} of binary; node trips {Wd / B:.2f})")


if __name__ == "__main__":
    main()
