// K-wide device tree derived from the reference's flattened binary tree (host side of hj_scene_upload).
//
// The reference walks its skip-link array in pre-order: an inner node's box is tested when the node is reached, with
// the tMax of that moment; a leaf's shape is tested unconditionally (reference shader/scene.glsl:102-131).  A wide node
// stands for one inner node P of that tree that the walk has ENTERED and holds up to K "slots", the reference's
// left-to-right order kept:
//   inner slot   box of a node C, link to C's wide node
//   pair slot    box of an inner node over two triangle leaves, link to its pair record (hj_kernels.h leaf_test)
//   leaf slot    a shape, guarded by the box of its PARENT E when E itself was dissolved into this wide node and the
//                leaf is E's first child (E's box test and the leaf's visit happen at the same moment in the
//                reference), or unguarded (link bit 30) when its parent is P: P has been entered, the reference tests
//                the shape whatever tMax is by now
// Slots come from P's two children by repeatedly replacing an inner child E (largest box first) by E's own two
// children.  That is exact when both of E's children bring a box test of their own that is at least as strict as E's
// (an inner or pair child whose box lies inside E's: every term of the slab test is monotone in the bounds and in tMax,
// so "child passes => E passed" and "E fails => child fails") or are a first-child leaf (guarded by E's box itself).  A
// leaf as SECOND child of E has neither, so such an E stays a slot of its own.
// The walk evaluates the slots from its current one on with the CURRENT tMax and takes the first that passes; a later
// slot is looked at again, with the tMax of that moment, when the walk returns (each slot's `next` names the node
// and slot to go on with) - which is when the reference tests it.  The root's box keeps a slot of its own in a
// one-slot top node: rays whose slab arithmetic degenerates (a zero direction component: inf - inf) miss the ROOT in
// the reference, and every argument above leans on that test having been made.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/hijiki_hip.h"

namespace hj_wide {

constexpr uint32_t kInner = 0x80000000u;      // link: inner slot (bit 30 clear) / pair slot (bit 30 set): hj_device.h kInnerFlag, kPairFlag
constexpr uint32_t kPair = 0x40000000u;
constexpr uint32_t kUnguarded = 0x40000000u;  // link of a leaf slot (bit 31 clear): tested without a box test
constexpr uint32_t kIndex = 0x3FFFFFFFu;
constexpr uint32_t kEmpty = 0xFFFFFFFFu;      // unused slot (its box is NaN: never passes)
constexpr uint32_t kEnd = kIndex;             // `next` of the walk's end: node index 0x3FFFFFFF, slot 0

struct Tree {
  std::vector<float> rec;        // per node K slots of 8 floats: (lo.xyz, link bits) (hi.xyz, next bits)
  std::vector<float> entry_area; // per node: surface area of the box that guards its entry (hot-first ordering)
  uint32_t num_nodes = 0;
};

// pair_of[i]: pair record of binary node i or 0xFFFFFFFF.  Returns false (tree untouched) when the array is not a
// well-formed pre-order binary tree (hj_scene_upload accepts any forward-linked array: those keep the binary walk).
inline bool build(const hj_bvh_node* bvh, size_t N, const std::vector<uint32_t>& pair_of, uint32_t K, Tree& out) {
  if (N < 3 || K < 2 || K > 4 || N >= kIndex / 2) return false;
  auto inner = [&](size_t i) { return bvh[i].shape_index == HJ_BVH_INNER; };
  if (!inner(0)) return false;
  // ---- shape of the tree: every inner node i has children l = i + 1 and r = exit(l), with l < r < end(i) and exit(r) == exit(i)
  std::vector<uint32_t> right(N, 0), end(N, 0);
  {
    struct F { uint32_t i, e; };
    std::vector<F> st;
    st.push_back({0u, (uint32_t)N});
    size_t visited = 0;
    while (!st.empty()) {
      const F f = st.back();
      st.pop_back();
      visited++;
      end[f.i] = f.e;
      const uint32_t ex = bvh[f.i].exit_index;
      if ((ex < N ? ex : (uint32_t)N) != f.e) return false;          // exit = the record behind the subtree
      if (!inner(f.i)) {
        if (f.e != f.i + 1) return false;
        continue;
      }
      const uint32_t l = f.i + 1;
      if (l >= f.e) return false;
      const uint32_t r = bvh[l].exit_index;
      if (r <= l || r >= f.e) return false;
      if (bvh[r].exit_index != bvh[f.i].exit_index) return false;    // a right child inherits its parent's exit
      right[f.i] = r;
      st.push_back({r, f.e});
      st.push_back({l, r});
    }
    if (visited != N) return false;
  }
  auto area = [&](size_t i) {
    const float dx = bvh[i].aabb_max[0] - bvh[i].aabb_min[0], dy = bvh[i].aabb_max[1] - bvh[i].aabb_min[1],
                dz = bvh[i].aabb_max[2] - bvh[i].aabb_min[2];
    return (dx >= 0 && dy >= 0 && dz >= 0) ? dx * dy + dy * dz + dz * dx : 0.f;
  };
  auto inside = [&](size_t c, size_t p) {   // false for NaN bounds
    bool ok = true;
    for (int k = 0; k < 3; k++) ok = ok && bvh[c].aabb_min[k] >= bvh[p].aabb_min[k] && bvh[c].aabb_max[k] <= bvh[p].aabb_max[k];
    return ok;
  };
  auto is_pair = [&](size_t i) { return pair_of[i] != 0xFFFFFFFFu; };
  // may E be dissolved into the wide node that holds it?
  auto expandable = [&](size_t e) {
    if (!inner(e) || is_pair(e)) return false;
    const size_t l = e + 1, r = right[e];
    if (!inner(r) || !inside(r, e)) return false;                    // the second child needs a box test of its own, inside E's
    if (inner(l) && !inside(l, e)) return false;                     // (a leaf first child is guarded by E's box itself)
    return true;
  };
  struct Ent { uint32_t node; uint32_t guard; };                     // guard != ~0u: a leaf guarded by the box of node `guard`
  struct Job { uint32_t P, w; };
  std::vector<Job> jobs;
  std::vector<float>& rec = out.rec;
  rec.clear();
  out.entry_area.clear();
  const float qnan = __builtin_nanf("");
  auto new_node = [&](float entry) {
    const uint32_t w = (uint32_t)(rec.size() / (8 * K));
    rec.resize(rec.size() + 8 * (size_t)K, qnan);
    for (uint32_t s = 0; s < K; s++) {
      uint32_t e = kEmpty, nx = kEnd;
      std::memcpy(&rec[(w * K + s) * 8 + 3], &e, 4);
      std::memcpy(&rec[(w * K + s) * 8 + 7], &nx, 4);
    }
    out.entry_area.push_back(entry);
    return w;
  };
  auto set_slot = [&](uint32_t w, uint32_t s, size_t box_node, uint32_t link) {
    float* p = &rec[((size_t)w * K + s) * 8];
    for (int k = 0; k < 3; k++) { p[k] = bvh[box_node].aabb_min[k]; p[4 + k] = bvh[box_node].aabb_max[k]; }
    std::memcpy(&p[3], &link, 4);
  };
  auto set_next = [&](uint32_t w, uint32_t s, uint32_t nx) { std::memcpy(&rec[((size_t)w * K + s) * 8 + 7], &nx, 4); };
  // the top node: one slot, the root's own box
  const uint32_t top = new_node(3.0e38f);
  std::vector<uint32_t> ret;                                         // per wide node: where the walk goes on when the node is done
  ret.push_back(kEnd);
  jobs.push_back({0u, 0u});
  {
    const uint32_t w0 = new_node(area(0));
    ret.push_back(kEnd);                                             // behind the root's subtree the walk ends
    set_slot(top, 0, 0, kInner | w0);
    jobs.back().w = w0;
  }
  std::vector<Ent> ents;
  while (!jobs.empty()) {
    const Job jb = jobs.back();
    jobs.pop_back();
    ents.clear();
    ents.push_back({jb.P + 1, 0xFFFFFFFFu});
    ents.push_back({right[jb.P], 0xFFFFFFFFu});
    while (ents.size() < K) {
      float best = -1.f;
      size_t bi = ents.size();
      for (size_t k = 0; k < ents.size(); k++)
        if (ents[k].guard == 0xFFFFFFFFu && expandable(ents[k].node) && area(ents[k].node) > best) { best = area(ents[k].node); bi = k; }
      if (bi == ents.size()) break;
      const uint32_t E = ents[bi].node, el = E + 1, er = right[E];
      const Ent a = inner(el) ? Ent{el, 0xFFFFFFFFu} : Ent{el, E};    // a leaf first child: guarded by E's box
      ents[bi] = a;
      ents.insert(ents.begin() + (std::ptrdiff_t)bi + 1, Ent{er, 0xFFFFFFFFu});
    }
    const uint32_t cnt = (uint32_t)ents.size();
    for (uint32_t s = 0; s < cnt; s++) {
      const Ent& e = ents[s];
      const uint32_t after = s + 1 < cnt ? (jb.w | ((s + 1) << 30)) : ret[jb.w];   // where the walk goes on behind slot s
      if (e.guard != 0xFFFFFFFFu) {
        set_slot(jb.w, s, e.guard, bvh[e.node].shape_index);
      } else if (!inner(e.node)) {
        // a leaf child of P itself.  As slot 0 it is reached only on arrival, with the tMax P's own box has just been
        // tested with: P's box as its guard repeats that test (same operands, same result).  Later slots are reached
        // with another tMax: no box test.
        if (s == 0) set_slot(jb.w, s, jb.P, bvh[e.node].shape_index);
        else set_slot(jb.w, s, e.node, bvh[e.node].shape_index | kUnguarded);
      } else if (is_pair(e.node)) {
        set_slot(jb.w, s, e.node, kInner | kPair | pair_of[e.node]);
      } else {
        const uint32_t cw = new_node(area(e.node));
        ret.push_back(after);
        set_slot(jb.w, s, e.node, kInner | cw);
        jobs.push_back({e.node, cw});
      }
      set_next(jb.w, s, after);
    }
    for (uint32_t s = cnt; s < K; s++) set_next(jb.w, s, ret[jb.w]);   // (the last slot's `next` is what the walk takes when nothing passes)
    if (cnt < K) set_next(jb.w, K - 1, ret[jb.w]);
  }
  // the top node: nothing passes -> the end; slot 0's next = the end
  for (uint32_t s = 0; s < K; s++) set_next(top, s, kEnd);
  out.num_nodes = (uint32_t)(rec.size() / (8 * K));
  return true;
}

// Re-index the nodes: `order[new] = old`.  Links of inner slots and every `next` follow.
inline void permute(Tree& t, uint32_t K, const std::vector<uint32_t>& order) {
  const uint32_t M = t.num_nodes;
  std::vector<uint32_t> where(M);
  for (uint32_t k = 0; k < M; k++) where[order[k]] = k;
  std::vector<float> rec(t.rec.size());
  std::vector<float> ea(M);
  for (uint32_t k = 0; k < M; k++) {
    std::memcpy(&rec[(size_t)k * K * 8], &t.rec[(size_t)order[k] * K * 8], sizeof(float) * 8 * K);
    ea[k] = t.entry_area[order[k]];
    for (uint32_t s = 0; s < K; s++) {
      uint32_t link, nx;
      std::memcpy(&link, &rec[((size_t)k * K + s) * 8 + 3], 4);
      std::memcpy(&nx, &rec[((size_t)k * K + s) * 8 + 7], 4);
      if (link != kEmpty && (link & (kInner | kPair)) == kInner) link = kInner | where[link & kIndex];
      if ((nx & kIndex) != kEnd) nx = (nx & ~kIndex) | where[nx & kIndex];
      std::memcpy(&rec[((size_t)k * K + s) * 8 + 3], &link, 4);
      std::memcpy(&rec[((size_t)k * K + s) * 8 + 7], &nx, 4);
    }
  }
  t.rec.swap(rec);
  t.entry_area.swap(ea);
}

}  // namespace hj_wide
