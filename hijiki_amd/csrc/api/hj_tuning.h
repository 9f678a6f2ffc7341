// Every environment switch of libhijiki_hip.so, in ONE table, read in ONE place (hjapi::Tuning::from_env): an entry point that
// uses a switch reads the table once at its start and passes the struct down - no getenv anywhere else in the library.
// None of the switches changes a result bit (tests/test_gpu_parity.py::test_tuning_switches_never_change_a_bit); the defaults
// are the measured optima (DESIGN.md section 4 "Tuning switches", profiles/NOTES.md).  Plain C++: api/light_grid.cpp (g++) includes it too.
#pragma once
#include <algorithm>
#include <cstdlib>

namespace hjapi {

struct Tuning {
  static constexpr int kUnset = -0x7fffffff;               // "chosen where it is used": by tree size, by call size
  // X(field, "NAME", default, lowest, highest)            what it is
#define HJ_TUNING_TABLE(X)                                                                                                              \
  /* context (read by hj_context_create) */                                                                                            \
  X(slots, "HJ_SLOTS", 3, 1, 4)                            /* batches in flight (kMaxSlots = 4) */                                      \
  X(pool, "HJ_POOL", 32768, 64, 1 << 20)                   /* record positions per workgroup */                                         \
  X(wg_per_cu, "HJ_WG_PER_CU", 8, 1, 32)                   /* persistent workgroups per CU, large render call */                        \
  X(wg_small, "HJ_WG_SMALL", 6, 1, 32)                     /* ... of a small one (a rank's share on 4 - 8 GPUs) */                      \
  X(recon_priority, "HJ_RECON_PRIORITY", 1, 0, 1)          /* reconstructions on a high-priority stream */                              \
  /* render calls */                                                                                                                   \
  X(batch_cap, "HJ_BATCH_CAP", 8192, 64, 32768)            /* blocks per batch, upper bound */                                          \
  X(wg_small_blocks, "HJ_WG_SMALL_BLOCKS", 12288, 0, 1 << 30) /* calls below this many blocks are "small" */                            \
  X(xcd_deal, "HJ_XCD_DEAL", 0, 0, 1)                      /* XCD-aware deal of sample groups (measured: -7 % on c4) */                 \
  X(lds_pad_kb, "HJ_LDS_PAD_KB", 0, 0, 64)                 /* dynamic LDS padding of the path kernel (occupancy experiments) */         \
  X(mem_limit_mb, "HJ_MEM_LIMIT_MB", 0, 0, 1 << 30)        /* test rig: pretend this little device memory is free */                    \
  X(alloc_limit_mb, "HJ_ALLOC_LIMIT_MB", 0, 0, 1 << 30)    /* test rig: allocations beyond this total fail (read once per process) */  \
  /* scene upload: what is derived from the uploaded tree, and where */                                                                \
  X(upload_device, "HJ_UPLOAD_DEVICE", -1, -1, 1)          /* re-layout on the device: -1 from upload_device_min nodes on */            \
  X(upload_device_min, "HJ_UPLOAD_DEVICE_MIN", 100000, 0, 1 << 30)                                                                      \
  X(upload_timing, "HJ_UPLOAD_TIMING", 0, 0, 1)            /* stage times on stderr */                                                  \
  X(pair_leaves, "HJ_PAIR_LEAVES", -1, -1, 1)              /* pair nodes: -1 from pair_min_nodes nodes on */                            \
  X(pair_min_nodes, "HJ_PAIR_MIN_NODES", 0, 0, 1 << 30)                                                                                 \
  X(collapse_pct, "HJ_COLLAPSE_PCT", 50, 0, 1000)          /* collapse threshold, per cent of the kept ancestor's area */               \
  X(leaf_guards, "HJ_LEAF_GUARDS", 2, 0, 3)                /* 2: triangle and quad leaves; 1 / 3 add the INEXACT sphere guards */       \
  X(node_order, "HJ_NODE_ORDER", -1, -1, 1)                /* cold nodes: 0 pre-order, 1 sibling groups, -1 by tree */                  \
  X(stream_state, "HJ_STREAM_STATE", -1, -1, 1)            /* non-temporal path-state accesses: -1 from stream_min_nodes nodes on */    \
  X(stream_min_nodes, "HJ_STREAM_MIN_NODES", 300000, 0, 1 << 30) /* "large tree" (also: 8 x 8 packets, no light grid, full shadow votes) */ \
  X(group_tile, "HJ_GROUP_TILE", -1, -1, 1)                /* camera packets of 8 x 8 pixels instead of 64 x 1: -1 on large trees */    \
  X(inner_burst, "HJ_INNER_BURST", kUnset, 1, 1 << 20)     /* box steps per round of the walk loop: 6 (8 on large trees) */            \
  X(refill_min, "HJ_REFILL_MIN", kUnset, 1, 64)            /* idle lanes that trigger a refill: 24 (32) */                              \
  X(light_grid, "HJ_LIGHT_GRID", kUnset, 0, 256)           /* cells per axis of the light-shaft grid: 64, none on large trees */        \
  X(light_grid_mesh, "HJ_LIGHT_GRID_MESH", 1, 0, 1)        /* also cells on meshes and in corners (flat shapes, not coplanar): bundle proofs */ \
  /* multi-GPU test rigs */                                                                                                            \
  X(comm_shared_gpu, "HJ_COMM_SHARED_GPU", 0, 0, 1)        /* several contexts of a communicator on one GPU (kernel sum) */             \
  X(comm_force_rccl, "HJ_COMM_FORCE_RCCL", 0, 0, 1)        /* RCCL also for a one-rank communicator */                                  \
  /* device BVH build + ray vote */                                                                                                    \
  X(lbvh_timing, "HJ_LBVH_TIMING", 0, 0, 1)                                                                                             \
  X(lbvh_big_pct, "HJ_LBVH_BIG_PCT", 2, 0, 100)            /* shapes above this share of the scene's box area stay out of the Morton tree */ \
  X(lbvh_cluster, "HJ_LBVH_CLUSTER", 512, 0, 1 << 20)      /* leaves per SAH-re-split cluster */                                        \
  X(lbvh_sah, "HJ_LBVH_SAH", 1, 0, 1)                                                                                                   \
  X(lbvh_top_rotate, "HJ_LBVH_TOP_ROTATE", -1, -1, 64)     /* rotation passes over the host-built top: -1 by size */                    \
  X(lbvh_vote_paths, "HJ_LBVH_VOTE_PATHS", 60000, 0, 1 << 24) /* camera paths of the ray vote at the end of the build */                \
  X(bvh_child_order, "HJ_BVH_CHILD_ORDER", 3, 0, 9)        /* static child order of the device build (3: fewer shapes first) */         \
  X(bvh_vote_shadow, "HJ_BVH_VOTE_SHADOW", kUnset, 0, 16)  /* a shadow ray's vote in quarters of a closest-hit ray's: 1 (4 on large trees) */
  // presence flags (debugging aids): set to anything = on
#define HJ_TUNING_FLAGS(X)                                                                                                              \
  X(trace_bounces, "HJ_TRACE_BOUNCES")                     /* per-bounce table of the split-kernel path on stderr */                    \
  X(relayout_debug, "HJ_RL_DEBUG")                         /* device re-layout: intermediate arrays checked on the host */              \
  X(light_grid_timing, "HJ_LIGHT_GRID_TIMING")             /* light-shaft grid: stage times on stderr */

#define HJ_X_FIELD(field, name, dflt, lo, hi) int field = dflt;
  HJ_TUNING_TABLE(HJ_X_FIELD)
#undef HJ_X_FIELD
#define HJ_X_FLAG(field, name) bool field = false;
  HJ_TUNING_FLAGS(HJ_X_FLAG)
#undef HJ_X_FLAG

  static int pick(int value, int where_unset) { return value == kUnset ? where_unset : value; }

  static Tuning from_env() {                                // THE place where the library reads its environment
    Tuning t;
    auto read = [](const char* name, int dflt, int lo, int hi) {
      const char* v = std::getenv(name);
      if (!v || !*v) return dflt;
      return std::min(hi, std::max(lo, std::atoi(v)));
    };
#define HJ_X_READ(field, name, dflt, lo, hi) t.field = read(name, dflt, lo, hi);
    HJ_TUNING_TABLE(HJ_X_READ)
#undef HJ_X_READ
#define HJ_X_READ_FLAG(field, name) t.field = std::getenv(name) != nullptr;
    HJ_TUNING_FLAGS(HJ_X_READ_FLAG)
#undef HJ_X_READ_FLAG
    return t;
  }
};

}  // namespace hjapi
