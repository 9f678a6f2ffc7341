/*
 * hijiki_hip.h — the drop-in boundary of the MI355X path-tracing hot path.
 *
 * Plain C ABI: POD structs, pointers and sizes only.  Every record below is
 * byte-for-byte the record the reference host (`/root/reference/src/main.rs`)
 * writes into its scene buffer and that the reference shaders read; every
 * entry point replaces one piece of the reference's host<->shader contract
 * (there is no plugin interface in the reference — the boundary is what
 * `Renderer::new/render/save_image` hand to wgpu).  A Rust host would bind to
 * this header with an `extern "C"` block.
 *
 * Citations `file:line` are relative to the reference checkout.
 */
#ifndef HIJIKI_HIP_H
#define HIJIKI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)   /* libraries are built with -fvisibility=hidden */
#endif

/* ------------------------------------------------------------------ status */

enum hj_status {
  HJ_OK = 0,
  HJ_ERR_INVALID = 1,     /* bad argument / inconsistent scene (reference: assert!/panic!) */
  HJ_ERR_DEVICE = 2,      /* HIP runtime error                                              */
  HJ_ERR_NOMEM = 3,       /* host or device allocation failed                               */
  HJ_ERR_STATE = 4,       /* call order violated (e.g. render before scene upload)          */
  HJ_ERR_UNSUPPORTED = 5  /* reference behaviour that is undefined / a panic upstream       */
};

/* ------------------------------------------------------- material tag words */

/* MaterialType discriminants, src/main.rs:34-45; shader macros MATERIAL_TAG_*
 * injected at src/main.rs:778-783.  A material word is (tag << 24) + index
 * (src/main.rs:275; decoded at shader/render.glsl:107-109). */
enum hj_material_tag {
  HJ_MAT_DIFFUSE = 0,
  HJ_MAT_DIFFUSECBOARD = 1,
  HJ_MAT_MIRROR = 2,
  HJ_MAT_DIELECTRIC = 3,
  HJ_MAT_EMISSIVE = 4
};
#define HJ_MATERIAL_TAG_SHIFT 24u
#define HJ_MATERIAL_INDEX_MASK 0x00FFFFFFu
#define HJ_BVH_INNER 0xFFFFFFFFu     /* shape_index of an inner node, src/main.rs:219 */
#define HJ_BVH_ROOT_EXIT 1000000u    /* exit index of the root,       src/main.rs:231 */
#define HJ_BLOCK_SIZE 128u           /* src/main.rs:1485 */

/* ------------------------------------------------- device records (std430) */

/* Camera, src/main.rs:154-160 / shader/render.glsl:12-16.  48 bytes. */
typedef struct hj_camera {
  float position[4];   /* xyz used, w = 0                                  */
  float rotation[4];   /* unit quaternion xyzw                             */
  float fov;           /* HORIZONTAL field of view in degrees              */
  float _pad[3];
} hj_camera;

/* SceneBufferInfo, src/main.rs:400-408 / shader/scene.glsl:1-8.  64 bytes. */
typedef struct hj_scene_info {
  hj_camera camera;
  uint32_t num_spheres, num_quads, num_triangles, num_emitters;
} hj_scene_info;

/* CompiledBVHNode, src/main.rs:92-99 / shader/scene.glsl:10-13.  32 bytes.
 * Pre-order skip-link array; see hj_host.h for the builder. */
typedef struct hj_bvh_node {
  float aabb_min[3];
  uint32_t shape_index;  /* global shape index of a leaf, HJ_BVH_INNER otherwise */
  float aabb_max[3];
  uint32_t exit_index;   /* node to visit when this subtree is skipped / done   */
} hj_bvh_node;

/* Direction classes of a ray (no counterpart upstream: shader/scene.glsl:97-133 visits the two children of a node in ARRAY order
 * whatever the ray's direction).  A "directional" tree is K skip-link arrays over the same boxes and leaves that differ only in
 * which child of a node comes first; a ray walks array hj_ray_direction_class(mode, d) from its root to its end.  The class is
 * a function of the SIGN BITS of the direction's components (mode 1..7: bit a of `mode` selects axis a, K = 2^axes, e.g. 7 = the
 * eight octants) or of its major axis and that component's sign (mode 8, K = 6).  One text for the host compiler, the oracle and
 * the kernels. */
#define HJ_DIR_MODE_NONE 0
#define HJ_DIR_MODE_OCTANTS 7
#define HJ_DIR_MODE_MAJOR_AXIS 8
#define HJ_DIR_MAX_CLASSES 8
static inline int hj_direction_classes(int mode) {
  if (mode <= 0 || mode > 8) return 1;
  if (mode == 8) return 6;
  return 1 << ((mode & 1) + ((mode >> 1) & 1) + ((mode >> 2) & 1));
}
static inline int hj_ray_direction_class(int mode, const float d[3]) {
  uint32_t b[3];
  int a, k = 0, cls = 0;
  if (mode <= 0 || mode > 8) return 0;
  for (a = 0; a < 3; a++) { union { float f; uint32_t u; } c; c.f = d[a]; b[a] = c.u; }
  if (mode == 8) {
    int major = 0;
    uint32_t best = b[0] & 0x7FFFFFFFu;
    for (a = 1; a < 3; a++) if ((b[a] & 0x7FFFFFFFu) > best) { best = b[a] & 0x7FFFFFFFu; major = a; }
    return 2 * major + (int)(b[major] >> 31);
  }
  for (a = 0; a < 3; a++) if (mode & (1 << a)) cls |= (int)(b[a] >> 31) << k++;
  return cls;
}

/* Sphere, src/shape.rs:6-11 / shader/shapes/sphere.glsl:1-3.  16 bytes. */
typedef struct hj_sphere { float center[3]; float radius; } hj_sphere;

/* Quad, src/shape.rs:22-31 / shader/shapes/quad.glsl:1-5.  48 bytes. */
typedef struct hj_quad {
  float origin[3]; float _pad1;
  float edge1[3];  float _pad2;
  float edge2[3];  float _pad3;
} hj_quad;

/* Triangle = 3 indices into the global vertex array, src/main.rs:51,386. */
typedef struct hj_triangle { uint32_t v[3]; } hj_triangle;

/* Vertex, src/main.rs:54-60 / shader/shapes/triangle.glsl:1-4.  32 bytes. */
typedef struct hj_vertex { float pos[3]; float u; float normal[3]; float v; } hj_vertex;

/* Emitter, src/main.rs:368-374 / shader/scene.glsl:33-38.  16 bytes. */
typedef struct hj_emitter { uint32_t shape; float pdf; float cdf; float _pad; } hj_emitter;

/* Material records, src/main.rs:102-146 / shader/materials/{diffuse,diffusecb,dielectric,emissive}.glsl. */
typedef struct hj_diffuse    { float color[3]; float _pad; } hj_diffuse;
typedef struct hj_diffuse_cb { float color_a[3]; float scale_u; float color_b[3]; float scale_v; } hj_diffuse_cb;
typedef struct hj_dielectric { float extinction[3]; float eta; } hj_dielectric;
typedef struct hj_emissive   { float power[3]; float _pad; } hj_emissive;

/* ImageBlock, src/main.rs:608-617 / shader/block.glsl:1-8.  40 bytes.
 * One per integrator+reconstruction dispatch pair in the reference
 * (src/main.rs:1322-1330). */
typedef struct hj_image_block {
  uint32_t id;
  uint32_t seed;
  uint32_t origin[2];
  uint32_t dimension[2];
  uint32_t original_dimension[2];
  float sample_offset[2];
} hj_image_block;

/* ------------------------------------------------------------ scene upload */

/* What CompiledScene::write_to_buffer (src/main.rs:561-605) packs into the
 * scene buffer, as 11 borrowed (pointer,count) arrays + the camera, in the
 * reference's sub-buffer order.  Global shape index space is
 * [spheres | quads | triangles]; `materials` has one word per shape in that
 * order (src/main.rs:278-287).  Arrays are borrowed for the duration of the
 * call only. */
typedef struct hj_scene_desc {
  hj_camera camera;
  const hj_bvh_node*   bvh;        size_t num_bvh_nodes;
  const hj_sphere*     spheres;    size_t num_spheres;
  const hj_quad*       quads;      size_t num_quads;
  const hj_triangle*   triangles;  size_t num_triangles;
  const hj_vertex*     vertices;   size_t num_vertices;
  const uint32_t*      materials;  size_t num_materials;   /* == number of shapes */
  const hj_emitter*    emitters;   size_t num_emitters;
  const hj_diffuse*    diffuse;    size_t num_diffuse;
  const hj_diffuse_cb* diffusecb;  size_t num_diffusecb;
  const hj_dielectric* dielectric; size_t num_dielectric;
  const hj_emissive*   emissive;   size_t num_emissive;
} hj_scene_desc;

/* Compile-time shader parameters of the reference, as run-time options:
 * USE_BVH (src/main.rs:769-777), RECONSTRUCTION_RADIUS / _STDDEV
 * (src/main.rs:916-921,1279-1286), the 1000-bounce cap and the `bounce > 3`
 * roulette start (shader/render.glsl:92,137). */
typedef struct hj_render_opts {
  uint32_t use_bvh;        /* 1 = skip-link BVH walk, 0 = linear scan (reference CLI default) */
  uint32_t recon_radius;   /* only 2 is supported (reference value)                           */
  float    recon_stddev;   /* 0.5 in the reference                                            */
  uint32_t max_bounces;    /* 1000                                                            */
  uint32_t rr_start;       /* roulette applies when bounce > rr_start - 1, i.e. 4 -> b > 3    */
  uint32_t batch_blocks;   /* blocks traced per wavefront batch; 0 = library default          */
  uint32_t flags;          /* HJ_RENDER_* bits                                                */
  uint32_t _reserved;
} hj_render_opts;

#define HJ_RENDER_TIME_KERNELS 1u   /* bracket every kernel with HIP events on its launch stream (fills *_ms) */
#define HJ_RENDER_SPLIT_KERNELS 2u  /* diagnostic: one launch per stage per bounce instead of the fused kernel */
#define HJ_RENDER_STATIC_DEAL 4u    /* hj_render_frame, world > 1: keep all passes of a block on one rank          */
#define HJ_RENDER_NO_DRAIN 8u       /* hj_render_frame: return when the frame's batches are ENQUEUED (hj_pipeline_wait) */
#define HJ_RENDER_NO_LIGHT_GRID 16u /* walk every shadow ray, also those the light-shaft grid proves unoccluded (same image) */

/* Per-render statistics (device counters; all in units of events). */
typedef struct hj_render_stats {
  uint64_t paths;            /* camera paths started (= sum of block areas)         */
  uint64_t closest_rays;     /* intersectScene(ray, its) calls                      */
  uint64_t shadow_rays;      /* intersectScene(ray) calls                           */
  uint64_t batches;          /* wavefront batches launched                          */
  uint64_t bounce_rounds;    /* per-bounce kernel rounds over all batches           */
  double   trace_closest_ms; /* HIP-event time of the closest-hit kernels (if timed)*/
  double   trace_shadow_ms;
  double   shade_ms;
  double   reconstruct_ms;
  double   total_ms;         /* host wall time of the call (all batches, both streams) */
  uint64_t closest_launches; /* number of closest-hit kernel launches timed         */
  double   path_ms;          /* HIP-event time of the fused k_path_wavefront launches (sum; launches overlap) */
  uint64_t path_launches;
  uint64_t hits;             /* closest-hit rays that hit a shape                   */
  uint64_t unoccluded_shadow_rays; /* shadow rays that reached their light           */
  double   path_busy_ms;     /* union of the path kernels' intervals = their exclusive GPU time (if timed) */
  uint64_t shadow_rays_proven_free; /* of shadow_rays (and of unoccluded_shadow_rays): answered by the light-shaft grid, no walk */
} hj_render_stats;

typedef struct hj_context hj_context;

/* ---------------------------------------------------------------- lifecycle */

/* Replaces GPU::new + Renderer::new resource creation (src/main.rs:684-713,
 * 1167-1314).  One context per GPU; no global state. */
int hj_context_create(int device_ordinal, hj_context** out_ctx);
void hj_context_destroy(hj_context* ctx);
/* Replaces unwrap()/panic! text: message of the last failing call on ctx, copied into a buffer of the CALLING thread
 * (valid until that thread's next hj_last_error call; the context's worker thread may fail at any time).  ctx may be
 * NULL for create errors. */
const char* hj_last_error(const hj_context* ctx);
/* Library / ABI version: (major<<16)|(minor<<8)|patch. */
uint32_t hj_version(void);
void hj_default_render_opts(hj_render_opts* opts);

/* -------------------------------------------------------------------- scene */

/* Replaces the staging copy of the packed scene buffer (src/main.rs:1186-1244)
 * and the bind-group plumbing (src/main.rs:808-855).  Validates the same
 * invariants the reference asserts (src/main.rs:562-565) plus index ranges,
 * copies, and re-lays the data out for the kernels (same boxes, same leaves, same visiting order per ray: DESIGN.md 4).
 * scene->bvh (NULL with num_bvh_nodes == 0: the tree hj_build_bvh_device left on the device, see there) must be what Scene::compile flattens (src/main.rs:203-231): a binary tree in pre-order with skip links (left child =
 * next record, its exit = the right child, a node's exit = the record behind its subtree or, on the right spine, any index >= the
 * node count); the boxes may be anything (a box that does not bound its subtree just culls what the reference would cull), other
 * link structures return HJ_ERR_INVALID.
 * Limit: the device node array - 32 bytes per record, TWO copies of the tree (the re-laid-out one and the reference's own, which
 * rays with a zero or non-finite direction component walk: DESIGN.md 4) - must fit in 4 GiB: about 67 M records per tree = 33 M shapes;
 * larger trees return HJ_ERR_UNSUPPORTED. */
int hj_scene_upload(hj_context* ctx, const hj_scene_desc* scene);

/* -------------------------------------------------------------- framebuffer */

/* Replaces creation of `final_texture` (W x H RGBA32F = (sum w*rgb, sum w),
 * src/main.rs:1209-1234).  Zero-initialised.  If `external_device_rgba` is
 * non-NULL the context accumulates into that caller-owned device buffer of
 * W*H*4 floats (tight pitch) instead of allocating its own — this is how a
 * torch/RCCL host shares the buffer for the multi-GPU reduce. */
int hj_framebuffer_create(hj_context* ctx, uint32_t width, uint32_t height,
                          void* external_device_rgba);
int hj_framebuffer_clear(hj_context* ctx);
/* Device pointer of the accumulation buffer (W*H*4 floats). */
void* hj_framebuffer_device_ptr(hj_context* ctx);
/* Replaces Renderer::save_image's texture->buffer copy + map (src/main.rs:
 * 1357-1394), tight pitch instead of ceil256(W) texels. */
int hj_framebuffer_read(hj_context* ctx, float* host_rgba /* W*H*4 */);
/* rgb / w of every pixel (src/main.rs:1395-1400, shader/preview.glsl:11). */
int hj_framebuffer_resolve(hj_context* ctx, float* host_rgb /* W*H*3 */);

/* ------------------------------------------------------------------- render */

/* Optional set-up step (no counterpart upstream: Renderer::new creates every resource it needs, src/main.rs:1167-1314):
 * allocates the batch slots - path state and sample buffers - that a render call of `total_blocks` ImageBlocks with
 * these options will use (for hj_render_frame: spp x blocks per pass / world), so that the first frame does not pay for
 * them (50 GB and 0.8 s for the benchmark's frames at the defaults).  A render call allocates whatever is missing.
 * Preconditions: a scene has been uploaded and a framebuffer created (HJ_ERR_STATE otherwise, as for a render call: the
 * sizes depend on both); no asynchronous frame in flight.  MEMORY: the defaults take up to 17 % of a 288 GB device (three
 * batch slots of 12.4 GB of path state + 4.3 GB of samples each, for calls of 32768 ImageBlocks and more; a call of n
 * blocks takes about n x 2.7 MB up to that; HJ_POOL).  Both hj_reserve and the render calls first fit their request to the free
 * device memory (hipMemGetInfo) and, when an allocation fails all the same - another context or the host application took
 * the memory in between -, give back every slot, halve the positions per workgroup (down to 1024), then the batch (down to
 * 64 ImageBlocks), then run ONE batch slot instead of three, and try again: HJ_ERR_NOMEM is returned only when the smallest
 * configuration (about 100 MB) does not fit, or hj_render_opts::batch_blocks fixed a batch that does not.  Slots keep what they hold until the context is destroyed or a larger request re-allocates them. */
int hj_reserve(hj_context* ctx, size_t total_blocks, const hj_render_opts* opts /* NULL = defaults */);

/* Replaces the body of Renderer::render (src/main.rs:1316-1355): for every
 * block IN ORDER, integrate (shader/render.glsl:149-175) and accumulate
 * (shader/reconstruction.glsl:22-66).  The result equals running the
 * reference's two dispatches per block sequentially; internally many blocks
 * are traced as one wavefront batch.  Blocks must satisfy
 * dimension <= 128 and original_dimension == framebuffer size.
 * Synchronous: returns when the framebuffer holds the result. */
int hj_render_blocks(hj_context* ctx, const hj_image_block* blocks, size_t num_blocks,
                     const hj_render_opts* opts /* NULL = defaults */,
                     hj_render_stats* stats /* may be NULL */);

/* Deterministic ImageBlockGenerator (src/main.rs:619-682) + render of the
 * blocks owned by `rank` out of `world` (hj_block_owner: a diagonal deal over the block grid that
 * moves one step per pass), for passes [pass_begin, pass_end).  With world == 1 this is the whole frame.
 * The reference seeds blocks from the OS RNG (src/main.rs:643,670,675);
 * here seeds/offsets come from hj_block_seed / hj_pass_offset below. */
int hj_render_frame(hj_context* ctx, uint32_t spp, uint64_t master_seed,
                    uint32_t pass_begin, uint32_t pass_end,
                    uint32_t rank, uint32_t world,
                    const hj_render_opts* opts, hj_render_stats* stats);

/* ------------------------------------------------------------- BVH on device */

/* The tree of `Scene::compile` (src/main.rs:199-231) built on the device instead of by the host's SAH builder
 * (SURVEY.md 8f #2): a Morton-code LBVH over the shapes of `scene` (scene->bvh is ignored; camera, materials and emitters feed the
 * ray vote at its end: hj_tune_bvh_device), written to
 * out_nodes in the reference's flattened format - one shape per leaf, pre-order with skip links, every record
 * holding the bounds of its own subtree, 2 * shapes - 1 records.  Start-up path for large meshes (1 M triangles
 * in milliseconds); its topology is not the host builder's, which changes images only through epsilon-ties and
 * traversal cost.  Put the result into scene->bvh before hj_scene_upload. */
int hj_build_bvh_device(hj_context* ctx, const hj_scene_desc* scene, hj_bvh_node* out_nodes /* may be NULL */, size_t capacity,
                        size_t* out_num_nodes /* may be NULL */);
/* The device route without the host in the middle: the tree hj_build_bvh_device builds also STAYS on the device, with the shape
 * arrays it was built over, until the next build, until an upload takes it over, or until the context ends.
 *   hj_build_bvh_device(ctx, scene, NULL, 0, &n)      builds; nothing is copied to the host
 *   hj_scene_upload(ctx, scene) with scene->bvh == NULL and scene->num_bvh_nodes == 0
 *                                                     derives the kernels' records from that tree on the device (the same shape
 *                                                     counts as at the build: HJ_ERR_INVALID otherwise; no tree: HJ_ERR_STATE)
 *   hj_bvh_device_read(ctx, out_nodes, capacity, &n)  a copy of the tree for a host that wants one (out_nodes NULL: only n)
 * 1 M triangles: build + upload in tens of milliseconds (profiles/r06_startup_1M_triangles.txt).  Results are those of the same
 * tree handed over through the host, bit for bit. */
int hj_bvh_device_read(hj_context* ctx, hj_bvh_node* out_nodes, size_t capacity, size_t* out_num_nodes /* may be NULL */);

/* The child order of a flattened tree, voted by a sample of the scene's own rays - on the device (no counterpart upstream: the
 * reference walks the tree the `bvh` crate hands it, src/main.rs:199-231, children in array order, shader/scene.glsl:97-133; host
 * form of the same pass: hjh_compiled_tune_bvh).  `vote_paths` camera paths of `scene` (camera, materials, emitters) are traced
 * through scene->bvh; every ray that hits votes, at the ancestors of its leaf, for the order that would have spared it more of
 * the other child; where the sample says so the two children of a node change places.  out_nodes receives the same tree - same
 * boxes, same leaves - as another valid pre-order skip-link array of scene->num_bvh_nodes records (60 000 paths: a few
 * milliseconds at 1 M triangles).  hj_build_bvh_device ends with this pass (HJ_LBVH_VOTE_PATHS, default 60000, 0 = off).  Images
 * change only through epsilon-ties, traversal cost falls. */
int hj_tune_bvh_device(hj_context* ctx, const hj_scene_desc* scene, hj_bvh_node* out_nodes, size_t capacity, size_t vote_paths);

/* ------------------------------------------------- asynchronous frame, progress */

/* hj_render_frame on the context's worker thread (one persistent thread per context, started by the first call):
 * returns at once, hj_sync waits for the frame and returns its status and statistics.  One host thread can so keep one
 * context per GPU rendering at the same time, and the drain of one context overlaps whatever the host does next (the
 * reduce waits for all of them).  At most one frame in flight per context: until it has finished every other entry
 * point that touches the context's device state (scene upload, framebuffer create / clear / read / resolve, the render
 * calls, the probes, hj_build_bvh_device) returns HJ_ERR_STATE.  The result of the last asynchronous frame stays
 * retrievable - hj_sync after hj_comm_reduce_framebuffers, which joins the frames itself, still returns the statistics -
 * until the next hj_render_frame_async; hj_sync on a context that never ran one returns HJ_OK and leaves *stats alone.
 * (No counterpart in the reference, whose submit loop src/main.rs:1316-1355 is synchronous with respect to the host but
 * never waits for the device.) */
int hj_render_frame_async(hj_context* ctx, uint32_t spp, uint64_t master_seed,
                          uint32_t pass_begin, uint32_t pass_end, uint32_t rank, uint32_t world,
                          const hj_render_opts* opts);
int hj_sync(hj_context* ctx, hj_render_stats* stats /* may be NULL */);

/* Frames BACK TO BACK without draining the batch pipeline between them.  (Upstream the queue never drains either: render()
 * submits one command buffer per block and never waits, src/main.rs:1341-1347; a blocking hj_render_frame ends with the
 * path-depth tail of its last batches running alone - 3 ms of a 160 ms frame on one GPU, of a 20 ms share on eight.)
 *   hj_render_frame(..., opts->flags | HJ_RENDER_NO_DRAIN, stats = NULL)
 *       returns when the frame's batches are enqueued; it blocks only while all batch slots are still busy with EARLIER
 *       batches (the natural back-pressure).  The frame accumulates into the framebuffer that is bound at the time of the
 *       call; it must have been zeroed (or hold what the frame is to be added to) before the call.
 *   hj_framebuffer_bind(ctx, device_ptr)
 *       frames submitted from now on accumulate into this caller-owned W x H RGBA32F buffer (16-byte aligned; same size as the
 *       one hj_framebuffer_create was given) - no synchronisation, frames already submitted keep their buffer: two buffers in
 *       turn let frame k + 1 render while frame k is reduced / read.
 *   hj_pipeline_wait(ctx, keep, totals)
 *       waits until at most `keep` of the submitted frames are still in flight (oldest first).  keep = 0 drains everything
 *       and returns in *totals (may be NULL) the statistics of ALL frames since the last drain (times: kernels of all of
 *       them); with keep > 0 *totals is left alone.
 * While frames are in flight the other entry points that touch the context's device state return HJ_ERR_STATE (as with an
 * asynchronous frame), except hj_framebuffer_bind and hj_framebuffer_device_ptr.  Results are the blocking call's, bit for bit. */
int hj_framebuffer_bind(hj_context* ctx, void* external_device_ptr);
int hj_pipeline_wait(hj_context* ctx, uint32_t keep, hj_render_stats* totals /* may be NULL */);

/* Replaces the window-title percentage the reference updates every `present_interval` blocks (src/main.rs:1335-1340):
 * `fn(user, blocks_done, blocks_total)` is called from the thread that drives the render whenever at least
 * `interval_blocks` more ImageBlocks have COMPLETED on the device (granularity: one wavefront batch), and once at the
 * end of the call.  fn == NULL switches reporting off. */
typedef void (*hj_progress_fn)(void* user, uint64_t blocks_done, uint64_t blocks_total);
void hj_set_progress_callback(hj_context* ctx, hj_progress_fn fn, void* user, uint32_t interval_blocks);

/* Number of HIP devices visible to the process (0 without a GPU). */
int hj_device_count(void);

/* ---------------------------------------------------------------- multi-GPU */

/* New with the multi-GPU tile sharding (no counterpart in the reference).  A communicator holds the `n` contexts of
 * THIS process (one per GPU) and their RCCL communicators (ncclCommInitAll once, reused by every reduce; RCCL is loaded
 * with dlopen on first use, a copy the process has already mapped is reused; n == 1 needs no RCCL). */
typedef struct hj_comm hj_comm;
int hj_comm_create(hj_context* const* ctxs, int n, hj_comm** out_comm);
void hj_comm_destroy(hj_comm* comm);
/* Element-wise SUM of the framebuffers (equal sizes) into ctxs[root]: waits for frames in flight (hj_sync) on every
 * context, then one ncclReduce per GPU inside a group call over xGMI.  Resolve rgb/w only after the reduce.  Errors
 * are reported on ctxs[root].  Hosts that run one process per GPU reduce the external framebuffer themselves instead
 * (hijiki_amd/dist.py does, through torch.distributed). */
int hj_comm_reduce_framebuffers(hj_comm* comm, int root);
/* The same without a communicator object: the communicators of the context list are created on the first call and
 * kept until one of the contexts is destroyed. */
int hj_reduce_framebuffers(hj_context* const* ctxs, int n, int root);

/* ------------------------------------------------------------------- probes */

/* Function-level probes used by the parity tests (no counterpart in the
 * reference; they expose intermediate values of the same kernels).
 * hj_debug_trace: intersectScene (shader/scene.glsl:92-175) for caller-given
 *   rays.  rays = n x 8 floats (origin.xyz, direction.xyz, tMin, tMax);
 *   hits = n x 4 floats (objectID as int32 bits or -1, t, u, v of the raw hit
 *   before populate*).  any_hit != 0 stops at the first accepted hit (the
 *   shadow-ray form): then only `objectID >= 0` is meaningful.
 * hj_debug_samples: the intermediate image of ONE block (layers 0 and 1 of
 *   shader/render.glsl:172-173): dimension.y x dimension.x x 8 floats
 *   (radiance rgb, 1, first-hit normal xyz, first-hit t), no reconstruction,
 *   framebuffer untouched. */
int hj_debug_trace(hj_context* ctx, const float* rays, size_t n, uint32_t use_bvh, uint32_t any_hit, float* hits);
int hj_debug_samples(hj_context* ctx, const hj_image_block* block, const hj_render_opts* opts, float* samples);
/* hj_debug_light_grid: the light-shaft visibility grid hj_scene_upload would build for `scene` (pure host code: no context,
 *   no GPU).  Returns the cells per axis (0: nothing can be proven for this scene).  bits (may be NULL) = res^3 bytes, x fastest:
 *   bit e of a cell set = every next-event shadow ray from a hit point in that cell to emitter e is unoccluded; the cell of a
 *   point p along axis k is (int)((p[k] - lo[k]) * inv[k]) in float arithmetic; stats = cells that hold a surface, cells whose
 *   surfaces lie in one plane, (cell, emitter) pairs proven free.  tests/test_light_grid.py attacks the claim with the oracle. */
int hj_debug_light_grid(const hj_scene_desc* scene, uint32_t res, uint8_t* bits, float lo[3], float inv[3], uint64_t stats[3]);
/* hj_debug_light_grid_planes: the same grid with its two kinds of proof apart.  planar (res^3 bytes): bits that hold for EVERY hit
 *   point of the cell (all shapes of the cell in one plane); mesh (res^3 bytes): bits of cells on meshes and in corners, which hold
 *   for a hit point that lies on its shape - a grazing hit leaves the reference's hit point off it, so the shade stage checks
 *   (DESIGN.md section 4): with recs, 8 floats per quad and triangle (shape index - num_spheres) = unit normal n, margin delta;
 *   vertex a, 0 (triangle) / 1 (quad), and limits = {sin_in, slide}: |d.n| >= sin_in |d|, (|n.(p - a)| + 3e-7 |p - a|_1) |d| <= slide |d.n|,
 *   min(u, v, 1 - u - v) >= delta (a quad: u, v, 1 - u, 1 - v) for the ray (o, d) that hit and its (t, u, v).
 *   hj_debug_light_grid's bits = planar | mesh.  Any output may be NULL. */
int hj_debug_light_grid_planes(const hj_scene_desc* scene, uint32_t res, uint8_t* planar, uint8_t* mesh, float* recs, float limits[2],
                               float lo[3], float inv[3], uint64_t stats[3]);

/* The deterministic replacement of `rand::random()` in the block generator.
 * Pure functions (no context); the same definitions are used by the host
 * library's ImageBlockGenerator and restated by the oracle. */
uint32_t hj_block_seed(uint64_t master_seed, uint32_t pass, uint32_t block_in_pass);
void hj_pass_offset(uint64_t master_seed, uint32_t offset_index, float out_xy[2]);
/* Rank that renders block j (raster index inside a pass) of pass `pass` of a width x height frame:
 * (column + row + pass) mod world.  With HJ_RENDER_STATIC_DEAL hj_render_frame uses pass = 0 for every pass. */
uint32_t hj_block_owner(uint32_t width, uint32_t height, uint32_t pass, uint32_t block_in_pass, uint32_t world);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* HIJIKI_HIP_H */
