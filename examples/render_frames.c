/* Frames back to back from a pure C host: two caller-owned device buffers in turn, the batch pipeline never drained between
 * frames (hj_render_frame with HJ_RENDER_NO_DRAIN, hj_framebuffer_bind, hj_pipeline_wait) - frame k + 1 renders while frame
 * k is read back.  Every frame is then compared, bit for bit, with the same frame rendered by one blocking call.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/render_frames.c -Lhijiki_amd/lib -lhijiki_hip \
 *       -lhijiki_host -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/hijiki_amd/lib -Wl,-rpath,/opt/rocm/lib -o render_frames && ./render_frames
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "hijiki_hip.h"
#include "hijiki_host.h"

static void die(const char* what, const char* why) {
  fprintf(stderr, "%s: %s\n", what, why);
  exit(1);
}
#define HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) die(#call, hipGetErrorString(e_)); } while (0)

int main(void) {
  enum { W = 384, H = 256, SPP = 16, FRAMES = 5 };
  const size_t bytes = (size_t)W * H * 4 * sizeof(float);
  hjh_scene* scene = NULL;
  hjh_compiled* compiled = NULL;
  hj_scene_desc desc;
  hj_context* ctx = NULL;
  hj_render_opts opts;
  hj_render_stats totals;
  void* fb[2] = {NULL, NULL};
  float* got[FRAMES];
  float* want = (float*)malloc(bytes);
  int k, bad = 0;

  if (hjh_scene_make_synthetic(HJH_SYNTH_CBOX, 320, 1, &scene) != HJ_OK) die("scene", hjh_last_error());
  if (hjh_scene_compile(scene, &compiled) != HJ_OK) die("compile", hjh_last_error());
  if (hjh_compiled_desc(compiled, &desc) != HJ_OK) die("desc", hjh_last_error());
  if (hj_context_create(0, &ctx) != HJ_OK) die("context", hj_last_error(NULL));
  if (hj_scene_upload(ctx, &desc) != HJ_OK) die("upload", hj_last_error(ctx));
  HIP(hipMalloc(&fb[0], bytes));
  HIP(hipMalloc(&fb[1], bytes));
  if (hj_framebuffer_create(ctx, W, H, fb[0]) != HJ_OK) die("framebuffer", hj_last_error(ctx));
  hj_default_render_opts(&opts);
  opts.flags |= HJ_RENDER_NO_DRAIN;

  for (k = 0; k < FRAMES; k++) {
    void* target = fb[k % 2];
    got[k] = (float*)malloc(bytes);
    HIP(hipMemset(target, 0, bytes));                                   /* (the frame before last was read from it already) */
    HIP(hipStreamSynchronize(NULL));                                    /* the zeroes are in place before the frame's first splat */
    if (hj_framebuffer_bind(ctx, target) != HJ_OK) die("bind", hj_last_error(ctx));
    if (hj_render_frame(ctx, SPP, 100u + (uint64_t)k, 0, SPP, 0, 1, &opts, NULL) != HJ_OK) die("submit", hj_last_error(ctx));
    if (k >= 1) {                                                       /* frame k - 1 is complete; frame k renders on */
      if (hj_pipeline_wait(ctx, 1, NULL) != HJ_OK) die("wait", hj_last_error(ctx));
      HIP(hipMemcpy(got[k - 1], fb[(k - 1) % 2], bytes, hipMemcpyDeviceToHost));
    }
  }
  if (hj_pipeline_wait(ctx, 0, &totals) != HJ_OK) die("drain", hj_last_error(ctx));
  HIP(hipMemcpy(got[FRAMES - 1], fb[(FRAMES - 1) % 2], bytes, hipMemcpyDeviceToHost));
  printf("%d frames back to back: %llu paths, %llu rays, %.2f ms\n", (int)FRAMES, (unsigned long long)totals.paths,
         (unsigned long long)(totals.closest_rays + totals.shadow_rays), totals.total_ms);

  if (hj_framebuffer_bind(ctx, fb[0]) != HJ_OK) die("bind", hj_last_error(ctx));
  for (k = 0; k < FRAMES; k++) {                                        /* the same frames, one blocking call each */
    HIP(hipMemset(fb[0], 0, bytes));
    if (hj_render_frame(ctx, SPP, 100u + (uint64_t)k, 0, SPP, 0, 1, NULL, NULL) != HJ_OK) die("render", hj_last_error(ctx));
    HIP(hipMemcpy(want, fb[0], bytes, hipMemcpyDeviceToHost));
    if (memcmp(want, got[k], bytes) != 0) { fprintf(stderr, "frame %d differs\n", k); bad++; }
    free(got[k]);
  }
  printf("%s\n", bad ? "MISMATCH" : "all frames bit-identical to the blocking calls");
  free(want);
  hj_context_destroy(ctx);
  HIP(hipFree(fb[0]));
  HIP(hipFree(fb[1]));
  hjh_compiled_destroy(compiled);
  hjh_scene_destroy(scene);
  return bad ? 1 : 0;
}
