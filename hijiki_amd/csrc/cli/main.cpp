// hijiki-hip — command line of the reference (`struct Opt` / `main`, src/main.rs:1426-1494) on the MI355X path:
//
//   hijiki-hip [--put-cbox-spheres] [--use-bvh] [-w/--width 800] [-h/--height 600] [--present-interval 128]
//              [-s/--sample-count 64] [-o/--output-image /tmp/output.exr] [--seed 1] <scene.obj | synthetic:KIND>
//
// Same flags and defaults (including `-h` meaning height and brute-force traversal unless --use-bvh).  There is
// no preview window, so --present-interval is accepted and ignored; --seed replaces the OS-seeded block RNG.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/hijiki_hip.h"
#include "../../../include/hijiki_host.h"
#include "../host/scene.hpp"

namespace {

struct Opt {
  bool put_cbox_spheres = false, use_bvh = false, device_bvh = false;
  uint32_t width = 800, height = 600, present_interval = 128, sample_count = 64;
  uint64_t seed = 1;
  std::string output_image = "/tmp/output.exr", scene;
};

[[noreturn]] void usage(const char* msg) {
  if (msg) std::fprintf(stderr, "error: %s\n\n", msg);
  std::fprintf(stderr,
               "USAGE: hijiki-hip [FLAGS] [OPTIONS] <scene>\n\n"
               "FLAGS:\n    --put-cbox-spheres    Add a mirror and glass sphere to the scene\n"
               "    --use-bvh             Use a BVH to optimize intersections\n"
               "    --device-bvh          (not upstream) build the tree on the GPU (LBVH, stays there: fast start) instead of on the host (SAH)\n\n"
               "OPTIONS:\n    -h, --height <height>                        [default: 600]\n"
               "    -o, --output-image <output-image>            [default: /tmp/output.exr] (.exr, .pfm or .png)\n"
               "        --present-interval <present-interval>    [default: 128] (ignored: no preview window)\n"
               "    -s, --sample-count <sample-count>            [default: 64]\n"
               "        --seed <seed>                            [default: 1]\n"
               "    -w, --width <width>                          [default: 800]\n\n"
               "ARGS:\n    <scene>    The scene (OBJ file) to render, or synthetic:cbox | synthetic:spheres | synthetic:mesh\n");
  std::exit(msg ? 2 : 0);
}

Opt parse(int argc, char** argv) {
  Opt o;
  auto value = [&](int& i) -> std::string {
    if (i + 1 >= argc) usage((std::string("missing value for ") + argv[i]).c_str());
    return argv[++i];
  };
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    if (a == "--put-cbox-spheres") o.put_cbox_spheres = true;
    else if (a == "--use-bvh") o.use_bvh = true;
    else if (a == "--device-bvh") o.device_bvh = true;
    else if (a == "-w" || a == "--width") o.width = (uint32_t)std::stoul(value(i));
    else if (a == "-h" || a == "--height") o.height = (uint32_t)std::stoul(value(i));
    else if (a == "--present-interval") o.present_interval = (uint32_t)std::stoul(value(i));
    else if (a == "-s" || a == "--sample-count") o.sample_count = (uint32_t)std::stoul(value(i));
    else if (a == "-o" || a == "--output-image") o.output_image = value(i);
    else if (a == "--seed") o.seed = std::stoull(value(i));
    else if (a == "--help") usage(nullptr);
    else if (!a.empty() && a[0] == '-') usage(("unknown flag " + a).c_str());
    else if (o.scene.empty()) o.scene = a;
    else usage("more than one scene given");
  }
  if (o.scene.empty()) usage("the <scene> argument is required");
  return o;
}

void check(hj_context* ctx, int rc, const char* what) {
  if (rc != HJ_OK) throw std::runtime_error(std::string(what) + ": " + hj_last_error(ctx));
}

}  // namespace

int main(int argc, char** argv) {
  try {
    const Opt opt = parse(argc, argv);
    hijiki::Scene scene;
    if (opt.scene.rfind("synthetic:", 0) == 0) {
      const std::string kind = opt.scene.substr(10);
      scene = hijiki::make_synthetic(kind == "spheres" ? HJH_SYNTH_CBOX_SPHERES : kind == "mesh" ? HJH_SYNTH_CBOX_MESH : HJH_SYNTH_CBOX,
                                     kind == "mesh" ? 1000000u : 0u, 1);
    } else {
      scene = hijiki::scene_from_obj(opt.scene);                       // Scene::from_obj, src/main.rs:1462
    }
    if (opt.put_cbox_spheres) hijiki::put_cbox_spheres(scene);         // src/main.rs:1463-1483
    std::printf("Building BVH\n");                                     // src/main.rs:198
    // --device-bvh: the fast start - no tree on the host at all; hj_build_bvh_device leaves its tree on the device and
    // hj_scene_upload (scene->bvh == NULL) takes it over there: a 1 M-triangle scene is renderable 40 ms after its shapes exist
    hijiki::CompiledScene cs = hijiki::compile(scene, !opt.device_bvh);  // src/main.rs:1486
    hj_context* ctx = nullptr;
    if (hj_context_create(0, &ctx) != HJ_OK) throw std::runtime_error(hj_last_error(nullptr));
    size_t num_nodes = cs.bvh.size();
    const hj_scene_desc desc = cs.desc();                              // (bvh == NULL, num_bvh_nodes == 0 on the device route)
    if (opt.device_bvh) check(ctx, hj_build_bvh_device(ctx, &desc, nullptr, 0, &num_nodes), "device BVH build");
    std::printf("Built BVH with %zu nodes\n", num_nodes);              // src/main.rs:200

    check(ctx, hj_scene_upload(ctx, &desc), "scene upload");
    check(ctx, hj_framebuffer_create(ctx, opt.width, opt.height, nullptr), "framebuffer");
    hj_render_opts ro;
    hj_default_render_opts(&ro);
    ro.use_bvh = opt.use_bvh ? 1u : 0u;                                // --use-bvh, src/main.rs:1432-1434
    hj_render_stats st;
    // the window title of the preview shows the percentage, updated every --present-interval blocks (src/main.rs:1335-1340)
    hj_set_progress_callback(ctx, [](void*, uint64_t done, uint64_t total) {
      std::fprintf(stderr, "\r%3.3f%% %llu/%llu", 100.0 * (double)done / (double)(total ? total : 1),      // the title's format
                   (unsigned long long)done, (unsigned long long)total);
      if (done >= total) std::fprintf(stderr, "\n");
    }, nullptr, opt.present_interval);
    {   // resource creation belongs to Renderer::new (src/main.rs:1167-1314), not to the timed render
      const uint64_t per_pass = (uint64_t)((opt.width + HJ_BLOCK_SIZE - 1) / HJ_BLOCK_SIZE) * ((opt.height + HJ_BLOCK_SIZE - 1) / HJ_BLOCK_SIZE);
      check(ctx, hj_reserve(ctx, (size_t)(per_pass * opt.sample_count), &ro), "reserve");
    }
    std::printf("Starting to render...\n");                            // src/main.rs:1488
    const auto t0 = std::chrono::steady_clock::now();
    check(ctx, hj_render_frame(ctx, opt.sample_count, opt.seed, 0, opt.sample_count, 0, 1, &ro, &st), "render");
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const uint64_t ray_count = (uint64_t)opt.width * opt.height * opt.sample_count;   // u32 upstream (overflows), u64 here
    std::printf("Integrated %llu rays in %.6fs (%.1f rays/s)\n", (unsigned long long)ray_count, secs, (double)ray_count / secs);

    std::vector<float> rgb((size_t)opt.width * opt.height * 3);
    check(ctx, hj_framebuffer_resolve(ctx, rgb.data()), "read-back");  // Renderer::save_image, src/main.rs:1493
    const std::string ext = opt.output_image.size() > 4 ? opt.output_image.substr(opt.output_image.size() - 4) : "";
    if (ext == ".pfm") hijiki::write_pfm(opt.output_image, opt.width, opt.height, rgb.data());
    else if (ext == ".png") hijiki::write_png(opt.output_image, opt.width, opt.height, rgb.data());   // the preview image
    else hijiki::write_exr(opt.output_image, opt.width, opt.height, rgb.data());
    hj_context_destroy(ctx);
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "hijiki-hip: %s\n", e.what());
    return 1;
  }
}
