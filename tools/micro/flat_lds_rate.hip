// Micro-benchmark: throughput of the three ways a lane can fetch a 32-byte BVH node whose copy may be in LDS.
//   mode 0  global loads only (table in L2)
//   mode 1  ds_read_b128 from the LDS copy (all lanes "hot")
//   mode 2  FLAT loads of a per-lane selected address: LDS copy for idx < hot, global otherwise  (what the walk does)
//   mode 3  branchy: ds_read for hot lanes, global_load for the others (two instruction sequences per step)
// Each lane chases idx = node[idx].w; `hot_pct` of the chain's targets fall in the LDS-resident range.
//   hipcc --offload-arch=gfx950 -O3 -w tools/micro/flat_lds_rate.hip -o /tmp/flat_lds_rate && /tmp/flat_lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>

constexpr uint32_t kHot = 256;
typedef float f4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) f4v* LdsF4;

template <int MODE>
__global__ __launch_bounds__(256) void chase(const float4* __restrict__ tab, uint32_t n, int iters, uint32_t* out) {
  __shared__ float4 s0[kHot], s1[kHot];
  for (uint32_t i = threadIdx.x; i < kHot; i += 256) { s0[i] = tab[2 * i]; s1[i] = tab[2 * i + 1]; }
  __syncthreads();
  uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n;
  float acc = 0.f;
  for (int i = 0; i < iters; i++) {
    float4 a, b;
    if (MODE == 0) { a = tab[2 * idx]; b = tab[2 * idx + 1]; }
    else if (MODE == 1) { const uint32_t j = idx & (kHot - 1); a = s0[j]; b = s1[j]; }
    else if (MODE == 2) {
      const float4* pa = idx < kHot ? &s0[idx] : &tab[2 * idx];
      const float4* pb = idx < kHot ? &s1[idx] : &tab[2 * idx + 1];
      a = *pa; b = *pb;
    } else {
      if (idx < kHot) {
        const f4v va = ((LdsF4)s0)[idx], vb = ((LdsF4)s1)[idx];
        a = make_float4(va.x, va.y, va.z, va.w); b = make_float4(vb.x, vb.y, vb.z, vb.w);
      } else { a = tab[2 * idx]; b = tab[2 * idx + 1]; }
    }
    acc += a.x + b.x;
    idx = __float_as_uint(a.w);
  }
  if (acc == 12345.f) out[0] = idx;
}

int main() {
  const uint32_t n = 10240;
  float4* d; uint32_t* out;
  hipMalloc(&d, 2 * n * sizeof(float4)); hipMalloc(&out, 16);
  hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
  const int iters = 2000, grid = 2048;
  for (int hot_pct : {0, 50, 84, 100}) {
    std::vector<float4> h(2 * n);
    std::mt19937 rng(1);
    for (uint32_t i = 0; i < n; i++) {
      const bool hot = (int)(rng() % 100) < hot_pct;
      const uint32_t nxt = hot ? rng() % kHot : kHot + rng() % (n - kHot);
      h[2 * i] = make_float4(1.f, 2.f, 3.f, __builtin_bit_cast(float, nxt));
      h[2 * i + 1] = make_float4(4.f, 5.f, 6.f, 0.f);
    }
    hipMemcpy(d, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice);
    for (int mode = 0; mode < 4; mode++) {
      if (mode == 1 && hot_pct != 100) continue;
      float best = 1e9f;
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(ea);
        switch (mode) {
          case 0: hipLaunchKernelGGL(chase<0>, dim3(grid), dim3(256), 0, 0, d, n, iters, out); break;
          case 1: hipLaunchKernelGGL(chase<1>, dim3(grid), dim3(256), 0, 0, d, n, iters, out); break;
          case 2: hipLaunchKernelGGL(chase<2>, dim3(grid), dim3(256), 0, 0, d, n, iters, out); break;
          default: hipLaunchKernelGGL(chase<3>, dim3(grid), dim3(256), 0, 0, d, n, iters, out); break;
        }
        hipEventRecord(eb); hipEventSynchronize(eb);
        float ms; hipEventElapsedTime(&ms, ea, eb); best = ms < best ? ms : best;
      }
      const double steps = (double)grid * 256 * iters;
      const char* names[4] = {"global only", "ds_read only", "flat select", "branchy ds/global"};
      printf("hot %3d%%  %-18s %7.3f ms  %7.1f G node fetches/s  (%.2f per cycle per CU)\n", hot_pct, names[mode], best,
             steps / best / 1e6, steps / (best * 1e-3) / 256 / 2.4e9);
    }
  }
  return 0;
}
