// Micro-benchmark: what does a divergent 16-byte-per-lane gather cost on one CU as a function of ACTIVE LANES?
// Every active lane chases indices through a 320 KB table (L2-resident, like the BVH nodes): idx = table[idx].w.
// If time per wave-instruction is flat in the lane count, the vector-memory pipeline charges per instruction;
// if it scales with lanes, it charges per lane (quad).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gather_rate.hip -o gpurun_out/gather_rate && gpurun_out/gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>

template <int LOADS>
__global__ __launch_bounds__(256) void chase(const float4* __restrict__ tab, uint32_t n, int iters, int lanes, uint32_t* out) {
  const uint32_t lane = threadIdx.x & 63u;
  if ((int)lane >= lanes) return;
  uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n;
  float acc = 0.f;
  for (int i = 0; i < iters; i++) {
    const float4 a = tab[2 * idx];
    if (LOADS == 2) { const float4 b = tab[2 * idx + 1]; acc += b.x; }
    acc += a.x;
    idx = __float_as_uint(a.w);
  }
  if (acc == 12345.f) out[0] = idx;
  if (idx == 0xFFFFFFFFu) out[1] = 1;
}

int main() {
  const uint32_t n = 10240;   // 10 k nodes x 32 B = 320 KB
  std::vector<float4> h(2 * n);
  std::mt19937 rng(1);
  for (uint32_t i = 0; i < n; i++) {
    uint32_t nxt = rng() % n;
    h[2 * i] = make_float4(1.f, 2.f, 3.f, __builtin_bit_cast(float, nxt));
    h[2 * i + 1] = make_float4(4.f, 5.f, 6.f, 0.f);
  }
  float4* d; uint32_t* out;
  hipMalloc(&d, h.size() * sizeof(float4)); hipMalloc(&out, 16);
  hipMemcpy(d, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 2000, grid = 2048 * 2;
  for (int loads = 1; loads <= 2; loads++)
    for (int lanes : {64, 48, 32, 16, 8}) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a);
        if (loads == 1) hipLaunchKernelGGL(chase<1>, dim3(grid), dim3(256), 0, 0, d, n, iters, lanes, out);
        else hipLaunchKernelGGL(chase<2>, dim3(grid), dim3(256), 0, 0, d, n, iters, lanes, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
      }
      const double wave_instr = (double)grid * 4 * iters * loads;
      printf("loads/iter %d lanes %2d: %7.3f ms  %6.2f G wave-instr/s  %7.2f G lane-loads/s  (%.1f cycles per wave-instr per CU at 2.4 GHz)\n",
             loads, lanes, best, wave_instr / best / 1e6, wave_instr * lanes / best / 1e6, best * 1e-3 * 2.4e9 * 256 / wave_instr);
    }
  return 0;
}
