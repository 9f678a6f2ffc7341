// Calibration of rocprofv3's FETCH_SIZE for THIS kernel's access widths (MI355X_MICROARCH.md, section HBM: "other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern"): every lane gathers random
// records of `rec` bytes (16, 32 or 48 = one, two or three 16-byte loads: path record / BVH node / triangle) from a
// 4 GiB table (far beyond L2 and the 256 MiB Infinity Cache), each record once.  The program prints the bytes it asked
// for; tools/fetch_calib.sh divides the counter by it.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/fetch_calib.hip -o gpurun_out/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

template <int F4>
__global__ __launch_bounds__(256) void gather(const float4* __restrict__ tab, uint64_t nrec, uint32_t per_lane, float* out) {
  const uint64_t gid = (uint64_t)blockIdx.x * 256u + threadIdx.x, total = (uint64_t)gridDim.x * 256u;
  float acc = 0.f;
  for (uint32_t k = 0; k < per_lane; k++) {
    // a permutation of the record indices (odd multiplier modulo a power of two): every record is read once
    const uint64_t r = ((gid + (uint64_t)k * total) * 0x9E3779B97F4A7C15ull) & (nrec - 1);
    const float4* p = tab + r * F4;
#pragma unroll
    for (int j = 0; j < F4; j++) acc += p[j].x;
  }
  if (acc == 12345.f) out[0] = acc;
}

int main(int argc, char** argv) {
  const int f4 = argc > 1 ? atoi(argv[1]) : 2;                 // float4 per record: 1, 2 or 3
  const uint64_t bytes = 4ull << 30;
  uint64_t nrec = 1;
  while (nrec * 2 * f4 * 16 <= bytes) nrec *= 2;               // power of two
  float4* d; float* out;
  if (hipMalloc(&d, nrec * f4 * 16) != hipSuccess || hipMalloc(&out, 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(d, 0, nrec * f4 * 16);
  const uint32_t grid = 8192, per_lane = (uint32_t)(nrec / 4 / ((uint64_t)grid * 256));   // a quarter of the records
  if (f4 == 1) hipLaunchKernelGGL(gather<1>, dim3(grid), dim3(256), 0, 0, d, nrec, per_lane, out);
  else if (f4 == 2) hipLaunchKernelGGL(gather<2>, dim3(grid), dim3(256), 0, 0, d, nrec, per_lane, out);
  else hipLaunchKernelGGL(gather<3>, dim3(grid), dim3(256), 0, 0, d, nrec, per_lane, out);
  hipDeviceSynchronize();
  printf("record_bytes %d requested_bytes %llu records %llu\n", f4 * 16, (unsigned long long)grid * 256 * per_lane * f4 * 16,
         (unsigned long long)grid * 256 * per_lane);
  return 0;
}
