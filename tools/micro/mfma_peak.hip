// Calibration micro-benchmark (gfx950): sustained rate of v_mfma_f32_16x16x4_f32 with N waves per SIMD and the
// shader clock under that load.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512) void k_mfma(float* out, int iters, long long* clk) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  float* out; long long* clk;
  hipMalloc(&out, sizeof(float) * 512 * cus * 4); hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves : {4, 8}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((k_mfma<8>), dim3(cus), dim3(64 * waves), 0, 0, out, iters, clk);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
      const double flops = 2.0 * 16 * 16 * 4 * 8.0 * iters * waves * cus;
      printf("CUs %d waves/CU %d: %.3f ms  %.1f TFLOP/s  shader clock %.0f MHz (clock64 %lld / wall_clock64 %lld @100MHz) cycles/MFMA/SIMD %.2f\n",
             cus, waves, ms, flops / ms * 1e-9, (double)h[0] / ((double)h[1] / 100.0), h[0], h[1],
             (double)h[0] / (8.0 * iters * (waves / 4.0)));
    }
  }
  return 0;
}
