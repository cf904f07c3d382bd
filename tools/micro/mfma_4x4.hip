// Micro-benchmark (gfx950): issue cost of v_mfma_f32_4x4x1_16b_f32 against v_mfma_f32_16x16x4_f32 (cycles per instruction
// and SIMD, 1 and 2 waves per SIMD, 8 independent accumulators / a dependent chain).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, bool SMALL>
__global__ __launch_bounds__(512) void k(float* out, int iters, long long* clk) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
  const long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (SMALL) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
  }
  const long long c1 = clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;
}
template <int NACC, bool SMALL>
void run(const char* name, int waves, int iters, float* out, long long* clk) {
  hipLaunchKernelGGL((k<NACC, SMALL>), dim3(256), dim3(64 * waves), 0, 0, out, iters, clk);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
  printf("%-28s waves/SIMD %d: %.2f cycles per instruction and SIMD (per wave: %.2f)\n", name, waves / 4,
         (double)h / ((double)NACC * iters * (waves / 4.0)), (double)h / ((double)NACC * iters));
}
int main() {
  float* out; long long* clk;
  hipMalloc(&out, sizeof(float) * 512 * 256); hipMalloc(&clk, 16);
  const int iters = 20000;
  for (int waves : {4, 8}) {
    run<8, false>("16x16x4 8 accumulators", waves, iters, out, clk);
    run<8, true>("4x4x1 8 accumulators", waves, iters, out, clk);
    run<2, true>("4x4x1 2 accumulators", waves, iters, out, clk);
    run<1, true>("4x4x1 dependent chain", waves, iters, out, clk);
    run<1, false>("16x16x4 dependent chain", waves, iters, out, clk);
  }
  return 0;
}
