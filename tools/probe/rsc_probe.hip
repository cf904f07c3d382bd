// Probe: chain of 4x4x1 MFMAs followed by the row / half swap reduce-scatter, against a host computation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// (the two-result builtins __builtin_amdgcn_permlane16_swap / permlane32_swap of this toolchain return the first result twice)
__device__ __forceinline__ float rsc(const f32x4& v) {
  float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
  asm volatile("s_nop 7\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
  float x = a0 + a1, y = a2 + a3;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(x), "+v"(y));
  return x + y;
}
// x[4 nodes][C], w[C][16 cols]; lane l: lr = l % 16, lq = l / 16; step s covers channels 4s + lq
__global__ void k(const float* x, const float* w, int C, float* out) {
  const int l = threadIdx.x, lr = l & 15, lq = l >> 4;
  f32x4 r[4];
  for (int t = 0; t < 4; ++t) r[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < C / 4; ++s)
    for (int t = 0; t < 4; ++t)
      r[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[(lr & 3) * C + 4 * s + lq] * (t + 1), w[(4 * s + lq) * 16 + lr], r[t], 0, 0, 0);
  __shared__ float y[4][4][16];
  for (int t = 0; t < 4; ++t) y[t][lq][lr] = rsc(r[t]);
  __syncthreads();
  for (int i = l; i < 256; i += 64) out[i] = (&y[0][0][0])[i];
}
int main() {
  const int C = 48;
  float hx[4 * C], hw[C * 16], ho[1024];
  for (int i = 0; i < 4 * C; ++i) hx[i] = (rand() % 17 - 8) / 8.f;
  for (int i = 0; i < C * 16; ++i) hw[i] = (rand() % 13 - 6) / 4.f;
  float *x, *w, *o;
  hipMalloc(&x, sizeof hx); hipMalloc(&w, sizeof hw); hipMalloc(&o, sizeof ho);
  hipMemcpy(x, hx, sizeof hx, hipMemcpyHostToDevice); hipMemcpy(w, hw, sizeof hw, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, w, C, o);
  hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 4; ++t) for (int n = 0; n < 4; ++n) for (int c = 0; c < 16; ++c) {
    double e = 0; for (int u = 0; u < C; ++u) e += (double)hx[n * C + u] * (t + 1) * hw[u * 16 + c];
    if (fabs(ho[(t * 4 + n) * 16 + c] - e) > 1e-3) { if (bad < 6) printf("t %d node %d col %d: %g expected %g\n", t, n, c, ho[(t * 4 + n) * 16 + c], e); ++bad; }
  }
  printf("reduce-scatter mismatches: %d\n", bad);
  return 0;
}
