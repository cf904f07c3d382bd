// Micro-benchmark (gfx950) for the next step of k_conv_fused: the same chunk-step work per workgroup issued by 8 waves (two
// virtual nodes each, as today) or by 16 waves (one virtual node each; wave pairs split the contraction of a k-row).
// Per chunk step and workgroup: 8 k-rows x NCT contraction MFMAs (v_mfma_f32_16x16x4_f32) whose results go to LDS, 16 virtual
// nodes x NED edge MFMAs whose B operands come from LDS, one 16-B global request per wave and MFMA group, one barrier.
// One workgroup per CU (LDS reservation).  Reports steps per second, MFMAs per SIMD cycle and the shader clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WAVES, int NCT, int NED>
__global__ __launch_bounds__(64 * WAVES) void k(const float* __restrict__ g, float* out, int steps, long long* clk) {
  extern __shared__ float lds[];                       // [2][16 nodes][8 rows][64] chunk buffers (+ padding to hold the CU)
  constexpr int VPW = 16 / WAVES;                      // virtual nodes per wave (2 or 1)
  constexpr int CPW = (NCT * 8 + WAVES - 1) / WAVES;   // contraction MFMAs per wave and step
  constexpr int EPW = NED * VPW;                       // edge MFMAs per wave and step
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x4 acc[EPW > 32 ? 32 : EPW];
  constexpr int NACC = EPW > 32 ? 32 : EPW;
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = 1e-3f * lane, b = 1.f + 1e-3f * wave;
  float xa[8];
  for (int i = 0; i < 8; ++i) xa[i] = a * (float)(i + 1) + g[i];   // opaque to the compiler (g is zero-filled at run time)
  const float* gp = g + ((size_t)blockIdx.x * 64 * WAVES + tid) * 4;
  float4 pre = *reinterpret_cast<const float4*>(gp);
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int s = 0; s < steps; ++s) {
    float* yw = lds + (s & 1) * 8192 + wave * (8192 / WAVES);
    const float* yr = lds + ((s & 1) ^ 1) * 8192;
    const float4 nxt = *reinterpret_cast<const float4*>(gp + (size_t)((s + 1) & 63) * 64 * WAVES * 4 * 256);   // next step's request
    f32x4 r[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
    for (int i = 0; i < CPW; ++i) r[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i & 7] + pre.x, b + pre.y, r[i & 3], 0, 0, 0);   // (distinct operands: no CSE)
#pragma unroll
    for (int i = 0; i < 4; ++i) yw[(i * 64 + lane) & (8192 / WAVES - 1)] = r[i][0] + r[i][1] + r[i][2] + r[i][3];
#pragma unroll
    for (int i = 0; i < EPW; ++i) {
      const float q = yr[((i * 16 + lane) * 5 + wave * 64) & 8191];
      acc[i % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(pre.z + a, q, acc[i % NACC], 0, 0, 0);
    }
    pre = nxt;
    __syncthreads();
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float sum = 0.f;
  for (int i = 0; i < NACC; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 64 * WAVES + tid] = sum;
  if (blockIdx.x == 0 && tid == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int WAVES, int NCT, int NED>
void run(const char* name, const float* g, float* out, long long* clk, int steps) {
  const size_t smem = 100 * 1024;                      // one workgroup per CU
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<WAVES, NCT, NED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<WAVES, NCT, NED>), dim3(256), dim3(64 * WAVES), smem, 0, g, out, steps, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double mfma_per_simd = (double)steps * (((NCT * 8 + WAVES - 1) / WAVES) * WAVES + NED * 16) / 4.0;
    const double clock_mhz = (double)h[0] / ((double)h[1] / 100.0);
    if (rep == 1)
      printf("%-34s %2d waves/WG: %7.3f ms, %6.1f ns per step, %5.1f TFLOP/s executed, shader clock %4.0f MHz, %5.2f cycles per MFMA and SIMD\n", name,
             WAVES, ms, ms * 1e6 / steps, 2048.0 * mfma_per_simd * 4 * 256 / (ms * 1e-3) * 1e-12, clock_mhz, ms * 1e-3 * clock_mhz * 1e6 / mfma_per_simd);
  }
}

int main() {
  float *g, *out; long long* clk;
  hipMalloc(&g, (size_t)256 * 1024 * 4 * 4 * 64 + (1 << 20)); hipMemset(g, 0, (size_t)256 * 1024 * 4 * 4 * 64 + (1 << 20));
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, 16);
  const int steps = 4000;
  // classic granule (12,3,3,3): 21 contraction MFMAs per k-row, 16 edge MFMAs per virtual node (2 row tiles x 2 k-halves x 4 blocks)
  run<8, 21, 16>("classic granule 21 + 16/vnode", g, out, clk, steps);
  run<16, 21, 16>("classic granule 21 + 16/vnode", g, out, clk, steps);
  // packed granule: 30 contraction, 20 edge per virtual node
  run<8, 30, 20>("packed granule 30 + 20/vnode", g, out, clk, steps);
  run<16, 30, 20>("packed granule 30 + 20/vnode", g, out, clk, steps);
  return 0;
}
