// Micro-benchmark (gfx950) behind round 4's restructuring of k_conv_fused: the chunk step of one tile with the kernel's real LDS
// traffic (every 16x16 contraction result leaves as four ds_write_b32 into the transposed chunk buffer, the edge product reads one
// B fragment per (virtual node, k half, column block)), its weight / hidden-row requests and one barrier per step -- and the
// variants that were candidates for the next structure:
//   bit 0  no barrier (timing only)                    bit 1  waves 4..7 run the step as [edge product, contraction] (the two waves
//   bit 2  s_setprio 1 for waves 4..7                          of a SIMD are then in complementary phases)
//   bit 3  no LDS traffic (registers only)             bit 4  barrier every second step only (timing only)
//   bit 5  contraction and edge MFMAs interleaved 1 : 1 instead of two phases
// One workgroup of 8 waves per CU.  Reports cycles per MFMA and SIMD (32 = the f32 matrix pipe's nominal rate).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}

template <int NCB> struct Chains;
template <> struct Chains<4> { static constexpr int len(int c) { return c == 0 ? 12 : 3; } static constexpr int NCT = 21; };
template <> struct Chains<5> { static constexpr int len(int c) { return c == 0 ? 12 : c < 3 ? 6 : 3; } static constexpr int NCT = 30; };
template <int NCB> constexpr int chain_of(int i) {   // contraction position i -> its chain (round-robin over the chains that still have steps)
  int left[NCB] = {};
  for (int q = 0; q < NCB; ++q) left[q] = Chains<NCB>::len(q);
  int n = 0;
  for (int cur = 0;; cur = (cur + 1) % NCB) {
    if (left[cur] == 0) continue;
    if (n == i) return cur;
    --left[cur]; ++n;
  }
}

template <int NCB, int MODE>
__global__ __launch_bounds__(512) void k(const float* __restrict__ g, float* out, int steps, long long* clk) {
  extern __shared__ float lds[];
  using C = Chains<NCB>;
  constexpr int NCT = C::NCT;
  constexpr int YROW = 16 * NCB + 8, YVN = 8 * YROW + 4, YB = 16 * YVN;
  constexpr int NW4 = (NCT + 3) / 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  if constexpr ((MODE & 4) != 0) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }
  f32x4 acc[2][2][NCB];
#pragma unroll
  for (int i = 0; i < 4 * NCB; ++i) acc[i / (2 * NCB)][(i / NCB) & 1][i % NCB] = f32x4{0.f, 0.f, 0.f, 0.f};
  float xa[NCT];
#pragma unroll
  for (int i = 0; i < NCT; ++i) xa[i] = 1e-3f * lane * (float)(i + 1) + g[i];   // x fragments: resident
  const float* gp = g + ((size_t)blockIdx.x * 512 + tid) * 4;
  float4 bw[NW4], bn[NW4], hC[2][2], hN[2][2];
#pragma unroll
  for (int i = 0; i < NW4; ++i) bw[i] = bn[i] = *reinterpret_cast<const float4*>(gp + (size_t)i * 2048 * 256);
#pragma unroll
  for (int i = 0; i < 4; ++i) hC[i >> 1][i & 1] = hN[i >> 1][i & 1] = *reinterpret_cast<const float4*>(gp + (size_t)(8 + i) * 2048 * 256);
  float* const ywr = lds + (4 * lq) * YVN + wave * YROW + lr;
  const float* const yrd = lds + (2 * wave) * YVN + (2 * lq) * YROW + lr;
  auto bwf = [&](int j) { const float4& v = bw[j >> 2]; return (j & 3) == 0 ? v.x : (j & 3) == 1 ? v.y : (j & 3) == 2 ? v.z : v.w; };
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int s = 0; s < steps; ++s) {
    const int eb = s & 1, cb = eb ^ 1;
    // requests of the next step: weights every step, hidden rows every second step
#pragma unroll
    for (int i = 0; i < NW4; ++i) bn[i] = *reinterpret_cast<const float4*>(gp + ((size_t)i + (size_t)((s + 1) & 7) * NW4) * 2048 * 256);
    if (s & 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) hN[i >> 1][i & 1] = *reinterpret_cast<const float4*>(gp + (size_t)(8 + i + 4 * ((s + 1) & 7)) * 2048 * 256 + (1 << 22));
    }
    f32x4 r[NCB];
    auto contract_one = [&](auto ic) __attribute__((always_inline)) {   // contraction MFMA i: chain of result c (round-robin over the live chains)
      constexpr int i = decltype(ic)::value;
      // position i -> (chain c, step j): the long chain alternates with the short ones
      constexpr int c = chain_of<NCB>(i);
      r[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i], bwf(i), r[c], 0, 0, 0);
    };
    auto store_res = [&](int c) __attribute__((always_inline)) {
      if constexpr ((MODE & 8) == 0) {
        float* yw = ywr + cb * YB + 16 * c;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) yw[rr * YVN] = r[c][rr];
      } else {
        acc[0][0][c][0] += r[c][0] + r[c][1] + r[c][2] + r[c][3];
      }
    };
    float q[NCB];
    auto readq = [&](int grp) __attribute__((always_inline)) {
      if constexpr ((MODE & 8) == 0) {
        const float* yb = yrd + eb * YB + (grp >> 1) * YVN + (grp & 1) * YROW;
#pragma unroll
        for (int c = 0; c < NCB; ++c) q[c] = yb[16 * c];
      } else {
#pragma unroll
        for (int c = 0; c < NCB; ++c) q[c] = hC[0][0].x + (float)(grp + c);
      }
    };
    auto edge_one = [&](auto mc) __attribute__((always_inline)) {
      constexpr int m = decltype(mc)::value;
      constexpr int grp = m / (2 * NCB), t8 = m % (2 * NCB), vi = grp >> 1, sub = grp & 1, rt = t8 / NCB, c = t8 % NCB;
      if constexpr (t8 == 0) readq(grp);
      const float av = eb ? (sub == 0 ? hC[vi][rt].z : hC[vi][rt].w) : (sub == 0 ? hC[vi][rt].x : hC[vi][rt].y);
      acc[vi][rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, q[c], acc[vi][rt][c], 0, 0, 0);
    };
    auto contract = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int c = 0; c < NCB; ++c) r[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      sfor<0, NCT>(contract_one);
#pragma unroll
      for (int c = 0; c < NCB; ++c) store_res(c);
    };
    auto edge = [&]() __attribute__((always_inline)) { sfor<0, 8 * NCB>(edge_one); };
    if constexpr ((MODE & 32) != 0) {
#pragma unroll
      for (int c = 0; c < NCB; ++c) r[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      sfor<0, 8 * NCB>([&](auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value;
        if constexpr (m < NCT) contract_one(mc);
        if constexpr (m == NCT) {
#pragma unroll
          for (int c = 0; c < NCB; ++c) store_res(c);
        }
        edge_one(mc);
      });
    } else if ((MODE & 2) != 0 && wave >= 4) { edge(); contract(); }
    else { contract(); edge(); }
#pragma unroll
    for (int i = 0; i < NW4; ++i) bw[i] = bn[i];
    if (s & 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) hC[i >> 1][i & 1] = hN[i >> 1][i & 1];
    }
    if constexpr ((MODE & 1) == 0) {
      if constexpr ((MODE & 16) != 0) { if (s & 1) __syncthreads(); }
      else __syncthreads();
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 4 * NCB; ++i) { const f32x4 v = acc[i / (2 * NCB)][(i / NCB) & 1][i % NCB]; sum += v[0] + v[1] + v[2] + v[3]; }
  out[blockIdx.x * 512 + tid] = sum;
  if (blockIdx.x == 0 && tid == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int NCB, int MODE>
void run(const char* name, const float* g, float* out, long long* clk, int steps) {
  const size_t smem = 128 * 1024;                      // one workgroup per CU
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<NCB, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NCB, MODE>), dim3(256), dim3(512), smem, 0, g, out, steps, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double mfma_per_simd = (double)steps * 2.0 * (Chains<NCB>::NCT + 8 * NCB);
    const double clock_mhz = (double)h[0] / ((double)h[1] / 100.0);
    if (rep == 1)
      printf("NCB %d mode %2d %-44s %7.3f ms  %6.1f ns/step  clock %4.0f MHz  %5.2f cycles per MFMA and SIMD\n", NCB, MODE, name, ms,
             ms * 1e6 / steps, clock_mhz, ms * 1e-3 * clock_mhz * 1e6 / mfma_per_simd);
  }
}

int main() {
  float *g, *out; long long* clk;
  const size_t gbytes = (size_t)64 * 2048 * 256 * 4 + (1 << 26);
  hipMalloc(&g, gbytes); hipMemset(g, 0, gbytes);
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 16);
  const int steps = 4000;
#define RUN(NCB, MODE, NAME) run<NCB, MODE>(NAME, g, out, clk, steps)
  RUN(4, 0, "baseline: phases, barrier per step");
  RUN(4, 1, "no barrier");
  RUN(4, 2, "waves 4-7 swapped phases");
  RUN(4, 6, "swapped phases + setprio 1 waves 4-7");
  RUN(4, 4, "setprio 1 waves 4-7");
  RUN(4, 8, "no LDS traffic");
  RUN(4, 9, "no LDS traffic, no barrier");
  RUN(4, 16, "barrier every 2nd step");
  RUN(4, 32, "interleaved contraction / edge");
  RUN(4, 34, "interleaved (waves 4-7 same)");
  RUN(5, 0, "baseline: phases, barrier per step");
  RUN(5, 1, "no barrier");
  RUN(5, 8, "no LDS traffic");
  RUN(5, 32, "interleaved contraction / edge");
  return 0;
}
