// Micro-benchmark (gfx950) behind the split-bf16 edge product of k_conv_fused (round 4): one chunk step of a classic granule --
// phase A = 21 contraction MFMAs (v_mfma_f32_16x16x4_f32), phase B = the edge product of the previous chunk, one barrier -- with
// the operand preparation of the bf16 form placed in different ways.  One workgroup of 8 waves per CU; x / weight / hidden-row
// fragments live in registers (no global traffic: this isolates issue slots, LDS and the matrix pipes).
//   mode 0  f32: 32 v_mfma_f32_16x16x4_f32 in phase B, results stored as they are
//   mode 1  bf16, first form: 16 v_mfma_f32_16x16x32_bf16; B tuples = 4 v_perm per (virtual node, column block) in phase B,
//           A tuples by register copies, results split (compiler's choice of packed f32 subtract) + stored in phase B
//   mode 2  mode 1 without the result split (timing only)          mode 3  mode 1 without the B perms (timing only)
//   mode 4  B tuples straight from LDS (ds_read2_b32 with equal offsets = the duplicated word), A tuples = perms in phase A,
//           results split with scalar subtracts + stored in phase B
//   mode 5  mode 4 with split + store of chunk t deferred into phase A of step t + 1 (edge product one step later)
//   mode 6  mode 5 without any LDS traffic / split (MFMAs and barrier only)
//   mode 7  mode 1 (scalar split) with waves 4..7 running [edge product, contraction]: the two waves of a SIMD are in complementary phases;
//           their results leave during the contraction, chain by chain (chains reordered so that they complete one after the other)
//   mode 8  mode 7 for ALL waves (order only, no complement)          mode 9  mode 1 with scalar subtracts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
__device__ __forceinline__ unsigned pk(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2)); }
__device__ __forceinline__ float fsub(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <bool SCALAR>
__device__ __forceinline__ void split2(float v1, float v2, float& w1, float& w2) {
  const unsigned hp = pk(v1, v2), h1 = hp << 16, h2 = hp & 0xffff0000u;
  const float l1 = SCALAR ? fsub(v1, __uint_as_float(h1)) : v1 - __uint_as_float(h1);
  const float l2 = SCALAR ? fsub(v2, __uint_as_float(h2)) : v2 - __uint_as_float(h2);
  const unsigned lp = pk(l1, l2);
  w1 = __uint_as_float((lp & 0xffffu) | h1);
  w2 = __uint_as_float(__builtin_amdgcn_perm(hp, lp, 0x07060302u));
}
__device__ __forceinline__ f32x4 mfma_bf(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// the word at lds[addr + OFF] twice, in two consecutive registers (offsets in dwords)
template <int OFF>
__device__ __forceinline__ u32x2 lds_dup(const float* p) {
  u32x2 r;
  const unsigned a = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)p;   // LDS byte address
  asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%2" : "=v"(r) : "v"(a), "n"(OFF));
  return r;
}
// issue order of the 21 contraction MFMAs: the 12-step chain alternating with the three 3-step chains
constexpr int slot_of(int i) { int c = 0; for (int j = 0; j < 12; ++j) { if (c == i) return 0; ++c; if (j < 9) { if (c == i) return 1 + j % 3; ++c; } } return 0; }

template <int MODE>
__global__ __launch_bounds__(512) void k(const float* __restrict__ g, float* out, int steps, long long* clk) {
  extern __shared__ float lds[];
  constexpr int NCB = 4, NCT = 21;
  constexpr int YROW = 16 * NCB + 8, YVN = 8 * YROW + 4, YB = 16 * YVN;
  constexpr bool SWAP = MODE == 7 || MODE == 8;
  constexpr bool BF = MODE != 0, PERM_B = (MODE >= 1 && MODE <= 2) || SWAP || MODE == 9, DUP_B = MODE >= 4 && MODE <= 6, A_EARLY = MODE >= 4 && MODE <= 6;
  constexpr bool SPLIT = MODE == 1 || MODE == 3 || MODE == 4 || MODE == 5 || SWAP || MODE == 9, SCALAR = MODE >= 4, DEFER = MODE == 5 || MODE == 6, NOLDS = MODE == 6;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  f32x4 acc[2][2][NCB];
#pragma unroll
  for (int i = 0; i < 4 * NCB; ++i) acc[i / (2 * NCB)][(i / NCB) & 1][i % NCB] = f32x4{0.f, 0.f, 0.f, 0.f};
  float xa[NCT], bw[NCT];
#pragma unroll
  for (int i = 0; i < NCT; ++i) { xa[i] = 1e-3f * lane * (float)(i + 1) + g[i]; bw[i] = g[64 + i] + 1e-4f * lane; }
  float4 hN[2][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) hN[i >> 1][i & 1] = *reinterpret_cast<const float4*>(g + 256 + 4 * (lane + 64 * i));
  for (int i = tid; i < 2 * YB; i += 512) lds[i] = g[i & 1023];
  __syncthreads();
  float* const ywr = lds + (4 * lq) * YVN + wave * YROW + lr;
  const float* const yrd = lds + (2 * wave) * YVN + (2 * lq) * YROW + lr;
  u32x4 at[2][2];   // A tuples of the current chunk
#pragma unroll
  for (int i = 0; i < 4; ++i) at[i >> 1][i & 1] = u32x4{0, 0, 0, 0};
  f32x4 r[NCB], rp[NCB];   // results of this step's contraction / of the previous one (DEFER)
#pragma unroll
  for (int c = 0; c < NCB; ++c) r[c] = rp[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long long c0 = clock64(), w0 = wall_clock64();
  auto step = [&](auto odd) __attribute__((always_inline)) {
    constexpr int eb = decltype(odd)::value, cb = eb ^ 1;
    constexpr int rb = DEFER ? cb : eb;   // buffer the edge product reads (DEFER: the chunk stored one step earlier)
    auto store_piece = [&](const f32x4 (&rr_)[NCB], int buf, int piece) __attribute__((always_inline)) {
      const int rr = piece & 3, h = piece >> 2;
      float v0 = rr_[2 * h][rr], v1 = rr_[2 * h + 1][rr];
      if constexpr (SPLIT) split2<SCALAR>(v0, v1, v0, v1);
      if constexpr (NOLDS) { acc[0][0][0][0] += v0 + v1; }
      else { float* yw = ywr + buf * YB; yw[rr * YVN + 16 * (2 * h)] = v0; yw[rr * YVN + 16 * (2 * h + 1)] = v1; }
    };
    if constexpr (eb == 0) {   // every second step new hidden-row words arrive (opaque: nothing derived from them is loop-invariant)
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(hN[i >> 1][i & 1].x), "+v"(hN[i >> 1][i & 1].y), "+v"(hN[i >> 1][i & 1].z), "+v"(hN[i >> 1][i & 1].w));
    }
#pragma unroll
    for (int i = 0; i < NCT; ++i) asm volatile("" : "+v"(bw[i]));   // new weight fragments every step
    bool swapped = false;
    if constexpr (SWAP) swapped = MODE == 8 || wave >= 4;
    if (swapped) {
      if constexpr (SWAP) {
        // ---- edge product first
        float q[2][2 * NCB];
        auto readq = [&](int vi) __attribute__((always_inline)) {
          const float* yb = yrd + eb * YB + vi * YVN;
#pragma unroll
          for (int c = 0; c < NCB; ++c) { q[vi][c] = yb[16 * c]; q[vi][NCB + c] = yb[YROW + 16 * c]; }
        };
        readq(0);
        sfor<0, 4 * NCB>([&](auto mc) __attribute__((always_inline)) {
          constexpr int m = decltype(mc)::value;
          constexpr int vi = m / (2 * NCB), t8 = m % (2 * NCB), c = t8 >> 1, rt = t8 & 1;
          const float4 h = hN[vi][rt];
          const unsigned w0_ = __float_as_uint(eb ? h.z : h.x), w1_ = __float_as_uint(eb ? h.w : h.y);
          const u32x4 A = u32x4{w0_, w0_, w1_, w1_};
          const unsigned b0 = __float_as_uint(q[vi][c]), b1 = __float_as_uint(q[vi][NCB + c]);
          const u32x4 B = u32x4{__builtin_amdgcn_perm(b0, b0, 0x03020302u), __builtin_amdgcn_perm(b0, b0, 0x01000100u),
                                __builtin_amdgcn_perm(b1, b1, 0x03020302u), __builtin_amdgcn_perm(b1, b1, 0x01000100u)};
          acc[vi][rt][c] = mfma_bf(A, B, acc[vi][rt][c]);
          if (m == 0) readq(1);
          FENCE();
        });
        // ---- contraction: chains 1, 2 first (alternating), then 3 with the long chain, then the rest of the long chain
        constexpr int ORD[21] = {1, 2, 1, 2, 1, 2, 3, 0, 3, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < NCB; ++c) r[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 r0b = f32x4{0.f, 0.f, 0.f, 0.f};
        float* yw = ywr + cb * YB;
        sfor<0, NCT>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = decltype(ic)::value;
          constexpr int sl = ORD[i];
          if constexpr (sl == 0 && (i & 1)) r0b = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i], bw[i], r0b, 0, 0, 0);
          else r[sl] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i], bw[i], r[sl], 0, 0, 0);
          if constexpr (i >= 8 && i < 12) {   // chains 1, 2 complete at position 5: pair (1, 2), node quarter i - 8
            constexpr int rr = i - 8;
            float v0 = r[1][rr], v1 = r[2][rr];
            split2<true>(v0, v1, v0, v1);
            yw[rr * YVN + 16] = v0; yw[rr * YVN + 32] = v1;
          }
          if constexpr (i == 14 || i == 16) {   // chain 3 complete at position 10: two node quarters per split
            constexpr int rr = i - 14;
            float v0 = r[3][rr], v1 = r[3][rr + 1];
            split2<true>(v0, v1, v0, v1);
            yw[rr * YVN + 48] = v0; yw[(rr + 1) * YVN + 48] = v1;
          }
          FENCE();
        });
        r[0] += r0b;
        {
          float v0 = r[0][0], v1 = r[0][1], v2 = r[0][2], v3 = r[0][3];
          split2<true>(v0, v1, v0, v1); split2<true>(v2, v3, v2, v3);
          yw[0] = v0; yw[YVN] = v1; yw[2 * YVN] = v2; yw[3 * YVN] = v3;
        }
      }
    } else {
    // ---- phase A: contraction of chunk s + 1
#pragma unroll
    for (int c = 0; c < NCB; ++c) { if constexpr (DEFER) rp[c] = r[c]; r[c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    sfor<0, NCT>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      constexpr int sl = slot_of(i);
      r[sl] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i], bw[i], r[sl], 0, 0, 0);
      if constexpr (A_EARLY && i < 4) {   // A tuples of this step's edge product: perms of the hidden-row words
        const float4 h = hN[i >> 1][i & 1];
        const unsigned w0_ = __float_as_uint(eb ? h.z : h.x), w1_ = __float_as_uint(eb ? h.w : h.y);
        at[i >> 1][i & 1] = u32x4{__builtin_amdgcn_perm(w0_, w0_, 0x03020302u), __builtin_amdgcn_perm(w0_, w0_, 0x01000100u),
                                  __builtin_amdgcn_perm(w1_, w1_, 0x03020302u), __builtin_amdgcn_perm(w1_, w1_, 0x01000100u)};
      }
      if constexpr (DEFER && i >= 4 && i < 12) store_piece(rp, eb, i - 4);   // previous contraction's results -> the buffer the NEXT step multiplies
      FENCE();
    });
    // ---- phase B: edge product
    if constexpr (!BF) {
      float q[2][NCB];
      auto readq = [&](int par, int grp) __attribute__((always_inline)) {
        const float* yb = yrd + eb * YB + (grp >> 1) * YVN + (grp & 1) * YROW;
#pragma unroll
        for (int c = 0; c < NCB; ++c) q[par][c] = yb[16 * c];
      };
      readq(0, 0);
      sfor<0, 8 * NCB>([&](auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value;
        constexpr int grp = m / (2 * NCB), t8 = m % (2 * NCB), vi = grp >> 1, sub = grp & 1, rt = t8 / NCB, c = t8 % NCB;
        const float av = eb ? (sub == 0 ? hN[vi][rt].z : hN[vi][rt].w) : (sub == 0 ? hN[vi][rt].x : hN[vi][rt].y);
        acc[vi][rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, q[grp & 1][c], acc[vi][rt][c], 0, 0, 0);
        if (m >= 2 && m < 10) store_piece(r, cb, m - 2);
        if (t8 == 1 && grp < 3) readq((grp + 1) & 1, grp + 1);
        FENCE();
      });
    } else {
      float q[2][2 * NCB];
      auto readq = [&](int vi) __attribute__((always_inline)) {
        const float* yb = yrd + rb * YB + vi * YVN;
#pragma unroll
        for (int c = 0; c < NCB; ++c) { q[vi][c] = yb[16 * c]; q[vi][NCB + c] = yb[YROW + 16 * c]; }
      };
      if constexpr (!DUP_B && !NOLDS) readq(0);
      sfor<0, 4 * NCB>([&](auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value;
        constexpr int vi = m / (2 * NCB), t8 = m % (2 * NCB), c = t8 >> 1, rt = t8 & 1;
        u32x4 A, B;
        if constexpr (A_EARLY) A = at[vi][rt];
        else {
          const float4 h = hN[vi][rt];
          const unsigned w0_ = __float_as_uint(eb ? h.z : h.x), w1_ = __float_as_uint(eb ? h.w : h.y);
          A = u32x4{w0_, w0_, w1_, w1_};
        }
        if constexpr (NOLDS) {
          B = u32x4{__float_as_uint(hN[0][0].x) + c, __float_as_uint(hN[0][0].y), __float_as_uint(hN[0][0].z), __float_as_uint(hN[0][0].w) + vi};
        } else if constexpr (DUP_B) {
          const float* yb = yrd + rb * YB + vi * YVN;
          const u32x2 b0 = lds_dup<16 * c>(yb), b1 = lds_dup<YROW + 16 * c>(yb);
          B = u32x4{b0[0], b0[1], b1[0], b1[1]};
        } else if constexpr (PERM_B) {
          const unsigned b0 = __float_as_uint(q[vi][c]), b1 = __float_as_uint(q[vi][NCB + c]);
          B = u32x4{__builtin_amdgcn_perm(b0, b0, 0x03020302u), __builtin_amdgcn_perm(b0, b0, 0x01000100u),
                    __builtin_amdgcn_perm(b1, b1, 0x03020302u), __builtin_amdgcn_perm(b1, b1, 0x01000100u)};
        } else {
          const unsigned b0 = __float_as_uint(q[vi][c]), b1 = __float_as_uint(q[vi][NCB + c]);
          B = u32x4{b0, b0, b1, b1};
        }
        acc[vi][rt][c] = mfma_bf(A, B, acc[vi][rt][c]);
        if constexpr (!DEFER) { if (m >= 1 && m < 9) store_piece(r, cb, m - 1); }
        if constexpr (!DUP_B && !NOLDS) { if (m == 0) readq(1); }
        FENCE();
      });
    }
    }
    __syncthreads();
  };
  for (int s = 0; s < steps; s += 2) { step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); }
  const long long c1 = clock64(), w1 = wall_clock64();
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 4 * NCB; ++i) { const f32x4 v = acc[i / (2 * NCB)][(i / NCB) & 1][i % NCB]; sum += v[0] + v[1] + v[2] + v[3]; }
#pragma unroll
  for (int c = 0; c < NCB; ++c) sum += r[c][0] + rp[c][1];
  out[blockIdx.x * 512 + tid] = sum;
  if (blockIdx.x == 0 && tid == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int MODE>
void run(const char* name, const float* g, float* out, long long* clk, int steps) {
  const size_t smem = 128 * 1024;                      // one workgroup per CU
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), smem, 0, g, out, steps, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double clock_mhz = (double)h[0] / ((double)h[1] / 100.0);
    const double cyc = ms * 1e-3 * clock_mhz * 1e6 / steps;   // cycles per step
    const double pipe = 2.0 * (21 * 32 + (MODE == 0 ? 32 * 32 : 16 * 16));   // matrix-pipe cycles per step and SIMD at the nominal rates
    if (rep == 1)
      printf("mode %d %-78s %7.3f ms  %6.1f ns/step  clock %4.0f MHz  %6.0f cycles/step (matrix pipe alone: %4.0f)\n", MODE, name, ms,
             ms * 1e6 / steps, clock_mhz, cyc, pipe);
  }
}

int main() {
  float *g, *out; long long* clk;
  const size_t gbytes = 1 << 22;
  hipMalloc(&g, gbytes); hipMemset(g, 0, gbytes);
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 16);
  const int steps = 4000;
  run<0>("f32 edge product (32 x 16x16x4)", g, out, clk, steps);
  run<1>("bf16: B perms + A copies + split (packed subtract) + stores, all in phase B", g, out, clk, steps);
  run<2>("  same without the split", g, out, clk, steps);
  run<3>("  same without the B perms", g, out, clk, steps);
  run<4>("bf16: B = duplicated LDS reads, A perms in phase A, scalar split + stores in phase B", g, out, clk, steps);
  run<5>("bf16: same, split + stores deferred into the next phase A", g, out, clk, steps);
  run<6>("bf16: MFMAs + barrier only", g, out, clk, steps);
  run<9>("bf16: mode 1 with scalar subtracts in the split", g, out, clk, steps);
  run<7>("bf16: mode 1 (scalar split), waves 4-7 in the complementary phase", g, out, clk, steps);
  run<8>("bf16: all waves [edge product, contraction + stores]", g, out, clk, steps);
  return 0;
}
