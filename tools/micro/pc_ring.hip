// Micro-benchmark (gfx950), round 5: does a producer / consumer split of k_conv_fused's chunk loop beat the lock-step loop?
// One workgroup per CU walks (tile, granule) pairs exactly as the kernel does: per granule NCH = 18 chunk steps of 8 k-rows, then
// a coupling epilogue.  Everything that costs issue slots or LDS / memory time in the real kernel is here with its real shape:
//   * contraction: NCT v_mfma_f32_16x16x4_f32 per k-row in the kernel's chains (12-step chain on two accumulators + 3-step
//     chains), weight fragments from a 6-MB packed-weight image through buffer-style 16-B / 12-B requests (1-KB pieces per wave,
//     L2-resident across workgroups), results leaving as transposing ds_write_b32 into the padded chunk layout [node][k][column];
//   * edge product: per virtual node 2 row tiles x 2 k-halves x NCB column blocks, B fragments = one ds_read_b32 each from the
//     chunk, A fragments = 16-B requests of a streamed hidden-row image (288 KB per tile, a new tile every granule: HBM / MALL);
//   * epilogue per (virtual node, row tile): the kernel's five dependent wave-local LDS phases (coupling rows written, read back
//     as 16-B pieces, message values staged, read back, 16-B global stores).
// Structures (MODE):
//   0 LOCK  the shipped structure: 8 waves, wave w contracts row w of chunk g+1, multiplies chunk g into its 2 virtual nodes,
//           one s_barrier per chunk, double-buffered chunk; all 8 waves run the epilogue (2 virtual nodes each).
//   1 PC    producer / consumer waves: waves 0-3 (one per SIMD) contract rows 2p, 2p+1 of every chunk into an R-deep ring,
//           waves 4-7 multiply (4 virtual nodes each) and run the epilogue while the producers run ahead into the next granule;
//           hand-off by monotonic LDS flags (data stores and flag store of one wave execute in order in the LDS), no s_barrier.
//   2 SOLO  4 waves, one per SIMD, 512 registers: each wave contracts 2 rows and multiplies 4 virtual nodes in ONE interleaved
//           instruction stream, s_barrier per chunk among 4 waves.
// Output: ns per (tile, granule), and cycles per MFMA and SIMD with / without the epilogue (32 = nominal f32 matrix-pipe rate).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o pc_ring pc_ring.hip ; run under `timeout 120` (spins are bounded).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
__device__ f32x4 raw_ld4(i32x4 rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ f32x3 raw_ld3(i32x4 rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.v3f32");
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define UNI(x) __builtin_amdgcn_readfirstlane(x)

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
__device__ __forceinline__ i32x4 mkbuf(const void* p, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(p);
  return i32x4{UNI((int)(unsigned)a), UNI((int)((a >> 32) & 0xffffu)), (int)bytes, 0x00020000};
}

constexpr int NCH = 18, NGR = 8, KSROW = 10240, HKROWS = 8 * NCH + 1;   // chunks per granule, granules per tile, floats per packed k-row
constexpr int XSTR = 162;

// granule shapes.  G = 0 classic (12,3,3,3), weights of the three short chains shared (DUP 1): 21 MFMAs, 4 column blocks, 16 stores
//                  G = 1 packed 7-slot (12 + 2 groups x 3 comps x 3 steps): 30 MFMAs, 5 column blocks, 28 stores
template <int G> struct Shape;
template <> struct Shape<0> { static constexpr int NCT = 21, NCB = 4, NSL = 4, NW3 = 1, NBK = 4; };
template <> struct Shape<1> { static constexpr int NCT = 30, NCB = 5, NSL = 7, NW3 = 2, NBK = 5; };
template <int NBK> struct Dim { static constexpr int YROW = 16 * NBK + 8, YVN = 8 * YROW + 4, YB = 16 * YVN; };

// chain of contraction position i: the long chain (slot 0) alternates with the short ones
template <int G> constexpr int slot_of(int i) {
  constexpr int NS = Shape<G>::NSL - 1;          // short slots
  int c = 0;
  for (int j = 0; j < 64; ++j) {
    if (j < 12) { if (c == i) return 0; ++c; }
    if (j < 3 * NS) { if (c == i) return 1 + j % NS; ++c; }
  }
  return 0;
}
template <int G> constexpr int step_of(int i) {
  constexpr int NS = Shape<G>::NSL - 1;
  int c = 0;
  for (int j = 0; j < 64; ++j) {
    if (j < 12) { if (c == i) return j; ++c; }
    if (j < 3 * NS) { if (c == i) return j / NS; ++c; }
  }
  return 0;
}

struct Args {
  const float* wpack; const float* hb; float* msg; float* out; long long* clk; int* err;
  int tiles;            // tiles per workgroup
  int epi;              // 1 = run the coupling epilogue
  int hb_tiles;         // tiles in the hidden-row image
};

// ---- the pieces shared by the three structures
template <int G>
struct Work {
  using S = Shape<G>;
  using D = Dim<S::NBK>;
  static constexpr int NCT = S::NCT, NCB = S::NCB, NSL = S::NSL, NW3 = S::NW3;
  // weight fragments of one k-row: slot 0 = 12 steps (3 requests of 4), short chains: NW3 requests of 3 (shared by 3 slots each)
  struct W { float b0[12]; float bs[NW3][3]; };
  static __device__ __forceinline__ void loadw(W& w, i32x4 wb, unsigned lane16, unsigned soff) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const f32x4 v = raw_ld4(wb, (int)(lane16 + 1024u * i), (int)soff, 0);
      w.b0[4 * i] = v[0]; w.b0[4 * i + 1] = v[1]; w.b0[4 * i + 2] = v[2]; w.b0[4 * i + 3] = v[3];
    }
#pragma unroll
    for (int i = 0; i < NW3; ++i) {
      const f32x3 v = raw_ld3(wb, (int)(lane16 / 16u * 12u + 3072u + 768u * i), (int)soff, 0);
      w.bs[i][0] = v[0]; w.bs[i][1] = v[1]; w.bs[i][2] = v[2];
    }
  }
  static __device__ __forceinline__ void loadw1(W& w, i32x4 wb, unsigned lane16, unsigned soff, int i) {   // request i of the row
    if (i < 3) {
      const f32x4 v = raw_ld4(wb, (int)(lane16 + 1024u * i), (int)soff, 0);
      w.b0[4 * i] = v[0]; w.b0[4 * i + 1] = v[1]; w.b0[4 * i + 2] = v[2]; w.b0[4 * i + 3] = v[3];
    } else {
      const f32x3 v = raw_ld3(wb, (int)(lane16 / 16u * 12u + 3072u + 768u * (i - 3)), (int)soff, 0);
      w.bs[i - 3][0] = v[0]; w.bs[i - 3][1] = v[1]; w.bs[i - 3][2] = v[2];
    }
  }
  static __device__ __forceinline__ float wfrag(const W& w, int slot, int step) {
    if (slot == 0) return w.b0[step];
    return w.bs[NW3 == 1 ? 0 : (slot - 1) / 3][step];
  }
  // contraction MFMA at position i of the row (r: one accumulator per slot, r0b: second accumulator of the long chain)
  template <int I>
  static __device__ __forceinline__ void cmma(f32x4 (&r)[NSL], f32x4& r0b, const float (&xa)[NCT], const W& w) {
    constexpr int t = slot_of<G>(I), j = step_of<G>(I);
    if constexpr (t == 0 && (j & 1)) r0b = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[I], wfrag(w, t, j), r0b, 0, 0, 0);
    else r[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[I], wfrag(w, t, j), r[t], 0, 0, 0);
  }
  // transposing store number p (0 .. 4 * NSL - 1) of a contracted row
  static __device__ __forceinline__ void store_piece(float* yw, int cstep, const f32x4 (&r)[NSL], int p) {
    const int s = p >> 2, rr = p & 3;
    yw[rr * D::YVN + cstep * s] = r[s][rr];
  }
};

// coupling epilogue of one (virtual node, row tile): five dependent wave-local LDS phases + 16-B global stores
template <int NCB>
__device__ __forceinline__ void epilogue_pair(const f32x4 (&acc)[NCB], float* gw, float* stg, float* msg_rows, int lane) {
  const int lr = lane & 15, lq = lane >> 4;
  // (1) coupling rows G[row][k'][slots]: lane = (row, part)
#pragma unroll
  for (int k = 0; k < 3; ++k) { gw[lr * 36 + 8 * k + lq] = 0.5f + 1e-3f * (float)(lane + k); gw[lr * 36 + 8 * k + 4 + lq] = 0.25f + 1e-3f * (float)lane; }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
  // (2) read back as 16-B pieces, couple with the accumulators, stage the message values
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * lq + r;
    float m[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float4 g4 = *reinterpret_cast<const float4*>(gw + row * 36 + 8 * k + 4 * (lr >> 3));
      float v = g4.x * acc[0][r];
      v = fmaf(g4.y, acc[1][r], v); v = fmaf(g4.z, acc[2][r], v); v = fmaf(g4.w, acc[3][r], v);
      if constexpr (NCB > 4) v = fmaf(g4.x, acc[4][r], v);
      m[k] = v + __shfl_xor(v, 8, 64);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) stg[row * 48 + lr * 3 + k] = m[k];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
  // (3) staged rows -> message rows, 16-B pieces: lane = 4 * row + q, pieces q, q + 4, q + 8
  const int row = lane >> 2, q = lane & 3;
#pragma unroll
  for (int cv = 0; cv < 3; ++cv) {
    const float4 v = *reinterpret_cast<const float4*>(stg + row * 48 + 4 * (q + 4 * cv));
    *reinterpret_cast<float4*>(msg_rows + (size_t)row * 160 + 4 * (q + 4 * cv)) = v;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
}

// =========================================================================================== MODE 0: lock-step, 8 waves
// XW: x fragments re-read from the x tile in LDS five contraction positions ahead of their MFMA (what the packed loops of k_conv_fused
// do to save 30 registers) instead of resident registers; SPREAD: one weight request per second edge-MFMA slot instead of a burst
template <int G, bool XW = false, bool SPREAD = false>
__global__ __launch_bounds__(512) void k_lock(Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using Wk = Work<G>; using S = Shape<G>; using D = Dim<S::NBK>;
  constexpr int NCT = S::NCT, NCB = S::NCB, NSL = S::NSL, NE = 8 * NCB, NP = 4 * NSL;
  const int tid = threadIdx.x, lane = tid & 63, wave = UNI(tid >> 6), lr = lane & 15, lq = lane >> 4;
  float* xbuf = lds;                         // [16][162]
  float* ybuf = xbuf + 16 * XSTR;            // [2][YB]
  float* gscr = ybuf + 2 * D::YB;            // per wave [16][36]
  float* stg = gscr + 8 * 16 * 36;           // per wave [16][48] (the kernel stages in the idle chunk buffers; same LDS traffic)
  for (int i = tid; i < 16 * XSTR; i += 512) xbuf[i] = 1e-3f * (float)(i % 97);
  __syncthreads();
  float xa[NCT];
#pragma unroll
  for (int i = 0; i < NCT; ++i) xa[i] = xbuf[lr * XSTR + 4 * i + lq];
  const i32x4 wb = mkbuf(a.wpack, (unsigned)HKROWS * KSROW * 4u);
  const unsigned lane16 = (unsigned)lane * 16u;
  const int cstep = G == 0 ? 16 : (lr < 8 ? 8 : lr < 10 ? 2 : 0);   // packed: channels 0-7 of slots 2b, 2b+1 share block b, channels 8, 9 the tail block
  float* const ywr = ybuf + (4 * lq) * D::YVN + wave * D::YROW + (G == 0 ? lr : (lr < 8 ? lr : lr < 10 ? 16 * (NCB - 1) + lr - 8 : 16 * S::NBK + lr - 10));
  const float* const yrd = ybuf + (2 * wave) * D::YVN + (2 * lq) * D::YROW + lr;
  float sum = 0.f;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int tl = 0; tl < a.tiles; ++tl) {
    const int tile = (blockIdx.x + 256 * tl) % a.hb_tiles;
    const i32x4 hbuf = mkbuf(a.hb + (size_t)tile * 16 * 2 * 9 * 256, 16u * 2u * 9u * 1024u);
    for (int gi = 0; gi < NGR; ++gi) {
      f32x4 acc[2][2][NCB];
#pragma unroll
      for (int i = 0; i < 4 * NCB; ++i) acc[i / (2 * NCB)][(i / NCB) & 1][i % NCB] = f32x4{0.f, 0.f, 0.f, 0.f};
      typename Wk::W w;
      unsigned woff = (unsigned)((wave * KSROW + gi * 1280) * 4);          // row k = 8 g + wave of this granule's columns
      unsigned hoff = (unsigned)(2 * wave) * 2u * 9u * 1024u;
      float4 hC[2][2], hN[2][2];
      Wk::loadw(w, wb, lane16, woff); woff += 8u * KSROW * 4u;
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) { const f32x4 v = raw_ld4(hbuf, (int)lane16, (int)(hoff + (unsigned)pc * 9u * 1024u), 0); hC[pc >> 1][pc & 1] = make_float4(v[0], v[1], v[2], v[3]); hN[pc >> 1][pc & 1] = hC[pc >> 1][pc & 1]; }
      hoff += 1024u;
      f32x4 r[NSL], r0b;
      {   // prologue: chunk 0 contracted, chunk 1 requested
#pragma unroll
        for (int t = 0; t < NSL; ++t) r[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        r0b = f32x4{0.f, 0.f, 0.f, 0.f};
        sfor<0, NCT>([&](auto ic) { Wk::template cmma<decltype(ic)::value>(r, r0b, xa, w); });
        r[0] += r0b;
#pragma unroll
        for (int p = 0; p < NP; ++p) Wk::store_piece(ywr, cstep, r, p);
        Wk::loadw(w, wb, lane16, woff); woff += 8u * KSROW * 4u;
      }
      __syncthreads();
      for (int g = 0; g < NCH; ++g) {
        const int eb = g & 1, cb = eb ^ 1;
        const bool do_c = g + 1 < NCH, do_w = g + 2 < NCH, do_h = (g & 1) && g + 1 < NCH;
        float q[2][NCB];
        auto readq = [&](int par, int grp) __attribute__((always_inline)) {
          const float* yb = yrd + eb * D::YB + (grp >> 1) * D::YVN + (grp & 1) * D::YROW;
#pragma unroll
          for (int c = 0; c < NCB; ++c) q[par][c] = yb[16 * c];
        };
        if (do_c) {
#pragma unroll
          for (int t = 0; t < NSL; ++t) r[t] = f32x4{0.f, 0.f, 0.f, 0.f};
          r0b = f32x4{0.f, 0.f, 0.f, 0.f};
          if constexpr (XW) {
            const float* xp = xbuf + lr * XSTR + lq;
#pragma unroll
            for (int i = 0; i < 5; ++i) xa[i] = xp[4 * i];
          }
          sfor<0, NCT>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (XW && i + 5 < NCT) xa[i + 5] = (xbuf + lr * XSTR + lq)[4 * (i + 5)];
            Wk::template cmma<i>(r, r0b, xa, w);
            if (i == NCT - 3) readq(0, 0);
            FENCE();
          });
          r[0] += r0b;
        } else {
          readq(0, 0);
        }
        if (!(g & 1) && g > 0) {
#pragma unroll
          for (int pc = 0; pc < 4; ++pc) hC[pc >> 1][pc & 1] = hN[pc >> 1][pc & 1];
          FENCE();
        }
        typename Wk::W wn = w;
        sfor<0, NE>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          constexpr int grp = m / (2 * NCB), t8 = m % (2 * NCB), vi = grp >> 1, sub = grp & 1, rt = t8 / NCB, c = t8 % NCB;
          const float4& h = hC[vi][rt];
          const float av = (g & 1) ? (sub == 0 ? h.z : h.w) : (sub == 0 ? h.x : h.y);
          acc[vi][rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, q[grp & 1][c], acc[vi][rt][c], 0, 0, 0);
          if constexpr (!SPREAD) { if constexpr (m == 0) { if (do_w) { Wk::loadw(wn, wb, lane16, woff); } } }
          else { if constexpr (m % 2 == 0 && m / 2 < 3 + Wk::NW3) { if (do_w) Wk::loadw1(wn, wb, lane16, woff, m / 2); } }
          if constexpr (m >= 8 && m < 12) {
            if (do_h) { const f32x4 v = raw_ld4(hbuf, (int)lane16, (int)(hoff + (unsigned)(m - 8) * 9u * 1024u), 0); hN[(m - 8) >> 1][(m - 8) & 1] = make_float4(v[0], v[1], v[2], v[3]); }
          }
          if constexpr (m >= 2 && m < 2 + NP) { if (do_c) Wk::store_piece(ywr + cb * D::YB, cstep, r, m - 2); }
          if constexpr (t8 == 1 && grp < 3) readq((grp + 1) & 1, grp + 1);
          FENCE();
        });
        w = wn;
        if (do_w) woff += 8u * KSROW * 4u;
        if (do_h) hoff += 1024u;
        __syncthreads();
      }
      if (a.epi) {
#pragma unroll
        for (int vi = 0; vi < 2; ++vi)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
            epilogue_pair<NCB>(acc[vi][rt], gscr + wave * 16 * 36, stg + wave * 16 * 48,
                               a.msg + ((size_t)(blockIdx.x * 16 + 2 * wave + vi) * 32 + 16 * rt) * 160, lane);
        __syncthreads();
      } else {
#pragma unroll
        for (int i = 0; i < 4 * NCB; ++i) sum += acc[i / (2 * NCB)][(i / NCB) & 1][i % NCB][0];
      }
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  a.out[blockIdx.x * 512 + tid] = sum;
  if (blockIdx.x == 0 && tid == 0) { a.clk[0] = c1 - c0; a.clk[1] = w1 - w0; }
}

// =========================================================================================== MODE 1: producer / consumer waves
__device__ __forceinline__ bool wait_ge(volatile int* f, int need, int* err) {   // all four flags >= need (bounded spin)
  for (int spin = 0; spin < (1 << 22); ++spin) {
    const int m = min(min(f[0], f[1]), min(f[2], f[3]));
    if (UNI(m) >= need) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  *err = 1;
  return false;
}
template <int G, int R>
__global__ __launch_bounds__(512) void k_pc(Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using Wk = Work<G>; using S = Shape<G>; using D = Dim<S::NBK>;
  constexpr int NCT = S::NCT, NCB = S::NCB, NSL = S::NSL, NP = 4 * NSL;
  const int tid = threadIdx.x, lane = tid & 63, wave = UNI(tid >> 6), lr = lane & 15, lq = lane >> 4;
  float* xbuf = lds;                         // [16][162]
  float* ring = xbuf + 16 * XSTR;            // [R][YB]
  float* gscr = ring + R * D::YB;            // per consumer wave [16][36]
  float* stg = gscr + 4 * 16 * 36;           // per consumer wave [16][48]
  volatile int* pflag = reinterpret_cast<volatile int*>(stg + 4 * 16 * 48);   // [4] chunks published by producer p
  volatile int* cflag = pflag + 4;                                             // [4] chunks released by consumer c
  for (int i = tid; i < 16 * XSTR; i += 512) xbuf[i] = 1e-3f * (float)(i % 97);
  if (tid < 8) pflag[tid] = 0;
  __syncthreads();
  const i32x4 wb = mkbuf(a.wpack, (unsigned)HKROWS * KSROW * 4u);
  const unsigned lane16 = (unsigned)lane * 16u;
  float sum = 0.f;
  const long long c0 = clock64(), w0 = wall_clock64();
  if (wave < 4) {
    // ---------------- producer p: rows 2p, 2p + 1 of every chunk
    const int p = wave;
    float xa[NCT];
#pragma unroll
    for (int i = 0; i < NCT; ++i) xa[i] = xbuf[lr * XSTR + 4 * i + lq];
    const int cstep = G == 0 ? 16 : (lr < 8 ? 8 : lr < 10 ? 2 : 0);
    float* const ywr = ring + (4 * lq) * D::YVN + (2 * p) * D::YROW + (G == 0 ? lr : (lr < 8 ? lr : lr < 10 ? 16 * (NCB - 1) + lr - 8 : 16 * S::NBK + lr - 10));
    int seq = 0;                              // chunks published so far (monotonic over granules and tiles)
    for (int tl = 0; tl < a.tiles; ++tl)
      for (int gi = 0; gi < NGR; ++gi) {
        typename Wk::W w0_, w1_;
        unsigned woff = (unsigned)((2 * p * KSROW + gi * 1280) * 4);
        Wk::loadw(w0_, wb, lane16, woff); Wk::loadw(w1_, wb, lane16, woff + KSROW * 4u);
        woff += 8u * KSROW * 4u;
        for (int g = 0; g < NCH; ++g, ++seq) {
          if (seq >= R) { if (!wait_ge(cflag, seq - R + 1, a.err)) return; }   // slot free: every consumer is past chunk seq - R
          float* yw = ywr + (seq % R) * D::YB;
          f32x4 r[NSL], r0b;
          typename Wk::W n0 = w0_, n1 = w1_;
          const bool more = g + 1 < NCH;
          // row 2p
#pragma unroll
          for (int t = 0; t < NSL; ++t) r[t] = f32x4{0.f, 0.f, 0.f, 0.f};
          r0b = f32x4{0.f, 0.f, 0.f, 0.f};
          sfor<0, NCT>([&](auto ic) { Wk::template cmma<decltype(ic)::value>(r, r0b, xa, w0_); if (decltype(ic)::value == 0 && more) Wk::loadw(n0, wb, lane16, woff); FENCE(); });
          r[0] += r0b;
          f32x4 s[NSL];
#pragma unroll
          for (int t = 0; t < NSL; ++t) s[t] = r[t];
          // row 2p + 1, the stores of row 2p behind its MFMAs
#pragma unroll
          for (int t = 0; t < NSL; ++t) r[t] = f32x4{0.f, 0.f, 0.f, 0.f};
          r0b = f32x4{0.f, 0.f, 0.f, 0.f};
          sfor<0, NCT>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            Wk::template cmma<i>(r, r0b, xa, w1_);
            if (i == 0 && more) Wk::loadw(n1, wb, lane16, woff + KSROW * 4u);
            if constexpr (i >= 1 && i < 1 + NP) Wk::store_piece(yw, cstep, s, i - 1);
            FENCE();
          });
          r[0] += r0b;
#pragma unroll
          for (int pc = 0; pc < NP; ++pc) Wk::store_piece(yw + D::YROW, cstep, r, pc);
          if (lane == 0) pflag[p] = seq + 1;   // in order behind this wave's data stores in the LDS
          w0_ = n0; w1_ = n1;
          if (more) woff += 8u * KSROW * 4u;
        }
      }
  } else {
    // ---------------- consumer c: virtual nodes 4c .. 4c + 3
    const int c_ = wave - 4;
    const float* const yrd = ring + (4 * c_) * D::YVN + (2 * lq) * D::YROW + lr;
    int seq = 0;
    for (int tl = 0; tl < a.tiles; ++tl) {
      const int tile = (blockIdx.x + 256 * tl) % a.hb_tiles;
      const i32x4 hbuf = mkbuf(a.hb + (size_t)tile * 16 * 2 * 9 * 256, 16u * 2u * 9u * 1024u);
      for (int gi = 0; gi < NGR; ++gi) {
        f32x4 acc[4][2][NCB];
#pragma unroll
        for (int i = 0; i < 8 * NCB; ++i) acc[i / (2 * NCB)][(i / NCB) & 1][i % NCB] = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned hoff = (unsigned)(4 * c_) * 2u * 9u * 1024u;
        // hidden rows: with 160 accumulator registers there is no room for a second copy -- the next pair of chunks of a virtual
        // node is requested IN PLACE right behind its last MFMA of the odd step (12 * NCB MFMAs = ~2000 cycles before its next use)
        f32x4 hC[4][2];
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) hC[pc >> 1][pc & 1] = raw_ld4(hbuf, (int)lane16, (int)(hoff + (unsigned)pc * 9u * 1024u), 0);
        hoff += 1024u;
        for (int g = 0; g < NCH; ++g, ++seq) {
          if (!wait_ge(pflag, seq + 1, a.err)) return;
          const float* yb0 = yrd + (seq % R) * D::YB;
          const bool do_h = (g & 1) && g + 1 < NCH;
          float q[2][NCB];
          auto readq = [&](int par, int grp) __attribute__((always_inline)) {
            const float* yb = yb0 + (grp >> 1) * D::YVN + (grp & 1) * D::YROW;
#pragma unroll
            for (int c = 0; c < NCB; ++c) q[par][c] = yb[16 * c];
          };
          readq(0, 0);
          sfor<0, 16 * NCB>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            constexpr int grp = m / (2 * NCB), t8 = m % (2 * NCB), vi = grp >> 1, sub = grp & 1, rt = t8 / NCB, c = t8 % NCB;
            const f32x4& h = hC[vi][rt];
            const float av = (g & 1) ? (sub == 0 ? h[2] : h[3]) : (sub == 0 ? h[0] : h[1]);
            acc[vi][rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, q[grp & 1][c], acc[vi][rt][c], 0, 0, 0);
            if constexpr (m % (4 * NCB) == 4 * NCB - 1) {   // last MFMA of virtual node m / (4 NCB) in this step
              if (do_h) {
                constexpr int v = m / (4 * NCB);
                hC[v][0] = raw_ld4(hbuf, (int)lane16, (int)(hoff + (unsigned)(2 * v) * 9u * 1024u), 0);
                hC[v][1] = raw_ld4(hbuf, (int)lane16, (int)(hoff + (unsigned)(2 * v + 1) * 9u * 1024u), 0);
              }
            }
            if constexpr (t8 == 1 && grp < 7) readq((grp + 1) & 1, grp + 1);
            FENCE();
          });
          if (do_h) hoff += 1024u;
          if (lane == 0) cflag[c_] = seq + 1;     // in order behind this wave's chunk reads in the LDS
        }
        if (a.epi) {
#pragma unroll
          for (int vi = 0; vi < 4; ++vi)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
              epilogue_pair<NCB>(acc[vi][rt], gscr + c_ * 16 * 36, stg + c_ * 16 * 48,
                                 a.msg + ((size_t)(blockIdx.x * 16 + 4 * c_ + vi) * 32 + 16 * rt) * 160, lane);
        } else {
#pragma unroll
          for (int i = 0; i < 8 * NCB; ++i) sum += acc[i / (2 * NCB)][(i / NCB) & 1][i % NCB][0];
        }
      }
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  a.out[blockIdx.x * 512 + tid] = sum;
  if (blockIdx.x == 0 && tid == 256) { a.clk[0] = c1 - c0; a.clk[1] = w1 - w0; }
}

// =========================================================================================== MODE 2: one wave per SIMD
template <int G>
__global__ __launch_bounds__(256) void k_solo(Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using Wk = Work<G>; using S = Shape<G>; using D = Dim<S::NBK>;
  constexpr int NCT = S::NCT, NCB = S::NCB, NSL = S::NSL, NP = 4 * NSL, NE = 16 * NCB;
  const int tid = threadIdx.x, lane = tid & 63, wave = UNI(tid >> 6), lr = lane & 15, lq = lane >> 4;
  float* xbuf = lds;
  float* ybuf = xbuf + 16 * XSTR;            // [2][YB]
  float* gscr = ybuf + 2 * D::YB;
  float* stg = gscr + 4 * 16 * 36;
  for (int i = tid; i < 16 * XSTR; i += 256) xbuf[i] = 1e-3f * (float)(i % 97);
  __syncthreads();
  float xa[NCT];
#pragma unroll
  for (int i = 0; i < NCT; ++i) xa[i] = xbuf[lr * XSTR + 4 * i + lq];
  const i32x4 wb = mkbuf(a.wpack, (unsigned)HKROWS * KSROW * 4u);
  const unsigned lane16 = (unsigned)lane * 16u;
  const int cstep = G == 0 ? 16 : (lr < 8 ? 8 : lr < 10 ? 2 : 0);
  float* const ywr = ybuf + (4 * lq) * D::YVN + (2 * wave) * D::YROW + (G == 0 ? lr : (lr < 8 ? lr : lr < 10 ? 16 * (NCB - 1) + lr - 8 : 16 * S::NBK + lr - 10));
  const float* const yrd = ybuf + (4 * wave) * D::YVN + (2 * lq) * D::YROW + lr;
  float sum = 0.f;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int tl = 0; tl < a.tiles; ++tl) {
    const int tile = (blockIdx.x + 256 * tl) % a.hb_tiles;
    const i32x4 hbuf = mkbuf(a.hb + (size_t)tile * 16 * 2 * 9 * 256, 16u * 2u * 9u * 1024u);
    for (int gi = 0; gi < NGR; ++gi) {
      f32x4 acc[4][2][NCB];
#pragma unroll
      for (int i = 0; i < 8 * NCB; ++i) acc[i / (2 * NCB)][(i / NCB) & 1][i % NCB] = f32x4{0.f, 0.f, 0.f, 0.f};
      typename Wk::W w0_, w1_;
      unsigned woff = (unsigned)((2 * wave * KSROW + gi * 1280) * 4);
      unsigned hoff = (unsigned)(4 * wave) * 2u * 9u * 1024u;
      f32x4 hC[4][2], hN[4][2];
      Wk::loadw(w0_, wb, lane16, woff); Wk::loadw(w1_, wb, lane16, woff + KSROW * 4u); woff += 8u * KSROW * 4u;
#pragma unroll
      for (int pc = 0; pc < 8; ++pc) { hC[pc >> 1][pc & 1] = raw_ld4(hbuf, (int)lane16, (int)(hoff + (unsigned)pc * 9u * 1024u), 0); hN[pc >> 1][pc & 1] = hC[pc >> 1][pc & 1]; }
      hoff += 1024u;
      f32x4 r[2][NSL], r0b[2];
      auto contract_all = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NSL; ++t) { r[0][t] = f32x4{0.f, 0.f, 0.f, 0.f}; r[1][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        r0b[0] = r0b[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        sfor<0, NCT>([&](auto ic) { Wk::template cmma<decltype(ic)::value>(r[0], r0b[0], xa, w0_); Wk::template cmma<decltype(ic)::value>(r[1], r0b[1], xa, w1_); });
        r[0][0] += r0b[0]; r[1][0] += r0b[1];
      };
      contract_all();
#pragma unroll
      for (int p = 0; p < NP; ++p) { Wk::store_piece(ywr, cstep, r[0], p); Wk::store_piece(ywr + D::YROW, cstep, r[1], p); }
      Wk::loadw(w0_, wb, lane16, woff); Wk::loadw(w1_, wb, lane16, woff + KSROW * 4u); woff += 8u * KSROW * 4u;
      __syncthreads();
      for (int g = 0; g < NCH; ++g) {
        const int eb = g & 1, cb = eb ^ 1;
        const bool do_c = g + 1 < NCH, do_w = g + 2 < NCH, do_h = (g & 1) && g + 1 < NCH;
        float q[2][NCB];
        auto readq = [&](int par, int grp) __attribute__((always_inline)) {
          const float* yb = yrd + eb * D::YB + (grp >> 1) * D::YVN + (grp & 1) * D::YROW;
#pragma unroll
          for (int c = 0; c < NCB; ++c) q[par][c] = yb[16 * c];
        };
        readq(0, 0);
        if (!(g & 1) && g > 0) {
#pragma unroll
          for (int pc = 0; pc < 8; ++pc) hC[pc >> 1][pc & 1] = hN[pc >> 1][pc & 1];
        }
#pragma unroll
        for (int t = 0; t < NSL; ++t) { r[0][t] = f32x4{0.f, 0.f, 0.f, 0.f}; r[1][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        r0b[0] = r0b[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        typename Wk::W n0 = w0_, n1 = w1_;
        // one stream: edge MFMA m, and a contraction MFMA of row (cpos & 1) at every slot where cpos * NE < m * 2 * NCT advances
        sfor<0, NE>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          constexpr int grp = m / (2 * NCB), t8 = m % (2 * NCB), vi = grp >> 1, sub = grp & 1, rt = t8 / NCB, c = t8 % NCB;
          const f32x4& h = hC[vi][rt];
          const float av = (g & 1) ? (sub == 0 ? h[2] : h[3]) : (sub == 0 ? h[0] : h[1]);
          acc[vi][rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, q[grp & 1][c], acc[vi][rt][c], 0, 0, 0);
          constexpr int lo = m * 2 * NCT / NE, hi = (m + 1) * 2 * NCT / NE;   // contraction positions [lo, hi) of the 2 * NCT
          if (do_c) {
            sfor<lo, hi>([&](auto pc_) {
              constexpr int cp = decltype(pc_)::value, row = cp & 1, i = cp >> 1;
              if constexpr (row == 0) Wk::template cmma<i>(r[0], r0b[0], xa, w0_); else Wk::template cmma<i>(r[1], r0b[1], xa, w1_);
            });
          }
          if constexpr (m == 1) { if (do_w) { Wk::loadw(n0, wb, lane16, woff); Wk::loadw(n1, wb, lane16, woff + KSROW * 4u); } }
          if constexpr (m >= 4 && m < 12) { if (do_h) hN[(m - 4) >> 1][(m - 4) & 1] = raw_ld4(hbuf, (int)lane16, (int)(hoff + (unsigned)(m - 4) * 9u * 1024u), 0); }
          if constexpr (t8 == 1 && grp < 7) readq((grp + 1) & 1, grp + 1);
          FENCE();
        });
        if (do_c) {
          r[0][0] += r0b[0]; r[1][0] += r0b[1];
#pragma unroll
          for (int p = 0; p < NP; ++p) { Wk::store_piece(ywr + cb * D::YB, cstep, r[0], p); Wk::store_piece(ywr + cb * D::YB + D::YROW, cstep, r[1], p); }
        }
        w0_ = n0; w1_ = n1;
        if (do_w) woff += 8u * KSROW * 4u;
        if (do_h) hoff += 1024u;
        __syncthreads();
      }
      if (a.epi) {
#pragma unroll
        for (int vi = 0; vi < 4; ++vi)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
            epilogue_pair<NCB>(acc[vi][rt], gscr + wave * 16 * 36, stg + wave * 16 * 48,
                               a.msg + ((size_t)(blockIdx.x * 16 + 4 * wave + vi) * 32 + 16 * rt) * 160, lane);
        __syncthreads();
      } else {
#pragma unroll
        for (int i = 0; i < 8 * NCB; ++i) sum += acc[i / (2 * NCB)][(i / NCB) & 1][i % NCB][0];
      }
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  a.out[blockIdx.x * 256 + tid] = sum;
  if (blockIdx.x == 0 && tid == 0) { a.clk[0] = c1 - c0; a.clk[1] = w1 - w0; }
}

// =========================================================================================== MODE 3: two 4-wave workgroups per CU
// TWO independent workgroups per CU, 4 waves each (one per SIMD), <= 256 registers and <= 80 KB of LDS per workgroup: the two
// waves of a SIMD then belong to different tiles and drift into different phases by themselves -- the coupling epilogue and
// the granule prologue of one tile run beside the main loop of the other.  Per workgroup the chunk is SINGLE-buffered and a
// step is [edge product of chunk g] barrier [contract rows 2w, 2w+1 of chunk g+1, one row at a time, weights just in time]
// barrier: nothing is overlapped inside a workgroup, the partner workgroup is the overlap.  Classic granules only: 4 virtual
// nodes x 2 row tiles x 4 column blocks = 128 accumulator registers per wave (the 5-block packed granule would need 160).
template <int G>
__global__ __launch_bounds__(256, 2) void k_duo(Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using Wk = Work<G>; using S = Shape<G>; using D = Dim<S::NBK>;
  constexpr int NCT = S::NCT, NCB = S::NCB, NSL = S::NSL, NP = 4 * NSL, NE = 16 * NCB;
  const int tid = threadIdx.x, lane = tid & 63, wave = UNI(tid >> 6), lr = lane & 15, lq = lane >> 4;
  float* xbuf = lds;
  float* ybuf = xbuf + 16 * XSTR;            // [YB] one chunk
  float* gscr = ybuf + D::YB;
  float* stg = gscr + 4 * 16 * 36;
  for (int i = tid; i < 16 * XSTR; i += 256) xbuf[i] = 1e-3f * (float)(i % 97);
  __syncthreads();
  const i32x4 wb = mkbuf(a.wpack, (unsigned)HKROWS * KSROW * 4u);
  const unsigned lane16 = (unsigned)lane * 16u;
  const int cstep = G == 0 ? 16 : (lr < 8 ? 8 : lr < 10 ? 2 : 0);
  float* const ywr = ybuf + (4 * lq) * D::YVN + (2 * wave) * D::YROW + (G == 0 ? lr : (lr < 8 ? lr : lr < 10 ? 16 * (NCB - 1) + lr - 8 : 16 * S::NBK + lr - 10));
  const float* const yrd = ybuf + (4 * wave) * D::YVN + (2 * lq) * D::YROW + lr;
  float sum = 0.f;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int tl = 0; tl < a.tiles; ++tl) {
    const int tile = (blockIdx.x + 512 * tl) % a.hb_tiles;
    const i32x4 hbuf = mkbuf(a.hb + (size_t)tile * 16 * 2 * 9 * 256, 16u * 2u * 9u * 1024u);
    for (int gi = 0; gi < NGR; ++gi) {
      f32x4 acc[4][2][NCB];
#pragma unroll
      for (int i = 0; i < 8 * NCB; ++i) acc[i / (2 * NCB)][(i / NCB) & 1][i % NCB] = f32x4{0.f, 0.f, 0.f, 0.f};
      typename Wk::W w;
      unsigned woff = (unsigned)((2 * wave * KSROW + gi * 1280) * 4);
      unsigned hoff = (unsigned)(4 * wave) * 2u * 9u * 1024u;
      f32x4 hC[4][2];
#pragma unroll
      for (int pc = 0; pc < 8; ++pc) hC[pc >> 1][pc & 1] = raw_ld4(hbuf, (int)lane16, (int)(hoff + (unsigned)pc * 9u * 1024u), 0);
      hoff += 1024u;
      auto contract_rows = [&]() __attribute__((always_inline)) {   // rows 2w, 2w + 1 of the next chunk, one after the other
        int xo = lr * XSTR + lq;
        asm volatile("" : "+v"(xo));           // (x fragments are re-read per chunk: resident they would not fit beside 128 accumulators)
        float xa[NCT];
#pragma unroll
        for (int i = 0; i < NCT; ++i) xa[i] = xbuf[xo + 4 * i];
#pragma unroll
        for (int row = 0; row < 2; ++row) {
          Wk::loadw(w, wb, lane16, woff + (unsigned)row * KSROW * 4u);
          f32x4 r[NSL], r0b;
#pragma unroll
          for (int t = 0; t < NSL; ++t) r[t] = f32x4{0.f, 0.f, 0.f, 0.f};
          r0b = f32x4{0.f, 0.f, 0.f, 0.f};
          sfor<0, NCT>([&](auto ic) { Wk::template cmma<decltype(ic)::value>(r, r0b, xa, w); FENCE(); });
          r[0] += r0b;
#pragma unroll
          for (int p = 0; p < NP; ++p) Wk::store_piece(ywr + row * D::YROW, cstep, r, p);
        }
        woff += 8u * KSROW * 4u;
      };
      contract_rows();
      __syncthreads();
      for (int g = 0; g < NCH; ++g) {
        const bool do_c = g + 1 < NCH, do_h = (g & 1) && g + 1 < NCH;
        float q[2][NCB];
        auto readq = [&](int par, int grp) __attribute__((always_inline)) {
          const float* yb = yrd + (grp >> 1) * D::YVN + (grp & 1) * D::YROW;
#pragma unroll
          for (int c = 0; c < NCB; ++c) q[par][c] = yb[16 * c];
        };
        readq(0, 0);
        sfor<0, NE>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          constexpr int grp = m / (2 * NCB), t8 = m % (2 * NCB), vi = grp >> 1, sub = grp & 1, rt = t8 / NCB, c = t8 % NCB;
          const f32x4& h = hC[vi][rt];
          const float av = (g & 1) ? (sub == 0 ? h[2] : h[3]) : (sub == 0 ? h[0] : h[1]);
          acc[vi][rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, q[grp & 1][c], acc[vi][rt][c], 0, 0, 0);
          if constexpr (m % (4 * NCB) == 4 * NCB - 1) {   // hidden rows of the next pair of chunks, in place behind the node's last MFMA
            if (do_h) {
              constexpr int v = m / (4 * NCB);
              hC[v][0] = raw_ld4(hbuf, (int)lane16, (int)(hoff + (unsigned)(2 * v) * 9u * 1024u), 0);
              hC[v][1] = raw_ld4(hbuf, (int)lane16, (int)(hoff + (unsigned)(2 * v + 1) * 9u * 1024u), 0);
            }
          }
          if constexpr (t8 == 1 && grp < 7) readq((grp + 1) & 1, grp + 1);
          FENCE();
        });
        if (do_h) hoff += 1024u;
        __syncthreads();                       // every wave is done reading the chunk
        if (do_c) { contract_rows(); __syncthreads(); }
      }
      if (a.epi) {
#pragma unroll
        for (int vi = 0; vi < 4; ++vi)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
            epilogue_pair<NCB>(acc[vi][rt], gscr + wave * 16 * 36, stg + wave * 16 * 48,
                               a.msg + ((size_t)((blockIdx.x & 255) * 16 + 4 * wave + vi) * 32 + 16 * rt) * 160, lane);
            FENCE();
          }
      } else {
#pragma unroll
        for (int i = 0; i < 8 * NCB; ++i) sum += acc[i / (2 * NCB)][(i / NCB) & 1][i % NCB][0];
      }
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  a.out[(blockIdx.x & 255) * 512 + tid] = sum;
  if (blockIdx.x == 0 && tid == 0) { a.clk[0] = c1 - c0; a.clk[1] = w1 - w0; }
}

// =========================================================================================== host
template <class K>
static void run(const char* name, K kern, int threads, size_t smem, Args a, int G, int grid = 256) {
  if (smem > 160 * 1024) { printf("%-44s LDS %zu KB > 160 KB: does not fit\n", name, smem / 1024); return; }
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f; double clock_mhz = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(a.err, 0, 4);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), smem, 0, a);
    hipEventRecord(e1);
    if (hipEventSynchronize(e1) != hipSuccess) { printf("%-44s launch failed: %s\n", name, hipGetErrorString(hipGetLastError())); return; }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, a.clk, 16, hipMemcpyDeviceToHost);
    int err; hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost);
    if (err) { printf("%-44s spin bound hit (protocol error)\n", name); return; }
    if (rep > 0 && ms < best) { best = ms; clock_mhz = (double)h[0] / ((double)h[1] / 100.0); }
  }
  const int NCT = G == 0 ? 21 : 30, NCB = G == 0 ? 4 : 5;
  const double granules = (double)a.tiles * NGR * (grid / 256);   // per CU
  const double mfma_simd = granules * ((NCH + 0) * (8.0 * NCT + 16.0 * 4 * NCB) + 0) / 4.0;   // per SIMD (prologue contraction ~ the last step's missing one)
  const double cyc = best * 1e-3 * clock_mhz * 1e6;
  printf("%-44s epi=%d  %8.3f ms  %7.1f ns/granule  clock %4.0f MHz  %6.2f cycles per MFMA and SIMD  (LDS %zu KB)\n", name, a.epi, best,
         best * 1e6 / granules, clock_mhz, cyc / mfma_simd, smem / 1024);
}

int main(int argc, char** argv) {
  const int tiles = argc > 1 ? atoi(argv[1]) : 6;
  Args a{};
  const int hb_tiles = 768;
  float *wpack, *hb, *msg, *out; long long* clk; int* err;
  hipMalloc(&wpack, (size_t)HKROWS * KSROW * 4 + 65536); hipMemset(wpack, 0, (size_t)HKROWS * KSROW * 4 + 65536);
  hipMalloc(&hb, (size_t)hb_tiles * 16 * 2 * 9 * 1024); hipMemset(hb, 0, (size_t)hb_tiles * 16 * 2 * 9 * 1024);
  hipMalloc(&msg, (size_t)256 * 16 * 32 * 160 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 16); hipMalloc(&err, 4);
  a.wpack = wpack; a.hb = hb; a.msg = msg; a.out = out; a.clk = clk; a.err = err; a.tiles = tiles; a.hb_tiles = hb_tiles;
  auto lds_lock = [](int nbk) { return (size_t)(16 * XSTR + 2 * (16 * (8 * (16 * nbk + 8) + 4)) + 8 * 16 * 36 + 8 * 16 * 48) * 4; };
  auto lds_pc = [](int nbk, int R) { return (size_t)(16 * XSTR + R * (16 * (8 * (16 * nbk + 8) + 4)) + 4 * 16 * 36 + 4 * 16 * 48 + 16) * 4; };
  auto lds_duo = [](int nbk) { return (size_t)(16 * XSTR + (16 * (8 * (16 * nbk + 8) + 4)) + 4 * 16 * 36 + 4 * 16 * 48) * 4; };
  auto lds_solo = [](int nbk) { return (size_t)(16 * XSTR + 2 * (16 * (8 * (16 * nbk + 8) + 4)) + 4 * 16 * 36 + 4 * 16 * 48) * 4; };
  for (int epi = 0; epi < 2; ++epi) {
    a.epi = epi;
    run("classic LOCK  8 waves, barrier per chunk", k_lock<0>, 512, lds_lock(4), a, 0);
    run("classic PC    4 + 4 waves, ring 2", k_pc<0, 2>, 512, lds_pc(4, 2), a, 0);
    run("classic PC    4 + 4 waves, ring 3", k_pc<0, 3>, 512, lds_pc(4, 3), a, 0);
    run("classic PC    4 + 4 waves, ring 4", k_pc<0, 4>, 512, lds_pc(4, 4), a, 0);
    run("classic SOLO  4 waves, one per SIMD", k_solo<0>, 256, std::max(lds_solo(4), (size_t)84 * 1024), a, 0);
    run("classic DUO   2 workgroups x 4 waves per CU", k_duo<0>, 256, lds_duo(4), a, 0, 512);
    run("packed  LOCK  8 waves, barrier per chunk", k_lock<1>, 512, lds_lock(5), a, 1);
    run("packed  LOCK  + x fragments through LDS", k_lock<1, true, false>, 512, lds_lock(5), a, 1);
    run("packed  LOCK  + x window + spread requests", k_lock<1, true, true>, 512, lds_lock(5), a, 1);
    run("classic LOCK  + spread weight requests", k_lock<0, false, true>, 512, lds_lock(4), a, 0);
    run("packed  PC    4 + 4 waves, ring 2", k_pc<1, 2>, 512, lds_pc(5, 2), a, 1);
    run("packed  PC    4 + 4 waves, ring 3", k_pc<1, 3>, 512, lds_pc(5, 3), a, 1);
    run("packed  SOLO  4 waves, one per SIMD", k_solo<1>, 256, std::max(lds_solo(5), (size_t)84 * 1024), a, 1);
  }
  return 0;
}
