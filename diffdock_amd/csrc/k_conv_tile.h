// Device code of the fused convolution tile (k_conv_fused / k_conv_grouped), shared by the kernel translation units:
//   k_conv_f32.hip   exact-f32 l <= 1 kernels (static chain shapes: sparse / dense / shared-node rows; the predicated generic one)
//   k_conv_bf.hip    the same loops with the split-bf16 edge product
//   k_conv_l2.hip    sh_lmax = 2 / second-order kernels
//   k_conv_grp.hip   the grouped dispatch: every edge group of an interaction layer in ONE launch
// (one TU per family so that they compile in parallel; k_conv.hip keeps the virtual-node lists, the dispatcher and k_reduce_bn).
// Reference: TensorProductConvLayer.forward models/tensor_layers.py:309-335 -- see the header of k_conv.hip for the re-association.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <type_traits>
#include <utility>
#include <vector>

#include "kernels.h"
#include "fc_layout.h"

namespace ddmi {


// x tile rows in LDS: stride 162 = 2 (mod 32).  An A fragment read has lane (node lr, quarter lq) at lr * stride + din * lq + c
// (din = 1 or 3: odd): the 16 nodes land on 16 distinct EVEN banks and the next quarter on the odd ones -- conflict-free per
// 32-lane half; an odd stride (161) put (lr, lq) and (lr + din, lq - 1) on one bank (2-way on almost every read).
constexpr int NC_XS = XS + 2;

// Profiling builds only (-DDDMI_PROFILING, tools/build_variant.sh): DDMI_ABLATE switches individual kernel phases off
// (garbage scores, timing only) and k_conv_fused accumulates per-phase cycle counts (fc_prof_report).  The shipped library
// contains neither: DDMI_ABL() folds to false and the stamps to nothing.
#ifdef DDMI_PROFILING
static int ablate_mask() { static int m = getenv("DDMI_ABLATE") ? atoi(getenv("DDMI_ABLATE")) : 0; return m; }
#define DDMI_ABL(mask, bit) (((mask) & (bit)) != 0)
#else
static int ablate_mask() { return 0; }
#define DDMI_ABL(mask, bit) false
#endif
#if defined(DDMI_PROFILING) && DDMI_PROFILING >= 2   // (the clocks cost registers: -DDDMI_PROFILING=1 builds carry the ablation hooks only)
#define DDMI_PHASE_CLOCKS 1
// per-wave phase clocks of k_conv_fused (s_memtime at the phase boundaries, summed per edge-group slot)
constexpr int FC_NPROF = 20, FC_PROF_SLOTS = 12;
static __device__ unsigned long long g_fc_prof[FC_PROF_SLOTS * FC_NPROF];
struct FcProf {
  unsigned t; unsigned acc[FC_NPROF];
  __device__ __forceinline__ void start() {
    t = (unsigned)__builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < FC_NPROF; ++i) acc[i] = 0;
  }
  __device__ __forceinline__ void stamp(int i) {
    __builtin_amdgcn_sched_barrier(0);
    const unsigned now = (unsigned)__builtin_readcyclecounter();
    acc[i] += now - t; t = now;
    __builtin_amdgcn_sched_barrier(0);
  }
};
#define FC_STAMP(pf, i) (pf).stamp(i)
#define FC_COUNT(pf, i) ((pf).acc[i] += 1)
#if DDMI_PROFILING >= 3   // also the barrier waits inside the main loop (two more clock reads per chunk: perturbs the loop)
#define FC_STAMP_FINE(pf, i) (pf).stamp(i)
#define DDMI_PROF_FINE 1
#else
#define FC_STAMP_FINE(pf, i) ((void)0)
#define DDMI_PROF_FINE 0
#endif
static const char* const fc_prof_names[FC_NPROF] = {"kernel_prologue", "granule_setup", "ml_prologue", "ml_steady", "bias", "barrier_pre_couple",
                                                    "G_rows", "couple_stage", "store_rows", "barrier_end", "wave_total", "ml_barrier_wait",
                                                    "ml_pro_chunk0", "ml_pro_weights1", "granules", "waves", "tile_pro_issue", "tile_pro_arrive", "-", "-"};
// (ml_pro_chunk0 + ml_pro_weights1 + ml_prologue = the main-loop prologue: up to chunk 0 contracted and stored | chunk 1's weights
// in registers | barrier and bias rows; tile_pro_issue + tile_pro_arrive + kernel_prologue = the tile prologue: every request
// issued | all of them arrived and copied to LDS | the rest)
// (the clocks and stamps are per translation unit: every kernel TU exports a report function, fc_prof_report() calls them all)
static void fc_prof_report_tu() {
  unsigned long long h[FC_PROF_SLOTS * FC_NPROF];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fc_prof), sizeof(h)) != hipSuccess) return;
  for (int sl = 0; sl < FC_PROF_SLOTS; ++sl) {
    const unsigned long long* r = h + sl * FC_NPROF;
    if (r[15] == 0) continue;
    fprintf(stderr, "FCPROF slot=%d waves=%llu granules_per_wave=%.2f", sl, r[15], (double)r[14] / (double)r[15]);
    for (int i = 0; i < 18; ++i) if (i != 14 && i != 15) fprintf(stderr, " %s=%.0f", fc_prof_names[i], (double)r[i] / (double)r[15]);
    fprintf(stderr, "\n");
  }
  memset(h, 0, sizeof(h));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fc_prof), h, sizeof(h));
}
#else
struct FcProf {};
#define FC_STAMP(pf, i) ((void)0)
#define FC_STAMP_FINE(pf, i) ((void)0)
#define FC_COUNT(pf, i) ((void)0)
#define DDMI_PROF_FINE 0
#endif
#if defined(DDMI_PROFILING) && defined(DDMI_WG_STAMPS)
// Workgroup stamps (round 6; -DDDMI_PROFILING=1 -DDDMI_WG_STAMPS -- WITHOUT the phase clocks, whose registers spill and throttle the
// number of resident workgroups: profiles/r06_p1_wg_idle_*.txt were taken with them and are not usable): constant-rate clock (s_memrealtime, 100 MHz) at the start and end of every live workgroup of
// k_conv_fused; DDMI_WG_DUMP=<file> appends them at every report as binary records (t0, t1 | slot << 56 | blockIdx.y << 52).
// tools/wg_idle.py integrates the CU-idle time while at least one fused workgroup is running (one workgroup per CU: LDS).
constexpr unsigned FC_WG_CAP = 1u << 21;
static __device__ unsigned long long g_fc_wg[2 * FC_WG_CAP];
static __device__ unsigned g_fc_wg_n;
__device__ __forceinline__ void fc_wg_record(unsigned long long t0, int slot, int by) {
  const unsigned long long t1 = wall_clock64();
  const unsigned i = atomicAdd(&g_fc_wg_n, 1u);
  if (i < FC_WG_CAP) {
    g_fc_wg[2 * i] = t0;
    g_fc_wg[2 * i + 1] = (t1 & 0xfffffffffffffull) | ((unsigned long long)(slot & 15) << 56) | ((unsigned long long)(by & 15) << 52);
  }
}
static void fc_wg_dump() {
  const char* path = getenv("DDMI_WG_DUMP");
  unsigned n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_fc_wg_n), sizeof(n)) != hipSuccess) return;
  n = std::min(n, FC_WG_CAP);
  if (path && n > 0) {
    std::vector<unsigned long long> h(2 * (size_t)n);
    if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_fc_wg), h.size() * 8) == hipSuccess) {
      if (FILE* f = fopen(path, "ab")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
    }
    fprintf(stderr, "FCWG %u workgroup records -> %s\n", n, path);
  }
  n = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fc_wg_n), &n, sizeof(n));
}
#define FC_WG_START() const unsigned long long wg_t0 = wall_clock64()
#define FC_WG_END(slot, by) do { if (threadIdx.x == 0) fc_wg_record(wg_t0, (slot), (by)); } while (0)
#else
#define FC_WG_START() ((void)0)
#define FC_WG_END(slot, by) ((void)0)
static void fc_wg_dump() {}
#endif

typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float* p) {
  const vf4 v = DDMI_NT_LOAD(reinterpret_cast<const vf4*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}

// fragment j of a lane's chain inside the packed weights (layout: see nc_lane_off)
__device__ __forceinline__ int nc_fo(int j, int pstride) { return (j >> 2) * pstride + (j & 3); }

// Packed second-layer weights of one (k, path), per 16-w tile (u = 4j + lq, w = 16*tile + lr, lane = 16*lq + lr):
//   chains of whole 4-step pieces (mul_in % 16 == 0):  [piece j/4][lane 64][j % 4]  -- a wave's 16-B request per piece is one
//       contiguous KB (a per-lane run of 12 steps would put the lanes 48 B apart: 64 partial cache lines per request);
//   other chains:                                        [lane 64][step j]          -- steps = 3: contiguous 12-B runs.
// Fragment j of a lane sits at loff + (j >> 2) * pstride + (j & 3) with pstride = 256 / 4 respectively.
__host__ __device__ __forceinline__ int nc_pstride(int steps) { return (steps & 3) == 0 ? 256 : 4; }
__device__ __forceinline__ int nc_lane_off(const NcSlot& S, int w0, int lr, int lq) {
  const int steps = S.u_pad >> 2;
  return (w0 >> 4) * 64 * steps + (lq * 16 + lr) * ((steps & 3) == 0 ? 4 : steps);
}
// real spherical harmonics of a unit edge vector (component normalisation, e3nn order), l <= lmax
__device__ __forceinline__ void edge_sh(const float* n, float sgn, int lmax, float* sh) {
  const float x = sgn * n[0], y = sgn * n[1], z = sgn * n[2];
  sh[0] = 1.f;
  const float s3 = 1.7320508075688772f;
  sh[1] = s3 * x; sh[2] = s3 * y; sh[3] = s3 * z;
  if (lmax >= 2) {
    const float s5 = 2.23606797749979f;
    sh[4] = s5 * (s3 * x * z);
    sh[5] = s5 * (s3 * x * y);
    sh[6] = s5 * (y * y - 0.5f * (x * x + z * z));
    sh[7] = s5 * (s3 * y * z);
    sh[8] = s5 * ((s3 / 2) * (z * z - x * x));
  } else {
    sh[4] = sh[5] = sh[6] = sh[7] = sh[8] = 0.f;
  }
}

constexpr int FC_VN = 16, FC_KC = 8, FC_WAVES = 8, FC_CAP0 = 12, FC_CAPN = 4;
static_assert(FC_VN == 16, "k_vn_rows pads the per-edge rows to whole 16-node tiles");
#ifdef FCV_NOBAR   // timing-only: main loops without the per-chunk barrier (garbage results)
#define FC_STEP_BARRIER() ((void)0)
#else
#define FC_STEP_BARRIER() __syncthreads()
#endif
constexpr int FC_GWORDS = sizeof(FGran) / 4;   // granule descriptor, in 32-bit words
constexpr int FC_MAXG = 24;                    // granule descriptors kept in LDS per workgroup (launches split larger ranges)
// chunk buffer in LDS: [16 nodes][8 rows][16 * NBK columns] (NBK = column blocks of the widest granule of the launch: 4, or 5
// with a packed 7-slot granule); classic granules: column = 16*slot + w.  The padded strides keep the transposing stores
// (lanes = 16 w x 4 node quarters) and the B-fragment loads (lanes = 16 w x rows 2q + sub) on 64 distinct banks
// (YROW = 8 mod 16, YVN = 4 mod 8).
template <int NBK> struct FcDim {
  static constexpr int YROW = 16 * NBK + 8, YVN = FC_KC * YROW + 4, YB = FC_VN * YVN;
  static constexpr int BST = 16 * NBK + 12;   // row stride of the bias rows [16 nodes][16 * NBK] (kept in the coupling scratch during a main loop)
};

// k-invariant per-lane part of one slot chain: uniform weight base + 32-bit lane offset (scalar-base global loads)
struct FcSlotRt { const float* wb; const float* xp; int loff, xstride, steps, ps; };
__device__ __forceinline__ FcSlotRt fc_slot_setup(const NcSlot S, const float* __restrict__ wpack, const float* __restrict__ xbuf,
                                                  int w0, int lr, int lq, int xr) {   // xr: the lane's row of the x tile
  FcSlotRt R;
  R.steps = S.din == 0 ? 0 : (S.u_pad >> 2);
  R.wb = wpack + S.wk_off;
  R.loff = nc_lane_off(S, w0, lr, lq);
  R.ps = nc_pstride(S.u_pad >> 2);
  R.xp = xbuf + xr * NC_XS + S.x_off + lq * S.din + S.comp;
  R.xstride = 4 * S.din;
  return R;
}
// Weight fragments of one node-contraction item (4 slot chains), fetched one chunk ahead of their MFMAs.  Slot 0 is the
// longest chain of the granule (host-sorted).  Granules whose chain lengths match one of the static shapes
// (12,3,3,3) / (3,3,3,3) / (12,-,-,-) run fully unrolled code (a shorter chain of such a shape is a padding column: its
// fragments are finite and its result is never read); anything else takes the predicated generic path.
struct FcPre { float b0[FC_CAP0], b1[FC_CAPN], b2[FC_CAPN], b3[FC_CAPN]; };

template <int N>
__device__ __forceinline__ void fc_fetch_n(const FcSlotRt& R, size_t koff, float* bv) {
  const float* __restrict__ wb = R.wb + koff;
#pragma unroll
  for (int j = 0; j < N; ++j) bv[j] = (wb + nc_fo(j, R.ps))[R.loff];
}
template <int N>
__device__ __forceinline__ f32x4 fc_apply_n(const FcSlotRt& R, const float* bv) {
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* __restrict__ xp = R.xp;
#pragma unroll
  for (int j = 0; j < N; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xp[j * R.xstride], bv[j], acc, 0, 0, 0);
  return acc;
}
template <int CAP>
__device__ __forceinline__ void fc_fetch(const FcSlotRt& R, size_t koff, float* bv) {
  const float* __restrict__ wb = R.wb + koff;
#pragma unroll
  for (int j = 0; j < CAP; ++j)
    if (j < R.steps) bv[j] = (wb + nc_fo(j, R.ps))[R.loff];
}
template <int CAP>
__device__ __forceinline__ f32x4 fc_apply(const FcSlotRt& R, size_t koff, const float* bv) {
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* __restrict__ xp = R.xp;
#pragma unroll
  for (int j = 0; j < CAP; ++j)
    if (j < R.steps) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xp[j * R.xstride], bv[j], acc, 0, 0, 0);
  for (int j = CAP; j < R.steps; ++j)   // chains longer than the prefetch capacity (ns > 48)
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xp[j * R.xstride], (R.wb + koff + nc_fo(j, R.ps))[R.loff], acc, 0, 0, 0);
  return acc;
}
template <int N>
__device__ __forceinline__ void fc_direct_n(const FcSlotRt& R, size_t koff, int j0, f32x4& acc) {
  float bv[N], av[N];   // all fragments requested before the first MFMA: one L2 round trip per chain piece
#pragma unroll
  for (int j = 0; j < N; ++j) { bv[j] = (R.wb + koff + nc_fo(j0 + j, R.ps))[R.loff]; av[j] = R.xp[(j0 + j) * R.xstride]; }
#pragma unroll
  for (int j = 0; j < N; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc, 0, 0, 0);
}
__device__ __forceinline__ f32x4 fc_direct(const FcSlotRt& R, size_t koff) {   // un-prefetched chain (bias row)
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  int j = 0;
  for (; j + 12 <= R.steps; j += 12) fc_direct_n<12>(R, koff, j, acc);
  for (; j + 4 <= R.steps; j += 4) fc_direct_n<4>(R, koff, j, acc);
  if (R.steps - j == 3) fc_direct_n<3>(R, koff, j, acc);
  else if (R.steps - j == 2) fc_direct_n<2>(R, koff, j, acc);
  else if (R.steps - j == 1) fc_direct_n<1>(R, koff, j, acc);
  return acc;
}
// transposing store of one chain result: lane (lr, lq) holds nodes 4lq .. 4lq+3 of column (slot, w = lr)
template <int NBK>
__device__ __forceinline__ void fc_store(float* yw, int col, const f32x4& v) {
#pragma unroll
  for (int r = 0; r < 4; ++r) yw[r * FcDim<NBK>::YVN + col] = v[r];
}

// Message columns of 16 edge rows, staged row-major in LDS ([16][RS]), streamed to their message rows with V-float
// accesses (V = widest vector the column offset and the row length allow).
template <int V>
__device__ __forceinline__ void fc_store_rows(const float* __restrict__ stg, int RS, int L, int nrows, const float* __restrict__ erow,
                                              int ES, int ts_col, float* __restrict__ msg, int c0, int accumulate, int lane) {
  // lane = 4 * row + q: a lane serves ONE row (message row index read once, no index division) and the V-float pieces
  // q, q + 4, q + 8, ... of it; per request the 4 lanes of a row cover 4 * V consecutive floats.
  const int per_row = L / V;
  const int row = lane >> 2, q = lane & 3;
  if (row >= nrows) return;
  const int ts = reinterpret_cast<const int*>(erow)[row * ES + ts_col];
  float* __restrict__ prow = msg + (size_t)ts * XS + c0;
  const float* __restrict__ qrow = stg + row * RS;
  for (int cv = q; cv < per_row; cv += 4) {
    float* __restrict__ p = prow + V * cv;
    const float* __restrict__ qq = qrow + V * cv;
    if (V == 4) {
      float4 v = *reinterpret_cast<const float4*>(qq);
      if (accumulate) { const float4 o = *reinterpret_cast<const float4*>(p); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
      *reinterpret_cast<float4*>(p) = v;
    } else if (V == 2) {
      float2 v = *reinterpret_cast<const float2*>(qq);
      if (accumulate) { const float2 o = *reinterpret_cast<const float2*>(p); v.x += o.x; v.y += o.y; }
      *reinterpret_cast<float2*>(p) = v;
    } else {
      float v = qq[0];
      if (accumulate) v += p[0];
      p[0] = v;
    }
  }
}

// Uniform-base requests (buffer resource in SGPRs + 32-bit lane offset + SGPR offset): no per-request 64-bit address
// arithmetic in the vector ALU and no address registers -- the dense main loop issues one such request per MFMA shadow.
#ifdef DDMI_HIPEMU
struct FcBuf { const char* p; };
__device__ __forceinline__ FcBuf fc_buf(const void* p, unsigned) { return FcBuf{reinterpret_cast<const char*>(p)}; }
__device__ __forceinline__ float4 fc_buf_ld4(const FcBuf& b, unsigned voff, unsigned soff) {
  return *reinterpret_cast<const float4*>(b.p + voff + soff);
}
__device__ __forceinline__ float3 fc_buf_ld3(const FcBuf& b, unsigned voff, unsigned soff) {
  const float* q = reinterpret_cast<const float*>(b.p + voff + soff);
  return make_float3(q[0], q[1], q[2]);
}
__device__ __forceinline__ f32x4 fc_buf_ld4v(const FcBuf& b, unsigned voff, unsigned soff) {
  const float* q = reinterpret_cast<const float*>(b.p + voff + soff);
  return f32x4{q[0], q[1], q[2], q[3]};
}
#else
// (declared by name: the __builtin_amdgcn_raw_buffer_load_b128 of this toolchain is lowered to a one-dword load)
typedef int fc_i32x4 __attribute__((ext_vector_type(4)));
typedef float fc_f32x3 __attribute__((ext_vector_type(3)));
__device__ f32x4 fc_raw_buffer_load_x4(fc_i32x4 rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ fc_f32x3 fc_raw_buffer_load_x3(fc_i32x4 rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.v3f32");
struct FcBuf { fc_i32x4 r; };
__device__ __forceinline__ FcBuf fc_buf(const void* p, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(p);
  const int lo = DDMI_UNIFORM((int)(unsigned)a), hi = DDMI_UNIFORM((int)((a >> 32) & 0xffffu));
  return FcBuf{fc_i32x4{lo, hi, (int)bytes, 0x00020000}};   // base, stride 0, bytes, raw dwords (gfx9 family)
}
__device__ __forceinline__ float4 fc_buf_ld4(const FcBuf& b, unsigned voff, unsigned soff) {
  const f32x4 v = fc_raw_buffer_load_x4(b.r, (int)voff, (int)soff, 0);
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float3 fc_buf_ld3(const FcBuf& b, unsigned voff, unsigned soff) {
  const fc_f32x3 v = fc_raw_buffer_load_x3(b.r, (int)voff, (int)soff, 0);
  return make_float3(v[0], v[1], v[2]);
}
// the four dwords kept as ONE register tuple (see roll() of the BF loops)
__device__ __forceinline__ f32x4 fc_buf_ld4v(const FcBuf& b, unsigned voff, unsigned soff) { return fc_raw_buffer_load_x4(b.r, (int)voff, (int)soff, 0); }
#endif
template <int I, int N, class F>
__device__ __forceinline__ void fc_sfor(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); fc_sfor<I + 1, N>(f); }
}
// Effective (DUP, NLV) of the hand-scheduled loop variant a granule runs.  Combinations without an instance treat padding
// slots like live ones.
__device__ __forceinline__ void fc_variant(const FGran& G, int& vdup, int& vnlv) {
  vdup = 0; vnlv = 3;
  if (G.shape == 1) { if (G.dup == 1 && (G.nlive == 3 || G.nlive == 2)) { vdup = 1; vnlv = G.nlive; } }
  else if (G.shape == 2) {
    if (G.dup == 3) { vdup = 3; vnlv = (G.nlive == 2 || G.nlive == 1) ? G.nlive : 3; }
    else if (G.dup == 2) vdup = 2;
  }
}
template <int S0, int SN, int NLV = 3>
struct FcOrder {   // issue order of the slot chains: slot 0 alternating with the NLV live slots 1.. (dependent MFMAs 40 cycles apart)
  static constexpr int NC = S0 + NLV * SN, NJ = S0 > NLV * SN ? S0 : NLV * SN;
  static constexpr int find(int i, bool want_slot) {
    int c = 0;
    for (int j = 0; j < NJ; ++j) {
      if (j < S0) { if (c == i) return want_slot ? 0 : j; ++c; }
      if (j < NLV * SN) { if (c == i) return want_slot ? 1 + j % NLV : j / NLV; ++c; }
    }
    return 0;
  }
  static constexpr int slot(int i) { return find(i, true); }
  static constexpr int step(int i) { return find(i, false); }
};
// DUP: slots sharing one set of weight fragments (FGran::dup): they are requested once and feed several chains.
// NLV: live slots among 1..3 (FGran::nlive; padding slots trail): a padding slot is neither contracted nor multiplied.
// SH (shared-node tiles, ligand gather nodes with several virtual nodes each): the x tile holds the DISTINCT gather nodes of
// the 16 virtual nodes (slot dsl[vi] of this wave's virtual node vi; at most four -- tiles with more take the 16-row form) and the contraction runs on the 4x4x1
// MFMA (16 blocks of 4 nodes x 4 columns, block = 4 * channel-in-quad + column quad with the SAME weight fragments as the
// 16x16x4 form): a quarter of the matrix-core time per pass.  The four channel partials of a column sit in the four lane
// rows; a reduce-scatter by row / half swaps leaves node slot lq in lane row lq, stored with one request per chain.
// ywr: SH ? chunk row `wave` of node slot lq, column lr : of node 4lq (+r); yrd: SH ? without the node term : node 2*wave.
#ifndef FC_R0B_ALL
#define FC_R0B_ALL 1
#endif
// BF: edge product of a chunk as ONE v_mfma_f32_16x16x32_bf16 per (virtual node, row tile, column block) on split operands
// (ddmi_common.h, bf_mfma): the hidden rows arrive as packed words, the contracted chunk is stored as packed words.
template <int NBK, int S0, int SN, bool DENSE, int DUP = 0, int NLV = 3, bool SH = false, bool BF = false>
__device__ __forceinline__ void fc_mainloop_dense(f32x4 (&acc)[2][2][NBK], const FcSlotRt (&sl)[4], const float* __restrict__ wpack,
                                                  int KS, int HK, int NG8, int wave, int lane,
                                                  const float* __restrict__ hb_tile, const int (&vne)[2], float* ywr,
                                                  const float* yrd, FcProf& pf, float* brow, const int (&dsl)[2]) {
  constexpr int FC_YROW = FcDim<NBK>::YROW, FC_YVN = FcDim<NBK>::YVN, FC_YB = FcDim<NBK>::YB, BST = FcDim<NBK>::BST;
  // sparse rows (!DENSE): the second 16-row tile of a virtual node with <= 16 edges is neither fetched nor multiplied
  const bool two[2] = {DENSE || vne[0] > 16, DENSE || vne[1] > 16};
  using O = FcOrder<S0, SN, NLV>;
  constexpr int NC = O::NC;
  // SAMEX (SN == 12, merged granule): slots 1.. are further channel tiles of slot 0's path -- one set of x fragments
  constexpr bool SAMEX = SN == 12;
  constexpr int NXA = SAMEX ? S0 : NC;
  auto xi = [](int i) constexpr { return SAMEX ? O::step(i) : i; };
  float xa_[NXA];
  float bw[4][S0 > 3 ? S0 : 3];   // weight fragments [slot][step] (slot 0: S0 steps, slots 1..3: SN steps)
  float bw1[4][S0 > 3 ? S0 : 3];  // prologue only: chunk 1's fragments, requested together with chunk 0's
  fc_sfor<0, NC>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if constexpr (!SAMEX || O::slot(i) == 0) xa_[xi(i)] = sl[O::slot(i)].xp[O::step(i) * sl[O::slot(i)].xstride];
  });
#define xa(i) xa_[xi(i)]
  unsigned woff[4];            // uniform byte offset of row k = 8g + wave of each slot's packed weights
  unsigned lo[4];              // per-lane byte offset of the lane's run of fragments
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    woff[t] = (unsigned)DDMI_UNIFORM((int)(sl[t].wb - wpack) + wave * KS) * 4u;
    lo[t] = (unsigned)sl[t].loff * 4u;
  }
  const unsigned gstep = 32u * (unsigned)KS;
  const FcBuf wbuf = fc_buf(wpack, (unsigned)HK * (unsigned)KS * 4u);
  // vector requests: slot 0 in pieces of 4 steps (or one piece of 3), slots 1..3 one piece of 3 each
  static_assert(DUP != 2 || NLV == 3, "DUP 2 keeps slot 3's own fragments");
  constexpr int PSN = SN >= 4 ? SN / 4 : 1;                                              // request pieces of a slot 1..3
  constexpr int NLN = SN == 0 ? 0 : DUP == 1 ? 1 : DUP == 2 ? 1 : DUP == 3 ? 0 : NLV * PSN;   // requests for slots 1..3
  constexpr int NL0 = S0 >= 4 ? S0 / 4 : (S0 > 0 ? 1 : 0), NL = NL0 + NLN;
  static_assert(SN != 12 || DUP == 0, "merged granules have their own weights per slot");
  // slot whose fragments slot t multiplies with
  auto wsl = [](int t) constexpr { return DUP == 1 ? (t == 0 ? 0 : 1) : DUP == 2 ? (t == 3 ? 3 : 0) : DUP == 3 ? 0 : t; };
  static_assert(S0 % 4 == 0 || S0 == 3, "slot-0 chains are whole 4-step pieces or one 3-step piece");
  static_assert(SN == 0 || SN == 3 || SN == 12, "slots 1..3 hold 3-step chains (or 12-step chains of slot 0's path)");
  auto loadw_to = [&](auto ic, float (&B)[4][S0 > 3 ? S0 : 3], unsigned ahead) __attribute__((always_inline)) {   // ahead: uniform byte offset on top of the current chunk's
    constexpr int i = decltype(ic)::value;
    constexpr int t = i < NL0 ? 0 : (DUP == 2 ? 3 : 1 + (i - NL0) / PSN);   // DUP 2: the one extra request is slot 3's
    if constexpr (t == 0 && S0 >= 4) {
      const float4 v = fc_buf_ld4(wbuf, lo[0] + 1024u * i, woff[0] + ahead);   // piece i of the chain: one contiguous KB per wave
      B[0][4 * i] = v.x; B[0][4 * i + 1] = v.y; B[0][4 * i + 2] = v.z; B[0][4 * i + 3] = v.w;
    } else if constexpr (SN >= 4) {
      constexpr int pc = (i - NL0) % PSN;
      const float4 v = fc_buf_ld4(wbuf, lo[t] + 1024u * pc, woff[t] + ahead);
      B[t][4 * pc] = v.x; B[t][4 * pc + 1] = v.y; B[t][4 * pc + 2] = v.z; B[t][4 * pc + 3] = v.w;
    } else {
      const float3 v = fc_buf_ld3(wbuf, lo[t], woff[t] + ahead);
      B[t][0] = v.x; B[t][1] = v.y; B[t][2] = v.z;
    }
  };
  auto loadw = [&](auto ic) __attribute__((always_inline)) { loadw_to(ic, bw, 0u); };
  constexpr bool FC_R0B = SH || FC_R0B_ALL;
  auto cmma = [](float av, float bv, f32x4 c) __attribute__((always_inline)) -> f32x4 {   // contraction MFMA
#ifdef FCV_NOCMMA   // timing-only: no contraction MFMAs (the weight requests stay alive)
    c[0] += bv; (void)av;
    if (true) return c;
#endif
    if constexpr (SH) return __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c, 0, 0, 0);
  };
  auto rsc = [](const f32x4& v) __attribute__((always_inline)) {   // SH: lane row lq <- node slot lq summed over the four lane rows
    float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
    DDMI_SWAP16(a0, a1, 1);   // (operands: copies of MFMA results that are at least two MFMAs old)
    DDMI_SWAP16(a2, a3, 1);
    float x = a0 + a1, y = a2 + a3;
    DDMI_SWAP32(x, y, 1);
    return x + y;
  };
  const int ynoff[2] = {SH ? dsl[0] * FC_YVN : 0, SH ? dsl[1] * FC_YVN : FC_YVN};   // chunk rows of this wave's two virtual nodes
  float4 hC[2][2], hN[2][2];   // hidden-row fragments of the current / next PAIR of chunks: [virtual node][row tile]
#pragma unroll
  for (int pc = 0; pc < 4; ++pc) hC[pc >> 1][pc & 1] = hN[pc >> 1][pc & 1] = make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned rts = (unsigned)(NG8 >> 1) * 1024u;                     // bytes per (virtual node, row tile)
  const FcBuf hbuf = fc_buf(hb_tile, (unsigned)FC_VN * 2u * rts);
  unsigned hoff = (unsigned)(2 * wave) * 2u * rts;                       // uniform: this wave's two virtual nodes, pair 0
  const unsigned hlane = (unsigned)lane * 16u;
  f32x4 hNv[2][2];             // BF: the next pair's words as they arrive (whole register tuples until roll())
#pragma unroll
  for (int pc = 0; pc < 4; ++pc) hNv[pc >> 1][pc & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto loadh = [&](float4 (&dst)[2][2], int piece) __attribute__((always_inline)) {
    if (DENSE || !(piece & 1) || two[piece >> 1]) {
      if (BF && &dst == &hN) hNv[piece >> 1][piece & 1] = fc_buf_ld4v(hbuf, hlane, hoff + (unsigned)piece * rts);
      else dst[piece >> 1][piece & 1] = fc_buf_ld4(hbuf, hlane, hoff + (unsigned)piece * rts);
    }
  };
  f32x4 r[4];
  // live 16-column blocks of the granule: a (12,-,-,-) granule (one item column) multiplies only block 0 in the edge product
  constexpr int NCB = SN == 0 ? 1 : 1 + NLV, NE = 8 * NCB, NP = SH ? NCB : (SN == 0 || NLV == 1) ? 4 : 8;
  constexpr int NEB = 4 * NCB;   // BF: edge-product MFMAs of a step
  float q[2][BF ? 2 * NCB : NCB];   // B fragments of the edge product: [parity of the (virtual node, k pair) group][column block]; BF: [virtual node][k0 | k1][column block]
  auto readq = [&](int par, int buf, int grp) __attribute__((always_inline)) {
    if constexpr (BF) {   // grp = virtual node: both k rows of the lane group
      const float* __restrict__ yb = yrd + buf * FC_YB + ynoff[grp];
#pragma unroll
      for (int c = 0; c < NCB; ++c) { q[par][c] = yb[16 * c]; q[par][NCB + c] = yb[FC_YROW + 16 * c]; }
    } else {
      const float* __restrict__ yb = yrd + buf * FC_YB + ynoff[grp >> 1] + (grp & 1) * FC_YROW;
#ifdef FCV_NOQ
#pragma unroll
      for (int c = 0; c < NCB; ++c) { q[par][c] = (float)(buf + grp); DDMI_OPAQUE(q[par][c]); }
      if (true) return;
#endif
#pragma unroll
      for (int c = 0; c < NCB; ++c) q[par][c] = yb[16 * c];
    }
  };
  auto store_piece = [&](int buf, int piece) __attribute__((always_inline)) {   // rows of node quarter `rr`, slots 2h and 2h+1
    float* yw = ywr + buf * FC_YB;
#ifdef FCV_NOYST
    if constexpr (SH) { float keep = r[piece][0]; DDMI_OPAQUE(keep); }
    else { float k0 = r[2 * (piece >> 2) < 4 ? 2 * (piece >> 2) : 0][piece & 3]; DDMI_OPAQUE(k0); }
    if (true) return;
#endif
    if constexpr (SH) {   // one chain per piece
      const float v = rsc(r[piece]);
      yw[16 * piece] = BF ? bf_split1(v) : v;
    } else {
      const int rr = piece & 3, h = piece >> 2;
      float v0 = 0.f, v1 = 0.f;
      if (2 * h < NCB) v0 = r[2 * h][rr];
      if (2 * h + 1 < NCB) v1 = r[2 * h + 1][rr];
      if constexpr (BF) {
        if (2 * h + 1 < NCB) bf_split2(v0, v1, v0, v1);
        else if (2 * h < NCB) v0 = bf_split1(v0);
      }
      if (2 * h < NCB) yw[rr * FC_YVN + 16 * (2 * h)] = v0;
      if (2 * h + 1 < NCB) yw[rr * FC_YVN + 16 * (2 * h + 1)] = v1;
    }
  };
  static_assert(NCB <= NBK, "column blocks of the granule exceed the chunk buffer");
  // one chunk step (chunk g = 2 * pair + ODD): contraction of chunk g+1 into buffer ODD ^ 1 (DO_C), weight requests for
  // chunk g+2 (DO_W), hidden rows of the next pair of chunks (DO_H), edge product of chunk g out of buffer ODD
  auto roll = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int vi = 0; vi < 2; ++vi)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        // BF: the A tuples {w, w, w', w'} are register copies of these words; unless the words pass through an opaque point HERE
        // the compiler places those copies right behind the requests of hN (an s_waitcnt vmcnt(0) per request in the middle of
        // the edge product, measured: 10 % slower than the f32 route)
        if constexpr (BF) {
          DDMI_OPAQUE(hNv[vi][rt]);   // (the request's own register tuple: the wait for it sits here, the copies behind it)
          hC[vi][rt] = make_float4(hNv[vi][rt][0], hNv[vi][rt][1], hNv[vi][rt][2], hNv[vi][rt][3]);
        } else {
          hC[vi][rt] = hN[vi][rt];
        }
      }
  };
  // (do_roll: the hidden rows requested during the previous pair become the current ones BEHIND this step's contraction --
  // the copy is where the compiler waits for those slow requests, and the contraction does not need them)
  auto step = [&](auto do_c, auto do_w, auto do_h, auto odd, auto do_roll) __attribute__((always_inline)) {
    constexpr bool DO_C = decltype(do_c)::value, DO_W = decltype(do_w)::value, DO_H = decltype(do_h)::value;
    constexpr int ODD = decltype(odd)::value, eb = ODD, cb = ODD ^ 1;
    if constexpr (DO_C) {
#pragma unroll
      for (int t = 0; t < 4; ++t) r[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 r0b = f32x4{0.f, 0.f, 0.f, 0.f};   // second accumulator of the long chain: its last three steps are consecutive, and a dependent
                                               // MFMA waits 40 (4x4x1) / 44-64 (16x16x4, one / two waves per SIMD) cycles instead of 12 / 33
      fc_sfor<0, NC>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int t = O::slot(i);
        if constexpr (FC_R0B && t == 0 && S0 > 3 && (O::step(i) & 1)) r0b = cmma(xa(i), bw[wsl(t)][O::step(i)], r0b);
        else r[t] = cmma(xa(i), bw[wsl(t)][O::step(i)], r[t]);
        if (i == NC - 3) readq(0, eb, 0);
        DDMI_SCHED_FENCE();
      });
      if constexpr (FC_R0B && S0 > 3) r[0] += r0b;
    } else {
      readq(0, eb, 0);
    }
    if constexpr (decltype(do_roll)::value) { roll(); DDMI_SCHED_FENCE(); }
    if constexpr (BF) {
      // slot m = (virtual node vi, column block c, row tile rt): the B tuple of (vi, c) serves both row tiles.  Side work per
      // slot: weight request m (slots 0 .. NL-1), the four hidden-row requests behind them, the row stores from slot 1 on
      fc_sfor<0, NEB>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        constexpr int vi = m / (2 * NCB), t8 = m % (2 * NCB), c = t8 >> 1, rt = t8 & 1;
        if (DENSE || rt == 0 || two[vi]) {
          const float a0 = ODD ? hC[vi][rt].z : hC[vi][rt].x, a1 = ODD ? hC[vi][rt].w : hC[vi][rt].y;
          acc[vi][rt][c] = bf_mfma(a0, a1, q[vi][c], q[vi][NCB + c], acc[vi][rt][c]);
        }
        if constexpr (DO_W) fc_sfor<0, NL>([&](auto ic) { if constexpr ((decltype(ic)::value < NEB ? decltype(ic)::value : NEB - 1) == m) loadw(ic); });
        if constexpr (DO_H) {   // (after this step's weight requests: vmcnt retires in order)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (m == (NL + j < NEB - 1 ? NL + j : NEB - 1)) loadh(hN, j);
        }
        if constexpr (DO_C) {
#pragma unroll
          for (int pc = 0; pc < NP; ++pc)
            if (m == (1 + pc < NEB - 1 ? 1 + pc : NEB - 1)) store_piece(cb, pc);
        }
        if (m == 0) readq(1, eb, 1);
        DDMI_SCHED_FENCE();
      });
    } else {
    static_assert(NE >= 2 * NL && NE >= NP + 2, "edge-product slots for the weight requests and the row stores");
    fc_sfor<0, NE>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      constexpr int grp = m / (2 * NCB), t8 = m % (2 * NCB), vi = grp >> 1, sub = grp & 1, rt = t8 / NCB, c = t8 % NCB;
      if (DENSE || rt == 0 || two[vi]) {
        const float av = ODD ? (sub == 0 ? hC[vi][rt].z : hC[vi][rt].w) : (sub == 0 ? hC[vi][rt].x : hC[vi][rt].y);
#ifdef FCV_NOEMMA   // timing-only: no edge-product MFMAs (upper bound of what a cheaper edge product of low-degree groups can buy)
        acc[vi][rt][c][0] += av * q[grp & 1][c];
#else
        acc[vi][rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, q[grp & 1][c], acc[vi][rt][c], 0, 0, 0);
#endif
      }
      if constexpr (DO_W) { if constexpr (m % 2 == 0 && m / 2 < NL) loadw(std::integral_constant<int, m / 2>{}); }
      // The hidden rows of the next pair of chunks (HBM / Infinity Cache, slow) are requested AFTER this step's weight
      // requests (L2, needed at the next contraction): vmcnt retires in order on this family, so a slow request issued
      // ahead of the weights would be waited for together with them, one contraction too early.
      if constexpr (DO_H) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (m == (2 * NL - 1 + 2 * j < NE - 1 ? 2 * NL - 1 + 2 * j : NE - 1)) loadh(hN, j);
      }
      if constexpr (DO_C) { if (m >= 2 && m < 2 + NP) store_piece(cb, m - 2); }
      if (t8 == 1 && grp < 3) readq((grp + 1) & 1, eb, grp + 1);
      DDMI_SCHED_FENCE();
    });
    }
#ifndef FCV_SAMEW
    if constexpr (DO_W) {
#pragma unroll
      for (int t = 0; t < 4; ++t) woff[t] += gstep;
    }
#endif
#ifndef FCV_SAMEH
    if constexpr (DO_H) hoff += 1024u;
#endif
  };
  using T = std::true_type;
  using F = std::false_type;
  using Even = std::integral_constant<int, 0>;
  using Odd = std::integral_constant<int, 1>;
  // prologue: chunk 0 contracted, chunk 1 requested  (NG8 is even and >= 2 here: the host selects this loop for H % 16 == 0)
  // Round 5: the requests go out as [weights of chunk 0][bias row][weights of chunk 1][hidden rows of pair 0] -- vector memory
  // returns in order, so the contraction of chunk 0 waits for the first two only, and the first in-loop contraction finds its
  // weights in registers (they used to be requested BEHIND the contraction of chunk 0: one more exposed round trip per granule).
  fc_sfor<0, NL>(loadw);
  // The bias row of the packed second layer (k = H, hidden value 1 for every edge) rides along with the prologue: wave t
  // contracts slot t with it and leaves the 16 x 16 result in the bias rows; after the barrier every accumulator STARTS from
  // its node's bias value.  (No separate phase behind the main loop: that one exposed an L2 round trip, a dependent MFMA chain
  // and two barriers per granule.)
  constexpr int NSLOT = SN == 0 ? 1 : 1 + NLV;                       // live slots
  int lane_b = lane;
  DDMI_OPAQUE(lane_b);                                               // (addresses of the bias rows are not hoisted out of the granule loop)
  float bb[S0 > 3 ? S0 : 3];
  const unsigned hrow = (unsigned)(HK - 1) * (unsigned)KS * 4u - (unsigned)wave * (unsigned)KS * 4u;   // row H instead of row `wave`
  fc_sfor<0, NSLOT>([&](auto tc) {
    constexpr int t = decltype(tc)::value, ws = wsl(t);
    if (wave == t) {
      if constexpr ((ws == 0 && S0 >= 4) || (ws > 0 && SN >= 4)) {
#pragma unroll
        for (int i = 0; i < (ws == 0 ? S0 : SN) / 4; ++i) {
          const float4 v = fc_buf_ld4(wbuf, lo[ws] + 1024u * i, woff[ws] + hrow);
          bb[4 * i] = v.x; bb[4 * i + 1] = v.y; bb[4 * i + 2] = v.z; bb[4 * i + 3] = v.w;
        }
      } else {
        const float3 v = fc_buf_ld3(wbuf, lo[ws], woff[ws] + hrow);
        bb[0] = v.x; bb[1] = v.y; bb[2] = v.z;
      }
    }
  });
  fc_sfor<0, NL>([&](auto ic) { loadw_to(ic, bw1, gstep); });
#pragma unroll
  for (int pc = 0; pc < 4; ++pc) loadh(hC, pc);
#pragma unroll
  for (int t = 0; t < 4; ++t) woff[t] += gstep;
  hoff += 1024u;
  fc_sfor<0, NSLOT>([&](auto tc) {   // bias chains (two accumulators for the long one: a dependent f32 MFMA waits 40 cycles)
    constexpr int t = decltype(tc)::value, LEN = t == 0 ? S0 : SN;
    if (wave == t) {
      f32x4 ba = f32x4{0.f, 0.f, 0.f, 0.f}, bq = f32x4{0.f, 0.f, 0.f, 0.f};
      fc_sfor<0, NC>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (O::slot(i) == t) {
          constexpr int j = O::step(i);
          if constexpr (LEN > 3 && (j & 1)) bq = cmma(xa(i), bb[j], bq);
          else ba = cmma(xa(i), bb[j], ba);
        }
      });
      if constexpr (SH) {
        brow[(lane_b >> 4) * BST + 16 * t + (lane_b & 15)] = rsc(ba + bq);
      } else {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) brow[(4 * (lane_b >> 4) + rr) * BST + 16 * t + (lane_b & 15)] = ba[rr] + bq[rr];
      }
    }
  });
#pragma unroll
  for (int t = 0; t < 4; ++t) r[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    f32x4 r0b = f32x4{0.f, 0.f, 0.f, 0.f};
    fc_sfor<0, NC>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int t = O::slot(i);
      if constexpr (FC_R0B && t == 0 && S0 > 3 && (O::step(i) & 1)) r0b = cmma(xa(i), bw[wsl(t)][O::step(i)], r0b);
      else r[t] = cmma(xa(i), bw[wsl(t)][O::step(i)], r[t]);
    });
    if constexpr (FC_R0B && S0 > 3) r[0] += r0b;
  }
#pragma unroll
  for (int pc = 0; pc < NP; ++pc) store_piece(0, pc);
  FC_STAMP(pf, 12);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int j = 0; j < (S0 > 3 ? S0 : 3); ++j) bw[t][j] = bw1[t][j];   // chunk 1's fragments (entries no request wrote are never read)
#ifdef DDMI_PHASE_CLOCKS
#pragma unroll
  for (int t = 0; t < 4; ++t) DDMI_OPAQUE(bw[t][0]);   // (the clock below then includes the wait for these requests)
  FC_STAMP(pf, 13);
#endif
#pragma unroll
  for (int t = 0; t < 4; ++t) woff[t] += gstep;
  __syncthreads();
#pragma unroll
  for (int vi = 0; vi < 2; ++vi)
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
      const float b = brow[(SH ? dsl[vi] : 2 * wave + vi) * BST + 16 * c + (lane_b & 15)];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) acc[vi][rt][c][rr] += b;
    }
  FC_STAMP(pf, 2);
  // first pair (hidden rows of pair 0 came with the prologue), pairs with a successor pair, last pair
  step(T{}, T{}, F{}, Even{}, F{});
  FC_STEP_BARRIER();
  step(T{}, T{}, T{}, Odd{}, F{});
  FC_STEP_BARRIER();
  for (int g = 2; g + 2 < NG8; g += 2) {
    step(T{}, T{}, F{}, Even{}, T{});
    FC_STAMP_FINE(pf, 3);
    FC_STEP_BARRIER();
    FC_STAMP_FINE(pf, 11);
    step(T{}, T{}, T{}, Odd{}, F{});
    FC_STAMP_FINE(pf, 3);
    FC_STEP_BARRIER();
    FC_STAMP_FINE(pf, 11);
  }
  step(T{}, F{}, F{}, Even{}, T{});        // last pair: one contraction left, nothing to request
  FC_STAMP_FINE(pf, 3);
  FC_STEP_BARRIER();
  FC_STAMP_FINE(pf, 11);
  step(F{}, F{}, F{}, Odd{}, F{});
  FC_STAMP_FINE(pf, 3);
  FC_STEP_BARRIER();
  FC_STAMP(pf, 3 + 8 * (DDMI_PROF_FINE));
}

#undef xa

// ---- packed granules (kernels.h, FGran): slots = [a 12-step chain (S0 = 12) or none (S0 = 0)] + NG groups of three 3-step
// chains, the three sharing one set of weight fragments (components of one vector path).  Column of (slot s, channel w):
// w < 8: 8*s + w (two slots per 16-column block), w = 8, 9: 16*(NB - 1) + 2*s + w - 8 (tail block), NB = ceil(NS / 2) + 1.
// k-invariant per-lane part of a packed granule: x fragments of the long chain at xp0[4 * step], of group g, component i
// at xg[g][i + 12 * step]; packed weights of request source t (0 = long chain, 1 + g = group g) at wk[t] (uniform) + loff[t]
struct FcPackRt { const float* xp0; const float* xg[2]; int wk[3], loff[3]; };
__device__ __forceinline__ FcPackRt fc_pack_setup(const FGran& G, const float* __restrict__ xbuf, int lr, int lq, int xr) {
  FcPackRt P;
  const int c0 = G.shape == 5 ? 0 : 1, ng = G.shape == 6 ? 1 : 2;
  const NcSlot& S0_ = G.slot[0];
  P.xp0 = xbuf + xr * NC_XS + S0_.x_off + lq;
  P.wk[0] = S0_.wk_off; P.loff[0] = nc_lane_off(S0_, 0, lr, lq);
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const NcSlot& S = G.slot[g < ng ? c0 + 3 * g : 0];
    P.xg[g] = xbuf + xr * NC_XS + S.x_off + 3 * lq;
    P.wk[1 + g] = S.wk_off; P.loff[1 + g] = nc_lane_off(S, 0, lr, lq);
  }
  return P;
}
template <int S0, int NG>
struct FcPackOrder {   // issue order of the contraction: the 12-step chain spread evenly between the short chains, which run
                       // group after group (step-major inside a group: dependent MFMAs >= 3 apart), so that a group's results
                       // can leave for the chunk buffer while the next group is still being contracted
  static constexpr int N3 = 9 * NG, NC = S0 + N3, C0 = S0 > 0 ? 1 : 0;
  static constexpr bool is_c0(int i) {   // position i carries a step of the long chain
    if (S0 == 0) return false;
    for (int k = 0; k < S0; ++k) if (k * NC / S0 == i) return true;
    return false;
  }
  static constexpr int rank(int i, bool c0) { int n = 0; for (int j = 0; j < i; ++j) n += is_c0(j) == c0; return n; }
  static constexpr int slot(int i) { return is_c0(i) ? 0 : C0 + 3 * (rank(i, false) / 9) + rank(i, false) % 3; }
  static constexpr int step(int i) { return is_c0(i) ? rank(i, true) : (rank(i, false) % 9) / 3; }
  static constexpr int last_pos(int t) { int p = 0; for (int i = 0; i < NC; ++i) if (slot(i) == t) p = i; return p; }   // position of slot t's last step
};
template <int NBK, int S0, int NG, bool DENSE, bool BF = false>
__device__ __forceinline__ void fc_mainloop_packed(f32x4 (&acc)[2][2][NBK], const FcPackRt& P, const float* __restrict__ wpack,
                                                   int KS, int HK, int NG8, int wave, int lane,
                                                   const float* __restrict__ hb_tile, const int (&vne)[2], float* ywr0,
                                                   const float* yrd, FcProf& pf, float* brow) {
  constexpr int FC_YROW = FcDim<NBK>::YROW, FC_YVN = FcDim<NBK>::YVN, FC_YB = FcDim<NBK>::YB, BST = FcDim<NBK>::BST;
  using O = FcPackOrder<S0, NG>;
  constexpr int C0 = O::C0, NS = C0 + 3 * NG, NB = (NS + 1) / 2 + 1, NC = O::NC;
  static_assert(NB <= NBK, "column blocks of the granule exceed the chunk buffer");
  static_assert(S0 == 0 || S0 == 12, "the long chain has 12 steps");
  const bool two[2] = {DENSE || vne[0] > 16, DENSE || vne[1] > 16};
  const int lr = lane & 15;
  // column of this lane's channel in slot s: cb + cs * s (channels >= 10 do not exist: those lanes store into the row padding)
  const int cb = lr < 8 ? lr : lr < 10 ? 16 * (NB - 1) + lr - 8 : 16 * NBK + lr - 10;
  const int cs = lr < 8 ? 8 : lr < 10 ? 2 : 0;
  float* const ywr = ywr0 + cb;
  // x fragments are re-read from the (read-only) x tile in LDS a few MFMAs ahead of their use -- 30 registers less than
  // keeping them, for one LDS read per contraction MFMA
  constexpr int XW = 5;          // read-ahead distance of the window, in contraction positions
  float xw[NC];
#ifdef FCV_NOXW
  float fcv_x0 = P.xp0[0];
#endif
  auto xread = [&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    if constexpr (i < NC) {
#ifdef FCV_NOXW   // timing-only: no x-fragment reads inside the loop (one register, kept opaque)
      xw[i] = fcv_x0; DDMI_OPAQUE(xw[i]);
      if (true) return;
#endif
      if constexpr (O::is_c0(i)) {
        xw[i] = P.xp0[4 * O::step(i)];                                       // scalar input: u = 4 * step + lane / 16
      } else {
        constexpr int t = O::slot(i) - C0;                                   // component t % 3 of group t / 3, vector input (3 floats per u)
        xw[i] = P.xg[t / 3][t % 3 + 12 * O::step(i)];
      }
    }
  };
  float bw0[S0 > 0 ? S0 : 1], bwg[NG][3];   // weight fragments: the long chain, one 3-step set per group
  float bw0n[S0 > 0 ? S0 : 1], bwgn[NG][3]; // prologue only: chunk 1's fragments, requested together with chunk 0's (fc_mainloop_dense)
  unsigned woff[1 + NG], lo[1 + NG];        // uniform byte offset of row k = 8g + wave / per-lane byte offset, per request source
#pragma unroll
  for (int t = 0; t < 1 + NG; ++t) {
    woff[t] = (unsigned)DDMI_UNIFORM(P.wk[t] + wave * KS) * 4u;
    lo[t] = (unsigned)P.loff[t] * 4u;
  }
  const unsigned gstep = 32u * (unsigned)KS;
  const FcBuf wbuf = fc_buf(wpack, (unsigned)HK * (unsigned)KS * 4u);
  constexpr int NL0 = S0 / 4, NL = NL0 + NG;
  auto loadw_to = [&](auto ic, float (&B0)[S0 > 0 ? S0 : 1], float (&BG)[NG][3], unsigned ahead) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    if constexpr (i < NL0) {
      const float4 v = fc_buf_ld4(wbuf, lo[0] + 1024u * i, woff[0] + ahead);
      B0[4 * i] = v.x; B0[4 * i + 1] = v.y; B0[4 * i + 2] = v.z; B0[4 * i + 3] = v.w;
    } else {
      constexpr int g = i - NL0;
      const float3 v = fc_buf_ld3(wbuf, lo[1 + g], woff[1 + g] + ahead);
      BG[g][0] = v.x; BG[g][1] = v.y; BG[g][2] = v.z;
    }
  };
  auto loadw = [&](auto ic) __attribute__((always_inline)) { loadw_to(ic, bw0, bwg, 0u); };
  float4 hC[2][2], hN[2][2];
#pragma unroll
  for (int pc = 0; pc < 4; ++pc) hC[pc >> 1][pc & 1] = hN[pc >> 1][pc & 1] = make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned rts = (unsigned)(NG8 >> 1) * 1024u;
  const FcBuf hbuf = fc_buf(hb_tile, (unsigned)FC_VN * 2u * rts);
  unsigned hoff = (unsigned)(2 * wave) * 2u * rts;
  const unsigned hlane = (unsigned)lane * 16u;
  f32x4 hNv[2][2];             // BF: the next pair's words as they arrive (whole register tuples until roll())
#pragma unroll
  for (int pc = 0; pc < 4; ++pc) hNv[pc >> 1][pc & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto loadh = [&](float4 (&dst)[2][2], int piece) __attribute__((always_inline)) {
    if (DENSE || !(piece & 1) || two[piece >> 1]) {
      if (BF && &dst == &hN) hNv[piece >> 1][piece & 1] = fc_buf_ld4v(hbuf, hlane, hoff + (unsigned)piece * rts);
      else dst[piece >> 1][piece & 1] = fc_buf_ld4(hbuf, hlane, hoff + (unsigned)piece * rts);
    }
  };
  f32x4 r[NS];
  // results leave for the chunk buffer as soon as their chain is complete: the slots finished >= 12 positions before the
  // end of the contraction (first group when there are two) during the contraction itself, the rest during the edge product
  constexpr int NEARLY = (NG == 2) ? 3 : 0;                      // slots C0 .. C0 + 2 (group 0)
  constexpr int EARLY0 = NEARLY ? O::last_pos(C0 + 2) + 2 : NC;  // first contraction position that carries an early store
  constexpr int EPP = NEARLY ? (4 * NEARLY + (NC - EARLY0) - 1) / (NC - EARLY0) : 1;   // early stores per contraction position
  static_assert(NEARLY == 0 || (EARLY0 < NC && EPP <= 2), "early stores fit behind their chains");
  constexpr int NE = 8 * NB, NP = 4 * (NS - NEARLY);
  constexpr int NEB = 4 * NB, SPS = (NP + NEB - 3) / (NEB - 2);   // BF: edge-product MFMAs of a step, late stores per slot
  static_assert(BF || (NE >= 2 * NL + 8 && NE >= NP + 2), "edge-product slots for the requests and the row stores");
  static_assert(NEB >= NL + 5 && NEB >= XW, "BF: edge-product slots for the requests and the x window");
  float q[2][BF ? 2 * NB : NB];   // (BF: [virtual node][k0 | k1][column block], see fc_mainloop_dense)
  auto readq = [&](int par, int buf, int grp) __attribute__((always_inline)) {
    if constexpr (BF) {
      const float* __restrict__ yb = yrd + buf * FC_YB + grp * FC_YVN;
#pragma unroll
      for (int c = 0; c < NB; ++c) { q[par][c] = yb[16 * c]; q[par][NB + c] = yb[FC_YROW + 16 * c]; }
    } else {
      const float* __restrict__ yb = yrd + buf * FC_YB + (grp >> 1) * FC_YVN + (grp & 1) * FC_YROW;
#ifdef FCV_NOQ    // timing-only: no chunk reads for the edge product
#pragma unroll
      for (int c = 0; c < NB; ++c) { q[par][c] = (float)(buf + grp); DDMI_OPAQUE(q[par][c]); }
      if (true) return;
#endif
#pragma unroll
      for (int c = 0; c < NB; ++c) q[par][c] = yb[16 * c];
    }
  };
  auto store_slot = [&](int buf, int s_, int rr) __attribute__((always_inline)) {   // node quarter rr of slot s_
#ifdef FCV_NOYST  // timing-only: contraction results are not stored to the chunk buffer
    { float keep = r[s_][rr]; DDMI_OPAQUE(keep); }
    if (true) return;
#endif
    ywr[buf * FC_YB + rr * FC_YVN + cs * s_] = BF ? bf_split1(r[s_][rr]) : r[s_][rr];
  };
  // late stores: slot 0 (long chain) and the slots behind the early ones
  auto late_slot = [](int piece) constexpr { const int k = piece >> 2; return (C0 && k == 0) ? 0 : C0 + NEARLY + (k - C0); };
  auto contract = [&](auto ic, int buf, bool stores) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    constexpr int t = O::slot(i), j = O::step(i);
    xread(std::integral_constant<int, i + XW>{});
    float b, x;
    x = xw[i];
    if constexpr (C0 == 1 && t == 0) b = bw0[j]; else b = bwg[(t - C0) / 3][j];
    r[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, b, r[t], 0, 0, 0);
    if constexpr (NEARLY > 0 && i >= EARLY0) {
      if (stores) {
#pragma unroll
        for (int e = EPP * (i - EARLY0); e < EPP * (i - EARLY0 + 1); ++e)
          if (e < 4 * NEARLY) store_slot(buf, C0 + e / 4, e % 4);
      }
    }
  };
  auto roll = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int vi = 0; vi < 2; ++vi)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        // BF: the A tuples {w, w, w', w'} are register copies of these words; unless the words pass through an opaque point HERE
        // the compiler places those copies right behind the requests of hN (an s_waitcnt vmcnt(0) per request in the middle of
        // the edge product, measured: 10 % slower than the f32 route)
        if constexpr (BF) {
          DDMI_OPAQUE(hNv[vi][rt]);   // (the request's own register tuple: the wait for it sits here, the copies behind it)
          hC[vi][rt] = make_float4(hNv[vi][rt][0], hNv[vi][rt][1], hNv[vi][rt][2], hNv[vi][rt][3]);
        } else {
          hC[vi][rt] = hN[vi][rt];
        }
      }
  };
  auto step = [&](auto do_c, auto do_w, auto do_h, auto odd, auto do_roll) __attribute__((always_inline)) {
    constexpr bool DO_C = decltype(do_c)::value, DO_W = decltype(do_w)::value, DO_H = decltype(do_h)::value;
    constexpr int ODD = decltype(odd)::value, eb = ODD, cb_ = ODD ^ 1;
    if constexpr (DO_C) {
#pragma unroll
      for (int t = 0; t < NS; ++t) r[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      fc_sfor<0, NC>([&](auto ic) {
        contract(ic, cb_, true);
        if (decltype(ic)::value == NC - 3) readq(0, eb, 0);
        DDMI_SCHED_FENCE();
      });
    } else {
      readq(0, eb, 0);
    }
    if constexpr (decltype(do_roll)::value) { roll(); DDMI_SCHED_FENCE(); }   // (see fc_mainloop_dense)
    if constexpr (BF) {
      fc_sfor<0, NEB>([&](auto mc) {   // slot m = (virtual node, column block, row tile), see fc_mainloop_dense
        constexpr int m = decltype(mc)::value;
        constexpr int vi = m / (2 * NB), t8 = m % (2 * NB), c = t8 >> 1, rt = t8 & 1;
        if (DENSE || rt == 0 || two[vi]) {
          const float a0 = ODD ? hC[vi][rt].z : hC[vi][rt].x, a1 = ODD ? hC[vi][rt].w : hC[vi][rt].y;
          acc[vi][rt][c] = bf_mfma(a0, a1, q[vi][c], q[vi][NB + c], acc[vi][rt][c]);
        }
        if constexpr (DO_W) { if constexpr (m < NL) loadw(std::integral_constant<int, m>{}); }
        if constexpr (DO_H) {   // (after this step's weight requests: vmcnt retires in order)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (m == NL + j) loadh(hN, j);
        }
        if constexpr (DO_C) {
#pragma unroll
          for (int pc = 0; pc < NP; ++pc)
            if (m == 2 + pc / SPS) store_slot(cb_, late_slot(pc), pc & 3);
        }
        if constexpr (DO_C && m >= NEB - XW) xread(std::integral_constant<int, m - (NEB - XW)>{});   // window of the next contraction
        if (m == 0) readq(1, eb, 1);
        DDMI_SCHED_FENCE();
      });
    } else {
    fc_sfor<0, NE>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      constexpr int grp = m / (2 * NB), t8 = m % (2 * NB), vi = grp >> 1, sub = grp & 1, rt = t8 / NB, c = t8 % NB;
      if (DENSE || rt == 0 || two[vi]) {
        const float av = ODD ? (sub == 0 ? hC[vi][rt].z : hC[vi][rt].w) : (sub == 0 ? hC[vi][rt].x : hC[vi][rt].y);
#ifdef FCV_NOEMMA
        acc[vi][rt][c][0] += av * q[grp & 1][c];
#else
        acc[vi][rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, q[grp & 1][c], acc[vi][rt][c], 0, 0, 0);
#endif
      }
      if constexpr (DO_W) { if constexpr (m % 2 == 0 && m / 2 < NL) loadw(std::integral_constant<int, m / 2>{}); }
      if constexpr (DO_H) {   // (after this step's weight requests: vmcnt retires in order)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (m == 2 * NL - 1 + 2 * j) loadh(hN, j);
      }
      if constexpr (DO_C) { if constexpr (m >= 2 && m < 2 + NP) store_slot(cb_, late_slot(m - 2), (m - 2) & 3); }
      if constexpr (DO_C && m >= NE - XW) xread(std::integral_constant<int, m - (NE - XW)>{});   // window of the next contraction (x tile: read-only)
      if (t8 == 1 && grp < 3) readq((grp + 1) & 1, eb, grp + 1);
      DDMI_SCHED_FENCE();
    });
    }
#ifndef FCV_SAMEW
    if constexpr (DO_W) {
#pragma unroll
      for (int t = 0; t < 1 + NG; ++t) woff[t] += gstep;
    }
#endif
#ifndef FCV_SAMEH
    if constexpr (DO_H) hoff += 1024u;
#endif
  };
  using T = std::true_type;
  using F = std::false_type;
  using Even = std::integral_constant<int, 0>;
  using Odd = std::integral_constant<int, 1>;
  // prologue: chunk 0 contracted, chunk 1 requested -- [weights 0][bias row][weights 1][hidden rows of pair 0], see fc_mainloop_dense
  fc_sfor<0, NL>(loadw);
  // bias row (k = H) with the prologue: wave t contracts slot t, every accumulator starts from its node's bias (fc_mainloop_dense)
  float bb[S0 > 3 ? S0 : 3];
  int lane_b = lane;
  DDMI_OPAQUE(lane_b);
  const int lr_b = lane_b & 15;
  const int cb_b = lr_b < 8 ? lr_b : lr_b < 10 ? 16 * (NB - 1) + lr_b - 8 : 16 * NBK + lr_b - 10, cs_b = lr_b < 8 ? 8 : lr_b < 10 ? 2 : 0;
  const unsigned hrow = (unsigned)(HK - 1) * (unsigned)KS * 4u - (unsigned)wave * (unsigned)KS * 4u;
  fc_sfor<0, NS>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    if (wave == t) {
      if constexpr (C0 == 1 && t == 0) {
#pragma unroll
        for (int i = 0; i < S0 / 4; ++i) {
          const float4 v = fc_buf_ld4(wbuf, lo[0] + 1024u * i, woff[0] + hrow);
          bb[4 * i] = v.x; bb[4 * i + 1] = v.y; bb[4 * i + 2] = v.z; bb[4 * i + 3] = v.w;
        }
      } else {
        constexpr int g = (t - C0) / 3;
        const float3 v = fc_buf_ld3(wbuf, lo[1 + g], woff[1 + g] + hrow);
        bb[0] = v.x; bb[1] = v.y; bb[2] = v.z;
      }
    }
  });
  fc_sfor<0, NL>([&](auto ic) { loadw_to(ic, bw0n, bwgn, gstep); });
#pragma unroll
  for (int pc = 0; pc < 4; ++pc) loadh(hC, pc);
#pragma unroll
  for (int t = 0; t < 1 + NG; ++t) woff[t] += gstep;
  hoff += 1024u;
  fc_sfor<0, NS>([&](auto tc) {   // bias chains
    constexpr int t = decltype(tc)::value;
    if (wave == t) {
      f32x4 ba = f32x4{0.f, 0.f, 0.f, 0.f}, bq = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (C0 == 1 && t == 0) {
#pragma unroll
        for (int j = 0; j < S0; j += 2) {
          ba = __builtin_amdgcn_mfma_f32_16x16x4f32(P.xp0[4 * j], bb[j], ba, 0, 0, 0);
          bq = __builtin_amdgcn_mfma_f32_16x16x4f32(P.xp0[4 * j + 4], bb[j + 1], bq, 0, 0, 0);
        }
      } else {
        constexpr int g = (t - C0) / 3, comp = (t - C0) % 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) ba = __builtin_amdgcn_mfma_f32_16x16x4f32(P.xg[g][comp + 12 * j], bb[j], ba, 0, 0, 0);
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) brow[(4 * (lane_b >> 4) + rr) * BST + cb_b + cs_b * t] = ba[rr] + bq[rr];
    }
  });
#pragma unroll
  for (int t = 0; t < NS; ++t) r[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  fc_sfor<0, XW>(xread);
  fc_sfor<0, NC>([&](auto ic) { contract(ic, 0, false); });
#pragma unroll
  for (int t = 0; t < NS; ++t)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) store_slot(0, t, rr);
  FC_STAMP(pf, 12);
#pragma unroll
  for (int j = 0; j < (S0 > 0 ? S0 : 1); ++j) bw0[j] = bw0n[j];   // chunk 1's fragments
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int j = 0; j < 3; ++j) bwg[g][j] = bwgn[g][j];
#ifdef DDMI_PHASE_CLOCKS
  DDMI_OPAQUE(bw0[0]);
#pragma unroll
  for (int g = 0; g < NG; ++g) DDMI_OPAQUE(bwg[g][0]);
  FC_STAMP(pf, 13);
#endif
#pragma unroll
  for (int t = 0; t < 1 + NG; ++t) woff[t] += gstep;
  fc_sfor<0, XW>(xread);   // window of the first in-loop contraction
  __syncthreads();
#pragma unroll
  for (int vi = 0; vi < 2; ++vi)
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      const float b = brow[(2 * wave + vi) * BST + 16 * c + (lane_b & 15)];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) acc[vi][rt][c][rr] += b;
    }
  FC_STAMP(pf, 2);
  step(T{}, T{}, F{}, Even{}, F{});
  FC_STEP_BARRIER();
  step(T{}, T{}, T{}, Odd{}, F{});
  FC_STEP_BARRIER();
  for (int g = 2; g + 2 < NG8; g += 2) {
    step(T{}, T{}, F{}, Even{}, T{});
    FC_STEP_BARRIER();
    step(T{}, T{}, T{}, Odd{}, F{});
    FC_STEP_BARRIER();
  }
  step(T{}, F{}, F{}, Even{}, T{});
  FC_STEP_BARRIER();
  step(F{}, F{}, F{}, Odd{}, F{});
  FC_STEP_BARRIER();
  FC_STAMP(pf, 3);
}

// Workgroup = 16 virtual nodes x the granules [gsplit[y], gsplit[y+1]), 8 waves: wave w owns virtual nodes 2w, 2w+1 in the
// edge GEMM and row w (k = 8g + w) of every 8-row k group g in the node contraction.  A lane's A values of the edge GEMM over
// a PAIR of groups (k = 8g + 2q + sub, g = 2p, 2p + 1) are one 16-B request of the fragment-ordered hidden rows.  The
// contracted group is double-buffered in LDS; per iteration a wave contracts row w of group g+1 (weights requested one
// iteration earlier), requests the weights of group g+2 and -- every second iteration -- the hidden fragments of the next
// pair, and multiplies group g into its edge accumulators: one barrier per 8 k (fc_mainloop_dense / fc_mainloop_packed).  The
// bias row of the packed second layer (h = 1) is added outside the MFMA loop; the epilogue of a granule couples the
// accumulators with sh_e, 16 edge rows at a time, and streams the message columns out through an LDS staging area
// (wave-local).
// MODE 0: static chain shapes, sparse rows (the second 16-edge tile of a virtual node with <= 16 edges is skipped),
// 1: generic (predicated, compiler-scheduled) contraction of classic granules, 3: static shapes, dense rows (both row tiles of
// every virtual node are multiplied: straight-line chunk body), 4: mode 3 for gather nodes with several virtual nodes each
// (ligand atoms in the rec<-lig group): the x tile holds the tile's DISTINCT gather nodes and the classic granules contract
// them on the 4x4x1 MFMA (fc_mainloop_dense<SH>).  NBK = column blocks of the chunk buffer (widest granule).
// fc_tile: one work item = (tile bx of 16 virtual nodes, granule range by) of the edge group described by `a`; the kernels below
// are thin wrappers (k_conv_fused: blockIdx = the item; k_conv_grouped, k_conv_grp.hip: items of several edge groups in one grid).
template <int MAXD, int SHD, int MODE, int NBK, bool BF = false>
__device__ __forceinline__ void fc_tile(const FusedConvArgs& a, const int bx, const int by, float* __restrict__ smem) {
  static_assert(!BF || (MODE != 1 && MAXD == 3 && SHD == 4), "the bf16 edge product exists in the static l <= 1 loops");
  using D = FcDim<NBK>;
  constexpr int FC_YROW = D::YROW, FC_YVN = D::YVN, FC_YB = D::YB;
  constexpr bool PACK = MODE != 1 && MAXD == 3 && SHD == 4;   // packed granules exist only with the static l <= 1 shapes
  // GS2: coupling row of one edge: classic [k'][4 slots]; packed [k'][even slots 0,2,4,6 | odd slots 1,3,5,7]
  // (row strides 20 / 36: the 16 rows written per request land on 8 distinct bank groups)
  constexpr int GS2 = PACK ? 8 * MAXD + 12 : 4 * MAXD + 8, ES = SHD == 4 ? 8 : SHD + 3, CGN = FC_MAXSLOT * MAXD * SHD;   // ES: edge-row stride (sh, weight, message row); 8 keeps 4 harmonics one 16-B read
  float* xbuf = smem;                                  // [16][XS+1]
  float* ybuf = xbuf + FC_VN * NC_XS;                  // [2][16 x FC_YVN]
  float* gscr = ybuf + 2 * FC_YB;                      // per wave: [16 edge rows][GS2] coupling rows of the current (granule, virtual node, row tile)
  float* escr = gscr + FC_WAVES * 16 * GS2;            // per wave: [2][32][ES] edge rows: sh (SHD), weight, message row
  int* gdesc = reinterpret_cast<int*>(escr + FC_WAVES * 2 * 32 * ES);   // [granules of this workgroup] FGran copies (see below)
  int* gorder = gdesc + FC_MAXG * FC_GWORDS;            // [granules of this workgroup] visiting order (rotated per workgroup, see below)
  int* prep = gorder + FC_MAXG;                         // [FC_TILE_NT] pre-reduction: message row of every target of the tile
  float* cgt = reinterpret_cast<float*>(prep + FC_TILE_NT);   // [granules of this workgroup][8 slots][MAXD][SHD] dense coupling rows
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = DDMI_UNIFORM(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int nvn = *a.nvn;
  const int v0 = bx * FC_VN;
  if (v0 >= nvn) return;
  FcProf pf;
#ifdef DDMI_PHASE_CLOCKS
  pf.start();
  const unsigned pf_t0 = pf.t;
#endif
  FC_WG_START();
  const int nv_live = min(FC_VN, nvn - v0);
  constexpr bool SHM = MODE == 4;
  // ---- tile prologue.  Every global request of the tile is in flight before the first dependent use: the one dependent chain is
  // [virtual node -> gather node -> x row]; granule descriptors, coupling rows, tile header and per-edge rows ride along with it.
  // (Round 5: the loop form -- descriptor / coupling / x-row copies as load -> LDS-store iterations with runtime trip counts -- was
  // ~10 dependent round trips, 25-30 k cycles per workgroup = 0.6 ms per forward, profiles/r05_p1_phase_clocks.txt.)
  int dsl[2] = {2 * wave, 2 * wave + 1};   // x-tile / chunk-buffer row of this wave's two virtual nodes
  int xr = lr;                             // the lane's x-tile row in the 16-row forms
  bool sh_tile = false;                    // SHM: at most four distinct gather nodes -> 4-row x tile, 4x4x1 contraction
  const int xnl = tid >> 5, xj = tid & 31; // x tile: thread = (row, 16-B piece); a row is 40 pieces (XS = 160)
  int xnd = -1;                            // gather node of x-tile row xnl (-1: zero row)
  int ndist = 0;
  if constexpr (SHM) {
    // Distinct gather nodes of the tile (virtual nodes of a node are consecutive): slot of virtual node j = number of node
    // changes up to j.  Every wave derives the same map from one request (lane lr <-> virtual node lr).
    const int vv = v0 + lr;
    const int nd_ = vv < nvn ? a.vn_node[vv] : -1;
    const int prev = __shfl(nd_, (lane & 48) + ((lr + 15) & 15));
    const bool first = lr == 0 || (nd_ >= 0 && nd_ != prev);
    const unsigned bal = (unsigned)(__ballot(first) & 0xffffull);
    auto slot_of = [&](int j) { return __popcll((unsigned long long)(bal & ((2u << j) - 1u))) - 1; };
    ndist = __popcll((unsigned long long)bal);
    sh_tile = DDMI_UNIFORM(ndist) <= 4;
    if (sh_tile) {
      xr = slot_of(lr);
      dsl[0] = DDMI_UNIFORM(slot_of(2 * wave));
      dsl[1] = DDMI_UNIFORM(slot_of(2 * wave + 1));
      // node of slot xnl = the virtual node at the xnl-th set bit of the map
      unsigned bb = bal;
      int p = 0;
      for (int q = 0; q <= (xnl & 3); ++q) { p = bb ? __builtin_ctz(bb) : 0; bb &= bb - 1u; }
      const int nds = __shfl(nd_, (lane & 48) + p);
      xnd = (xnl < 4 && xnl < ndist) ? nds : -1;
    } else {   // more than four distinct nodes: the tile runs in the 16-row form of mode 3
      const int nds = __shfl(nd_, (lane & 48) + (xnl & 15));
      xnd = xnl < nv_live ? nds : -1;
    }
  } else {
    if (xnl < nv_live) xnd = a.vn_node[v0 + xnl];
  }
  const int g_begin = a.gsplit[by], g_end = a.gsplit[by + 1];
  const int n_loc = g_end - g_begin;
  // The granule descriptors are copied to LDS before the first message store: on gfx9-family parts loads and stores share
  // the in-order vmcnt counter, so a descriptor field fetched from global memory AFTER a burst of message stores would wait
  // for every one of them to be acknowledged (the compiler cannot keep the fields in registers across stores that may alias).
  constexpr int GD_IT = (FC_MAXG * FC_GWORDS + 64 * FC_WAVES - 1) / (64 * FC_WAVES);
  int gd_r[GD_IT];
  {
    const int* __restrict__ gsrc = reinterpret_cast<const int*>(a.gran + g_begin);
#pragma unroll
    for (int it = 0; it < GD_IT; ++it) {
      const int idx = tid + 64 * FC_WAVES * it;
      gd_r[it] = idx < n_loc * FC_GWORDS ? gsrc[idx] : 0;
    }
  }
  // dense coupling rows of this workgroup's granules (host-built, weights.cpp): cgt[g][s][k'][j]
  constexpr bool CG_REG = CGN <= 96;       // l <= 1 kernels: through registers with the other requests; wider tables: plain loop below
  constexpr int CG_IT = CG_REG ? (FC_MAXG * CGN + 64 * FC_WAVES - 1) / (64 * FC_WAVES) : 1;
  float cg_r[CG_IT];
  if constexpr (CG_REG) {
#pragma unroll
    for (int it = 0; it < CG_IT; ++it) {
      const int idx = tid + 64 * FC_WAVES * it;
      cg_r[it] = idx < n_loc * CGN ? a.cgt[(size_t)g_begin * CGN + idx] : 0.f;
    }
  }
  // In-tile pre-reduction (launch_vn_tiles): this tile's targets span <= 32 rows -> every wave sums its message rows per target
  // in LDS, the eight partial sums meet in a fixed order and ONE row per target leaves the tile.
  constexpr bool PRE_OK = (MODE == 0 || MODE == 3) && SHD == 4;
  bool pre = false;
  int pre_t0 = 0, pre_nt = 0;
  int pre_rep_r = -1;
  if constexpr (PRE_OK) {
    if (a.tile_hdr) {
      const int* __restrict__ th = a.tile_hdr + (size_t)bx * FC_TILE_HDR;
      pre = DDMI_UNIFORM(th[0]) != 0;
      pre_t0 = DDMI_UNIFORM(th[1]); pre_nt = DDMI_UNIFORM(th[2]);
      if (tid < FC_TILE_NT) pre_rep_r = th[4 + tid];
    }
  }
  (void)pre_t0; (void)pre_nt;
  int vne[2];
  float* gw = gscr + wave * 16 * GS2;
  float* ew_ = escr + wave * 2 * 32 * ES;
  // per-edge rows (harmonics, weight, message row) of the wave's two virtual nodes: prepared by k_vn_rows, one coalesced copy
  constexpr int ER_IT = 2 * 32 * ES / 4 / 64;
  static_assert(2 * 32 * ES % 256 == 0, "per-edge rows of a wave are whole 16-B pieces per lane");
  float4 er_r[ER_IT];
  {
    const float4* __restrict__ rsrc = reinterpret_cast<const float4*>(a.vrows + (size_t)(v0 + 2 * wave) * 32 * ES);
#pragma unroll
    for (int it = 0; it < ER_IT; ++it) er_r[it] = rsrc[lane + 64 * it];
    vne[0] = a.vn_ne[v0 + 2 * wave];
    vne[1] = a.vn_ne[v0 + 2 * wave + 1];
  }
  // x rows (behind the node ids)
  float4 xv0 = make_float4(0.f, 0.f, 0.f, 0.f), xv1 = xv0;
  if (xnd >= 0) {
    const float4* __restrict__ xrow = reinterpret_cast<const float4*>(a.X + (size_t)(a.gbase + xnd) * XS);
    xv0 = xrow[xj];
    if (xj < XS / 4 - 32) xv1 = xrow[32 + xj];
  }
  static_assert(XS % 4 == 0 && XS / 4 > 32 && XS / 4 <= 64 && (NC_XS % 2) == 0, "x tile copy: 40 pieces per row, 8-B aligned LDS rows");
  FC_STAMP(pf, 16);
  // ---- everything to LDS
#pragma unroll
  for (int it = 0; it < GD_IT; ++it) {
    const int idx = tid + 64 * FC_WAVES * it;
    if (idx < n_loc * FC_GWORDS) gdesc[idx] = gd_r[it];
  }
  const FGran* __restrict__ gran_l = reinterpret_cast<const FGran*>(gdesc) - g_begin;   // gran_l[gi], gi in [g_begin, g_end)
  // Visiting order of the granules, rotated by whole units per workgroup: tiles start together and take equal time, so with
  // one common order all 256 CUs would issue their message stores (98 KB per tile and granule) in the same microseconds and
  // then wait for that burst to drain; rotated, the stores of the chip spread over the whole granule period.
  if (tid < n_loc) {   // (unit starts and the units of this granule range come with the kernel arguments: no global loads)
    const int n = n_loc, nu = a.ucount[by];
    const int want = DDMI_ABL(a.dbg, 2048) || nu == 0 ? 0 : (int)(bx % (unsigned)nu);
    const int start = nu == 0 ? 0 : a.ustart[a.ufirst[by] + want] - g_begin;
    gorder[tid] = g_begin + (start + tid) % n;
  }
  if constexpr (CG_REG) {
#pragma unroll
    for (int it = 0; it < CG_IT; ++it) {
      const int idx = tid + 64 * FC_WAVES * it;
      if (idx < n_loc * CGN) cgt[idx] = cg_r[it];
    }
  } else {
    for (int idx = tid; idx < n_loc * CGN; idx += 64 * FC_WAVES) cgt[idx] = a.cgt[(size_t)g_begin * CGN + idx];
  }
  if constexpr (PRE_OK) {
    if (pre && tid < FC_TILE_NT) prep[tid] = pre_rep_r;   // (to LDS before the first message store, like the granule descriptors)
  }
  (void)prep;
#pragma unroll
  for (int it = 0; it < ER_IT; ++it) reinterpret_cast<float4*>(ew_)[lane + 64 * it] = er_r[it];
  if (!(SHM && sh_tile) || xnl < 4) {   // (shared-node tiles keep their distinct nodes in rows 0..3)
    float* xd = xbuf + xnl * NC_XS + 4 * xj;
    *reinterpret_cast<float2*>(xd) = make_float2(xv0.x, xv0.y);
    *reinterpret_cast<float2*>(xd + 2) = make_float2(xv0.z, xv0.w);
    if (xj < XS / 4 - 32) {
      *reinterpret_cast<float2*>(xd + 128) = make_float2(xv1.x, xv1.y);
      *reinterpret_cast<float2*>(xd + 130) = make_float2(xv1.z, xv1.w);
    }
  }
  FC_STAMP(pf, 17);
  __syncthreads();
  const int H = a.HK - 1;
  const int NG8 = a.NG8;
  const float* __restrict__ hb_tile = a.Hb + fc_hb_off(v0, 0, 0, 0, fc_ngp(NG8));   // uniform: hidden rows of this tile
  (void)hb_tile;
  const float* __restrict__ hfrag = a.Hb + fc_hb_off(v0 + 2 * wave, 0, 0, lane, fc_ngp(NG8));   // + fc_hb_off(vi, rt, g, 0)
  float* const ywr0 = ybuf + (4 * lq) * FC_YVN + wave * FC_YROW;               // node 4lq (+r), row = wave; + the lane's column
  float* const ywr = ywr0 + lr;                                                // classic granules: column 16*slot + lr
  const float* const yrd = ybuf + (2 * wave) * FC_YVN + (2 * lq) * FC_YROW + lr;   // node 2wave (+vi), row 2lq (+sub), column 16c + lr
  float* const ywr_sh = ybuf + lq * FC_YVN + wave * FC_YROW + lr;                  // shared-node form: node slot lq (+4 per pass)
  const float* const yrd_sh = ybuf + (2 * lq) * FC_YROW + lr;                      // + the node slot of the virtual node
  (void)ywr_sh; (void)yrd_sh;
  FC_STAMP(pf, 0);
  for (int go = g_begin; go < g_end; ++go) {
    FC_COUNT(pf, 14);
    const int gi = gorder[go - g_begin];
    const FGran& Gd = gran_l[gi];
    if (DDMI_ABL(a.dbg, 4096) && Gd.accumulate) continue;   // timing-only: the second granule of a wide unit dropped
    const bool packed = PACK && Gd.shape >= 4 && Gd.shape <= 6;
    const int NB = packed ? Gd.nb : 4;                       // live column blocks of this granule
    f32x4 acc[2][2][NBK];
#pragma unroll
    for (int vi = 0; vi < 2; ++vi)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int c = 0; c < NBK; ++c) acc[vi][rt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!Gd.empty && !DDMI_ABL(a.dbg, 128)) {
      if (PACK && packed) {
        const FcPackRt P = fc_pack_setup(Gd, xbuf, lr, lq, xr);
        FC_STAMP(pf, 1);
        constexpr bool DN = MODE == 3 || MODE == 4;
#define FC_MLP(S0_, NG_) fc_mainloop_packed<NBK, S0_, NG_, DN, BF>(acc, P, a.wpack, a.KS, a.HK, NG8, wave, lane, hb_tile, vne, ywr0, yrd, pf, gscr)
        if constexpr (NBK >= 5) { if (Gd.shape == 4) FC_MLP(12, 2); }
        if (Gd.shape == 5) FC_MLP(0, 2);
        else if (Gd.shape == 6) FC_MLP(12, 1);
#undef FC_MLP
      } else {
      const FcSlotRt s0 = fc_slot_setup(Gd.slot[0], a.wpack, xbuf, Gd.shape == 7 ? 0 : Gd.w0, lr, lq, sh_tile ? (lr & 3) : xr);
      const FcSlotRt s1 = fc_slot_setup(Gd.slot[1], a.wpack, xbuf, Gd.shape == 7 ? 16 : Gd.w0, lr, lq, sh_tile ? (lr & 3) : xr);
      const FcSlotRt s2 = fc_slot_setup(Gd.slot[2], a.wpack, xbuf, Gd.shape == 7 ? 32 : Gd.w0, lr, lq, sh_tile ? (lr & 3) : xr);
      const FcSlotRt s3 = fc_slot_setup(Gd.slot[3], a.wpack, xbuf, Gd.shape == 7 ? 48 : Gd.w0, lr, lq, sh_tile ? (lr & 3) : xr);
      if (MODE == 0 || MODE == 3 || MODE == 4) {   // static chain shapes: hand-scheduled loop, dense (3, 4) or sparse (0) rows
        constexpr bool DN = MODE == 3 || MODE == 4;
        const FcSlotRt sl[4] = {s0, s1, s2, s3};
        FC_STAMP(pf, 1);
#define FC_ML(S0_, SN_, DUP_, NLV_)                                                                                                  \
  do {                                                                                                                                \
    if (SHM && sh_tile) fc_mainloop_dense<NBK, S0_, SN_, DN, DUP_, NLV_, SHM, BF>(acc, sl, a.wpack, a.KS, a.HK, NG8, wave, lane, hb_tile, vne, ywr_sh, yrd_sh, pf, gscr, dsl); \
    else fc_mainloop_dense<NBK, S0_, SN_, DN, DUP_, NLV_, false, BF>(acc, sl, a.wpack, a.KS, a.HK, NG8, wave, lane, hb_tile, vne, ywr, yrd, pf, gscr, dsl); \
  } while (0)
        int dup, nlv;
        fc_variant(Gd, dup, nlv);
        if (Gd.shape == 7) FC_ML(12, 12, 0, 2);                // three channel tiles of one 12-step path
        else if (Gd.shape == 1 && dup == 1 && nlv == 3) FC_ML(12, 3, 1, 3);
        else if (Gd.shape == 1 && dup == 1) FC_ML(12, 3, 1, 2);
        else if (Gd.shape == 1) FC_ML(12, 3, 0, 3);          // (a padding slot is contracted like a live one: its result is never read)
        else if (Gd.shape == 2 && dup == 3 && nlv == 2) FC_ML(3, 3, 3, 2);
        else if (Gd.shape == 2 && dup == 3 && nlv == 1) FC_ML(3, 3, 3, 1);
        else if (Gd.shape == 2 && dup == 3) FC_ML(3, 3, 3, 3);
        else if (Gd.shape == 2 && dup == 2) FC_ML(3, 3, 2, 3);
        else if (Gd.shape == 2) FC_ML(3, 3, 0, 3);
        else FC_ML(12, 0, 0, 3);
#undef FC_ML
      } else {
      FcPre pre;
      const int shape = Gd.shape;
      auto fetch = [&](int k) __attribute__((always_inline)) {
        if (k >= H) return;
        const size_t koff = (size_t)k * a.KS;
        (void)shape;
        fc_fetch<FC_CAP0>(s0, koff, pre.b0); fc_fetch<FC_CAPN>(s1, koff, pre.b1);
        fc_fetch<FC_CAPN>(s2, koff, pre.b2); fc_fetch<FC_CAPN>(s3, koff, pre.b3);
      };
      // contraction of row k (this wave's row of a group) into buffer `buf`; rows past the hidden width are zero
      auto contract = [&](int k, int buf) __attribute__((always_inline)) {
        float* yw = ywr + buf * FC_YB;
        f32x4 r0 = f32x4{0.f, 0.f, 0.f, 0.f}, r1 = r0, r2 = r0, r3 = r0;
        if (k < H) {
          const size_t koff = (size_t)k * a.KS;
          r0 = fc_apply<FC_CAP0>(s0, koff, pre.b0); r1 = fc_apply<FC_CAPN>(s1, koff, pre.b1);
          r2 = fc_apply<FC_CAPN>(s2, koff, pre.b2); r3 = fc_apply<FC_CAPN>(s3, koff, pre.b3);
        }
        fc_store<NBK>(yw, 0, r0); fc_store<NBK>(yw, 16, r1); fc_store<NBK>(yw, 32, r2); fc_store<NBK>(yw, 48, r3);
      };
      // edge GEMM of this wave's 2 virtual nodes on the group in buffer `buf` (two k-steps: sub = 0, 1)
      auto edge_gemm = [&](int buf, const float2 (&hA)[2][2]) __attribute__((always_inline)) {
        const float* __restrict__ yb0 = yrd + buf * FC_YB;
#pragma unroll
        for (int vi = 0; vi < 2; ++vi) {
          if (vne[vi] == 0) continue;
#pragma unroll
          for (int sub = 0; sub < 2; ++sub) {
            const float* __restrict__ yb = yb0 + vi * FC_YVN + sub * FC_YROW;
            const float b0 = yb[0], b1 = yb[16], b2 = yb[32], b3 = yb[48];
            const float a0 = sub == 0 ? hA[vi][0].x : hA[vi][0].y;
            acc[vi][0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[vi][0][0], 0, 0, 0);
            acc[vi][0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[vi][0][1], 0, 0, 0);
            acc[vi][0][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b2, acc[vi][0][2], 0, 0, 0);
            acc[vi][0][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b3, acc[vi][0][3], 0, 0, 0);
            if (vne[vi] > 16) {
              const float a1 = sub == 0 ? hA[vi][1].x : hA[vi][1].y;
              acc[vi][1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[vi][1][0], 0, 0, 0);
              acc[vi][1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[vi][1][1], 0, 0, 0);
              acc[vi][1][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b2, acc[vi][1][2], 0, 0, 0);
              acc[vi][1][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b3, acc[vi][1][3], 0, 0, 0);
            }
          }
        }
      };
      auto load_h = [&](int g, float2 (&hA)[2][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int vi = 0; vi < 2; ++vi)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
            hA[vi][rt] = make_float2(0.f, 0.f);
            if (g < NG8 && vne[vi] > 16 * rt)
              hA[vi][rt] = *reinterpret_cast<const float2*>(hfrag + fc_hb_off(vi, rt, g, 0, fc_ngp(NG8)));
          }
      };
      float2 hC[2][2], hN[2][2];
      fetch(wave);
      load_h(0, hC);
      contract(wave, 0);
      fetch(8 + wave);
      __syncthreads();
      for (int g = 0; g < NG8; ++g) {
        const int kn = 8 * (g + 1) + wave;
        if (g + 1 < NG8) contract(kn, (g + 1) & 1);     // weights requested one iteration ago
        if (g + 2 < NG8) fetch(kn + 8);
        load_h(g + 1, hN);
        edge_gemm(g & 1, hC);
        __syncthreads();
#pragma unroll
        for (int vi = 0; vi < 2; ++vi)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) hC[vi][rt] = hN[vi][rt];
      }
      }
      if (MODE == 1) {   // generic loop: the bias row (k = H, h = 1) behind the main loop -- waves 0..3 contract one slot each,
                         // every edge row receives the node's bias row (the static loops take it along in their prologue)
        if (wave < 4) {
          const FcSlotRt sb = fc_slot_setup(Gd.slot[wave], a.wpack, xbuf, Gd.w0, lr, lq, xr);
          const f32x4 rb = fc_direct(sb, (size_t)H * a.KS);
          fc_store<NBK>(ybuf + (4 * lq) * FC_YVN + lr, 16 * wave, rb);    // row 0 of buffer 0
        }
        __syncthreads();
#pragma unroll
        for (int vi = 0; vi < 2; ++vi) {
          const float* __restrict__ yb = ybuf + (2 * wave + vi) * FC_YVN + lr;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float b = yb[16 * c];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[vi][rt][c][r] += b;
          }
        }
      }
      }
    }
    FC_STAMP(pf, 4);
    __syncthreads();   // the coupling phase stages message rows in the (now idle) chunk buffers
    FC_STAMP(pf, 5);
    // ---- coupling with the spherical harmonics and message stores, 16 edge rows at a time (wave-local: no workgroup barrier)
    // (the lane id is made opaque here: every per-lane address of the epilogue is then recomputed per granule instead of being
    // hoisted out of the granule loop, where it would stay live across the register-tight main loops)
    int lane_e = lane;
    DDMI_OPAQUE(lane_e);
    const int lr_e = lane_e & 15, lq_e = lane_e >> 4;
    const float* __restrict__ cg = cgt + (gi - g_begin) * CGN;
    float* stg = ybuf + wave * ((2 * FC_YB) / FC_WAVES);   // the chunk buffers are idle during the coupling phase: [16][RS] message rows
    float* const pw = stg;                                 // pre-reduction: this wave's partial sums [targets of the tile][RS] (instead of the staged rows)
    const bool tri = Gd.shape == 7;                        // merged granule: slot c = output channels 16c .. 16c+15 (dout = 1)
    // (row stride of the staged rows / partial sums: the widest row of the kernel's granules for EVERY granule -- a compile-time
    // stride keeps the row addressing out of the integer multiplier; scalar blocks then use 16 of the 48 columns)
    constexpr int RS = 16 * MAXD;
    const int L = Gd.n_w * Gd.dout, c0 = Gd.o_off + Gd.w0 * Gd.dout;
    const int V = ((c0 | L) & 3) == 0 ? 4 : ((c0 | L) & 1) == 0 ? 2 : 1;
    // Round 5: BOTH 16-row tiles of a virtual node go through every phase together (coupling rows -> coupling -> tail block ->
    // row stores): the phases are wave-local LDS hand-offs whose latency is exposed (all eight waves of the CU are in the same
    // phase), so two row tiles per hand-off halve the exposures.  Wave-private scratch in the idle chunk buffers, WST floats:
    //   [0, 1536)    staged rows of row tile 0 | 1 ([16][RS <= 48] each), or the partial sums of the pre-reduction ([<= 32 targets][RS])
    //   [1536, 2048) tail-block transposes of row tile 0 | 1 (packed granules)
    //   [2048, ...)  coupling rows of row tile 1 (row tile 0: the wave's gscr rows)
    // Launches whose chunk buffers are too small for that (4-column-block kernels with a packed granule, l = 2 kernels) keep
    // the one-row-tile sequence.
    constexpr int WST = (2 * FC_YB) / FC_WAVES;
    constexpr bool BATCH_FITS = PACK && WST >= 2048 + 16 * GS2;          // tail blocks included
    constexpr bool BATCH_FITS_NOTAIL = PACK && WST >= 1536 + 16 * GS2;   // granules without a tail block (classic, merged)
    float* const gw2 = stg + ((BATCH_FITS || !BATCH_FITS_NOTAIL) ? 2048 : 1536);
    static_assert(!PACK || FC_TILE_NT * 48 <= 1536, "partial sums of the pre-reduction fit the staged-row area");
    static_assert(WST - 68 >= (PACK ? (BATCH_FITS ? 2048 + 16 * GS2 : BATCH_FITS_NOTAIL ? 1536 + 16 * GS2 : 1792) : 16 * 16 * MAXD + 256),
                  "the per-lane dump words of the masked stores lie behind everything else in the wave's scratch");
    const bool packed_rt = PACK && packed, tri_rt = tri, pre_rt = PRE_OK && pre;
    // (Tried and dropped, profiles/r05_e12_ab.txt: touching the NEXT granule's first weight rows here, one dword per 16-B piece,
    // so that its prologue's requests -- 4 k cycles of exposed wait per granule -- find the lines in L2: 151.5 against 154.1 poses/s.)
    // One instance of the coupling phase per granule kind and pre-reduction state (round 5): the wave-uniform tests on them --
    // per value, per row, per phase in the run-time form -- fold at compile time inside an instance (the epilogue is bound by
    // its instruction count).  KIND: 0 packed, 1 merged (three scalar channel tiles), 2 scalar block,
    // 3 vector block, 4 any other output width.  The locals below SHADOW the run-time values of the same name.
    auto epi_body = [&](auto kindc, auto prec) __attribute__((always_inline)) {
    constexpr int KIND = decltype(kindc)::value;
    constexpr bool pre = decltype(prec)::value;
    constexpr bool packed = KIND == 0;
    const int dout = KIND == 1 || KIND == 2 ? 1 : KIND == 3 ? 3 : Gd.dout;
    const bool batch2 = !DDMI_ABL(a.dbg, 16384) && (BATCH_FITS || (BATCH_FITS_NOTAIL && !(PACK && packed)));
    if (PRE_OK && pre) {   // partial sums start from zero
      for (int idx = lane_e; idx < pre_nt * RS / 4; idx += 64) reinterpret_cast<float4*>(pw)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // A lane's message values of one row tile: staged in the row's place, or (pre-reduction) ADDED to the partial sum of the
    // row's target.  The adds of a row tile go out as [all loads][all stores]: within a virtual node every edge has its own
    // target (a gather node has at most ONE edge to a target in every graph the builders produce: radius graphs, kNN lists and
    // all-pairs cross graphs hold a pair once), so the addresses of one tile are distinct -- value by value (load, add, store; the compiler must keep possibly
    // aliasing LDS accesses in order) each of the 12 values paid an LDS round trip.
    constexpr int NPUT = 4 * MAXD;
    struct Puts { float* p[NPUT]; float v[NPUT]; bool ok[NPUT]; };
    // Masked-out values go to a per-lane dump word instead of being branched around: a predicated LDS access costs an exec-mask
    // save / branch / restore sequence (four scalar instructions) per value, the select costs one v_cndmask.
    // (One select per ROW on a base pointer, the values at base + k, was tried next: the compiler then keeps the pointer array
    // in scratch memory and addresses LDS through flat instructions -- 62 instead of 147 poses/s, profiles/r05_e5_ab.txt.)
    float* const dump = stg + WST - 68 + lane_e;   // (+ 2 floats behind the lane's word for a row's values k = 1, 2: all of it scratch)
    // entries [row r][k]: rows r < NR, values k < KN of a row (compile time: a scalar block stores ONE value per row, not MAXD
    // slots of which MAXD - 1 go to the dump word).  SEL: the per-entry select on P.ok; without it the caller has already
    // pointed the masked ROWS at the dump word (one select per row: P.p[r][k] = (ok_r ? row address : dump) + k).
    auto flush = [&](Puts& P, auto nrc, auto knc, auto selc) __attribute__((always_inline)) {
      constexpr int NR = decltype(nrc)::value, KN = decltype(knc)::value;
      constexpr bool SEL = decltype(selc)::value;
#define FC_LIVE(i) ((i) / MAXD < NR && (i) % MAXD < KN)
#pragma unroll
      for (int i = 0; i < NPUT; ++i) if (SEL && FC_LIVE(i)) P.p[i] = P.ok[i] ? P.p[i] : dump;
      if (PRE_OK && pre) {
        float old[NPUT];
#pragma unroll
        for (int i = 0; i < NPUT; ++i) if (FC_LIVE(i)) old[i] = *P.p[i];
#pragma unroll
        for (int i = 0; i < NPUT; ++i) if (FC_LIVE(i)) *P.p[i] = old[i] + P.v[i];
      } else {
#pragma unroll
        for (int i = 0; i < NPUT; ++i) if (FC_LIVE(i)) *P.p[i] = P.v[i];
      }
#undef FC_LIVE
    };
    using I1 = std::integral_constant<int, 1>;
    using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;
    using IM = std::integral_constant<int, MAXD>;
    using SelY = std::true_type;
    using SelN = std::false_type;
#pragma unroll
    for (int vi = 0; vi < 2; ++vi) {
      const int ne = vne[vi];
      if (ne == 0 || DDMI_ABL(a.dbg, 32)) continue;
      const float* __restrict__ erow = ew_ + vi * 32 * ES;
      // ---- phase A: G[row][s][k'] = we_row * sum_j cg[s][k'][j] * sh_row[j] : lane = (edge row, quarter of the slots); the edge weight rides along
      auto phase_g = [&](int rt, float* __restrict__ gwx) __attribute__((always_inline)) {
        const int row = lane_e & 15, part = lane_e >> 4, el = 16 * rt + row;
        float sh[SHD];
        if constexpr (SHD == 4) {
          const float4 s4 = *reinterpret_cast<const float4*>(erow + el * ES);
          sh[0] = s4.x; sh[1] = s4.y; sh[2] = s4.z; sh[3] = s4.w;
        } else {
#pragma unroll
          for (int j = 0; j < SHD; ++j) sh[j] = erow[el * ES + j];
        }
        const float we = erow[el * ES + SHD];
        auto gval = [&](int s_, int k) __attribute__((always_inline)) {
          float v = 0.f;
#pragma unroll
          for (int j = 0; j < SHD; ++j) v = fmaf(cg[(s_ * MAXD + k) * SHD + j], sh[j], v);
          return v * we;
        };
        if (PACK && packed) {   // slots 2*part (even half, position part) and 2*part + 1 (odd half, position part)
#pragma unroll
          for (int k = 0; k < MAXD; ++k) {
            gwx[row * GS2 + 8 * k + part] = gval(2 * part, k);
            gwx[row * GS2 + 8 * k + 4 + part] = gval(2 * part + 1, k);
          }
        } else {
#pragma unroll
          for (int k = 0; k < MAXD; ++k)
            if (k < dout) gwx[row * GS2 + 4 * k + part] = gval(part, k);   // the four slots of (row, k') side by side (scalar blocks: k' = 0 only)
        }
      };
      // ---- phase B: coupling of the lane's four rows 4 lq + r with their accumulators; destination = the staged row, or
      // (pre-reduction) the partial-sum row of the row's target
      auto phase_c = [&](auto rtc, const float* __restrict__ gwx, float* __restrict__ stgx, float* __restrict__ tTx) __attribute__((always_inline)) {
        constexpr int rt = decltype(rtc)::value;
        float* orow[4]; bool rok[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * lq_e + r, el = 16 * rt + row;
          rok[r] = true; orow[r] = stgx + row * RS;
          if (PRE_OK && pre) {
            rok[r] = el < ne;
            orow[r] = pw + (rok[r] ? reinterpret_cast<const int*>(erow)[el * ES + 7] - pre_t0 : 0) * RS;
          }
        }
        Puts P;
        if (PACK && packed) {
          if constexpr (PACK) {
            const int hi = lr_e >> 3, NS = Gd.nslot;
            // pair blocks: lanes lr_e and lr_e + 8 hold the even / odd slots of channel lr_e & 7; their partial sums meet by a row rotate
            // (one body per packed shape with the block / slot counts at compile time measured 1 % SLOWER: profiles/r05_e7_ab.txt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = 4 * lq_e + r;
              const float* __restrict__ G = gwx + row * GS2 + 4 * hi;
#pragma unroll
              for (int k = 0; k < MAXD; ++k) {
                const float4 g4 = *reinterpret_cast<const float4*>(G + 8 * k);
                const float gq[4] = {g4.x, g4.y, g4.z, g4.w};
                float v = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c)   // pair block c: slot 2c + hi (a slot past the last one is a padding column: not read)
                  if (c < NB - 1) v = fmaf(gq[c], 2 * c + hi < NS ? acc[vi][rt][c][r] : 0.f, v);
                P.v[r * MAXD + k] = v + DDMI_ROW_XOR8(v);
              }
              float* const pr = (lr_e < 8 && rok[r]) ? orow[r] + lr_e * MAXD : dump;   // (packed granules: dout = MAXD = 3)
#pragma unroll
              for (int k = 0; k < MAXD; ++k) P.p[r * MAXD + k] = pr + k;
              // tail block: lane_e lr_e = 2*slot + (channel - 8) -> transposed through LDS
              float tv = 0.f;
#pragma unroll
              for (int c = 2; c < NBK; ++c) if (c == NB - 1) tv = acc[vi][rt][c][r];
              tTx[row * 16 + (lr_e & 1) * 8 + (lr_e >> 1)] = (lr_e >> 1) < NS ? tv : 0.f;
            }
            flush(P, I4{}, IM{}, SelN{});
          }
        } else {
          // message value of (row, k') from the lane_e's four slot accumulators
          auto couple = [&](const float* __restrict__ G, int k, float t0, float t1, float t2, float t3) __attribute__((always_inline)) {
            const float4 g4 = *reinterpret_cast<const float4*>(G + 4 * k);
            float v = g4.x * t0;
            v = fmaf(g4.y, t1, v);
            v = fmaf(g4.z, t2, v);
            return fmaf(g4.w, t3, v);
          };
          // one body per output width: DO = 0 the merged granule (three scalar channel tiles: every slot is
          // its own output, columns lr, 16 + lr, 32 + lr), 1 scalar blocks, 3 vector blocks (the components of (row, w) side by
          // side), -1 any other width (l = 2 kernels)
          auto body = [&](auto doc) __attribute__((always_inline)) {
            constexpr int DO = decltype(doc)::value;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = 4 * lq_e + r;
              const float* __restrict__ G = gwx + row * GS2;
              const float t0 = acc[vi][rt][0][r], t1 = acc[vi][rt][1][r], t2 = acc[vi][rt][2][r], t3 = acc[vi][rt][3][r];
              if constexpr (DO == 0) {
                const float4 g4 = *reinterpret_cast<const float4*>(G);
                const float gv[3] = {g4.x * t0, g4.y * t1, g4.z * t2};
#pragma unroll
                for (int k = 0; k < MAXD; ++k) {
                  P.ok[r * MAXD + k] = rok[r];
                  P.p[r * MAXD + k] = orow[r] + lr_e + 16 * k;
                  P.v[r * MAXD + k] = k < 3 ? gv[k < 3 ? k : 0] : 0.f;
                }
              } else if constexpr (DO > 0) {   // one select per row: masked rows point at the dump word
                float* const pr = rok[r] ? orow[r] + lr_e * DO : dump;
#pragma unroll
                for (int k = 0; k < MAXD; ++k) {
                  P.p[r * MAXD + k] = pr + k;
                  P.v[r * MAXD + k] = k < DO ? couple(G, k, t0, t1, t2, t3) : 0.f;
                }
              } else {
#pragma unroll
                for (int k = 0; k < MAXD; ++k) {
                  const bool live = k < dout;
                  P.ok[r * MAXD + k] = rok[r] && live;
                  P.p[r * MAXD + k] = orow[r] + lr_e * dout + k;
                  P.v[r * MAXD + k] = live ? couple(G, k, t0, t1, t2, t3) : 0.f;
                }
              }
            }
            if constexpr (DO == 0) flush(P, I4{}, I3{}, SelY{});
            else if constexpr (DO == 3) flush(P, I4{}, I3{}, SelN{});
            else if constexpr (DO == 1) flush(P, I4{}, I1{}, SelN{});
            else flush(P, I4{}, IM{}, SelY{});
          };
          body(std::integral_constant<int, KIND == 1 ? 0 : KIND == 2 ? 1 : KIND == 3 ? 3 : -1>{});
        }
      };
      // ---- phase B': tail block of a packed granule: lane = (row, channel 8 + wb), all slots of one output channel
      auto phase_t = [&](int rt, const float* __restrict__ gwx, float* __restrict__ stgx, const float* __restrict__ tTx) __attribute__((always_inline)) {
        if constexpr (PACK) {
          if (lane_e < 32) {
            const int row = lane_e >> 1, wb = lane_e & 1;
            float* o2 = stgx + row * RS; bool ok2 = true;
            if (PRE_OK && pre) {
              ok2 = 16 * rt + row < ne;
              o2 = pw + (ok2 ? reinterpret_cast<const int*>(erow)[(16 * rt + row) * ES + 7] - pre_t0 : 0) * RS;
            }
            ok2 = ok2 && 8 + wb < Gd.n_w;
            const float4 ta = *reinterpret_cast<const float4*>(tTx + row * 16 + wb * 8), tb = *reinterpret_cast<const float4*>(tTx + row * 16 + wb * 8 + 4);
            Puts P;
#pragma unroll
            for (int k = 0; k < MAXD; ++k) {
              const float4 ge = *reinterpret_cast<const float4*>(gwx + row * GS2 + 8 * k), go_ = *reinterpret_cast<const float4*>(gwx + row * GS2 + 8 * k + 4);
              float v = ge.x * ta.x;           // slots 0, 2, 4, 6 = ge.xyzw; 1, 3, 5, 7 = go_.xyzw; tT holds slot order 0..7
              v = fmaf(go_.x, ta.y, v); v = fmaf(ge.y, ta.z, v); v = fmaf(go_.y, ta.w, v);
              v = fmaf(ge.z, tb.x, v); v = fmaf(go_.z, tb.y, v); v = fmaf(ge.w, tb.z, v); v = fmaf(go_.w, tb.w, v);
              P.v[k] = v;
            }
            float* const pr = ok2 ? o2 + (8 + wb) * MAXD : dump;
#pragma unroll
            for (int k = 0; k < MAXD; ++k) P.p[k] = pr + k;
            flush(P, I1{}, IM{}, SelN{});
          }
        }
      };
      // ---- phase C: staged rows -> message rows
      auto phase_s = [&](int rt, const float* __restrict__ stgx) __attribute__((always_inline)) {
        if (!DDMI_ABL(a.dbg, 256) && !(PRE_OK && pre)) {
          const int nrows = min(16, ne - 16 * rt);
          const float* __restrict__ er = erow + rt * 16 * ES;
          const int accum = DDMI_ABL(a.dbg, 8192) ? 0 : Gd.accumulate;   // (timing-only: later granules of a unit overwrite instead of adding)
          // (every staged piece read before the first store instead of this rolled loop: null, profiles/r05_e12_ab.txt -- the 7 % of a
          // rec-rec launch spent here wait on the store queue, not on LDS)
          if (V == 4) fc_store_rows<4>(stgx, RS, L, nrows, er, ES, SHD + 1, a.msg, c0, accum, lane_e);
          else if (V == 2) fc_store_rows<2>(stgx, RS, L, nrows, er, ES, SHD + 1, a.msg, c0, accum, lane_e);
          else fc_store_rows<1>(stgx, RS, L, nrows, er, ES, SHD + 1, a.msg, c0, accum, lane_e);
        }
      };
      using RT0 = std::integral_constant<int, 0>;
      using RT1 = std::integral_constant<int, 1>;
      float* const tT0 = stg + 1536;        // (one-row-tile sequence: both row tiles use the first slots in turn)
      if (batch2 && ne > 16) {
        DDMI_WAVE_SYNC();
        phase_g(0, gw); phase_g(1, gw2);
        DDMI_WAVE_SYNC();
        FC_STAMP(pf, 6);
        phase_c(RT0{}, gw, stg, tT0); phase_c(RT1{}, gw2, stg + 768, tT0 + 256);
        if (PACK && packed) {
          DDMI_WAVE_SYNC();
          phase_t(0, gw, stg, tT0); phase_t(1, gw2, stg + 768, tT0 + 256);
        }
        DDMI_WAVE_SYNC();
        FC_STAMP(pf, 7);
        phase_s(0, stg); phase_s(1, stg + 768);
        DDMI_WAVE_SYNC();
        FC_STAMP(pf, 8);
      } else {
        float* const tT1 = (PRE_OK && pre) ? tT0 : stg + 16 * 16 * MAXD;   // (as before round 5: behind the staged rows / the partial sums)
        DDMI_WAVE_SYNC();
        phase_g(0, gw);
        DDMI_WAVE_SYNC();
        FC_STAMP(pf, 6);
        phase_c(RT0{}, gw, stg, tT1);
        if (PACK && packed) { DDMI_WAVE_SYNC(); phase_t(0, gw, stg, tT1); }
        DDMI_WAVE_SYNC();
        FC_STAMP(pf, 7);
        phase_s(0, stg);
        DDMI_WAVE_SYNC();
        FC_STAMP(pf, 8);
        if (ne > 16) {
          phase_g(1, gw);
          DDMI_WAVE_SYNC();
          FC_STAMP(pf, 6);
          phase_c(RT1{}, gw, stg, tT1);
          if (PACK && packed) { DDMI_WAVE_SYNC(); phase_t(1, gw, stg, tT1); }
          DDMI_WAVE_SYNC();
          FC_STAMP(pf, 7);
          phase_s(1, stg);
          DDMI_WAVE_SYNC();
          FC_STAMP(pf, 8);
        }
      }
    }
    };   // epi_body
    {
      using P0 = std::false_type;
      using P1 = std::true_type;
      auto run_kind = [&](auto prec) __attribute__((always_inline)) {
        if constexpr (PACK) {
          if (packed_rt) { epi_body(std::integral_constant<int, 0>{}, prec); return; }
          if (tri_rt) { epi_body(std::integral_constant<int, 1>{}, prec); return; }
        }
        if (Gd.dout == 1) epi_body(std::integral_constant<int, 2>{}, prec);
        else if (Gd.dout == 3) epi_body(std::integral_constant<int, 3>{}, prec);
        else epi_body(std::integral_constant<int, 4>{}, prec);
      };
      if constexpr (PRE_OK) { if (pre_rt) run_kind(P1{}); else run_kind(P0{}); }
      else run_kind(P0{});
    }
    FC_STAMP(pf, 7);
    if (PRE_OK && pre) {   // the eight partial sums of every (target, column), in wave order; one message row per target leaves the tile
      __syncthreads();
      constexpr int WST = (2 * FC_YB) / FC_WAVES;
      const int accum = Gd.accumulate;
      for (int e = tid; e < pre_nt * L; e += 64 * FC_WAVES) {
        const int j = e / L, cc = e - j * L;
        const float* __restrict__ pp = ybuf + j * RS + cc;
        float sum = pp[0];
#pragma unroll
        for (int w = 1; w < FC_WAVES; ++w) sum += pp[w * WST];
        const int row = prep[j];
        if (row >= 0) {
          float* __restrict__ q = a.msg + (size_t)row * XS + c0 + cc;
          *q = accum ? *q + sum : sum;
        }
      }
    }
    __syncthreads();   // chunk buffers / coupling scratch are reused by the next granule
    FC_STAMP(pf, 9);
  }
#ifdef DDMI_PHASE_CLOCKS
  pf.acc[10] = pf.t - pf_t0;
  pf.acc[15] = 1;
  if (lane == 0 && a.prof_slot >= 0 && a.prof_slot < FC_PROF_SLOTS) {
#pragma unroll
    for (int i = 0; i < FC_NPROF; ++i) atomicAdd(&g_fc_prof[a.prof_slot * FC_NPROF + i], (unsigned long long)pf.acc[i]);
  }
#endif
  FC_WG_END(a.prof_slot, by);
}

template <int MAXD, int SHD, int MODE, int NBK, bool BF = false>
__global__ __launch_bounds__(512) void k_conv_fused(FusedConvArgs a) {
  DDMI_DYN_SMEM(float, smem);
  fc_tile<MAXD, SHD, MODE, NBK, BF>(a, (int)blockIdx.x, (int)blockIdx.y, smem);
}

// LDS bytes of a workgroup that keeps max_local granule descriptors (the widest granule range of the launch)
template <int MAXD, int SHD, int MODE, int NBK>
inline size_t fc_smem_bytes(int max_local) {
  constexpr bool PACK = MODE != 1 && MAXD == 3 && SHD == 4;
  constexpr int GS2 = PACK ? 8 * MAXD + 12 : 4 * MAXD + 8, ES = SHD == 4 ? 8 : SHD + 3, CGN = FC_MAXSLOT * MAXD * SHD;
  return (size_t)(FC_VN * NC_XS + 2 * FcDim<NBK>::YB + FC_WAVES * 16 * GS2 + FC_WAVES * 2 * 32 * ES + FC_MAXG * FC_GWORDS + FC_MAXG + FC_TILE_NT + max_local * CGN) * sizeof(float);
}
inline int fc_max_local(const FusedConvArgs& a) {
  int max_local = 0;
  for (int y = 0; y < a.ysplit; ++y) max_local = std::max(max_local, a.gsplit[y + 1] - a.gsplit[y]);
  if (max_local > FC_MAXG) throw Error(DDMI_ERR_CAPACITY, "k_conv_fused: more granules per workgroup than descriptor slots (raise ddmi_config.exec.tile_split)");
  return max_local;
}

// (explicitly instantiated in the kernel TUs, declared `extern template` in k_conv.hip)
template <int MAXD, int SHD, int MODE, int NBK, bool BF = false>
void launch_conv_fused_k(const FusedConvArgs& a, hipStream_t s) {
  const size_t smem = fc_smem_bytes<MAXD, SHD, MODE, NBK>(fc_max_local(a));
  if (smem > 160 * 1024) throw Error(DDMI_ERR_CAPACITY, "k_conv_fused: LDS budget exceeded (raise ddmi_config.exec.tile_split)");
  static bool lds_opt_in = false;   // > 64 KB of dynamic LDS per workgroup needs the attribute (once per instantiation)
  if (!lds_opt_in) {
    DDMI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_fused<MAXD, SHD, MODE, NBK, BF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    lds_opt_in = true;
  }
  dim3 grid(cdiv(a.vcap, FC_VN), a.ysplit);
  hipLaunchKernelGGL((k_conv_fused<MAXD, SHD, MODE, NBK, BF>), grid, dim3(64 * FC_WAVES), smem, s, a);
  DDMI_CHECK_HIP(hipGetLastError());
}

}  // namespace ddmi
