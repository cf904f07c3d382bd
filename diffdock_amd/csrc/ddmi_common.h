// Shared host/device helpers for libddmi (gfx950).  Built with hipcc --offload-arch=gfx950;
// the CPU test-suite builds the same sources against tests/hipemu (DDMI_HIPEMU).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

#include "../../include/ddmi.h"

#ifdef DDMI_HIPEMU
typedef f32x4_emu f32x4;
typedef f32x16_emu f32x16;
#define DDMI_DYN_SMEM(type, name) type* name = reinterpret_cast<type*>(hipemu_dyn_smem())
#else
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define DDMI_DYN_SMEM(type, name)                                        \
  extern __shared__ __attribute__((aligned(16))) unsigned char ddmi_dyn_smem_[]; \
  type* name = reinterpret_cast<type*>(ddmi_dyn_smem_)
#endif

// streaming (read-once / write-once) accesses of the contracted rows Y: keep them from evicting the L2-resident
// operands (packed weights, first-layer rows) -- `nt` cache policy on gfx950
#ifdef DDMI_HIPEMU
#define DDMI_NT_STORE(val, ptr) (*(ptr) = (val))
#define DDMI_NT_LOAD(ptr) (*(ptr))
#define DDMI_WAVE_SYNC() hipemu_wave_sync()
#define DDMI_UNIFORM(x) (x)
#define DDMI_SCHED_FENCE() ((void)0)
#define DDMI_WAIT_VMEM() ((void)0)
#define DDMI_ROW_XOR8(v) __shfl_xor((v), 8, 64)
#define DDMI_OPAQUE(x) ((void)0)
#define DDMI_SWAP32(a, b, WAIT)                                                        \
  do {                                                                                 \
    const float ta_ = __shfl_xor((a), 32, 64), tb_ = __shfl_xor((b), 32, 64);          \
    const bool up_ = (threadIdx.x & 32) != 0;                                          \
    const float na_ = up_ ? tb_ : (a), nb_ = up_ ? (b) : ta_;                          \
    (a) = na_; (b) = nb_;                                                              \
  } while (0)
#define DDMI_SWAP16(a, b, WAIT)                                                        \
  do {                                                                                 \
    const float ta_ = __shfl_xor((a), 16, 64), tb_ = __shfl_xor((b), 16, 64);          \
    const bool odd_ = (threadIdx.x & 16) != 0;                                         \
    const float na_ = odd_ ? tb_ : (a), nb_ = odd_ ? (b) : ta_;                        \
    (a) = na_; (b) = nb_;                                                              \
  } while (0)
#else
// the compiler may not assume anything about x past this point (keeps loop-invariant address arithmetic inside the loop)
#define DDMI_OPAQUE(x) asm volatile("" : "+v"(x))
// wave-uniform value -> scalar register
#define DDMI_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
// LDS hand-off between lanes of ONE wave (lock-step on the hardware: order the ds ops, keep the compiler from moving them)
#define DDMI_WAVE_SYNC()                                                                     \
  do {                                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* LDS only: global stores stay in flight */ \
    __builtin_amdgcn_wave_barrier();                                                         \
  } while (0)
// nothing is scheduled across this point (hand-placed issue order of the MFMA main loops)
#ifdef FCV_NOFENCE   // timing experiment: the compiler's own order instead of the hand-placed one
#define DDMI_SCHED_FENCE() ((void)0)
#else
#define DDMI_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
// every outstanding vector-memory request of the wave has completed (vmcnt(0); expcnt / lgkmcnt untouched) -- a real
// S_WAITCNT, which the compiler's own wait-count insertion takes into account
#define DDMI_WAIT_VMEM() __builtin_amdgcn_s_waitcnt(0x0F70)
// value of lane ^ 8 (the other half of the lane's row of 16): DPP row rotate by 8, no LDS traffic
#define DDMI_ROW_XOR8(v) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (float)(v)), 0x128, 0xf, 0xf, false))
// gfx950 half / row swaps of two registers (v_permlane32_swap / v_permlane16_swap; semantics checked on the MI355X,
// tools/probe/mfma4x4.hip, rsc_probe.hip).  SWAP32: a <- [a.lanes 0-31 | b.lanes 0-31], b <- [a.lanes 32-63 | b.lanes 32-63].
// SWAP16 (rows of 16 lanes): a <- [a.row0 | b.row0 | a.row2 | b.row2], b <- [a.row1 | b.row1 | a.row3 | b.row3].
// Inline assembly: the two-result builtins of this toolchain (__builtin_amdgcn_permlane16_swap / 32_swap) hand back the first
// result twice.  The hazard recognizer does not look inside; WAIT = wait states the operands' producers still need
// (7 covers a 2-pass MFMA result, 1 a VALU result).
#define DDMI_SWAP32(a, b, WAIT) asm volatile("s_nop " #WAIT "\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b))
#define DDMI_SWAP16(a, b, WAIT) asm volatile("s_nop " #WAIT "\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b))
#define DDMI_NT_STORE(val, ptr) __builtin_nontemporal_store((val), (ptr))
#define DDMI_NT_LOAD(ptr) __builtin_nontemporal_load((ptr))
#endif

namespace ddmi {

constexpr int WAVE = 64;
constexpr int XS = 160;  // row stride (floats) of node feature tables, >= largest irreps dim, 16B-aligned rows

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define DDMI_CHECK_HIP(expr)                                                                     \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      throw ::ddmi::Error(DDMI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
  } while (0)

#define DDMI_REQUIRE(cond, code, msg)                      \
  do {                                                     \
    if (!(cond)) throw ::ddmi::Error((code), (msg));       \
  } while (0)

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
inline long round_up(long a, long b) { return (a + b - 1) / b * b; }

// ---- split-bf16 edge product (ddmi_config.edge_product = 1).  A float v is carried as ONE 32-bit word: bf16(v) in the upper
// half, bf16(v - bf16(v)) in the lower half (both round-to-nearest-even: 16 significand bits, |error| <= 2^-17 |v|).  With
// a = ah + al and b = bh + bl the four products ah*bh + al*bh + ah*bl + al*bl fill four K slots of v_mfma_f32_16x16x32_bf16
// (products exact, f32 accumulation): A registers (wa, wa), B registers (bh|bh, bl|bl) per k, so one instruction multiplies an
// 8-row k chunk (two k per lane group) -- the work of two v_mfma_f32_16x16x4_f32 (64 cycles) in 16.
#ifdef DDMI_HIPEMU
typedef u32x4_emu u32x4;
static inline unsigned bf_rne16(float f) {   // bf16 bits of f, round to nearest even (finite inputs)
  unsigned u; memcpy(&u, &f, 4);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
static inline unsigned bf_pk(float a, float b) { return (bf_rne16(a) & 0xffffu) | (bf_rne16(b) << 16); }
static inline unsigned bf_perm(unsigned s0, unsigned s1, unsigned sel) {
  const unsigned long long v = ((unsigned long long)s0 << 32) | s1;
  unsigned r = 0;
  for (int i = 0; i < 4; ++i) { const unsigned b = (sel >> (8 * i)) & 0xffu; r |= (b < 8 ? (unsigned)((v >> (8 * b)) & 0xffu) : 0u) << (8 * i); }
  return r;
}
static inline f32x4 bf_mfma_raw(u32x4 a, u32x4 b, f32x4 c) { return hipemu_mfma_f32_16x16x32_bf16(a, b, c); }
static inline float bf_sub(float a, float b) { return a - b; }
#else
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 ddmi_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 ddmi_bf16x8 __attribute__((ext_vector_type(8)));
typedef float ddmi_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bf_pk(float a, float b) {   // v_cvt_pk_bf16_f32: (bf16(a) | bf16(b) << 16)
  return __builtin_bit_cast(unsigned, __builtin_convertvector(ddmi_f32x2{a, b}, ddmi_bf16x2));
}
__device__ __forceinline__ unsigned bf_perm(unsigned s0, unsigned s1, unsigned sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
__device__ __forceinline__ f32x4 bf_mfma_raw(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ddmi_bf16x8, a), __builtin_bit_cast(ddmi_bf16x8, b), c, 0, 0, 0);
}
// a - b as ONE v_sub_f32: left to itself the compiler pairs the subtractions of a split into v_pk_add_f32 (plus the register
// copies that pair the operands), which costs issue slots next to MFMAs (MI355X_MICROARCH.md: packed f32 VALU is an anti-lever)
__device__ __forceinline__ float bf_sub(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#endif
__device__ __forceinline__ float bf_word_f(unsigned w) { return __uint_as_float(w); }
__device__ __forceinline__ unsigned bf_f_word(float f) { return __float_as_uint(f); }
// packed words of two / one float(s)
__device__ __forceinline__ void bf_split2(float v1, float v2, float& w1, float& w2) {
  const unsigned hp = bf_pk(v1, v2);
  const unsigned h1 = hp << 16, h2 = hp & 0xffff0000u;
  const unsigned lp = bf_pk(bf_sub(v1, bf_word_f(h1)), bf_sub(v2, bf_word_f(h2)));
  w1 = bf_word_f((lp & 0xffffu) | h1);
  w2 = bf_word_f(bf_perm(hp, lp, 0x07060302u));   // bytes [hp.3, hp.2, lp.3, lp.2]
}
__device__ __forceinline__ float bf_split1(float v) {
  const unsigned h = bf_pk(v, v) << 16;
  const unsigned lp = bf_pk(bf_sub(v, bf_word_f(h)), 0.f);
  return bf_word_f((lp & 0xffffu) | h);
}
// c += sum over the lane group's two k of a(k) * b(k): wa0, wa1 = packed A words, q0, q1 = packed B words of k0, k1
__device__ __forceinline__ f32x4 bf_mfma(float wa0, float wa1, float q0, float q1, f32x4 c) {
  const unsigned a0 = bf_f_word(wa0), a1 = bf_f_word(wa1), b0 = bf_f_word(q0), b1 = bf_f_word(q1);
  const u32x4 A = {a0, a0, a1, a1};
  const u32x4 B = {bf_perm(b0, b0, 0x03020302u), bf_perm(b0, b0, 0x01000100u), bf_perm(b1, b1, 0x03020302u), bf_perm(b1, b1, 0x01000100u)};
  return bf_mfma_raw(A, B, c);
}

// wave-wide sum (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

}  // namespace ddmi
