// fp32 GEMM on the f32-input matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chain).
//   C[M,N] = act(A[M,K] * W[N,K]^T + bias + rowbias[ridx[m]])
// Serves every dense layer of the path that is not the fused edge contraction: node
// encoders, edge-embedding second layers, the per-layer edge/node halves of the first
// tensor-product-weight layer (reference models/tensor_layers.py:140,211 `fc_layer`, split
// algebraically -- DESIGN.md), read-out MLPs.  K here is 48..1328 and the operands are
// L2-resident, so operands go straight to registers: lane l of a wave owns row (l&31) of
// its 32x32 tile for both A and W and loads 4 consecutive k (16 B) per step; the k <-> lane
// pairing of the MFMA (lanes 0-31: k, lanes 32-63: k') is free as long as A and B agree.
#include <algorithm>

#include "kernels.h"

namespace ddmi {

template <bool VEC>
__device__ __forceinline__ void gemm_tile(const GemmArgs& a, int bx, int by) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  int M = a.M;
  if (a.m_dev) { int mv = *a.m_dev; M = mv < M ? mv : M; }
  const int m0 = by * 64 + (wave >> 1) * 32;
  const int n0 = bx * 64 + (wave & 1) * 32;
  if (m0 >= M || n0 >= a.N) return;  // wave-uniform
  const int row = min(m0 + r, M - 1), col = min(n0 + r, a.N - 1);
  const float* __restrict__ Ap = a.A + (size_t)row * a.lda;
  const float* __restrict__ Wp = a.W + (size_t)col * a.ldw;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int K = a.K;
  for (int k0 = 0; k0 < K; k0 += 8) {
    const int k = k0 + 4 * h;
    float av[4], bv[4];
    if (VEC && k + 3 < K) {
      const float4 a4 = *reinterpret_cast<const float4*>(Ap + k);
      const float4 b4 = *reinterpret_cast<const float4*>(Wp + k);
      av[0] = a4.x; av[1] = a4.y; av[2] = a4.z; av[3] = a4.w;
      bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = (k + j) < K;
        av[j] = ok ? Ap[k + j] : 0.f;
        bv[j] = ok ? Wp[k + j] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
  }
  const int c = n0 + r;
  if (c >= a.N) return;
  const float bias = a.bias ? a.bias[c] : 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int m = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
    if (m < M) {
      float v = acc[i] + bias;
      if (a.rowbias) v += a.rowbias[(size_t)(a.ridx ? a.ridx[m] : 0) * a.ldrb + c];
      if (a.act == 1) v = v > 0.f ? v : 0.f;
      else if (a.act == 2) v = tanhf(v);
      a.C[(size_t)m * a.ldc + c] = v;
    }
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void k_gemm_nt(GemmArgs a) { gemm_tile<VEC>(a, blockIdx.x, blockIdx.y); }
// up to GEMM_BATCH_MAX independent small GEMMs in one launch (blockIdx.z = problem): the per-node terms of one edge group's first layer
template <bool VEC>
__global__ __launch_bounds__(256) void k_gemm_nt_batch(GemmBatch b) {
  const GemmArgs& a = b.g[blockIdx.z];
  if ((int)blockIdx.x * 64 >= a.N || (int)blockIdx.y * 64 >= a.M) return;
  gemm_tile<VEC>(a, blockIdx.x, blockIdx.y);
}

void launch_gemm_batch(const GemmBatch& b, hipStream_t s) {
  int mx = 0, my = 0;
  bool vec = true;
  for (int i = 0; i < b.n; ++i) {
    const GemmArgs& a = b.g[i];
    mx = std::max(mx, cdiv(a.N, 64)); my = std::max(my, cdiv(a.M, 64));
    vec = vec && (a.lda % 4 == 0) && (a.ldw % 4 == 0) && (((uintptr_t)a.A | (uintptr_t)a.W) % 16 == 0);
  }
  if (b.n <= 0 || mx <= 0 || my <= 0) return;
  dim3 grid(mx, my, b.n);
  if (vec) hipLaunchKernelGGL(k_gemm_nt_batch<true>, grid, dim3(256), 0, s, b);
  else hipLaunchKernelGGL(k_gemm_nt_batch<false>, grid, dim3(256), 0, s, b);
  DDMI_CHECK_HIP(hipGetLastError());
}

void launch_gemm(const GemmArgs& a, hipStream_t s) {
  if (a.M <= 0 || a.N <= 0) return;
  dim3 grid(cdiv(a.N, 64), cdiv(a.M, 64));
  const bool vec = (a.lda % 4 == 0) && (a.ldw % 4 == 0) && (((uintptr_t)a.A | (uintptr_t)a.W) % 16 == 0);
  if (vec) hipLaunchKernelGGL(k_gemm_nt<true>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(k_gemm_nt<false>, grid, dim3(256), 0, s, a);
  DDMI_CHECK_HIP(hipGetLastError());
}

}  // namespace ddmi
