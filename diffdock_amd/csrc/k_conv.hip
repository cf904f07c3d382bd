// The equivariant graph-convolution layer (reference TensorProductConvLayer.forward,
// models/tensor_layers.py:309-335, tp_scatter_simple/_multigroup :125-231,
// FasterTensorProduct :71-122 / e3nn FullyConnectedTensorProduct) re-associated for the
// matrix cores.  Per edge e = (target s <- gather node d) the reference computes
//     w_e  = W2 * relu(W1 * [edge_attr_e, x_s[:ns], x_d[:ns]] + b1) + b2        [weight_numel]
//     m_e  = TP(x_d, sh_e; w_e)              out_s = BN(mean_e m_e) + pad(x_s)
// which is tri-linear in (h_e = relu(..) (+) 1, x_d, sh_e).  Contracting x_d with W2 FIRST
// (per gather node, shared by all of its edges) cuts the multiply-adds per edge from
// K*weight_numel (~1.0 M at ns=48) to K*NT (~76 k), K = 3ns+1, NT = sum_paths din*mul_out:
//   k_node_contract : Y[d][k][n]  = sum_u x_d[u,i] * W2[k][slot(u,w)]          (n = (path,i,w))
//   k_edge_conv     : T[e][n]     = sum_k h_e[k] * Y[d(e)][k][n]               (MFMA 16x16x4 f32)
//                     m_e[o,w,k'] = sum_{paths,i,j} C[i][j][k'] sh_e[j] T[e][path,i,w]
//   k_reduce_bn     : deterministic segmented mean over the target-CSR, BatchNorm, residual
// Results equal the reference up to fp32 re-association.
#include "kernels.h"

namespace ddmi {

// ------------------------------------------------------------------ node pre-contraction
// grid (col blocks of 128 over HK*mul_out, node blocks of 32, path-components); 4 waves, each a
// 32 x 32 tile on v_mfma_f32_32x32x2_f32 with K = mul_in (10..48): the kernel is write-bound.
__global__ __launch_bounds__(256) void k_node_contract(const float* __restrict__ X, int gbase, int gcount,
                                                       const float* __restrict__ wpack,
                                                       const PathComp* __restrict__ pcs, int HK, int HKp, int NTs,
                                                       float* __restrict__ Y) {
  const PathComp pc = pcs[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int ncols = HK * pc.mul_out;
  const int n0 = blockIdx.x * 128 + wave * 32;
  const int m0 = blockIdx.y * 32;
  if (n0 >= ncols || m0 >= gcount) return;
  const int node = min(m0 + r, gcount - 1);
  const int col = min(n0 + r, ncols - 1);
  const float* __restrict__ xp = X + (size_t)(gbase + node) * XS + pc.x_off;
  const float* __restrict__ wp = wpack + pc.wp_off + (size_t)col * pc.mul_in_pad;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < pc.mul_in_pad; k0 += 8) {
    const int k = k0 + 4 * h;
    float av[4], bv[4];
    if (k + 3 < pc.mul_in_pad) {
      const float4 b4 = *reinterpret_cast<const float4*>(wp + k);
      bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w;
    } else {
      bv[0] = bv[1] = bv[2] = bv[3] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) av[j] = (k + j) < pc.mul_in ? xp[(k + j) * pc.din] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
  }
  const int n = n0 + r;
  if (n >= ncols) return;
  const int k = n / pc.mul_out, w = n - k * pc.mul_out;
  float* __restrict__ yp = Y + (size_t)k * NTs + pc.n_off + w;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int m = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
    if (m < gcount) yp[(size_t)m * HKp * NTs] = acc[i];
  }
}

void launch_node_contract(const float* X, int gbase, int gcount, const float* wpack, const PathComp* pcs, int n_pc,
                          int max_mul_out, int HK, int HKp, int NTs, float* Y, hipStream_t s) {
  if (gcount <= 0 || n_pc <= 0) return;
  dim3 grid(cdiv((long)HK * max_mul_out, 128), cdiv(gcount, 32), n_pc);
  hipLaunchKernelGGL(k_node_contract, grid, dim3(256), 0, s, X, gbase, gcount, wpack, pcs, HK, HKp, NTs, Y);
  DDMI_CHECK_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------ edge kernel
// One workgroup (8 waves) per gather node d; its edges are processed 16 at a time:
//   phase 1  h[16][HKp]   = relu(HE[arow] + P[tgt] + Q[d]) (+) 1            -> LDS
//   phase 2  T[16][NT]    = h * Y_d   wave w owns column super-tiles w, w+8, .. of 64 columns;
//            per k-step of 4 one ds_read (A), one 16-B global load (B, 4 column tiles) and
//            four v_mfma_f32_16x16x4_f32; accumulators -> LDS as float4 rows
//   phase 3  coupling with the edge's spherical harmonics, one thread per (edge, out block, w),
//            message written to its slot of the target-CSR (no atomics, deterministic)
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
  }
}

__global__ __launch_bounds__(512) void k_edge_conv(EdgeConvArgs a) {
  DDMI_DYN_SMEM(float, smem);
  const int HS = a.HKp + 1;                 // odd-ish row stride: conflict-free column reads
  float* hbuf = smem;                       // [16][HS]
  float* tbuf = smem + ((16 * HS + 3) & ~3);  // [16][NTs]
  float* shbuf = tbuf + 16 * a.NTs;         // [16][12]: sh(9) + ew + pad
  int* ibuf = reinterpret_cast<int*>(shbuf + 16 * 12);  // [16] tslot
  const int d = blockIdx.x;
  const int e_begin = a.goff[d], e_end = a.goff[d + 1];
  if (e_begin >= e_end) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* __restrict__ Yd = a.Y + (size_t)d * a.HKp * a.NTs;
  const float* __restrict__ Qd = a.Q + (size_t)d * a.H;
  const int n_super = (a.NT + 63) >> 6;
  for (int e0 = e_begin; e0 < e_end; e0 += 16) {
    const int ne = min(16, e_end - e0);
    // ---- phase 1
    for (int idx = tid; idx < 16 * a.HKp; idx += 512) {
      const int el = idx / a.HKp, k = idx - el * a.HKp;
      float v = 0.f;
      if (el < ne) {
        if (k < a.H) {
          const int e = e0 + el;
          const int ar = a.arow ? a.arow[e] : e;
          v = a.HE[(size_t)ar * a.H + k] + a.P[(size_t)(a.tgt[e] - a.tbase) * a.H + k] + Qd[k];
          v = v > 0.f ? v : 0.f;
        } else if (k == a.H) {
          v = 1.f;
        }
      }
      hbuf[el * HS + k] = v;
    }
    if (tid < 16) {
      float sh[9] = {1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float w = 0.f;
      int slot = 0;
      if (tid < ne) {
        const int e = e0 + tid;
        const int ar = a.arow ? a.arow[e] : e;
        edge_sh(a.nvec + (size_t)ar * 3, a.sgn, a.sh_lmax, sh);
        w = a.ew ? a.ew[ar] : 1.f;
        slot = a.tslot[e];
      }
#pragma unroll
      for (int j = 0; j < 9; ++j) shbuf[tid * 12 + j] = sh[j];
      shbuf[tid * 12 + 9] = w;
      ibuf[tid] = slot;
    }
    __syncthreads();
    // ---- phase 2
    for (int st = wave; st < n_super; st += 8) {
      const int col0 = st * 64 + 4 * (lane & 15);
      f32x4 acc[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* __restrict__ hp = hbuf + (lane & 15) * HS + (lane >> 4);
      const float* __restrict__ yp = Yd + (size_t)(lane >> 4) * a.NTs + col0;
      for (int k0 = 0; k0 < a.HKp; k0 += 4) {
        const float av = hp[k0];
        const float4 b4 = *reinterpret_cast<const float4*>(yp + (size_t)k0 * a.NTs);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b4.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b4.y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b4.z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b4.w, acc[3], 0, 0, 0);
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int row = 4 * (lane >> 4) + rr;
        *reinterpret_cast<float4*>(tbuf + row * a.NTs + col0) = make_float4(acc[0][rr], acc[1][rr], acc[2][rr], acc[3][rr]);
      }
    }
    __syncthreads();
    // ---- phase 3
    for (int idx = tid; idx < ne * a.n_items; idx += 512) {
      const int el = idx / a.n_items, it = idx - el * a.n_items;
      const CgItem item = a.items[it];
      const float* __restrict__ T = tbuf + el * a.NTs;
      const float* __restrict__ sh = shbuf + el * 12;
      float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
      for (int p = item.path_begin; p < item.path_end; ++p) {
        const DevPath P = a.paths[p];
        const float* __restrict__ C = a.ctab + P.c_off;
        for (int i = 0; i < P.din; ++i) {
          const float t = T[P.n_off + i * P.mul_out + item.w];
          for (int j = 0; j < P.ds; ++j) {
            const float ts = t * sh[P.s_off + j];
            for (int k = 0; k < P.dout; ++k) m[k] = fmaf(C[(i * P.ds + j) * P.dout + k], ts, m[k]);
          }
        }
      }
      const float w = sh[9];
      float* __restrict__ out = a.msg + (size_t)ibuf[el] * XS + item.o_off + item.w * item.dout;
      for (int k = 0; k < item.dout; ++k) out[k] = w * m[k];
    }
    __syncthreads();
  }
}

void launch_edge_conv(const EdgeConvArgs& a, hipStream_t s) {
  if (a.gcount <= 0) return;
  const size_t smem = (size_t)(((16 * (a.HKp + 1) + 3) & ~3) + 16 * a.NTs + 16 * 12 + 16) * sizeof(float);
  hipLaunchKernelGGL(k_edge_conv, dim3(a.gcount), dim3(512), smem, s, a);
  DDMI_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------- reduce + BN
__global__ __launch_bounds__(192) void k_reduce_bn(const ReduceGroup* __restrict__ groups, int n_groups, int nbase,
                                                   int D_in, int D_out, const float* __restrict__ bn_mean,
                                                   const float* __restrict__ bn_scale, const float* __restrict__ bn_bias,
                                                   int residual, const float* __restrict__ X_in,
                                                   float* __restrict__ X_out, int out_stride) {
  const int s = nbase + blockIdx.x;
  const int c = threadIdx.x;
  float acc = 0.f;
  int cnt = 0;
  for (int g = 0; g < n_groups; ++g) {
    const ReduceGroup G = groups[g];
    const int sl = s - G.tbase;
    if (sl < 0 || sl >= G.tcount) continue;
    const int b = G.toff[sl], e = G.toff[sl + 1];
    cnt += e - b;
    if (c < D_out)
      for (int r = b; r < e; ++r) acc += G.msg[(size_t)r * XS + c];
  }
  if (c >= out_stride) return;
  float v = 0.f;
  if (c < D_out) {
    v = cnt > 0 ? acc / (float)cnt : 0.f;
    if (bn_scale) v = (v - bn_mean[c]) * bn_scale[c] + bn_bias[c];
    if (residual && c < D_in) v += X_in[(size_t)s * XS + c];
  }
  X_out[(size_t)s * out_stride + c] = v;
}

void launch_reduce_bn(const ReduceGroup* groups_dev, int n_groups, int nbase, int ncount, int D_in, int D_out,
                      const float* bn_mean, const float* bn_scale, const float* bn_bias, int residual,
                      const float* X_in, float* X_out, int out_stride, hipStream_t s) {
  if (ncount <= 0) return;
  hipLaunchKernelGGL(k_reduce_bn, dim3(ncount), dim3(192), 0, s, groups_dev, n_groups, nbase, D_in, D_out, bn_mean,
                     bn_scale, bn_bias, residual, X_in, X_out, out_stride);
  DDMI_CHECK_HIP(hipGetLastError());
}

}  // namespace ddmi
