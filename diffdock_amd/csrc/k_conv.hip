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
// Workgroup = 16 gather nodes x KC consecutive k (rows of W2^T incl. the bias row).  Per k the four waves
// share the (path, 16-wide w tile) work items: A = x rows from LDS (one fragment per input component i),
// B = the k-th slab of the packed second-layer weights (L2-resident, 4 MB per edge group), v_mfma_f32_16x16x4_f32,
// results scattered into an LDS row image in the item-major column order and then streamed out as 256-B runs
// Y[node][super-tile][k][64] -- the exact order k_edge_conv reads them back.
constexpr int NC_NODES = 16, NC_KC = 5, NC_XS = XS + 1;

__global__ __launch_bounds__(256) void k_node_contract(const float* __restrict__ X, int gbase, int gcount,
                                                       const float* __restrict__ wpack,
                                                       const NcItem* __restrict__ items, int n_items, int KS, int HK,
                                                       int HKp, int NTs, float* __restrict__ Y) {
  DDMI_DYN_SMEM(float, smem);
  float* xbuf = smem;                                   // [16][XS+1]
  float* obuf = smem + ((NC_NODES * NC_XS + 3) & ~3);   // [16][NTs + 4] (row shift of 4 banks)
  const int OS = NTs + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int node0 = blockIdx.x * NC_NODES;
  const int n_super = NTs >> 6;
  for (int idx = tid; idx < NC_NODES * XS; idx += 256) {
    const int nl = idx / XS, c = idx - nl * XS;
    xbuf[nl * NC_XS + c] = (node0 + nl) < gcount ? X[(size_t)(gbase + node0 + nl) * XS + c] : 0.f;
  }
  for (int idx = tid; idx < NC_NODES * OS; idx += 256) obuf[idx] = 0.f;
  __syncthreads();
  const int lr = lane & 15, lq = lane >> 4;
  for (int kk = 0; kk < NC_KC; ++kk) {
    const int k = blockIdx.y * NC_KC + kk;
    if (k >= HK) break;
    const float* __restrict__ slab = wpack + (size_t)k * KS;
    for (int it = wave; it < n_items; it += 4) {
      const NcItem I = items[it];
      f32x4 acc[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* __restrict__ bp = slab + I.wk_off + (size_t)lq * I.w_pad + I.w0 + lr;
      const float* __restrict__ xp = xbuf + lr * NC_XS + I.x_off;
      // all B fragments of the item are requested before the first MFMA (u_pad <= 64): one L2 round trip per item
      for (int ub = 0; ub < I.u_pad; ub += 64) {
        float bv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) bv[j] = (ub + 4 * j) < I.u_pad ? bp[(size_t)(ub + 4 * j) * I.w_pad] : 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          if (ub + 4 * j >= I.u_pad) break;
          const int u = ub + 4 * j + lq;
          const bool ok = u < I.mul_in;
          for (int i = 0; i < I.din; ++i) {
            const float a = ok ? xp[u * I.din + i] : 0.f;
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[j], acc[i], 0, 0, 0);
          }
        }
      }
      if (lr < I.n_w) {
        float* __restrict__ op = obuf + I.col_base + (I.w0 + lr) * I.itemw;
        for (int i = 0; i < I.din; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) op[(4 * lq + r) * OS + i] = acc[i][r];
      }
    }
    __syncthreads();
    for (int idx = tid; idx < NC_NODES * (NTs >> 2); idx += 256) {
      const int nl = idx / (NTs >> 2), q = idx - nl * (NTs >> 2);
      const int node = node0 + nl;
      if (node >= gcount) continue;
      const int col = q << 2, st = col >> 6, c = col & 63;
      *reinterpret_cast<float4*>(Y + (((size_t)node * n_super + st) * HKp + k) * 64 + c) =
          *reinterpret_cast<const float4*>(obuf + nl * OS + col);
    }
    __syncthreads();
  }
}

void launch_node_contract(const float* X, int gbase, int gcount, const float* wpack, const NcItem* items, int n_items,
                          int KS, int HK, int HKp, int NTs, float* Y, hipStream_t s) {
  if (gcount <= 0 || n_items <= 0) return;
  const size_t smem = (size_t)(((NC_NODES * NC_XS + 3) & ~3) + NC_NODES * (NTs + 4)) * sizeof(float);
  dim3 grid(cdiv(gcount, NC_NODES), cdiv(HK, NC_KC));
  hipLaunchKernelGGL(k_node_contract, grid, dim3(256), smem, s, X, gbase, gcount, wpack, items, n_items, KS, HK, HKp, NTs, Y);
  DDMI_CHECK_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------ edge kernel
// One workgroup per (gather node d, split); edges are processed 32 at a time (two 16-row MFMA tiles):
//   phase 1  h[32][HKp] = relu(HE[arow] + P[tgt] + Q[d]) (+) 1, per-edge coupling vectors
//            G[e][path][i][k'] = sum_j C[i][j][k'] sh_e[j]                                       -> LDS
//   phase 2  wave w owns the 64-column super-tiles w, w+W, ..: T = h * Y_d on v_mfma_f32_16x16x4_f32, Y_d streamed
//            from HBM in its storage order with a 2-deep register prefetch (one 16-B load feeds 8 MFMAs); the
//            accumulators never leave registers: lane (l&15) of a 16-lane group holds one quad of an item for 4
//            edges, contracts it with G, sums the item's quads with wave shuffles -> message image in LDS
//   phase 3  message rows streamed to their slots of the target-ordered buffer (no atomics, deterministic)
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

constexpr int EC_E = 32;   // edges per pass

__global__ __launch_bounds__(768) void k_edge_conv(EdgeConvArgs a) {
  DDMI_DYN_SMEM(float, smem);
  const int HS = a.HKp + 1;                          // odd row stride: conflict-free A-fragment reads
  const int GS = a.GN | 1, MS = a.D_out | 1;
  float* hbuf = smem;                                // [32][HS]
  float* gbuf = hbuf + EC_E * HS;                    // [32][GS]
  float* mbuf = gbuf + EC_E * GS;                    // [32][MS]
  float* wbuf = mbuf + EC_E * MS;                    // [32] edge weight
  int* ibuf = reinterpret_cast<int*>(wbuf + EC_E);   // [32] tslot
  const int d = blockIdx.x;
  const int e_begin = a.goff[d], e_end = a.goff[d + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwave = nthr >> 6;
  const int n_super = a.NTs >> 6;
  const float* __restrict__ Yd = a.Y + (size_t)d * n_super * a.HKp * 64;
  const float* __restrict__ Qd = a.Q + (size_t)d * a.H;
  const int lr = lane & 15, lq = lane >> 4;
  int pass = 0;
  for (int e0 = e_begin; e0 < e_end; e0 += EC_E, ++pass) {
    if (pass % a.esplit != (int)blockIdx.y) continue;
    const int ne = min(EC_E, e_end - e0);
    // ---- phase 1
    for (int idx = tid; idx < EC_E * a.HKp; idx += nthr) {
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
    for (int idx = tid; idx < ne * a.GN; idx += nthr) {
      const int el = idx / a.GN, g = idx - el * a.GN;
      const int e = e0 + el;
      const int ar = a.arow ? a.arow[e] : e;
      float sh[9];
      edge_sh(a.nvec + (size_t)ar * 3, a.sgn, a.sh_lmax, sh);
      const GEntry G = a.gmap[g];
      float acc = 0.f;
      for (int j = 0; j < G.ds; ++j) acc = fmaf(a.ctab[G.c_idx + j * G.dout], sh[G.s_off + j], acc);
      gbuf[el * GS + g] = acc;
    }
    for (int idx = tid; idx < EC_E * a.D_out; idx += nthr) mbuf[(idx / a.D_out) * MS + (idx % a.D_out)] = 0.f;
    if (tid < EC_E) {
      float w = 0.f;
      int slot = 0;
      if (tid < ne) {
        const int e = e0 + tid;
        const int ar = a.arow ? a.arow[e] : e;
        w = a.ew ? a.ew[ar] : 1.f;
        slot = a.tslot[e];
      }
      wbuf[tid] = w;
      ibuf[tid] = slot;
    }
    __syncthreads();
    // ---- phase 2
    const bool two = ne > 16;
    for (int st = wave; st < n_super; st += nwave) {
      const int col0 = st * 64 + 4 * lr;
      f32x4 acc0[4], acc1[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) { acc0[c] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      const float* __restrict__ hp0 = hbuf + lr * HS + lq;
      const float* __restrict__ hp1 = hp0 + 16 * HS;
      const float* __restrict__ yp = Yd + ((size_t)st * a.HKp + lq) * 64 + 4 * lr;
      // Y_d super-tile streamed in blocks of 4 k-steps (4 x 16 B per lane), next block in flight during the 32 MFMAs
      const int nsteps = a.HKp >> 2;
      float4 cur[4], nxt[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) cur[j] = j < nsteps ? *reinterpret_cast<const float4*>(yp + (size_t)j * 256) : make_float4(0.f, 0.f, 0.f, 0.f);
      for (int s0 = 0; s0 < nsteps; s0 += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          nxt[j] = (s0 + 4 + j) < nsteps ? *reinterpret_cast<const float4*>(yp + (size_t)(s0 + 4 + j) * 256) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (s0 + j >= nsteps) break;
          const float4 b = cur[j];
          const float a0 = hp0[(s0 + j) * 4];
          acc0[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b.x, acc0[0], 0, 0, 0);
          acc0[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b.y, acc0[1], 0, 0, 0);
          acc0[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b.z, acc0[2], 0, 0, 0);
          acc0[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b.w, acc0[3], 0, 0, 0);
          if (two) {
            const float a1 = hp1[(s0 + j) * 4];
            acc1[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b.x, acc1[0], 0, 0, 0);
            acc1[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b.y, acc1[1], 0, 0, 0);
            acc1[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b.z, acc1[2], 0, 0, 0);
            acc1[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b.w, acc1[3], 0, 0, 0);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
      }
      // coupling in registers: this lane's quad = columns col0 .. col0+3
      int ob = -1;
      for (int b = 0; b < a.n_ob; ++b)
        if (col0 >= a.obs[b].base && col0 < a.obs[b].base + a.obs[b].mul * a.obs[b].itemw) ob = b;
      ObInfo O{0, 4, 0, 0, 1};
      int w = 0, qi = 0;
      QuadDesc qd{{-1, -1, -1, -1}, {0, 0, 0, 0}};
      if (ob >= 0) {
        O = a.obs[ob];
        const int rel = col0 - O.base;
        w = rel / O.itemw;
        qi = (rel - w * O.itemw) >> 2;
        qd = a.qdesc[ob * 4 + qi];
      }
      int goffs[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) goffs[c] = qd.path[c] >= 0 ? a.paths[qd.path[c]].g_off + qd.comp[c] * O.dout : -1;
      const int nq = O.itemw >> 2;
      for (int rt = 0; rt < (two ? 2 : 1); ++rt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int el = rt * 16 + 4 * lq + r;
          float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
          const float* __restrict__ G = gbuf + el * GS;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (goffs[c] < 0) continue;
            const float t = rt == 0 ? acc0[c][r] : acc1[c][r];
            for (int k = 0; k < O.dout; ++k) m[k] = fmaf(G[goffs[c] + k], t, m[k]);
          }
          for (int k = 0; k < a.maxd; ++k) {      // sum the item's quads (items never straddle a 16-lane group)
            const float m1 = __shfl_down(m[k], 1, 64);
            if (nq >= 2) m[k] += m1;
            const float m2 = __shfl_down(m[k], 2, 64);
            if (nq >= 4) m[k] += m2;
          }
          if (ob >= 0 && qi == 0 && el < ne) {
            float* __restrict__ mp = mbuf + el * MS + O.o_off + w * O.dout;
            const float we = wbuf[el];
            for (int k = 0; k < O.dout; ++k) mp[k] = we * m[k];
          }
        }
      }
    }
    __syncthreads();
    // ---- phase 3
    for (int idx = tid; idx < ne * a.D_out; idx += nthr) {
      const int el = idx / a.D_out, c = idx - el * a.D_out;
      a.msg[(size_t)ibuf[el] * XS + c] = mbuf[el * MS + c];
    }
    __syncthreads();
  }
}

void launch_edge_conv(const EdgeConvArgs& a, hipStream_t s) {
  if (a.gcount <= 0) return;
  const int HS = a.HKp + 1, GS = a.GN | 1, MS = a.D_out | 1;
  const size_t smem = (size_t)(EC_E * (HS + GS + MS) + 2 * EC_E) * sizeof(float);
  const int n_super = a.NTs >> 6;
  const int waves = n_super < 12 ? n_super : 12;
  hipLaunchKernelGGL(k_edge_conv, dim3(a.gcount, a.esplit), dim3(64 * waves), smem, s, a);
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
