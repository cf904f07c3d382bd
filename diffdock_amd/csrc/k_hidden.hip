// Hidden rows of the edge MLP (first Linear + ReLU of every TensorProductConvLayer's FCBlock, models/tensor_layers.py:140,211 /
// models/layers.py:10-17) written in the A-fragment order k_conv_fused streams them in (fc_layout.h).
//   k_edge_hidden_mm : the first Linear on the matrix cores straight from the edge attributes (ddmi_exec_options.hidden_mm = 0)
//   k_edge_hidden    : re-ordering pass behind plain GEMMs (hidden_mm = 1, deeper edge MLPs);  k_edge_rows: plain rows for those
#include <algorithm>

#include "kernels.h"
#include "fc_layout.h"

namespace ddmi {

typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float* p) {
  const vf4 v = DDMI_NT_LOAD(reinterpret_cast<const vf4*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}

__global__ __launch_bounds__(256) void k_edge_hidden(const int* __restrict__ nvn, const int* __restrict__ vn_node,
                                                    const int* __restrict__ vn_e0, const int* __restrict__ goff,
                                                    const int* __restrict__ arow, const int* __restrict__ tgt, int tbase,
                                                    const float* __restrict__ HE, const float* __restrict__ P,
                                                    const float* __restrict__ Q, int H, int NG8, float* __restrict__ Hb, int bf) {
  const int v = blockIdx.x;
  if (v >= *nvn) return;
  const int d = vn_node[v], e0 = vn_e0[v];
  const int ne = min(32, goff[d + 1] - e0);
  const int NGP = fc_ngp(NG8), q4 = 4 * NGP;   // 4-k pieces per row, padded to whole group pairs
  for (int idx = threadIdx.x; idx < 32 * q4; idx += blockDim.x) {
    const int el = idx / q4, k = 4 * (idx - el * q4);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (el < ne && k < H) {
      const int e = e0 + el;
      const int ar = arow ? arow[e] : e;
      const float4 x = nt_load4(HE + (size_t)ar * H + k);
      if (P) {   // (P == nullptr: HE already holds the finished hidden rows of a deeper edge MLP, k_edge_rows + GEMMs)
        const float4 p = *reinterpret_cast<const float4*>(P + (size_t)(tgt[e] - tbase) * H + k);
        const float4 q = *reinterpret_cast<const float4*>(Q + (size_t)d * H + k);
        o.x = fmaxf(x.x + p.x + q.x, 0.f); o.y = fmaxf(x.y + p.y + q.y, 0.f);
        o.z = fmaxf(x.z + p.z + q.z, 0.f); o.w = fmaxf(x.w + p.w + q.w, 0.f);
      } else {
        o = x;
      }
    }
    if (bf) { bf_split2(o.x, o.y, o.x, o.y); bf_split2(o.z, o.w, o.z, o.w); }   // split-bf16 edge product: packed (hi | lo) words
    const int rt = el >> 4, r = el & 15, g = k >> 3, q0 = (k & 7) >> 1;
    *reinterpret_cast<float2*>(Hb + fc_hb_off(v, rt, g, q0 * 16 + r, NGP)) = make_float2(o.x, o.y);
    *reinterpret_cast<float2*>(Hb + fc_hb_off(v, rt, g, (q0 + 1) * 16 + r, NGP)) = make_float2(o.z, o.w);
  }
}
// First hidden layer of a deeper edge MLP (tp_weights_layers > 2, models/layers.py:10-17), one plain row per edge in gather
// order: rows[e] = relu(HE[arow[e]] + P[tgt[e]] + Q[d]); the hidden Linear layers then are ordinary GEMMs over these rows and
// k_edge_hidden (P == nullptr) only re-orders the last one into fragment order.
__global__ __launch_bounds__(256) void k_edge_rows(const int* __restrict__ nvn, const int* __restrict__ vn_node,
                                                  const int* __restrict__ vn_e0, const int* __restrict__ goff,
                                                  const int* __restrict__ arow, const int* __restrict__ tgt, int tbase,
                                                  const float* __restrict__ HE, const float* __restrict__ P,
                                                  const float* __restrict__ Q, int H, float* __restrict__ rows) {
  const int v = blockIdx.x;
  if (v >= *nvn) return;
  const int d = vn_node[v], e0 = vn_e0[v];
  const int ne = min(32, goff[d + 1] - e0);
  for (int idx = threadIdx.x; idx < ne * H; idx += blockDim.x) {
    const int el = idx / H, k = idx - el * H, e = e0 + el;
    const int ar = arow ? arow[e] : e;
    rows[(size_t)e * H + k] = fmaxf(HE[(size_t)ar * H + k] + P[(size_t)(tgt[e] - tbase) * H + k] + Q[(size_t)d * H + k], 0.f);
  }
}
void launch_edge_rows(const int* nvn, int vcap, const int* vn_node, const int* vn_e0, const int* goff, const int* arow,
                      const int* tgt, int tbase, const float* HE, const float* P, const float* Q, int H, float* rows, hipStream_t s) {
  if (vcap <= 0) return;
  hipLaunchKernelGGL(k_edge_rows, dim3(vcap), dim3(256), 0, s, nvn, vn_node, vn_e0, goff, arow, tgt, tbase, HE, P, Q, H, rows);
  DDMI_CHECK_HIP(hipGetLastError());
}
void launch_edge_hidden(const int* nvn, int vcap, const int* vn_node, const int* vn_e0, const int* goff, const int* arow,
                        const int* tgt, int tbase, const float* HE, const float* P, const float* Q, int H, int NG8,
                        float* Hb, hipStream_t s, int bf) {
  if (vcap <= 0) return;
  hipLaunchKernelGGL(k_edge_hidden, dim3(vcap), dim3(256), 0, s, nvn, vn_node, vn_e0, goff, arow, tgt, tbase, HE, P, Q, H,
                     NG8, Hb, bf);
  DDMI_CHECK_HIP(hipGetLastError());
}

__device__ __forceinline__ float f4c(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// One wave per virtual node (grid-stride), the edge-attribute block of the first Linear in LDS.  The product is taken
// transposed -- D[hidden k][edge row] = W[k][:] . attr[row][:] with the weights as the A operand -- so that lane
// (row, q) ends up with 4 consecutive hidden values of ITS row: exactly two (k pair) slots of the fragment order
// Hb[v][rt][g][16q' + row][sub], k = 8g + 2q' + sub.  The target-node term P[tgt], Q[d] (+ the per-graph term) are
// added as 16-B row pieces, relu applied, and each lane writes two 8-B pieces (16 lanes = 128 contiguous bytes).
// k-permuted MFMA steps: lane (row, q) fetches floats [q*ns/4, (q+1)*ns/4) of its attribute row with 16-B loads; step t
// multiplies attr[row][q*ns/4 + t] with W[k][q*ns/4 + t].
template <int NSQ>   // ns = 16 * NSQ, H = 3 * ns; workgroup `block` of `nblocks` of the edge group described by `a`
__device__ __forceinline__ void eh_body(const EdgeHiddenArgs& a, const int block, const int nblocks, float* __restrict__ smem) {
  constexpr int KS = 4 * NSQ;                        // MFMA steps = floats per lane quarter
  constexpr int H = 48 * NSQ, NB = H / 16, NG8 = H / 8;
  constexpr int HP = H + 1;                          // odd row stride: the staging writes (consecutive threads = consecutive rows) spread over the banks
  float* wl = smem;                                  // [KS][4][HP]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = DDMI_UNIFORM(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int nvn = *a.nvn;
  if (block * 4 >= nvn) return;
  // a.W1 is the permuted copy of the first layer (weights.cpp): output position 4a + i of a 16-block holds hidden unit
  // 8 (i >> 1) + 2a + (i & 1), so lane (row, quarter a) ends with k = 8g + 2a + {0, 1} of BOTH 8-k groups g = 2nb, 2nb + 1 --
  // the float4 of fragment lane 16a + row; P, Q and the sigma rows arrive in the same order.
  {   // k fastest: coalesced reads of the weight rows; every request of the thread is issued before the first LDS store (the
      // rolled loop was a chain of 27 request -> store round trips per workgroup: ~a quarter of the kernel)
    constexpr int NW = (H * 4 * KS + 255) / 256;
    float wreg[NW];
#pragma unroll
    for (int it = 0; it < NW; ++it) {
      const int idx = tid + 256 * it;
      wreg[it] = idx < H * 4 * KS ? a.W1[(size_t)(idx / (4 * KS)) * a.ldw + idx % (4 * KS)] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < NW; ++it) {
      const int idx = tid + 256 * it;
      const int k = idx % (4 * KS), n = idx / (4 * KS);
      const int q = k / KS, t = k - q * KS;
      if (idx < H * 4 * KS) wl[(t * 4 + q) * HP + n] = wreg[it];
    }
  }
  __syncthreads();
  // (Round 5, measured and dropped: the header of a virtual node -- gather node, edge count, attribute / target rows of both row
  // tiles -- requested one virtual node ahead, two dependent round trips per node instead of five: 1.49 ms per forward either
  // way, profiles/r05_e13_ab.txt.  The kernel moves 215 MB out and 125 MB in per large launch at 3.95 TB/s; what is left is the
  // write-heavy stream itself, not the request chain.)
  for (int v = block * 4 + wave; v < nvn; v += nblocks * 4) {
    const int d = a.vn_node[v], e0 = a.vn_e0[v];
    const int ne = a.vn_ne ? a.vn_ne[v] : min(32, a.goff[d + 1] - e0);
    // All requests of a (virtual node, row tile) are issued before the first use and nothing in the tile body branches:
    // a load -> wait -> MFMA -> store chain per 16 hidden units made this kernel latency-bound (0.24 of the HBM write
    // roofline in round 1).  Rows past the node's edge count read the tile's first edge (valid memory) and store zeros.
    float4 qv[NB];
    {
      const float* __restrict__ qrow = a.Q + (size_t)d * H + 4 * lq;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) qv[nb] = *reinterpret_cast<const float4*>(qrow + 16 * nb);
      if (a.rowbias && ne > 0) {   // wave-uniform
        const float* __restrict__ rbrow = a.rowbias + (size_t)a.ridx[a.arow ? a.arow[e0] : e0] * H + 4 * lq;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const float4 t = *reinterpret_cast<const float4*>(rbrow + 16 * nb);
          qv[nb].x += t.x; qv[nb].y += t.y; qv[nb].z += t.z; qv[nb].w += t.w;
        }
      }
    }
#pragma unroll 1   // (unrolled, the second row tile's requests do not move ahead of the first one's MFMAs anyway and the kernel loses 9 %: r03_e51)
    for (int rt = 0; rt < 2; ++rt) {
      float* __restrict__ hp = a.Hb + fc_hb_off(v, rt, 0, lane, NG8 / 2);   // + 256 per pair of 8-k groups
      if (16 * rt >= ne) {   // empty row tile (wave-uniform): zero fragments where the consumer multiplies them
        if (!a.zero_fill) continue;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) *reinterpret_cast<float4*>(hp + (size_t)nb * 256) = make_float4(0.f, 0.f, 0.f, 0.f);
        continue;
      }
      const int el = 16 * rt + lr;
      const bool live = el < ne;
      int ar, tg;
      if (a.vrows) {   // attribute row / target row of the lane's edge row, prepared by k_vn_rows: one dependent request less
        const int2 at = *reinterpret_cast<const int2*>(a.vrows + ((size_t)v * 32 + el) * 8 + 6);
        ar = at.x; tg = at.y;
      } else {
        const int e = e0 + (live ? el : 16 * rt);
        ar = a.arow ? a.arow[e] : e;
        tg = a.tgt[e] - a.tbase;
      }
#ifdef EHV_SEQATTR   // timing-only variant (garbage scores; run with frozen poses): attribute rows read in gather order -- the upper bound
                     // of what re-ordering the attribute stores (rec-rec at set_complex, a receptor-major copy of the cross rows) can buy
      ar = e0 + (live ? el : 16 * rt);
#endif
      const float* __restrict__ ep = a.ea + (size_t)ar * a.ns + KS * lq;
      const float* __restrict__ prow = a.P + (size_t)tg * H + 4 * lq;
      float4 ae[NSQ], pv[NB];
#pragma unroll
      for (int j = 0; j < NSQ; ++j) ae[j] = *reinterpret_cast<const float4*>(ep + 4 * j);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) pv[nb] = *reinterpret_cast<const float4*>(prow + 16 * nb);
      DDMI_SCHED_FENCE();   // every request of the tile is in flight before the first MFMA (the scheduler would sink them to their uses)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        f32x4 acc = f32x4{qv[nb].x + pv[nb].x, qv[nb].y + pv[nb].y, qv[nb].z + pv[nb].z, qv[nb].w + pv[nb].w};
        f32x4 acc2 = f32x4{0.f, 0.f, 0.f, 0.f};   // two chains: a dependent f32 MFMA waits 40 cycles
        const float* __restrict__ wp = wl + lq * HP + 16 * nb + lr;
#pragma unroll
        for (int j = 0; j < NSQ; ++j) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[(4 * j + 0) * 4 * HP], ae[j].x, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[(4 * j + 1) * 4 * HP], ae[j].y, acc2, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[(4 * j + 2) * 4 * HP], ae[j].z, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[(4 * j + 3) * 4 * HP], ae[j].w, acc2, 0, 0, 0);
        }
        float4 o;
        o.x = live ? fmaxf(acc[0] + acc2[0], 0.f) : 0.f; o.y = live ? fmaxf(acc[1] + acc2[1], 0.f) : 0.f;
        o.z = live ? fmaxf(acc[2] + acc2[2], 0.f) : 0.f; o.w = live ? fmaxf(acc[3] + acc2[3], 0.f) : 0.f;
        if (a.bf) { bf_split2(o.x, o.y, o.x, o.y); bf_split2(o.z, o.w, o.z, o.w); }   // split-bf16 edge product: packed (hi | lo) words
        *reinterpret_cast<float4*>(hp + (size_t)nb * 256) = o;
      }
    }
  }
}

template <int NSQ>
__global__ __launch_bounds__(256) void k_edge_hidden_mm(EdgeHiddenArgs a) {
  DDMI_DYN_SMEM(float, smem);
  eh_body<NSQ>(a, (int)blockIdx.x, (int)gridDim.x, smem);
}
// the hidden rows of several edge groups in one launch: workgroups [first[g], first[g + 1]) serve group g
template <int NSQ>
__global__ __launch_bounds__(256) void k_edge_hidden_mm_grouped(EdgeHiddenGroupedArgs G) {
  DDMI_DYN_SMEM(float, smem);
  const int b = (int)blockIdx.x;
  int g = 0;
#pragma unroll
  for (int i = 1; i < EH_GROUPS_MAX; ++i)
    if (i < G.n && b >= G.first[i]) g = i;
  eh_body<NSQ>(G.g[g], b - G.first[g], G.first[g + 1] - G.first[g], smem);
}

void launch_edge_hidden_mm_grouped(const EdgeHiddenGroupedArgs& G_in, hipStream_t s) {
  EdgeHiddenGroupedArgs G = G_in;
  if (G.n <= 0) return;
  if (G.n > EH_GROUPS_MAX) throw Error(DDMI_ERR_ARG, "k_edge_hidden_mm_grouped: too many edge groups in one launch");
  int blocks = 0;
  const int ns = G.g[0].ns, H = G.g[0].H;
  for (int i = 0; i < G.n; ++i) {
    const EdgeHiddenArgs& a = G.g[i];
    if (a.ns != ns || a.H != H || a.ns % 16 != 0 || a.ns > 64 || a.H != 3 * a.ns || a.NG8 * 8 != a.H)
      throw Error(DDMI_ERR_ARG, "k_edge_hidden_mm_grouped: unsupported / mixed widths");
    G.first[i] = blocks;
    // the launch has the chip to itself (no fused workgroups next to it): every group gets its share of a.grid workgroups by
    // virtual-node capacity, at least one
    blocks += a.vcap > 0 ? std::max(1, std::min(cdiv(a.vcap, 4), a.grid > 0 ? a.grid : 2048)) : 0;
  }
  G.first[G.n] = blocks;
  if (blocks <= 0) return;
  const size_t smem = (size_t)(ns * (H + 1)) * sizeof(float);
  switch (ns / 16) {
    case 1: hipLaunchKernelGGL(k_edge_hidden_mm_grouped<1>, dim3(blocks), dim3(256), smem, s, G); break;
    case 2: hipLaunchKernelGGL(k_edge_hidden_mm_grouped<2>, dim3(blocks), dim3(256), smem, s, G); break;
    case 3: hipLaunchKernelGGL(k_edge_hidden_mm_grouped<3>, dim3(blocks), dim3(256), smem, s, G); break;
    default: hipLaunchKernelGGL(k_edge_hidden_mm_grouped<4>, dim3(blocks), dim3(256), smem, s, G); break;
  }
  DDMI_CHECK_HIP(hipGetLastError());
}

void launch_edge_hidden_mm(const EdgeHiddenArgs& a, hipStream_t s) {
  if (a.vcap <= 0) return;
  if (a.ns % 16 != 0 || a.ns > 64 || a.H != 3 * a.ns || a.NG8 * 8 != a.H)
    throw Error(DDMI_ERR_ARG, "k_edge_hidden_mm: unsupported width");
  const size_t smem = (size_t)(a.ns * (a.H + 1)) * sizeof(float);
  // Many short workgroups (not a persistent grid of 3 per CU, which is 6 % faster alone): with the two streams a long-lived
  // workgroup holds 27 KB of LDS on its CU and keeps the concurrent k_conv_fused workgroups (131 KB) off it.
  const int grid = std::min(cdiv(a.vcap, 4), a.grid > 0 ? a.grid : 2048);
  switch (a.ns / 16) {
    case 1: hipLaunchKernelGGL(k_edge_hidden_mm<1>, dim3(grid), dim3(256), smem, s, a); break;
    case 2: hipLaunchKernelGGL(k_edge_hidden_mm<2>, dim3(grid), dim3(256), smem, s, a); break;
    case 3: hipLaunchKernelGGL(k_edge_hidden_mm<3>, dim3(grid), dim3(256), smem, s, a); break;
    default: hipLaunchKernelGGL(k_edge_hidden_mm<4>, dim3(grid), dim3(256), smem, s, a); break;
  }
  DDMI_CHECK_HIP(hipGetLastError());
}


}  // namespace ddmi
