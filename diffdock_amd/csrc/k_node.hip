// Fused node update of an interaction layer (round 6): for a tile of 16 target nodes
//   X_out[s] = BatchNorm(mean over ALL incoming messages of s) + pad(X_in[s])          (k_reduce_bn: torch_scatter mean over the joint
//                                                                                        edge list, e3nn BatchNorm, residual --
//                                                                                        models/tensor_layers.py:220-229,327-332)
//   P_g[s] = W1s_g . X_out[s][:ns],   Q_g[s] = W1d_g . X_out[s][:ns] + b1_g              (the per-node terms of the NEXT layer's first
//                                                                                        Linear on [edge_attr | x_target | x_gather],
//                                                                                        models/tensor_layers.py:140,211, models/layers.py:10-17)
// in one kernel: the node row is still in the workgroup when the next layer's first-Linear terms are taken from it, so the
// k_gemm_nt_batch launch at the head of every layer (and its dependency level) disappears.  The reduction keeps k_reduce_bn's
// summation tree (four partial sums by row index mod 4 per group, (s0 + s1) + (s2 + s3)): X_out is bit-identical to the unfused
// path; P / Q come off the 16x16x4 f32 MFMA with the k index permuted for 16-byte weight loads -- equal to the GEMM's up to the
// order of a 48-term fp32 sum.
#include <algorithm>

#include "kernels.h"

namespace ddmi {

typedef float vf4n __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nu_load4(const float* p) {
  const vf4n v = DDMI_NT_LOAD(reinterpret_cast<const vf4n*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void nu_add(float4& a, const float4& v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }

constexpr int NU_TILE = 16;

// WPN = waves per node: 1 -> workgroup = 16 nodes x 16 waves (chip-filling node counts); 4 -> 4 nodes x 16 waves, every node's rows
// dealt over four waves exactly as k_reduce_bn deals them (small batches: a node's reduction is a chain of dependent round trips,
// and 1650 nodes in 103 sixteen-node workgroups leave most of the chip idle)
template <int NSQ, int WPN>   // ns = 16 * NSQ
__global__ __launch_bounds__(1024) void k_node_update(NodeUpdateArgs a) {
  constexpr int NS = 16 * NSQ, XST = NS + 4, NPW = NU_TILE / WPN;
  __shared__ float xs[NU_TILE][XST];
  __shared__ float red[WPN == 1 ? 1 : NPW][WPN == 1 ? 1 : 4][WPN == 1 ? 4 : XS + 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = DDMI_UNIFORM(tid >> 6);
  const int n0 = a.nbase + (int)blockIdx.x * NPW, n_end = a.nbase + a.ncount;
  const bool live = 4 * lane < a.D_out;      // columns past D_out inside the XS-wide row are never used
  // BatchNorm + residual + row store of node s from the four partial sums (k_reduce_bn's tree: (s0 + s1) + (s2 + s3))
  auto finish = [&](int slot, int s, int cnt, const float4& p0, const float4& p1, const float4& p2, const float4& p3) __attribute__((always_inline)) {
    if (4 * lane < XS) {
      float sum[4] = {(p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w)};
      float ov[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 4 * lane + i;
        float v = 0.f;
        if (c < a.D_out) {
          v = cnt > 0 ? sum[i] / (float)cnt : 0.f;
          if (a.bn_scale) v = (v - a.bn_mean[c]) * a.bn_scale[c] + a.bn_bias[c];
          if (a.residual && c < a.D_in) v += a.X_in[(size_t)s * XS + c];
        }
        ov[i] = v;
      }
      const float4 o = make_float4(ov[0], ov[1], ov[2], ov[3]);
      *reinterpret_cast<float4*>(a.X_out + (size_t)s * XS + 4 * lane) = o;
      if (4 * lane < NS) *reinterpret_cast<float4*>(&xs[slot][4 * lane]) = o;
    }
  };
  if constexpr (WPN == 4) {
    // ---- phase 1 (small node counts): wave (slot, part) sums the rows of node `slot` whose index within their group is part mod 4
    const int slot = wave >> 2, part = wave & 3, s = n0 + slot;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int cnt = 0;
    if (s < n_end) {
      for (int g = 0; g < a.n_groups; ++g) {
        const ReduceGroup G = a.groups[g];
        const int sl = s - G.tbase;
        if (sl < 0 || sl >= G.tcount) continue;
        const int b = G.toff[sl], e = G.toff[sl + 1];
        cnt += e - b;
        const float* __restrict__ mp = G.msg + 4 * lane;
        if (G.live) {
          int ord = 0;
          for (int base = b; base < e; base += 64) {
            const int rr = base + lane;
            unsigned long long mask = __ballot(rr < e && G.live[rr] != 0);
            while (mask) {
              int rows[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                rows[i] = -1;
                while (mask && rows[i] < 0) {
                  const int bit = __builtin_ctzll(mask);
                  mask &= mask - 1;
                  if ((ord++ & 3) == part) rows[i] = base + bit;
                }
              }
              float4 v[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) v[i] = (live && rows[i] >= 0) ? nu_load4(mp + (size_t)rows[i] * XS) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (rows[i] >= 0) nu_add(acc, v[i]);
            }
          }
        } else if (live) {
          int r = b + part;
          for (; r + 12 < e; r += 16) {
            const float4 v0 = nu_load4(mp + (size_t)r * XS), v1 = nu_load4(mp + (size_t)(r + 4) * XS);
            const float4 v2 = nu_load4(mp + (size_t)(r + 8) * XS), v3 = nu_load4(mp + (size_t)(r + 12) * XS);
            nu_add(acc, v0); nu_add(acc, v1); nu_add(acc, v2); nu_add(acc, v3);
          }
          for (; r < e; r += 4) nu_add(acc, nu_load4(mp + (size_t)r * XS));
        }
      }
      if (4 * lane < XS) *reinterpret_cast<float4*>(&red[slot][part][4 * lane]) = acc;
    }
    // rows of the MFMA tile behind the workgroup's nodes are zero
    for (int idx = tid; idx < (NU_TILE - NPW) * XST; idx += 64 * NU_TILE) xs[NPW + idx / XST][idx % XST] = 0.f;
    __syncthreads();
    if (part == 0) {
      if (s < n_end) {
        float4 p[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) p[q] = 4 * lane < XS ? *reinterpret_cast<const float4*>(&red[slot][q][4 * lane]) : make_float4(0.f, 0.f, 0.f, 0.f);
        finish(slot, s, cnt, p[0], p[1], p[2], p[3]);
      } else if (4 * lane < NS) {
        *reinterpret_cast<float4*>(&xs[slot][4 * lane]) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  } else {
  // ---- phase 1: wave w reduces node w of the tile (one wave per node: a node's chain group -> offsets -> rows is a dozen dependent
  // round trips, so the nodes of a tile run side by side; a first form with four nodes per wave took 97 us per launch at 5 poses
  // against k_reduce_bn's 17, profiles/r06_p3_*)
  {
    const int slot = wave, s = n0 + slot;
    if (s >= n_end) {
      if (4 * lane < NS) *reinterpret_cast<float4*>(&xs[slot][4 * lane]) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
    float4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    int cnt = 0;
    for (int g = 0; g < a.n_groups; ++g) {
      const ReduceGroup G = a.groups[g];
      const int sl = s - G.tbase;
      if (sl < 0 || sl >= G.tcount) continue;
      const int b = G.toff[sl], e = G.toff[sl + 1];
      cnt += e - b;
      const float* __restrict__ mp = G.msg + 4 * lane;
      if (G.live) {   // pre-reduced group: only the flagged rows hold (partial) sums; the row of ordinal o goes to partial sum o mod 4
        int ord = 0;
        for (int base = b; base < e; base += 64) {
          const int rr = base + lane;
          unsigned long long mask = __ballot(rr < e && G.live[rr] != 0);
          while (mask) {
            int rows[4], which[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              rows[i] = -1; which[i] = 0;
              if (mask) { rows[i] = base + __builtin_ctzll(mask); mask &= mask - 1; which[i] = ord++ & 3; }
            }
            float4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (live && rows[i] >= 0) ? nu_load4(mp + (size_t)rows[i] * XS) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // (which[i] is wave-uniform; four consecutive ordinals hit four different partial sums)
              if (rows[i] < 0) continue;
              if (which[i] == 0) nu_add(acc[0], v[i]);
              else if (which[i] == 1) nu_add(acc[1], v[i]);
              else if (which[i] == 2) nu_add(acc[2], v[i]);
              else nu_add(acc[3], v[i]);
            }
          }
        }
      } else if (live) {
        int r = b;
        for (; r + 8 <= e; r += 8) {   // rows r + i -> partial sum i mod 4, in increasing row order per partial sum
          float4 v[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = nu_load4(mp + (size_t)(r + i) * XS);
#pragma unroll
          for (int i = 0; i < 8; ++i) nu_add(acc[i & 3], v[i]);
        }
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = r + i < e ? nu_load4(mp + (size_t)(r + i) * XS) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (r + i < e) nu_add(acc[i & 3], v[i]);
      }
    }
    finish(slot, s, cnt, acc[0], acc[1], acc[2], acc[3]);
    }
  }
  }
  // ---- phase 2: the next layer's first-Linear terms of the tile: out_t[16 nodes][H] = xs[16][ns] . W_t^T (+ bias_t), 16 hidden units
  // per MFMA column block.  Lane (lr, lq) supplies k = 16 jj + 4 lq + i in step (jj, i) for both operands, so its weight
  // fragments are whole 16-byte pieces of a weight row.  A wave's (term, column block) tasks are known before the barrier: their
  // weight fragments are requested in front of it (the L2 round trip hides behind the slowest node of the tile), and every task
  // runs two MFMA chains (even / odd jj... i) so that a dependent f32 MFMA does not wait out its predecessor.
  const int lr = lane & 15, lq = lane >> 4;
  const int nb_h = a.H / 16, n_tasks = a.n_terms * nb_h;
  constexpr int NT_MAX = 5;     // tasks per wave kept in registers (8 terms x 9 column blocks / 16 waves = 4.5)
  float4 bw[NT_MAX][NSQ];
  float tb[NT_MAX];
  bool on[NT_MAX];
#pragma unroll
  for (int q = 0; q < NT_MAX; ++q) {
    const int task = wave + NU_TILE * q;
    on[q] = false; tb[q] = 0.f;
#pragma unroll
    for (int jj = 0; jj < NSQ; ++jj) bw[q][jj] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (task < n_tasks) {
      const int t = task / nb_h, h0 = 16 * (task - t * nb_h);
      const NodeTerm T = a.term[t];
      on[q] = !(n0 + NPW <= T.base || n0 >= T.base + T.count);   // (wave-uniform)
      if (on[q]) {
        const float* __restrict__ wrow = T.W + (size_t)(h0 + lr) * a.ldw + 4 * lq;
#pragma unroll
        for (int jj = 0; jj < NSQ; ++jj) bw[q][jj] = *reinterpret_cast<const float4*>(wrow + 16 * jj);
        tb[q] = T.bias ? T.bias[h0 + lr] : 0.f;
      }
    }
  }
  __syncthreads();
  float4 xa[NSQ];
#pragma unroll
  for (int jj = 0; jj < NSQ; ++jj) xa[jj] = *reinterpret_cast<const float4*>(&xs[lr][16 * jj + 4 * lq]);
  auto run_task = [&](int task, const float4 (&w)[NSQ], float bias) __attribute__((always_inline)) {
    const int t = task / nb_h, h0 = 16 * (task - t * nb_h);
    const NodeTerm T = a.term[t];
    f32x4 acc = f32x4{bias, bias, bias, bias}, acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < NSQ; ++jj) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[jj].x, w[jj].x, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[jj].y, w[jj].y, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[jj].z, w[jj].z, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[jj].w, w[jj].w, acc2, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int s = n0 + 4 * lq + r;
      if (4 * lq + r < NPW && s >= T.base && s < T.base + T.count && s < n_end) T.out[(size_t)(s - T.base) * a.H + h0 + lr] = acc[r] + acc2[r];
    }
  };
#pragma unroll
  for (int q = 0; q < NT_MAX; ++q)
    if (on[q]) run_task(wave + NU_TILE * q, bw[q], tb[q]);
  for (int task = wave + NU_TILE * NT_MAX; task < n_tasks; task += NU_TILE) {   // (wider layers: the rest without the prefetch)
    const int t = task / nb_h, h0 = 16 * (task - t * nb_h);
    const NodeTerm T = a.term[t];
    if (n0 + NPW <= T.base || n0 >= T.base + T.count) continue;
    float4 w[NSQ];
    const float* __restrict__ wrow = T.W + (size_t)(h0 + lr) * a.ldw + 4 * lq;
#pragma unroll
    for (int jj = 0; jj < NSQ; ++jj) w[jj] = *reinterpret_cast<const float4*>(wrow + 16 * jj);
    run_task(task, w, T.bias ? T.bias[h0 + lr] : 0.f);
  }
}

void launch_node_update(const NodeUpdateArgs& a, hipStream_t s) {
  if (a.ncount <= 0) return;
  if (a.ns % 16 != 0 || a.ns > 64 || a.H % 16 != 0 || a.n_terms > NU_TERMS_MAX || a.ldw % 4 != 0)
    throw Error(DDMI_ERR_ARG, "k_node_update: unsupported width");
  // four waves per node while the node count cannot fill the chip with sixteen-node workgroups
  const bool small = a.wpn == 4 || (a.wpn == 0 && a.ncount < 16 * 2 * 256);
  const dim3 grid((unsigned)cdiv(a.ncount, small ? NU_TILE / 4 : NU_TILE));
#define NU_LAUNCH(NSQ_)                                                                                   \
  do {                                                                                                    \
    if (small) hipLaunchKernelGGL((k_node_update<NSQ_, 4>), grid, dim3(64 * NU_TILE), 0, s, a);           \
    else hipLaunchKernelGGL((k_node_update<NSQ_, 1>), grid, dim3(64 * NU_TILE), 0, s, a);                 \
  } while (0)
  switch (a.ns / 16) {
    case 1: NU_LAUNCH(1); break;
    case 2: NU_LAUNCH(2); break;
    case 3: NU_LAUNCH(3); break;
    default: NU_LAUNCH(4); break;
  }
#undef NU_LAUNCH
  DDMI_CHECK_HIP(hipGetLastError());
}

}  // namespace ddmi
