// Score read-outs (reference models/cg_model.py:368-423): centre-of-mass convolution
// (build_center_conv_graph :610-623, final_conv), translation / rotation heads with the
// sigma and IGSO(3) score-norm scalings (:377-395, utils/so3.py:89-93), rotatable-bond
// convolution (build_bond_conv_graph :625-639, FullTensorProduct(sh, "2e") :412,
// tor_bond_conv, tor_final_layer, utils/torus.py:79-83).  These graphs have B*Nl and
// ~B*R*10 edges -- three orders of magnitude fewer than the interaction layers -- so they use
// the direct form: per-edge weights from the GEMM, then a table-driven tensor product.
#include "kernels.h"

namespace ddmi {

__device__ __forceinline__ void sh_from_unit(float x, float y, float z, int lmax, float* sh) {
  const float s3 = 1.7320508075688772f, s5 = 2.23606797749979f;
  sh[0] = 1.f;
  sh[1] = s3 * x; sh[2] = s3 * y; sh[3] = s3 * z;
  if (lmax >= 2) {
    sh[4] = s5 * (s3 * x * z);
    sh[5] = s5 * (s3 * x * y);
    sh[6] = s5 * (y * y - 0.5f * (x * x + z * z));
    sh[7] = s5 * (s3 * y * z);
    sh[8] = s5 * ((s3 / 2) * (z * z - x * x));
  }
}

// thread per atom: edge (graph b <- atom a), vec = pos[a] - centroid[b]
__global__ void k_center_edges(const float* __restrict__ pos, const int* __restrict__ batch, const int* __restrict__ ptr,
                               int nL, float* __restrict__ dist, float* __restrict__ nvec) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= nL) return;
  const int b = batch[a], lo = ptr[b], hi = ptr[b + 1];
  float cx = 0.f, cy = 0.f, cz = 0.f;
  for (int i = lo; i < hi; ++i) { cx += pos[3 * i]; cy += pos[3 * i + 1]; cz += pos[3 * i + 2]; }
  const float n = (float)(hi - lo);
  cx /= n; cy /= n; cz /= n;
  const float vx = pos[3 * a] - cx, vy = pos[3 * a + 1] - cy, vz = pos[3 * a + 2] - cz;
  const float d = sqrtf(vx * vx + vy * vy + vz * vz);
  const float inv = 1.f / fmaxf(d, 1e-12f);
  dist[a] = d;
  nvec[3 * a] = vx * inv; nvec[3 * a + 1] = vy * inv; nvec[3 * a + 2] = vz * inv;
}
void launch_center_edges(const float* pos, const int* batch, const int* ptr, int B, int nL, float* dist, float* nvec,
                         hipStream_t s) {
  (void)B;
  if (nL <= 0) return;
  hipLaunchKernelGGL(k_center_edges, dim3(cdiv(nL, 64)), dim3(64), 0, s, pos, batch, ptr, nL, dist, nvec);
  DDMI_CHECK_HIP(hipGetLastError());
}

// centre convolution, everything per ligand atom that does not need the edge MLP (cg_model.py:368-376): vector to the graph's
// centroid, its harmonics, and columns [ns, 2 ns) of the attribute row = the scalars of node xrow[a]
__global__ __launch_bounds__(64) void k_center_prep(const float* __restrict__ pos, const int* __restrict__ batch, const int* __restrict__ ptr,
                                                   int nL, int lmax, const float* __restrict__ X, const int* __restrict__ xrow, int ns,
                                                   float* __restrict__ dist, float* __restrict__ nvec, float* __restrict__ sh, int lds_,
                                                   float* __restrict__ attr, int lda) {
  const int a = blockIdx.x;           // one wave per atom: lane 0 the geometry, all lanes the scalar columns
  if (a >= nL) return;
  const int lane = threadIdx.x;
  if (lane == 0) {
    const int b = batch[a], lo = ptr[b], hi = ptr[b + 1];
    float cx = 0.f, cy = 0.f, cz = 0.f;
    for (int i = lo; i < hi; ++i) { cx += pos[3 * i]; cy += pos[3 * i + 1]; cz += pos[3 * i + 2]; }
    const float n = (float)(hi - lo);
    cx /= n; cy /= n; cz /= n;
    const float vx = pos[3 * a] - cx, vy = pos[3 * a + 1] - cy, vz = pos[3 * a + 2] - cz;
    const float d = sqrtf(vx * vx + vy * vy + vz * vz);
    const float inv = 1.f / fmaxf(d, 1e-12f);
    dist[a] = d;
    const float ux = vx * inv, uy = vy * inv, uz = vz * inv;
    nvec[3 * a] = ux; nvec[3 * a + 1] = uy; nvec[3 * a + 2] = uz;
    float v[9];
    sh_from_unit(ux, uy, uz, lmax, v);
    const int nsh = (lmax + 1) * (lmax + 1);
    for (int j = 0; j < nsh; ++j) sh[(size_t)a * lds_ + j] = v[j];
  }
  const float* __restrict__ xr = X + (size_t)xrow[a] * XS;
  for (int c = lane; c < ns; c += 64) attr[(size_t)a * lda + ns + c] = xr[c];
}
void launch_center_prep(const float* pos, const int* batch, const int* ptr, int nL, int lmax, const float* X, const int* xrow, int ns,
                        float* dist, float* nvec, float* sh, int lds_, float* attr, int lda, hipStream_t s) {
  if (nL <= 0) return;
  hipLaunchKernelGGL(k_center_prep, dim3(nL), dim3(64), 0, s, pos, batch, ptr, nL, lmax, X, xrow, ns, dist, nvec, sh, lds_, attr, lda);
  DDMI_CHECK_HIP(hipGetLastError());
}

__global__ void k_sh_rows(const float* __restrict__ nvec, float sgn, int E, int lmax, float* __restrict__ sh, int lds_) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  float v[9];
  sh_from_unit(sgn * nvec[3 * e], sgn * nvec[3 * e + 1], sgn * nvec[3 * e + 2], lmax, v);
  const int n = (lmax + 1) * (lmax + 1);
  for (int j = 0; j < n; ++j) sh[(size_t)e * lds_ + j] = v[j];
}
void launch_sh_rows(const float* nvec, float sgn, int E, int lmax, float* sh, int lds_, hipStream_t s) {
  if (E <= 0) return;
  hipLaunchKernelGGL(k_sh_rows, dim3(cdiv(E, 64)), dim3(64), 0, s, nvec, sgn, E, lmax, sh, lds_);
  DDMI_CHECK_HIP(hipGetLastError());
}

// out[e][k] = sum_{i,j} T[i][j][k] * sh_edge[e][i] * sh2(bond(e))[j]      (e = t*cap + r)
__global__ void k_tor_sh(const float* __restrict__ edge_nvec, const float* __restrict__ bond_nvec, int nT, int cap,
                         int lmax, const float* __restrict__ T, int ds, int dts, float* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nT * cap) return;
  const int t = e / cap;
  float se[9], sb9[9];
  sh_from_unit(edge_nvec[3 * e], edge_nvec[3 * e + 1], edge_nvec[3 * e + 2], lmax, se);
  sh_from_unit(bond_nvec[3 * t], bond_nvec[3 * t + 1], bond_nvec[3 * t + 2], 2, sb9);
  const float* sb = sb9 + 4;  // the 2e block
  // a zero edge vector (padding slot) has sh_l>0 = 0 but sh_0 = 1, as F.normalize(0) = 0
  for (int k = 0; k < dts; ++k) {
    float acc = 0.f;
    for (int i = 0; i < ds; ++i)
      for (int j = 0; j < 5; ++j) acc = fmaf(T[(i * 5 + j) * dts + k], se[i] * sb[j], acc);
    out[(size_t)e * dts + k] = acc;
  }
}
void launch_tor_sh(const float* edge_nvec, const float* bond_nvec, int nT, int cap, int lmax, const float* T, int ds,
                   int dts, float* out, hipStream_t s) {
  if (nT <= 0) return;
  hipLaunchKernelGGL(k_tor_sh, dim3(cdiv(nT * cap, 64)), dim3(64), 0, s, edge_nvec, bond_nvec, nT, cap, lmax, T, ds, dts, out);
  DDMI_CHECK_HIP(hipGetLastError());
}

// torsion read-out, everything per (bond, neighbour) pair that does not need the edge MLP (cg_model.py:404-416): the FullTensorProduct
// harmonics of k_tor_sh and columns [ns, 3 ns) of the attribute row = scalars of the neighbour atom | sum of the bond's two atoms
__global__ __launch_bounds__(64) void k_tor_prep(const float* __restrict__ edge_nvec, const float* __restrict__ bond_nvec, int nT, int cap, int lmax,
                                                const float* __restrict__ T, int ds, int dts, float* __restrict__ out,
                                                const float* __restrict__ X, const int* __restrict__ atom, const int* __restrict__ eu,
                                                const int* __restrict__ ev, int ns, float* __restrict__ attr, int lda) {
  const int e = blockIdx.x;           // one wave per pair: lanes over the output harmonics, then over the scalar columns
  if (e >= nT * cap) return;
  const int t = e / cap, lane = threadIdx.x;
  float se[9], sb9[9];
  sh_from_unit(edge_nvec[3 * e], edge_nvec[3 * e + 1], edge_nvec[3 * e + 2], lmax, se);
  sh_from_unit(bond_nvec[3 * t], bond_nvec[3 * t + 1], bond_nvec[3 * t + 2], 2, sb9);
  const float* sb = sb9 + 4;  // the 2e block
  for (int k = lane; k < dts; k += 64) {   // (same sum order as k_tor_sh)
    float acc = 0.f;
    for (int i = 0; i < ds; ++i)
      for (int j = 0; j < 5; ++j) acc = fmaf(T[(i * 5 + j) * dts + k], se[i] * sb[j], acc);
    out[(size_t)e * dts + k] = acc;
  }
  const float* __restrict__ xa = X + (size_t)atom[e] * XS;
  const float* __restrict__ xu = X + (size_t)eu[e] * XS;
  const float* __restrict__ xv = X + (size_t)ev[e] * XS;
  float* __restrict__ row = attr + (size_t)e * lda;
  for (int c = lane; c < ns; c += 64) { row[ns + c] = xa[c]; row[2 * ns + c] = xu[c] + xv[c]; }
}
void launch_tor_prep(const float* edge_nvec, const float* bond_nvec, int nT, int cap, int lmax, const float* T, int ds, int dts, float* out,
                     const float* X, const int* atom, const int* eu, const int* ev, int ns, float* attr, int lda, hipStream_t s) {
  if (nT <= 0) return;
  hipLaunchKernelGGL(k_tor_prep, dim3(nT * cap), dim3(64), 0, s, edge_nvec, bond_nvec, nT, cap, lmax, T, ds, dts, out, X, atom, eu, ev, ns, attr, lda);
  DDMI_CHECK_HIP(hipGetLastError());
}

// dst[r][col0 + c] = src[rowidx[r]][c] (+ src[rowidx2[r]][c])
__global__ void k_gather_cols(float* __restrict__ dst, int ldd, int col0, const float* __restrict__ src, int lds_,
                              const int* __restrict__ rowidx, int rows, int cols, const int* __restrict__ rowidx2) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)rows * cols) return;
  const int r = (int)(t / cols), c = (int)(t - (long)r * cols);
  float v = src[(size_t)(rowidx ? rowidx[r] : r) * lds_ + c];
  if (rowidx2) v += src[(size_t)rowidx2[r] * lds_ + c];
  dst[(size_t)r * ldd + col0 + c] = v;
}
void launch_gather_cols(float* dst, int ldd, int col0, const float* src, int lds_, const int* rowidx, int rows, int cols,
                        const int* rowidx2, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return;
  hipLaunchKernelGGL(k_gather_cols, dim3(cdiv((long)rows * cols, 256)), dim3(256), 0, s, dst, ldd, col0, src, lds_, rowidx,
                     rows, cols, rowidx2);
  DDMI_CHECK_HIP(hipGetLastError());
}

// thread per (edge, (out block, w)):  out[e][o_off + w*dout + k] = ew * sum_paths sum_u Wt[e][w_off+u*mul_out+w] *
//                                       sum_{i,j} C[i][j][k] * x[xrow[e]][i_off + u*din + i] * sh[e][s_off + j]
__global__ void k_tp_apply(TpApplyArgs a) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)a.E * a.n_items) return;
  const int e = (int)(t / a.n_items), it = (int)(t - (long)e * a.n_items);
  const CgItem item = a.items[it];
  float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  const bool live = !a.valid_cnt || (e % a.cap) < a.valid_cnt[e / a.cap];
  if (live) {
    const float* __restrict__ x = a.X + (size_t)a.xrow[e] * XS;
    const float* __restrict__ sh = a.sh + (size_t)e * a.lds_;
    const float* __restrict__ wt = a.Wt + (size_t)e * a.ldw;
    for (int p = item.path_begin; p < item.path_end; ++p) {
      const DevPath P = a.paths[p];
      const float* __restrict__ C = a.ctab + P.c_off;
      for (int u = 0; u < P.mul_in; ++u) {
        const float w = wt[P.w_off + u * P.mul_out + item.w];
        for (int i = 0; i < P.din; ++i) {
          const float xw = x[P.i_off + u * P.din + i] * w;
          for (int j = 0; j < P.ds; ++j) {
            const float v = xw * sh[P.s_off + j];
            for (int k = 0; k < P.dout; ++k) m[k] = fmaf(C[(i * P.ds + j) * P.dout + k], v, m[k]);
          }
        }
      }
    }
  }
  const float w = (live && a.ew) ? a.ew[e] : 1.f;
  for (int k = 0; k < item.dout; ++k) a.out[(size_t)e * a.ldo + item.o_off + item.w * item.dout + k] = w * m[k];
}
// Same arithmetic with one WAVE per (edge, item): the lanes share the (path, u) terms of the item and meet in a shuffle
// reduction.  For launches with few (edge, item) pairs (final_conv: one edge per ligand atom, 4 items) the thread-per-item
// form above is a chain of ~100 dependent-latency iterations on a nearly empty chip (160 us); this form takes ~10 us.
__global__ __launch_bounds__(256) void k_tp_apply_wave(TpApplyArgs a) {
  const int lane = threadIdx.x & 63;
  const long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= (long)a.E * a.n_items) return;
  const int e = (int)(t / a.n_items), it = (int)(t - (long)e * a.n_items);
  const CgItem item = a.items[it];
  float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  const bool live = !a.valid_cnt || (e % a.cap) < a.valid_cnt[e / a.cap];
  if (live) {
    const float* __restrict__ x = a.X + (size_t)a.xrow[e] * XS;
    const float* __restrict__ sh = a.sh + (size_t)e * a.lds_;
    const float* __restrict__ wt = a.Wt + (size_t)e * a.ldw;
    for (int p = item.path_begin; p < item.path_end; ++p) {
      const DevPath P = a.paths[p];
      const float* __restrict__ C = a.ctab + P.c_off;
      for (int u = lane; u < P.mul_in; u += 64) {
        const float w = wt[P.w_off + u * P.mul_out + item.w];
        for (int i = 0; i < P.din; ++i) {
          const float xw = x[P.i_off + u * P.din + i] * w;
          for (int j = 0; j < P.ds; ++j) {
            const float v = xw * sh[P.s_off + j];
            for (int k = 0; k < P.dout; ++k) m[k] = fmaf(C[(i * P.ds + j) * P.dout + k], v, m[k]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 5; ++k)
    for (int off = 32; off > 0; off >>= 1) m[k] += __shfl_down(m[k], off, 64);
  if (lane == 0) {
    const float w = (live && a.ew) ? a.ew[e] : 1.f;
    for (int k = 0; k < item.dout; ++k) a.out[(size_t)e * a.ldo + item.o_off + item.w * item.dout + k] = w * m[k];
  }
}
// Same arithmetic with one WORKGROUP per edge, in two stages: z[path][u][k] = sum_{i,j} C[i][j][k] * x[u,i] * sh[j] once per edge
// (it does not depend on the output channel w), then a thread per item sums Wt[e][w_off + u*mul_out + w] * z over its paths.
// The thread-per-item form recomputes z for every one of the 96 output channels of the torsion read-out: ~200 dependent
// requests per thread, 148 us for 19 200 edges at 40 poses (address-unit bound); this form issues the weight reads only.
constexpr int TPE_THREADS = 128, TPE_MAXPATHS = 32;
__global__ __launch_bounds__(TPE_THREADS) void k_tp_apply_edge(TpApplyArgs a) {
  DDMI_DYN_SMEM(float, z);                       // [sum_paths mul_in * dout]
  __shared__ int zoff[TPE_MAXPATHS + 1];
  const int e = blockIdx.x, tid = threadIdx.x;
  const bool live = !a.valid_cnt || (e % a.cap) < a.valid_cnt[e / a.cap];
  if (tid <= a.n_paths) {
    int off = 0;
    for (int q = 0; q < tid; ++q) off += a.paths[q].mul_in * a.paths[q].dout;
    zoff[tid] = off;
  }
  __syncthreads();
  if (live) {
    const float* __restrict__ x = a.X + (size_t)a.xrow[e] * XS;
    const float* __restrict__ sh = a.sh + (size_t)e * a.lds_;
    for (int p = 0; p < a.n_paths; ++p) {
      const DevPath P = a.paths[p];
      const float* __restrict__ C = a.ctab + P.c_off;
      for (int t = tid; t < P.mul_in * P.dout; t += TPE_THREADS) {
        const int u = t / P.dout, k = t - u * P.dout;
        float v = 0.f;
        for (int i = 0; i < P.din; ++i) {
          const float xv = x[P.i_off + u * P.din + i];
          for (int j = 0; j < P.ds; ++j) v = fmaf(C[(i * P.ds + j) * P.dout + k], xv * sh[P.s_off + j], v);
        }
        z[zoff[p] + t] = v;
      }
    }
  }
  __syncthreads();
  const float w_e = (live && a.ew) ? a.ew[e] : 1.f;
  const float* __restrict__ wt = a.Wt + (size_t)e * a.ldw;
  for (int it = tid; it < a.n_items; it += TPE_THREADS) {
    const CgItem item = a.items[it];
    float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
      for (int p = item.path_begin; p < item.path_end; ++p) {
        const DevPath P = a.paths[p];
        const float* __restrict__ zp = z + zoff[p];
        for (int u = 0; u < P.mul_in; ++u) {
          const float w = wt[P.w_off + u * P.mul_out + item.w];
          for (int k = 0; k < P.dout; ++k) m[k] = fmaf(w, zp[u * P.dout + k], m[k]);
        }
      }
    }
    for (int k = 0; k < item.dout; ++k) a.out[(size_t)e * a.ldo + item.o_off + item.w * item.dout + k] = w_e * m[k];
  }
}
void launch_tp_apply(const TpApplyArgs& a, hipStream_t s) {
  if (a.E <= 0 || a.n_items <= 0) return;
  const long pairs = (long)a.E * a.n_items;
  const bool edge_ok = a.n_paths <= TPE_MAXPATHS && a.z_floats > 0 && a.z_floats <= 8192;
  // a.form (tests, DDMI_TP_APPLY at ddmi_create): 0 wave / 1 edge / 2 thread instead of the size rule
  const int form = a.form == 0 || a.form == 2 ? a.form : a.form == 1 && edge_ok ? 1 : pairs <= 32768 ? 0 : edge_ok ? 1 : 2;
  if (form == 0) hipLaunchKernelGGL(k_tp_apply_wave, dim3(cdiv(pairs, 4)), dim3(256), 0, s, a);
  else if (form == 1)
    hipLaunchKernelGGL(k_tp_apply_edge, dim3(a.E), dim3(TPE_THREADS), (size_t)a.z_floats * sizeof(float), s, a);
  else hipLaunchKernelGGL(k_tp_apply, dim3(cdiv(pairs, 128)), dim3(128), 0, s, a);
  DDMI_CHECK_HIP(hipGetLastError());
}

// block per segment, thread per column: mean over the segment's rows, then BatchNorm (eval)
__global__ void k_segment_mean_bn(const float* __restrict__ rows, int ldr, const int* __restrict__ seg_off,
                                  const int* __restrict__ seg_cnt, int cap, int D, const float* __restrict__ bn_mean,
                                  const float* __restrict__ bn_scale, const float* __restrict__ bn_bias,
                                  float* __restrict__ out, int ldo) {
  const int sg = blockIdx.x;
  const int start = seg_off ? seg_off[sg] : sg * cap;
  const int cnt = seg_off ? seg_off[sg + 1] - start : seg_cnt[sg];
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    float acc = 0.f;
    for (int r = 0; r < cnt; ++r) acc += rows[(size_t)(start + r) * ldr + c];
    float v = cnt > 0 ? acc / (float)cnt : 0.f;
    if (bn_scale) v = (v - bn_mean[c]) * bn_scale[c] + bn_bias[c];
    out[(size_t)sg * ldo + c] = v;
  }
}
void launch_segment_mean_bn(const float* rows, int ldr, const int* seg_off, const int* seg_cnt, int cap, int n_seg, int D,
                            const float* bn_mean, const float* bn_scale, const float* bn_bias, float* out, int ldo,
                            hipStream_t s) {
  if (n_seg <= 0) return;
  hipLaunchKernelGGL(k_segment_mean_bn, dim3(n_seg), dim3(64), 0, s, rows, ldr, seg_off, seg_cnt, cap, D, bn_mean, bn_scale,
                     bn_bias, out, ldo);
  DDMI_CHECK_HIP(hipGetLastError());
}

__device__ __forceinline__ float sigma_of_t(float smin, float smax, float t) { return powf(smin, 1.f - t) * powf(smax, t); }

// nearest-bin index in log-sigma, rounded half-to-even like np.around / np.round
__device__ __forceinline__ int table_index(double v, int lo, int hi) {
  long i = (long)nearbyint(v);
  return (int)(i < lo ? lo : (i > hi ? hi : i));
}

__global__ void k_score_heads(ScoreHeadArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  const float* g = a.gp + (size_t)b * (a.odd_parity ? 6 : 12);
  float tr[3], rot[3];
  for (int k = 0; k < 3; ++k) {
    tr[k] = g[k] + (a.odd_parity ? 0.f : g[6 + k]);
    rot[k] = g[3 + k] + (a.odd_parity ? 0.f : g[9 + k]);
  }
  const float trn = sqrtf(tr[0] * tr[0] + tr[1] * tr[1] + tr[2] * tr[2]);
  const float rotn = sqrtf(rot[0] * rot[0] + rot[1] * rot[1] + rot[2] * rot[2]);
  float ftr = a.tr_b3[0], frot = a.rot_b3[0];
  for (int c = 0; c < a.ns; ++c) {
    float h = fmaf(a.tr_w0n[(size_t)c * a.ldw0], trn, a.tr_sig[(size_t)b * a.ns + c]);
    ftr = fmaf(a.tr_w3[c], h > 0.f ? h : 0.f, ftr);
    h = fmaf(a.rot_w0n[(size_t)c * a.ldw0], rotn, a.rot_sig[(size_t)b * a.ns + c]);
    frot = fmaf(a.rot_w3[c], h > 0.f ? h : 0.f, frot);
  }
  float str = 1.f, srot = 1.f;
  if (a.scale_by_sigma) {
    str = 1.f / sigma_of_t(a.tr_smin, a.tr_smax, a.t_tr[b]);
    const float eps = sigma_of_t(a.rot_smin, a.rot_smax, a.t_rot[b]);
    const double lo = log10(0.0005), hi = log10(4.0);
    const int idx = table_index(((double)log10f(eps) - lo) / (hi - lo) * (double)a.so3_n, 0, a.so3_n - 1);
    srot = a.so3_table[idx];
  }
  for (int k = 0; k < 3; ++k) {
    a.tr_out[3 * b + k] = tr[k] / trn * ftr * str;
    a.rot_out[3 * b + k] = rot[k] / rotn * frot * srot;
  }
}
void launch_score_heads(const ScoreHeadArgs& a, hipStream_t s) {
  if (a.B <= 0) return;
  hipLaunchKernelGGL(k_score_heads, dim3(cdiv(a.B, 64)), dim3(64), 0, s, a);
  DDMI_CHECK_HIP(hipGetLastError());
}

__global__ __launch_bounds__(64) void k_conf_head(ConfHeadArgs a) {
  __shared__ float feat[256], h0[128], h1[128];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int lo = a.lig_ptr[b], hi = a.lig_ptr[b + 1], n_in = a.ns + a.n_tail;
  for (int c = lane; c < n_in; c += 64) {
    const int col = c < a.ns ? a.col0 + c : a.tail_off + (c - a.ns);
    float s = 0.f;
    for (int i = lo; i < hi; ++i) s += a.X[(size_t)i * a.ldx + col];
    feat[c] = s / (float)max(hi - lo, 1);
  }
  __syncthreads();
  for (int c = lane; c < a.ns; c += 64) {
    float v = a.b0[c];
    for (int j = 0; j < n_in; ++j) v = fmaf(a.W0[(size_t)c * n_in + j], feat[j], v);
    h0[c] = fmaxf(v * a.sc0[c] + a.sh0[c], 0.f);
  }
  __syncthreads();
  for (int c = lane; c < a.ns; c += 64) {
    float v = a.b1[c];
    for (int j = 0; j < a.ns; ++j) v = fmaf(a.W1[(size_t)c * a.ns + j], h0[j], v);
    h1[c] = fmaxf(v * a.sc1[c] + a.sh1[c], 0.f);
  }
  __syncthreads();
  for (int o = lane; o < a.n_out; o += 64) {
    float v = a.b2[o];
    for (int j = 0; j < a.ns; ++j) v = fmaf(a.W2[(size_t)o * a.ns + j], h1[j], v);
    a.out[(size_t)b * a.n_out + o] = v;
  }
}
void launch_conf_head(const ConfHeadArgs& a, hipStream_t s) {
  if (a.B <= 0) return;
  if (a.ns > 128 || a.ns + a.n_tail > 256) throw Error(DDMI_ERR_ARG, "confidence head: ns too large");
  hipLaunchKernelGGL(k_conf_head, dim3(a.B), dim3(64), 0, s, a);
  DDMI_CHECK_HIP(hipGetLastError());
}

// one wave per torsion bond: lane c owns hidden unit c (c, c + 64, ..), tanh, weighted wave sum
__global__ __launch_bounds__(64) void k_tor_head(TorHeadArgs a) {
  const int t = blockIdx.x, lane = threadIdx.x;
  if (t >= a.nT) return;
  const float* __restrict__ f = a.feat + (size_t)t * a.in_dim;
  float part = 0.f;
  for (int c = lane; c < a.ns; c += 64) {
    const float* __restrict__ w = a.W0 + (size_t)c * a.in_dim;
    float h = 0.f;
    for (int j = 0; j < a.in_dim; ++j) h = fmaf(w[j], f[j], h);
    part = fmaf(a.W3[c], tanhf(h), part);
  }
  float out = wave_sum(part);
  if (lane != 0) return;
  if (a.scale_by_sigma) {
    const float sig = sigma_of_t(a.smin, a.smax, a.t_tor[a.tor_batch[t]]);
    const double PI = 3.14159265358979323846;
    const double lo = log(3e-3), hi = log(2.0);
    double v = ((double)logf(sig / (float)PI) - lo) / (hi - lo) * (double)(a.torus_n - 1);
    v = v < 0.0 ? 0.0 : (v > (double)(a.torus_n - 1) ? (double)(a.torus_n - 1) : v);
    out *= sqrtf(a.torus_table[table_index(v, 0, a.torus_n - 1)]);
  }
  a.out[t] = out;
}
void launch_tor_head(const TorHeadArgs& a, hipStream_t s) {
  if (a.nT <= 0) return;
  hipLaunchKernelGGL(k_tor_head, dim3(a.nT), dim3(64), 0, s, a);
  DDMI_CHECK_HIP(hipGetLastError());
}

}  // namespace ddmi
