// Node / edge encoders: sinusoidal timestep embedding (utils/diffusion_utils.py:99-110),
// AtomEncoder categorical sums (models/layers.py:33-67), GaussianSmearing + two-layer edge MLPs
// (models/layers.py:20-30; lig/rec/cross/center/final edge embeddings models/cg_model.py:86-98,
// 213-218,234-239).  The first Linear of every edge MLP is applied as
//   W0 * [feat, sigma_emb(graph), gauss(d)] + b0 = W0f*feat + (W0s*sigma_emb[b] + b0) + W0g*gauss(d)
// with the per-graph bracket precomputed once per forward (gvec).
#include "kernels.h"

namespace ddmi {

// sinusoidal_embedding(scale * t) (utils/diffusion_utils.py:99-110; freq = its exp table) or, fourier = 1,
// GaussianFourierProjection (:113-127; freq = the frozen parameter W): phase ((t * W) * 2) * pi in the reference's float32 order.
__global__ void k_time_embedding(const float* __restrict__ t, int B, const float* __restrict__ freq, int half,
                                 float scale, int fourier, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * half) return;
  const int b = idx / half, k = idx - b * half;
  const float pi = 3.14159274101257324f;   // float32(np.pi)
#ifdef DDMI_HIPEMU
  volatile float ts = fourier ? t[b] * freq[k] : scale * t[b];
  volatile float p2 = fourier ? ts * 2.f : ts * freq[k];
  volatile float ph = fourier ? p2 * pi : p2;
#else
  const float ts = fourier ? __fmul_rn(t[b], freq[k]) : __fmul_rn(scale, t[b]);
  const float p2 = fourier ? __fmul_rn(ts, 2.f) : __fmul_rn(ts, freq[k]);
  const float ph = fourier ? __fmul_rn(p2, pi) : p2;
#endif
  out[b * 2 * half + k] = sinf(ph);
  out[b * 2 * half + half + k] = cosf(ph);
}
void launch_time_embedding(const float* t, int B, const float* freq, int half, float scale, int fourier, float* out, hipStream_t s) {
  hipLaunchKernelGGL(k_time_embedding, dim3(cdiv(B * half, 64)), dim3(64), 0, s, t, B, freq, half, scale, fourier, out);
  DDMI_CHECK_HIP(hipGetLastError());
}

// Time embedding + every per-graph linear term of it + the two-layer rec_sigma MLP in ONE launch (round 6; workgroup = graph): the
// head of a forward was k_time_embedding -> k_gemm_nt_batch -> k_gemm_nt, three dependent launches of B rows each
// (cg_model.py:298-301,312-322: timestep_emb_func, lig / rec node sigma terms, the sigma columns of the edge embeddings and read-outs).
__global__ __launch_bounds__(128) void k_time_terms(TimeTermsArgs a) {
  __shared__ float temb[256], hid[128];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float pi = 3.14159274101257324f;   // float32(np.pi)
  for (int k = tid; k < a.half; k += 128) {
#ifdef DDMI_HIPEMU
    volatile float ts = a.fourier ? a.t[b] * a.freq[k] : a.scale * a.t[b];
    volatile float p2 = a.fourier ? ts * 2.f : ts * a.freq[k];
    volatile float ph = a.fourier ? p2 * pi : p2;
#else
    const float ts = a.fourier ? __fmul_rn(a.t[b], a.freq[k]) : __fmul_rn(a.scale, a.t[b]);
    const float p2 = a.fourier ? __fmul_rn(ts, 2.f) : __fmul_rn(ts, a.freq[k]);
    const float ph = a.fourier ? __fmul_rn(p2, pi) : p2;
#endif
    const float sv = sinf(ph), cv = cosf(ph);
    temb[k] = sv; temb[a.half + k] = cv;
    a.temb[(size_t)b * 2 * a.half + k] = sv;
    a.temb[(size_t)b * 2 * a.half + a.half + k] = cv;
  }
  __syncthreads();
  const int sd = 2 * a.half;
  for (int idx = tid; idx < a.n * a.ns; idx += 128) {
    const int q = idx / a.ns, col = idx - q * a.ns;
    const float* __restrict__ w = a.term[q].W + (size_t)col * a.term[q].ldw;
    float acc = 0.f;
    for (int k = 0; k < sd; ++k) acc = fmaf(temb[k], w[k], acc);
    if (a.term[q].bias) acc += a.term[q].bias[col];
    if (a.term[q].act) acc = fmaxf(acc, 0.f);
    a.term[q].C[(size_t)b * a.ns + col] = acc;
    if (q == a.hid_term) hid[col] = acc;
  }
  __syncthreads();
  if (a.W3)
    for (int col = tid; col < a.ns; col += 128) {
      const float* __restrict__ w = a.W3 + (size_t)col * a.ns;
      float acc = 0.f;
      for (int k = 0; k < a.ns; ++k) acc = fmaf(hid[k], w[k], acc);
      a.out3[(size_t)b * a.ns + col] = acc + (a.b3 ? a.b3[col] : 0.f);
    }
}
void launch_time_terms(const TimeTermsArgs& a, hipStream_t s) {
  if (a.B <= 0) return;
  if (a.half > 128 || a.ns > 128 || a.n > TIME_TERMS_MAX) throw Error(DDMI_ERR_ARG, "k_time_terms: unsupported width");
  hipLaunchKernelGGL(k_time_terms, dim3(a.B), dim3(128), 0, s, a);
  DDMI_CHECK_HIP(hipGetLastError());
}

__global__ void k_lig_node_embed(const int* __restrict__ x, int nL, const float* __restrict__ emb,
                                 const int* __restrict__ emb_off, int n_feat, int ns, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nL * ns) return;
  const int i = idx / ns, c = idx - i * ns;
  float acc = 0.f;
  for (int f = 0; f < n_feat; ++f) acc += emb[(size_t)(emb_off[f] + x[i * n_feat + f]) * ns + c];
  out[idx] = acc;
}
void launch_lig_node_embed(const int* x, int nL, const float* emb, const int* emb_off, int n_feat, int ns, float* out,
                           hipStream_t s) {
  if (nL <= 0) return;
  hipLaunchKernelGGL(k_lig_node_embed, dim3(cdiv((long)nL * ns, 256)), dim3(256), 0, s, x, nL, emb, emb_off, n_feat, ns, out);
  DDMI_CHECK_HIP(hipGetLastError());
}

// X[r][c] = base[r][c] + (c < vcols ? vec[idx[r]][c] : 0), c < cols
__global__ void k_add_rowvec(float* __restrict__ X, int ldx, const float* __restrict__ base, int ldb,
                             const float* __restrict__ vec, int ldv, const int* __restrict__ idx, int rows, int cols,
                             int vcols) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)rows * cols) return;
  const int r = (int)(t / cols), c = (int)(t - (long)r * cols);
  float v = base[(size_t)r * ldb + c];
  if (vec && c < vcols) v += vec[(size_t)(idx ? idx[r] : 0) * ldv + c];
  X[(size_t)r * ldx + c] = v;
}
void launch_add_rowvec(float* X, int ldx, const float* base, int ldb, const float* vec, int ldv, const int* idx, int rows,
                       int cols, int vcols, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return;
  hipLaunchKernelGGL(k_add_rowvec, dim3(cdiv((long)rows * cols, 256)), dim3(256), 0, s, X, ldx, base, ldb, vec, ldv, idx,
                     rows, cols, vcols);
  DDMI_CHECK_HIP(hipGetLastError());
}

// one wave per edge; weights transposed into LDS once per workgroup
__global__ __launch_bounds__(256) void k_edge_mlp(EdgeMlpArgs a) {
  DDMI_DYN_SMEM(float, smem);
  const int ns = a.ns, D = a.D, nf = a.nfeat;
  float* w0g = smem;                 // [D][ns]
  float* w0f = w0g + D * ns;         // [nf][ns]
  float* w1 = w0f + nf * ns;         // [ns][ns]  (w1[j][c] = W1[c][j])
  float* scratch = w1 + ns * ns;     // 4 x (D + ns)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < D * ns; i += 256) { const int k = i / ns, c = i - k * ns; w0g[i] = a.W0g[(size_t)c * a.ldw0g + k]; }
  for (int i = tid; i < nf * ns; i += 256) { const int f = i / ns, c = i - f * ns; w0f[i] = a.W0f[(size_t)c * a.ldw0f + f]; }
  for (int i = tid; i < ns * ns; i += 256) { const int j = i / ns, c = i - j * ns; w1[i] = a.W1[(size_t)c * ns + j]; }
  float* g = scratch + wave * (D + ns);
  float* hid = g + D;
  int E = a.E;
  if (a.e_dev) { const int ev = *a.e_dev; E = ev < E ? ev : E; }
  __syncthreads();
  for (int base = blockIdx.x * 4; base < E; base += gridDim.x * 4) {
    const int e = base + wave;
    const bool live = e < E;
    if (live) {
      const float dist = a.dist[e];
      for (int k = lane; k < D; k += 64) { const float dd = dist - a.offsets[k]; g[k] = expf(a.coeff * (dd * dd)); }
    }
    __syncthreads();
    if (live) {
      const float* gv = a.gvec + (size_t)(a.gidx ? a.gidx[e] : 0) * ns;
      const int fr = (nf > 0 && a.featidx) ? a.featidx[e] : (nf > 0 ? e : -1);
      for (int c = lane; c < ns; c += 64) {
        float acc = gv[c];
        for (int k = 0; k < D; ++k) acc = fmaf(w0g[k * ns + c], g[k], acc);
        if (fr >= 0)
          for (int f = 0; f < nf; ++f) acc = fmaf(w0f[f * ns + c], a.feat[(size_t)fr * nf + f], acc);
        hid[c] = acc > 0.f ? acc : 0.f;
      }
    }
    __syncthreads();
    if (live) {
      for (int c = lane; c < ns; c += 64) {
        float acc = a.b1[c];
        for (int j = 0; j < ns; ++j) acc = fmaf(w1[j * ns + c], hid[j], acc);
        a.out[(size_t)e * a.ldo + c] = acc;
      }
    }
    __syncthreads();
  }
}
// Matrix-core form of the same two-layer MLP for ns % 16 == 0, D % 4 == 0, nfeat <= 4: one wave per 16 edges, both
// products taken transposed (weights = A operand from LDS, edges = the N dimension), so that lane (edge, q) owns hidden
// units 16cb + 4q + {0..3} of ITS edge after the first layer -- which is exactly the B fragment (k = 16cb + 4q + r,
// n = edge) the second layer needs: no transpose, no LDS round trip for the activations.  The Gaussians are the first
// layer's B fragments, computed in registers.  Output: 16 B per lane (64 B per edge per column block).
template <int NSB>   // ns = 16 * NSB
__global__ __launch_bounds__(256) void k_edge_mlp_mm(EdgeMlpArgs a) {
  DDMI_DYN_SMEM(float, smem);
  constexpr int ns = 16 * NSB, WS = ns + (16 - ns % 64 + 64) % 64;   // LDS row stride = 16 mod 64: the 4 lane quarters hit distinct banks
  const int D = a.D, nf = a.nfeat, T0 = D >> 2;
  float* w0 = smem;                  // [D + 4][WS]: w0[k][c] = W0g[c][k], rows D.. = W0f (zero padded to 4 features)
  float* w1 = w0 + (D + 4) * WS;     // [ns][WS]:    w1[j][c] = W1[c][j]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = DDMI_UNIFORM(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  for (int i = tid; i < (D + 4) * ns; i += 256) {
    const int c = i / (D + 4), k = i - c * (D + 4);
    float v = 0.f;
    if (k < D) v = a.W0g[(size_t)c * a.ldw0g + k];
    else if (k - D < nf) v = a.W0f[(size_t)c * a.ldw0f + (k - D)];
    w0[k * WS + c] = v;
  }
  for (int i = tid; i < ns * ns; i += 256) { const int c = i / ns, j = i - c * ns; w1[j * WS + c] = a.W1[i]; }
  int E = a.E;
  if (a.e_dev) { const int ev = *a.e_dev; E = ev < E ? ev : E; }
  __syncthreads();
  f32x4 b1v[NSB];
#pragma unroll
  for (int nb = 0; nb < NSB; ++nb)
    b1v[nb] = f32x4{a.b1[16 * nb + 4 * lq], a.b1[16 * nb + 4 * lq + 1], a.b1[16 * nb + 4 * lq + 2], a.b1[16 * nb + 4 * lq + 3]};
  for (int base = (blockIdx.x * 4 + wave) * 16; base < E; base += gridDim.x * 64) {
    const int e = base + lr;
    const bool live = e < E;
    const float dist = live ? a.dist[e] : 0.f;
    const float* __restrict__ gv = a.gvec + (size_t)((live && a.gidx) ? a.gidx[e] : 0) * ns + 4 * lq;
    float fv = 0.f;   // this lane's bond feature (feature index = lq)
    if (live && nf > 0 && lq < nf) {
      const int fr = a.featidx ? a.featidx[e] : e;
      if (fr >= 0) fv = a.feat[(size_t)fr * nf + lq];
    }
    f32x4 h[NSB];
#pragma unroll
    for (int cb = 0; cb < NSB; ++cb) {
      const float4 g4 = *reinterpret_cast<const float4*>(gv + 16 * cb);
      h[cb] = f32x4{g4.x, g4.y, g4.z, g4.w};
    }
    for (int t = 0; t < T0; ++t) {   // Gaussian k = 4t + lq of this lane's edge
      const float dd = dist - a.offsets[4 * t + lq];
      const float g = expf(a.coeff * (dd * dd));
      const float* __restrict__ wp = w0 + (4 * t + lq) * WS + lr;
#pragma unroll
      for (int cb = 0; cb < NSB; ++cb) h[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[16 * cb], g, h[cb], 0, 0, 0);
    }
    if (nf > 0) {
      const float* __restrict__ wp = w0 + (D + lq) * WS + lr;
#pragma unroll
      for (int cb = 0; cb < NSB; ++cb) h[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[16 * cb], fv, h[cb], 0, 0, 0);
    }
    f32x4 o[NSB];
#pragma unroll
    for (int nb = 0; nb < NSB; ++nb) o[nb] = b1v[nb];
#pragma unroll
    for (int cb = 0; cb < NSB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float hv = fmaxf(h[cb][r], 0.f);                       // hidden unit j = 16cb + 4lq + r of this lane's edge
        const float* __restrict__ wp = w1 + (16 * cb + 4 * lq + r) * WS + lr;
#pragma unroll
        for (int nb = 0; nb < NSB; ++nb) o[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[16 * nb], hv, o[nb], 0, 0, 0);
      }
    if (live) {
      float* __restrict__ op = a.out + (size_t)e * a.ldo + 4 * lq;
#pragma unroll
      for (int nb = 0; nb < NSB; ++nb) *reinterpret_cast<float4*>(op + 16 * nb) = make_float4(o[nb][0], o[nb][1], o[nb][2], o[nb][3]);
    }
  }
}

void launch_edge_mlp(const EdgeMlpArgs& a, hipStream_t s) {
  if (a.E <= 0) return;
  if (a.ns % 16 == 0 && a.ns <= 64 && a.D % 4 == 0 && a.nfeat <= 4 && a.ldo % 4 == 0) {
    const int ns = a.ns, WS = ns + (16 - ns % 64 + 64) % 64;
    const size_t smem = (size_t)((a.D + 4) * WS + ns * WS) * sizeof(float);
    const int grid = min(cdiv(a.E, 64), 2048);
    switch (ns / 16) {
      case 1: hipLaunchKernelGGL(k_edge_mlp_mm<1>, dim3(grid), dim3(256), smem, s, a); break;
      case 2: hipLaunchKernelGGL(k_edge_mlp_mm<2>, dim3(grid), dim3(256), smem, s, a); break;
      case 3: hipLaunchKernelGGL(k_edge_mlp_mm<3>, dim3(grid), dim3(256), smem, s, a); break;
      default: hipLaunchKernelGGL(k_edge_mlp_mm<4>, dim3(grid), dim3(256), smem, s, a); break;
    }
    DDMI_CHECK_HIP(hipGetLastError());
    return;
  }
  const size_t smem = (size_t)(a.D * a.ns + a.nfeat * a.ns + a.ns * a.ns + 4 * (a.D + a.ns)) * sizeof(float);
  const int grid = min(cdiv(a.E, 4), 2048);
  hipLaunchKernelGGL(k_edge_mlp, dim3(grid), dim3(256), smem, s, a);
  DDMI_CHECK_HIP(hipGetLastError());
}

__global__ void k_rec_edge_geom(const float* __restrict__ pos, const float* __restrict__ pos_d, const int* __restrict__ src,
                                const int* __restrict__ dst, int E, float smooth_max, float* __restrict__ dist,
                                float* __restrict__ nvec, float* __restrict__ ew) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int s = src[e], d = dst[e];
  const float vx = pos_d[3 * d] - pos[3 * s], vy = pos_d[3 * d + 1] - pos[3 * s + 1], vz = pos_d[3 * d + 2] - pos[3 * s + 2];
  const float dd = sqrtf(vx * vx + vy * vy + vz * vz);
  const float inv = 1.f / fmaxf(dd, 1e-12f);
  dist[e] = dd;
  nvec[3 * e] = vx * inv; nvec[3 * e + 1] = vy * inv; nvec[3 * e + 2] = vz * inv;
  if (ew) {
    const float PI = 3.14159265358979323846f;
    ew[e] = smooth_max > 0.f ? 0.5f * (cosf(fminf(dd * PI / smooth_max, PI)) + 1.f) : 1.f;
  }
}
void launch_rec_edge_geom(const float* pos, const int* src, const int* dst, int E, float smooth_max, float* dist,
                          float* nvec, float* ew, hipStream_t s, const float* pos_dst) {
  if (E <= 0) return;
  hipLaunchKernelGGL(k_rec_edge_geom, dim3(cdiv(E, 256)), dim3(256), 0, s, pos, pos_dst ? pos_dst : pos, src, dst, E,
                     smooth_max, dist, nvec, ew);
  DDMI_CHECK_HIP(hipGetLastError());
}

// X_out[r][c] = pad(X_in[r])[c] + U1[r][c] + U2[r][c]  (legacy class: node features + the two layer updates, old_cg_model.py:276-285)
__global__ void k_add3(float* __restrict__ out, const float* __restrict__ x, int d_in, const float* __restrict__ u1,
                       const float* __restrict__ u2, int rows, int d_out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)rows * d_out) return;
  const int r = (int)(t / d_out), c = (int)(t - (long)r * d_out);
  const size_t o = (size_t)r * XS + c;
  out[o] = (c < d_in ? x[o] : 0.f) + u1[o] + u2[o];
}
void launch_add3(float* out, const float* x, int d_in, const float* u1, const float* u2, int rows, int d_out, hipStream_t s) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(k_add3, dim3(cdiv((long)rows * d_out, 256)), dim3(256), 0, s, out, x, d_in, u1, u2, rows, d_out);
  DDMI_CHECK_HIP(hipGetLastError());
}

// out[j] = [ emb[restype_j] (ns) | rec_x[j][1:1+lm] ]  -- input of additional_features_embedder
__global__ void k_concat_rec_input(const float* __restrict__ rec_x, int ldx, const float* __restrict__ emb, int ns,
                                   int lm, int nR, float* __restrict__ out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int w = ns + lm;
  if (t >= (long)nR * w) return;
  const int j = (int)(t / w), c = (int)(t - (long)j * w);
  float v;
  if (c < ns) v = emb[(size_t)((int)rec_x[(size_t)j * ldx]) * ns + c];
  else v = rec_x[(size_t)j * ldx + 1 + (c - ns)];
  out[t] = v;
}
void launch_concat_rec_input(const float* rec_x, int ldx, const float* emb, int ns, int lm, int nR, float* out,
                             hipStream_t s) {
  if (nR <= 0) return;
  hipLaunchKernelGGL(k_concat_rec_input, dim3(cdiv((long)nR * (ns + lm), 256)), dim3(256), 0, s, rec_x, ldx, emb, ns, lm, nR, out);
  DDMI_CHECK_HIP(hipGetLastError());
}

}  // namespace ddmi
