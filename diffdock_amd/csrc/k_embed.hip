// Node / edge encoders: sinusoidal timestep embedding (utils/diffusion_utils.py:99-110),
// AtomEncoder categorical sums (models/layers.py:33-67), GaussianSmearing + two-layer edge MLPs
// (models/layers.py:20-30; lig/rec/cross/center/final edge embeddings models/cg_model.py:86-98,
// 213-218,234-239).  The first Linear of every edge MLP is applied as
//   W0 * [feat, sigma_emb(graph), gauss(d)] + b0 = W0f*feat + (W0s*sigma_emb[b] + b0) + W0g*gauss(d)
// with the per-graph bracket precomputed once per forward (gvec).
#include "kernels.h"

namespace ddmi {

__global__ void k_time_embedding(const float* __restrict__ t, int B, const float* __restrict__ freq, int half,
                                 float scale, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * half) return;
  const int b = idx / half, k = idx - b * half;
#ifdef DDMI_HIPEMU
  volatile float ts = scale * t[b];
  volatile float ph = ts * freq[k];
#else
  const float ts = __fmul_rn(scale, t[b]);
  const float ph = __fmul_rn(ts, freq[k]);
#endif
  out[b * 2 * half + k] = sinf(ph);
  out[b * 2 * half + half + k] = cosf(ph);
}
void launch_time_embedding(const float* t, int B, const float* freq, int half, float scale, float* out, hipStream_t s) {
  hipLaunchKernelGGL(k_time_embedding, dim3(cdiv(B * half, 64)), dim3(64), 0, s, t, B, freq, half, scale, out);
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
void launch_edge_mlp(const EdgeMlpArgs& a, hipStream_t s) {
  if (a.E <= 0) return;
  const size_t smem = (size_t)(a.D * a.ns + a.nfeat * a.ns + a.ns * a.ns + 4 * (a.D + a.ns)) * sizeof(float);
  const int grid = min(cdiv(a.E, 4), 2048);
  hipLaunchKernelGGL(k_edge_mlp, dim3(grid), dim3(256), smem, s, a);
  DDMI_CHECK_HIP(hipGetLastError());
}

__global__ void k_rec_edge_geom(const float* __restrict__ pos, const int* __restrict__ src, const int* __restrict__ dst,
                                int E, float smooth_max, float* __restrict__ dist, float* __restrict__ nvec,
                                float* __restrict__ ew) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int s = src[e], d = dst[e];
  const float vx = pos[3 * d] - pos[3 * s], vy = pos[3 * d + 1] - pos[3 * s + 1], vz = pos[3 * d + 2] - pos[3 * s + 2];
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
                          float* nvec, float* ew, hipStream_t s) {
  if (E <= 0) return;
  hipLaunchKernelGGL(k_rec_edge_geom, dim3(cdiv(E, 256)), dim3(256), 0, s, pos, src, dst, E, smooth_max, dist, nvec, ew);
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
