// Per-step graph construction on the device: radius graphs with deterministic CSR layouts.
// Replaces torch_cluster.radius / radius_graph (third-party CUDA, call sites
// models/cg_model.py:477 ligand graph, :543-548 cross graph with per-graph cutoff through
// scaled coordinates, :630 bond graph) and the edge bookkeeping of CGModel.forward :329-338.
// Every graph is produced directly in the two orders the convolution needs:
//   gather order  (edges of one gather node contiguous)  -> k_edge_conv streams them
//   target order  (slots of one target node contiguous)  -> k_reduce_bn sums them in a fixed order
// Ranks come from counting scans (no atomics), so layouts -- and therefore fp32 summation
// orders -- are reproducible run to run.  Neighbour caps keep the first `cap` hits in ascending
// index order (torch_cluster's CUDA kernel behaviour).
#include "kernels.h"

namespace ddmi {

// distances use separately rounded products/sums like the ATen expressions they replace
__device__ __forceinline__ float dist2_rn(float ax, float ay, float az, float bx, float by, float bz) {
#ifdef DDMI_HIPEMU
  volatile float dx = ax - bx, dy = ay - by, dz = az - bz;
  volatile float xx = dx * dx, yy = dy * dy, zz = dz * dz;
  volatile float s = xx + yy;
  return s + zz;
#else
  const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
#endif
}

// ---------------------------------------------------------------------------- scan
__global__ __launch_bounds__(1024) void k_exclusive_scan(const int* __restrict__ in, int* __restrict__ out, int n) {
  __shared__ int part[1024];
  const int t = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int lo = min(t * per, n), hi = min(lo + per, n);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += in[i];
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - s;
  for (int i = lo; i < hi; ++i) { const int v = in[i]; out[i] = run; run += v; }
  if (t == 1023) out[n] = part[1023];
}
// two independent scans in one launch (gather- and target-side degree counts of a graph build: one dependency level instead of two)
__global__ __launch_bounds__(1024) void k_exclusive_scan2(const int* __restrict__ in0, int* __restrict__ out0, int n0,
                                                          const int* __restrict__ in1, int* __restrict__ out1, int n1) {
  __shared__ int part[1024];
  const int* __restrict__ in = blockIdx.x == 0 ? in0 : in1;
  int* __restrict__ out = blockIdx.x == 0 ? out0 : out1;
  const int n = blockIdx.x == 0 ? n0 : n1;
  const int t = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int lo = min(t * per, n), hi = min(lo + per, n);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += in[i];
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - s;
  for (int i = lo; i < hi; ++i) { const int v = in[i]; out[i] = run; run += v; }
  if (t == 1023) out[n] = part[1023];
}
void launch_exclusive_scan2(const int* in0, int* out0, int n0, const int* in1, int* out1, int n1, hipStream_t s) {
  hipLaunchKernelGGL(k_exclusive_scan2, dim3(2), dim3(1024), 0, s, in0, out0, n0, in1, out1, n1);
  DDMI_CHECK_HIP(hipGetLastError());
}
void launch_exclusive_scan(const int* in, int* out, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, s, in, out, n);
  DDMI_CHECK_HIP(hipGetLastError());
}

// ------------------------------------------------------------------- ligand radius graph
// thread per query atom d (= gather node, edge_index[1]); adjrank[d][s_local] = rank of s among d's
// kept neighbours or -1.  cap counts the query itself like radius(x, x, r, max_num_neighbors + 1).
__global__ void k_lig_radius(const float* __restrict__ pos, const int* __restrict__ batch, const int* __restrict__ ptr,
                             int nL, int maxNl, float r2, int cap, int* __restrict__ adjrank, int* __restrict__ cnt_g) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= nL) return;
  const int b = batch[d], lo = ptr[b], hi = ptr[b + 1];
  const float x = pos[3 * d], y = pos[3 * d + 1], z = pos[3 * d + 2];
  int hits = 0, kept = 0;
  int* row = adjrank + (size_t)d * maxNl;
  for (int s = lo; s < hi; ++s) {
    int rank = -1;
    if (hits < cap && dist2_rn(pos[3 * s], pos[3 * s + 1], pos[3 * s + 2], x, y, z) < r2) {
      ++hits;
      if (s != d) rank = kept++;
    }
    row[s - lo] = rank;
  }
  cnt_g[d] = kept;
}
void launch_lig_radius(const float* pos, const int* batch, const int* ptr, int nL, int maxNl, float r, int cap,
                       int* adjrank, int* cnt_g, hipStream_t s) {
  if (nL <= 0) return;
  hipLaunchKernelGGL(k_lig_radius, dim3(cdiv(nL, 64)), dim3(64), 0, s, pos, batch, ptr, nL, maxNl, r * r, cap, adjrank, cnt_g);
  DDMI_CHECK_HIP(hipGetLastError());
}

// thread per atom: in-degree as a target (bonds + radius edges), out-degree as gather node
__global__ void k_ll_count(const int* __restrict__ adjrank, const int* __restrict__ batch, const int* __restrict__ ptr,
                           int nL, int maxNl, const int* __restrict__ bg, const int* __restrict__ bt,
                           int* __restrict__ cnt_g, int* __restrict__ cnt_t) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nL) return;
  const int b = batch[s], lo = ptr[b], hi = ptr[b + 1];
  int c = 0;
  for (int d = lo; d < hi; ++d) c += adjrank[(size_t)d * maxNl + (s - lo)] >= 0;
  cnt_t[s] = c + bt[s];
  cnt_g[s] += bg[s];
}
void launch_ll_count(const int* adjrank, const int* batch, const int* ptr, int nL, int maxNl, const int* bg,
                     const int* bt, int* cnt_g_inout, int* cnt_t, hipStream_t s) {
  if (nL <= 0) return;
  hipLaunchKernelGGL(k_ll_count, dim3(cdiv(nL, 64)), dim3(64), 0, s, adjrank, batch, ptr, nL, maxNl, bg, bt, cnt_g_inout, cnt_t);
  DDMI_CHECK_HIP(hipGetLastError());
}

__device__ __forceinline__ void write_geom(float vx, float vy, float vz, float smooth_max, int e, float* dist,
                                           float* nvec, float* ew) {
  const float d = sqrtf(vx * vx + vy * vy + vz * vz);
  const float inv = 1.f / fmaxf(d, 1e-12f);
  dist[e] = d;
  nvec[3 * e] = vx * inv; nvec[3 * e + 1] = vy * inv; nvec[3 * e + 2] = vz * inv;
  if (ew) {
    const float PI = 3.14159265358979323846f;
    ew[e] = smooth_max > 0.f ? 0.5f * (cosf(fminf(d * PI / smooth_max, PI)) + 1.f) : 1.f;
  }
}

// threads [0,nL): one per target atom s (radius edges into s); threads [nL, nL+n_bonds): one per bond edge
__global__ void k_ll_fill(const float* __restrict__ pos, const int* __restrict__ batch, const int* __restrict__ ptr,
                          int nL, int maxNl, const int* __restrict__ adjrank, const int* __restrict__ goff,
                          const int* __restrict__ toff, const int* __restrict__ bg, const int* __restrict__ bt,
                          int n_bonds, const int* __restrict__ bond_src, const int* __restrict__ bond_dst,
                          const int* __restrict__ bond_grank, const int* __restrict__ bond_trank, float smooth_max,
                          int* __restrict__ tgt, int* __restrict__ tslot, int* __restrict__ featidx,
                          int* __restrict__ ebatch, float* __restrict__ dist, float* __restrict__ nvec,
                          float* __restrict__ ew) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < nL) {
    const int s = t, b = batch[s], lo = ptr[b], hi = ptr[b + 1];
    int r = bt[s];
    for (int d = lo; d < hi; ++d) {
      const int a = adjrank[(size_t)d * maxNl + (s - lo)];
      if (a < 0) continue;
      const int e = goff[d] + bg[d] + a;
      tgt[e] = s;
      tslot[e] = toff[s] + r++;
      featidx[e] = -1;
      ebatch[e] = b;
      write_geom(pos[3 * d] - pos[3 * s], pos[3 * d + 1] - pos[3 * s + 1], pos[3 * d + 2] - pos[3 * s + 2], smooth_max, e,
                 dist, nvec, ew);
    }
  } else if (t < nL + n_bonds) {
    const int k = t - nL, s = bond_src[k], d = bond_dst[k];
    const int e = goff[d] + bond_grank[k];
    tgt[e] = s;
    tslot[e] = toff[s] + bond_trank[k];
    featidx[e] = k;
    ebatch[e] = batch[s];
    write_geom(pos[3 * d] - pos[3 * s], pos[3 * d + 1] - pos[3 * s + 1], pos[3 * d + 2] - pos[3 * s + 2], smooth_max, e,
               dist, nvec, ew);
  }
}
void launch_ll_fill(const float* pos, const int* batch, const int* ptr, int nL, int maxNl, const int* adjrank,
                    const int* goff, const int* toff, const int* bg, const int* bt, int n_bonds, const int* bond_src,
                    const int* bond_dst, const int* bond_grank, const int* bond_trank, float smooth_max, int* tgt,
                    int* tslot, int* featidx, int* ebatch, float* dist, float* nvec, float* ew, hipStream_t s) {
  const int n = nL + n_bonds;
  if (n <= 0) return;
  hipLaunchKernelGGL(k_ll_fill, dim3(cdiv(n, 64)), dim3(64), 0, s, pos, batch, ptr, nL, maxNl, adjrank, goff, toff, bg, bt,
                     n_bonds, bond_src, bond_dst, bond_grank, bond_trank, smooth_max, tgt, tslot, featidx, ebatch, dist,
                     nvec, ew);
  DDMI_CHECK_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------- cross graph
// radius(rec.pos / cut, lig.pos / cut, 1): pairs of one graph with |r/c - l/c|^2 < 1 (cg_model.py:543-545)
__device__ __forceinline__ bool cross_in_range(const float* lp, const float* rp, float cut, bool scaled) {
  if (scaled)
    return dist2_rn(rp[0] / cut, rp[1] / cut, rp[2] / cut, lp[0] / cut, lp[1] / cut, lp[2] / cut) < 1.0f;
  return dist2_rn(rp[0], rp[1], rp[2], lp[0], lp[1], lp[2]) < cut * cut;
}

// blocks [0, ceil(nL / 4)): one WAVE per ligand atom -- its residues are tested 64 at a time and ranked with a shuffle scan
// (ranks along the receptor index; a thread per atom walking 300-1500 residues was an 80-us latency chain per forward);
// the remaining blocks: one thread per receptor node (its graph's few ligand atoms).
__global__ __launch_bounds__(256) void k_cross_count(const float* __restrict__ lpos, const float* __restrict__ rpos,
                              const int* __restrict__ lbatch, const int* __restrict__ rbatch,
                              const int* __restrict__ lptr, const int* __restrict__ rptr, int nL, int nR, int maxNr,
                              const float* __restrict__ cutoff, float const_cutoff, const int* __restrict__ keep,
                              int* __restrict__ pairrank, int* __restrict__ cnt_l, int* __restrict__ cnt_r) {
  const int lig_blocks = (nL + 3) / 4;
  if ((int)blockIdx.x < lig_blocks) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= nL) return;
    const int b = lbatch[i], lo = rptr[b], hi = rptr[b + 1];
    const float cut = cutoff ? cutoff[b] : const_cutoff;
    const float lp[3] = {lpos[3 * i], lpos[3 * i + 1], lpos[3 * i + 2]};
    int* row = pairrank + (size_t)i * maxNr;
    int running = 0;
    for (int base = lo; base < hi; base += 64) {
      const int j = base + lane;
      const int hit = (j < hi && (!keep || keep[j]) && cross_in_range(lp, rpos + 3 * j, cut, cutoff != nullptr)) ? 1 : 0;
      int incl = hit;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl(incl, lane >= d ? lane - d : lane, 64);
        if (lane >= d) incl += up;
      }
      if (j < hi) row[j - lo] = hit ? running + incl - 1 : -1;
      running += __shfl(incl, 63, 64);
    }
    if (lane == 0) cnt_l[i] = running;
  } else {
    const int j = ((int)blockIdx.x - lig_blocks) * 256 + threadIdx.x;
    if (j >= nR) return;
    const int b = rbatch[j], lo = lptr[b], hi = lptr[b + 1];
    const float cut = cutoff ? cutoff[b] : const_cutoff;
    int c = 0;
    if (!keep || keep[j])
      for (int i = lo; i < hi; ++i) c += cross_in_range(lpos + 3 * i, rpos + 3 * j, cut, cutoff != nullptr);
    cnt_r[j] = c;
  }
}
void launch_cross_count(const float* lpos, const float* rpos, const int* lbatch, const int* rbatch, const int* lptr,
                        const int* rptr, int nL, int nR, int maxNr, const float* cutoff, float const_cutoff,
                        const int* keep, int* pairrank, int* cnt_l, int* cnt_r, hipStream_t s) {
  if (nL + nR <= 0) return;
  hipLaunchKernelGGL(k_cross_count, dim3(cdiv(nL, 4) + cdiv(nR, 256)), dim3(256), 0, s, lpos, rpos, lbatch, rbatch, lptr, rptr, nL,
                     nR, maxNr, cutoff, const_cutoff, keep, pairrank, cnt_l, cnt_r);
  DDMI_CHECK_HIP(hipGetLastError());
}

// thread per receptor node j.  L-major id e_l = offs_l[i] + rank_l(i,j) is the pair's attribute row;
// R-major id e_r = offs_r[j] + rank_r(j,i).   group 1 (target lig <- gather rec) lives in R-major order,
// group 3 (target rec <- gather lig) in L-major order; each one's tslot is the other's id.
__global__ void k_cross_fill(const float* __restrict__ lpos, const float* __restrict__ rpos,
                             const int* __restrict__ rbatch, const int* __restrict__ lptr, const int* __restrict__ rptr,
                             int nL, int nR, int maxNr, const int* __restrict__ pairrank, const int* __restrict__ offs_l,
                             const int* __restrict__ offs_r, const float* __restrict__ cutoff, float const_cutoff,
                             int smooth, int* __restrict__ g1_tgt, int* __restrict__ g1_tslot, int* __restrict__ g3_tgt,
                             int* __restrict__ g3_tslot, int* __restrict__ pbatch, float* __restrict__ pdist,
                             float* __restrict__ pnvec, float* __restrict__ pew, int rbase) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nR) return;
  const int b = rbatch[j], lo = lptr[b], hi = lptr[b + 1], jl = j - rptr[b];
  const float cut = cutoff ? cutoff[b] : const_cutoff;
  int rr = 0;
  for (int i = lo; i < hi; ++i) {
    const int rl = pairrank[(size_t)i * maxNr + jl];
    if (rl < 0) continue;
    const int e_l = offs_l[i] + rl, e_r = offs_r[j] + rr++;
    g1_tgt[e_r] = i;
    g1_tslot[e_r] = e_l;
    g3_tgt[e_l] = rbase + j;   // global id of the second node type (residues: nL, atoms: nL + nR)
    g3_tslot[e_l] = e_r;
    pbatch[e_l] = b;
    write_geom(rpos[3 * j] - lpos[3 * i], rpos[3 * j + 1] - lpos[3 * i + 1], rpos[3 * j + 2] - lpos[3 * i + 2],
               smooth ? cut : 0.f, e_l, pdist, pnvec, pew);
  }
}
void launch_cross_fill(const float* lpos, const float* rpos, const int* rbatch, const int* lptr, const int* rptr, int nL,
                       int nR, int maxNr, const int* pairrank, const int* offs_l, const int* offs_r, const float* cutoff,
                       float const_cutoff, int smooth, int* g1_tgt, int* g1_tslot, int* g3_tgt, int* g3_tslot,
                       int* pbatch, float* pdist, float* pnvec, float* pew, hipStream_t s, int rbase) {
  if (nR <= 0) return;
  hipLaunchKernelGGL(k_cross_fill, dim3(cdiv(nR, 64)), dim3(64), 0, s, lpos, rpos, rbatch, lptr, rptr, nL, nR, maxNr,
                     pairrank, offs_l, offs_r, cutoff, const_cutoff, smooth, g1_tgt, g1_tslot, g3_tgt, g3_tslot, pbatch,
                     pdist, pnvec, pew, rbase < 0 ? nL : rbase);
  DDMI_CHECK_HIP(hipGetLastError());
}

// cutoff_b = 3 * tr_sigma(t_b) + 20   (models/cg_model.py:321-322, utils/diffusion_utils.py:28-32)
__global__ void k_cross_cutoff(const float* __restrict__ t_tr, int B, float smin, float smax, float* __restrict__ out, int raw) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float sigma = raw ? t_tr[b] : powf(smin, 1.f - t_tr[b]) * powf(smax, t_tr[b]);   // confidence models: complex_t raw (cg_model.py:314-315)
  out[b] = sigma * 3.f + 20.f;
}
void launch_cross_cutoff(const float* t_tr, int B, float smin, float smax, float* out, hipStream_t s, int raw) {
  hipLaunchKernelGGL(k_cross_cutoff, dim3(cdiv(B, 64)), dim3(64), 0, s, t_tr, B, smin, smax, out, raw);
  DDMI_CHECK_HIP(hipGetLastError());
}

// -------------------------------------------------- per-step receptor cropping (crop_beyond)
// utils/utils.py:388-413 via utils/sampling.py:104-109: a residue survives iff it lies within `cutoff` of ANY
// ligand atom of its graph; contact edges survive iff both ends do.  The reference rebuilds the PyG batch on
// the host every step (deepcopy -> to_data_list -> crop -> from_data_list); here it is a mask over the static
// receptor plus a re-compaction of the static contact-graph CSRs (dropped residues simply keep no edges).
__global__ void k_crop_mask(const float* __restrict__ lpos, const float* __restrict__ rpos, const int* __restrict__ rbatch,
                            const int* __restrict__ lptr, int nR, float cut2, int* __restrict__ keep) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nR) return;
  const int b = rbatch[j];
  int k = 0;
  for (int i = lptr[b]; i < lptr[b + 1] && !k; ++i)
    k = dist2_rn(lpos[3 * i], lpos[3 * i + 1], lpos[3 * i + 2], rpos[3 * j], rpos[3 * j + 1], rpos[3 * j + 2]) < cut2;
  keep[j] = k;
}
void launch_crop_mask(const float* lpos, const float* rpos, const int* rbatch, const int* lptr, int nR, float cut2, int* keep,
                      hipStream_t s) {
  if (nR <= 0) return;
  hipLaunchKernelGGL(k_crop_mask, dim3(cdiv(nR, 64)), dim3(64), 0, s, lpos, rpos, rbatch, lptr, nR, cut2, keep);
  DDMI_CHECK_HIP(hipGetLastError());
}

// static CSRs: gather order (goff, src = tgt - nL, dst = the gather node) and target order (toff, tlist = gather-order id)
__global__ void k_rr_filter_count(const int* __restrict__ keep, const int* __restrict__ goff, const int* __restrict__ tgt,
                                  const int* __restrict__ toff, const int* __restrict__ tlist,
                                  const int* __restrict__ gnode, int nL, int nR, int* __restrict__ cnt_g,
                                  int* __restrict__ cnt_t) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nR) return;
  int cg = 0, ct = 0;
  if (keep[n]) {
    for (int e = goff[n]; e < goff[n + 1]; ++e) cg += keep[tgt[e] - nL];              // gather node n, target tgt[e]
    for (int ts = toff[n]; ts < toff[n + 1]; ++ts) ct += keep[gnode[tlist[ts]]];      // target node n, gather gnode[e]
  }
  cnt_g[n] = cg;
  cnt_t[n] = ct;
}
__global__ void k_rr_filter_tslot(const int* __restrict__ keep, const int* __restrict__ toff, const int* __restrict__ tlist,
                                  const int* __restrict__ gnode, const int* __restrict__ toff2, int nR,
                                  int* __restrict__ tslot_tmp) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nR || !keep[n]) return;
  int r = toff2[n];
  for (int ts = toff[n]; ts < toff[n + 1]; ++ts) {
    const int e = tlist[ts];
    if (keep[gnode[e]]) tslot_tmp[e] = r++;
  }
}
__global__ void k_rr_filter_fill(const int* __restrict__ keep, const int* __restrict__ goff, const int* __restrict__ tgt,
                                 const int* __restrict__ arow, const int* __restrict__ tslot_tmp,
                                 const int* __restrict__ goff2, int nL, int nR, int* __restrict__ tgt2,
                                 int* __restrict__ tslot2, int* __restrict__ arow2) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nR || !keep[n]) return;
  int q = goff2[n];
  for (int e = goff[n]; e < goff[n + 1]; ++e)
    if (keep[tgt[e] - nL]) { tgt2[q] = tgt[e]; tslot2[q] = tslot_tmp[e]; arow2[q] = arow[e]; ++q; }
}
void launch_rr_filter(const int* keep, const int* goff, const int* tgt, const int* arow, const int* toff, const int* tlist,
                      const int* gnode, int nL, int nR, int* cnt_g, int* cnt_t, int* goff2, int* toff2, int* tslot_tmp,
                      int* tgt2, int* tslot2, int* arow2, hipStream_t s) {
  if (nR <= 0) return;
  const dim3 grid(cdiv(nR, 64)), block(64);
  hipLaunchKernelGGL(k_rr_filter_count, grid, block, 0, s, keep, goff, tgt, toff, tlist, gnode, nL, nR, cnt_g, cnt_t);
  launch_exclusive_scan(cnt_g, goff2, nR, s);
  launch_exclusive_scan(cnt_t, toff2, nR, s);
  hipLaunchKernelGGL(k_rr_filter_tslot, grid, block, 0, s, keep, toff, tlist, gnode, toff2, nR, tslot_tmp);
  hipLaunchKernelGGL(k_rr_filter_fill, grid, block, 0, s, keep, goff, tgt, arow, tslot_tmp, goff2, nL, nR, tgt2, tslot2, arow2);
  DDMI_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- rotatable-bond graph
// thread per rotatable bond t: atoms of the same graph within r of the bond midpoint, padded [nT][cap]
__global__ void k_tor_radius(const float* __restrict__ pos, const int* __restrict__ ptr, const int* __restrict__ tor_u,
                             const int* __restrict__ tor_v, const int* __restrict__ tor_batch, int nT, float r2, int cap,
                             float smooth_max, int* __restrict__ cnt, int* __restrict__ atom, float* __restrict__ dist,
                             float* __restrict__ nvec, float* __restrict__ ew, float* __restrict__ bond_nvec) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nT) return;
  const int u = tor_u[t], v = tor_v[t], b = tor_batch[t];
  const float mx = (pos[3 * u] + pos[3 * v]) / 2, my = (pos[3 * u + 1] + pos[3 * v + 1]) / 2,
              mz = (pos[3 * u + 2] + pos[3 * v + 2]) / 2;
  {
    const float bx = pos[3 * v] - pos[3 * u], by = pos[3 * v + 1] - pos[3 * u + 1], bz = pos[3 * v + 2] - pos[3 * u + 2];
    const float inv = 1.f / fmaxf(sqrtf(bx * bx + by * by + bz * bz), 1e-12f);
    bond_nvec[3 * t] = bx * inv; bond_nvec[3 * t + 1] = by * inv; bond_nvec[3 * t + 2] = bz * inv;
  }
  int c = 0;
  for (int a = ptr[b]; a < ptr[b + 1] && c < cap; ++a) {
    if (dist2_rn(pos[3 * a], pos[3 * a + 1], pos[3 * a + 2], mx, my, mz) < r2) {
      const int e = t * cap + c++;
      atom[e] = a;
      write_geom(pos[3 * a] - mx, pos[3 * a + 1] - my, pos[3 * a + 2] - mz, smooth_max, e, dist, nvec, ew);
    }
  }
  cnt[t] = c;
  for (int q = c; q < cap; ++q) {  // padding slots: valid indices, zero geometry, zero weight
    const int e = t * cap + q;
    atom[e] = u;
    dist[e] = 0.f;
    nvec[3 * e] = nvec[3 * e + 1] = nvec[3 * e + 2] = 0.f;
    if (ew) ew[e] = 0.f;
  }
}
void launch_tor_radius(const float* pos, const int* ptr, const int* tor_u, const int* tor_v, const int* tor_batch, int nT,
                       float r, int cap, float smooth_max, int* cnt, int* atom, float* dist, float* nvec, float* ew,
                       float* bond_nvec, hipStream_t s) {
  if (nT <= 0) return;
  hipLaunchKernelGGL(k_tor_radius, dim3(cdiv(nT, 64)), dim3(64), 0, s, pos, ptr, tor_u, tor_v, tor_batch, nT, r * r, cap,
                     smooth_max, cnt, atom, dist, nvec, ew, bond_nvec);
  DDMI_CHECK_HIP(hipGetLastError());
}

}  // namespace ddmi
