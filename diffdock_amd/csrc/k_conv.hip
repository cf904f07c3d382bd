// The equivariant graph-convolution layer (reference TensorProductConvLayer.forward,
// models/tensor_layers.py:309-335, tp_scatter_simple/_multigroup :125-231,
// FasterTensorProduct :71-122 / e3nn FullyConnectedTensorProduct) re-associated for the
// matrix cores.  Per edge e = (target s <- gather node d) the reference computes
//     w_e  = W2 * relu(W1 * [edge_attr_e, x_s[:ns], x_d[:ns]] + b1) + b2        [weight_numel]
//     m_e  = TP(x_d, sh_e; w_e)              out_s = BN(mean_e m_e) + pad(x_s)
// which is tri-linear in (h_e = relu(..) (+) 1, x_d, sh_e).  Contracting x_d with W2 FIRST
// (per gather node, shared by all of its edges) cuts the multiply-adds per edge from
// K*weight_numel (~1.0 M at ns=48) to K*NT (~76 k), K = 3ns+1, NT = sum_paths din*mul_out:
//   Y[d][k][n]  = sum_u x_d[u,i] * W2[k][slot(u,w)]          (n = (path,i,w))            node contraction
//   T[e][n]     = sum_k h_e[k] * Y[d(e)][k][n]               (MFMA 16x16x4 f32)           edge product
//   m_e[o,w,k'] = sum_{paths,i,j} C[i][j][k'] sh_e[j] T[e][path,i,w]                       coupling with sh_e
//   k_reduce_bn : deterministic segmented mean over the target-CSR, BatchNorm, residual
// k_conv_fused does the first three per tile of 16 virtual nodes with Y never leaving LDS (every edge group of every layer);
// k_edge_hidden(_mm) produces h_e in its A-fragment order.  Results equal the reference up to fp32 re-association.
#include "k_conv_tile.h"

namespace ddmi {

// ------------------------------------------------------------------ fused contraction + edge kernel
// Virtual nodes: gather node d with edges [goff[d], goff[d+1]) becomes ceil(deg/32) entries (d, first edge).
__global__ void k_vn_count(const int* __restrict__ goff, int gcount, int* __restrict__ cnt) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= gcount) return;
  cnt[d] = (goff[d + 1] - goff[d] + 31) >> 5;
}
__global__ void k_vn_fill(const int* __restrict__ goff, const int* __restrict__ voff, int gcount, int* __restrict__ vn_node,
                          int* __restrict__ vn_e0) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= gcount) return;
  const int v0 = voff[d], n = voff[d + 1] - v0, e0 = goff[d], e1 = goff[d + 1];
  for (int b = 0; b < n; ++b) { vn_node[v0 + b] = d; vn_e0[v0 + b] = min(e0 + 32 * b, e1); }
}
// tile_per_pose (ddmi_exec_options): the same lists with every graph of the batch padded to whole 16-node tiles by DEAD virtual
// nodes (node = the graph's last gather node, no edges), so that no tile of k_conv_fused spans two graphs: which virtual nodes
// share a tile -- and with it the summation trees inside the tile -- then depends on the graph alone, not on its neighbours in
// the batch.  graph_ptr: first gather node of every graph (local index), node_batch: graph of every gather node; *nvn_pad = the
// padded list length.
__global__ void k_vn_fill_pp(const int* __restrict__ goff, const int* __restrict__ voff, int gcount, const int* __restrict__ node_batch,
                             const int* __restrict__ graph_ptr, int n_graphs, int* __restrict__ vn_node, int* __restrict__ vn_e0,
                             int* __restrict__ nvn_pad) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= gcount) return;
  const int b = node_batch[d];
  int pad = 0;
  for (int q = 0; q < b; ++q) { const int n = voff[graph_ptr[q + 1]] - voff[graph_ptr[q]]; pad += ((n + 15) & ~15) - n; }
  const int v0 = voff[d] + pad, n = voff[d + 1] - voff[d], e0 = goff[d], e1 = goff[d + 1];
  for (int j = 0; j < n; ++j) { vn_node[v0 + j] = d; vn_e0[v0 + j] = min(e0 + 32 * j, e1); }
  if (d == graph_ptr[b + 1] - 1) {   // last gather node of its graph: dead virtual nodes up to the tile boundary
    const int nb = voff[graph_ptr[b + 1]] - voff[graph_ptr[b]];
    const int padb = ((nb + 15) & ~15) - nb;
    for (int j = 0; j < padb; ++j) { vn_node[v0 + n + j] = d; vn_e0[v0 + n + j] = e1; }
  }
  if (d == gcount - 1) {   // padded list length: from ONE thread over ALL graphs (a last graph without gather nodes of this type must not leave it unwritten)
    int tot = 0;
    for (int q = 0; q < n_graphs; ++q) { const int nq = voff[graph_ptr[q + 1]] - voff[graph_ptr[q]]; tot += ((nq + 15) & ~15) - nq; }
    *nvn_pad = voff[gcount] + tot;
  }
}
// Per-edge rows of the fused kernel, once per forward and edge group (the six layers share the graph): virtual node v, edge
// row r -> [spherical harmonics (SHD) | edge weight | message row | pad] at stride ES, zero rows behind the node's last edge
// and for the dead virtual nodes of the last 16-node tile; vn_ne[v] = edges of the virtual node.  The tile prologue of
// k_conv_fused then is one coalesced copy instead of the chain vn_e0 -> arow -> nvec / weight / slot.
template <int SHD, int ES, int VPB>   // VPB virtual nodes per workgroup: 8, or 16 = one tile (then the workgroup also writes the tile's pre-reduction header)
__device__ __forceinline__ void vn_rows_body(const VnRowsArgs& a, const int block) {
  const int v = block * VPB + (threadIdx.x >> 5), r = threadIdx.x & 31;
  const int nvn = *a.nvn;
  if (v >= ((nvn + 15) & ~15)) return;   // whole 16-node tiles (FC_VN); workgroup-uniform for VPB = 16
  int ne = 0, e0 = 0;
  if (v < nvn) { e0 = a.vn_e0[v]; ne = min(32, a.goff[a.vn_node[v] + 1] - e0); }
  float sh[9] = {1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float we = 0.f;
  int ts = 0, ar = 0, tg = 0;
  if (r < ne) {
    const int e = e0 + r;
    ar = a.arow ? a.arow[e] : e;
    edge_sh(a.nvec + (size_t)ar * 3, a.sgn, a.sh_lmax, sh);
    we = a.ew ? a.ew[ar] : 1.f;
    ts = a.tslot[e];
    tg = a.tgt[e] - a.tbase;
  } else if (ne > 0) {   // rows past the node's last edge: the tile's first edge (valid memory for k_edge_hidden_mm, which stores zeros there)
    const int e = e0 + ((r & 16) < ne ? (r & 16) : 0);
    ar = a.arow ? a.arow[e] : e;
    tg = a.tgt[e] - a.tbase;
  }
  float* er = a.rows + ((size_t)v * 32 + r) * ES;
#pragma unroll
  for (int j = 0; j < SHD; ++j) er[j] = sh[j];
  er[SHD] = we;
  reinterpret_cast<int*>(er)[SHD + 1] = ts;
#pragma unroll
  for (int j = SHD + 2; j < ES; ++j) er[j] = 0.f;
  if (ES >= SHD + 4) {   // (l <= 1 rows: attribute row and target row of the edge for the first-layer kernel)
    reinterpret_cast<int*>(er)[SHD + 2] = ar;
    reinterpret_cast<int*>(er)[SHD + 3] = tg;
  }
  if (r == 0) a.vn_ne[v] = ne;
  if constexpr (VPB == 16 && SHD == 4) {
    // tile header + live flags of the in-tile pre-reduction (k_vn_tiles' logic on the rows this workgroup has just produced: thread
    // = edge row e = 32 * (virtual node in the tile) + row, the key k_vn_tiles uses)
    if (a.tile_hdr) {   // (workgroup-uniform)
      __shared__ int smin, smax, key[FC_TILE_NT], sts[512];
      const int tid = threadIdx.x;
      sts[tid] = ts;
      if (tid == 0) { smin = 0x7fffffff; smax = -0x7fffffff; }
      if (tid < FC_TILE_NT) key[tid] = 0x7fffffff;
      __syncthreads();
      const bool ok = v < nvn && r < ne;
      if (ok) { atomicMin(&smin, tg); atomicMax(&smax, tg); }
      __syncthreads();
      const int t0 = smin, nt = smax - smin + 1;
      const bool pre = smin <= smax && nt <= FC_TILE_NT;
      if (pre && ok) atomicMin(&key[tg - t0], tid);
      __syncthreads();
      if (ok) a.live[ts] = (!pre || key[tg - t0] == tid) ? 1 : 0;
      int* __restrict__ h = a.tile_hdr + (size_t)block * FC_TILE_HDR;
      if (tid < FC_TILE_NT) {
        const int k = key[tid];
        h[4 + tid] = (pre && k != 0x7fffffff) ? sts[k] : -1;   // the representative's message row
      }
      if (tid == 0) { h[0] = pre ? 1 : 0; h[1] = t0; h[2] = pre ? nt : 0; h[3] = 0; }
    }
  }
}
template <int SHD, int ES>
__global__ __launch_bounds__(256) void k_vn_rows(VnRowsArgs a) { vn_rows_body<SHD, ES, 8>(a, (int)blockIdx.x); }
// the per-edge rows of several edge groups in one launch (grouped dispatch): workgroups [first[g], first[g + 1]) serve group g
template <int SHD, int ES>
__global__ __launch_bounds__(512) void k_vn_rows_grouped(VnRowsGroupedArgs G) {   // workgroup = one tile of 16 virtual nodes
  const int b = (int)blockIdx.x;
  int g = 0;
#pragma unroll
  for (int i = 1; i < VN_GROUPS_MAX; ++i)
    if (i < G.n && b >= G.first[i]) g = i;
  vn_rows_body<SHD, ES, 16>(G.g[g], b - G.first[g]);
}

// Virtual-node lists of several edge groups in ONE launch (round 6): workgroup = edge group; count -> scan -> fill in one pass
// over the group's gather nodes (thread t owns the nodes [t * per, (t + 1) * per): its running offset is the block scan of the
// chunk sums), instead of k_vn_count -> k_exclusive_scan -> k_vn_fill per group (three dependent launches each, 12 per forward).
// node_batch != nullptr: tile_per_pose lists (every graph padded to whole 16-node tiles by dead virtual nodes, k_vn_fill_pp).
__global__ __launch_bounds__(1024) void k_vn_lists(VnListsArgs A) {
  __shared__ int part[1024];
  const VnListArgs& a = A.g[blockIdx.x];
  const int t = threadIdx.x, n = a.gcount;
  const int per = (n + 1023) / 1024;
  const int lo = min(t * per, n), hi = min(lo + per, n);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += (a.goff[i + 1] - a.goff[i] + 31) >> 5;
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - s;
  const int total = part[1023];
  if (!a.node_batch) {
    for (int i = lo; i < hi; ++i) {
      const int e0 = a.goff[i], e1 = a.goff[i + 1], c = (e1 - e0 + 31) >> 5;
      a.voff[i] = run;
      for (int b = 0; b < c; ++b) { a.vn_node[run + b] = i; a.vn_e0[run + b] = min(e0 + 32 * b, e1); }
      run += c;
    }
    if (t == 1023) a.voff[n] = total;
    return;
  }
  for (int i = lo, r2 = run; i < hi; ++i) { a.voff[i] = r2; r2 += (a.goff[i + 1] - a.goff[i] + 31) >> 5; }
  if (t == 1023) a.voff[n] = total;
  __syncthreads();   // voff of the whole group (written by this workgroup) before the per-graph sizes are read
  for (int i = lo; i < hi; ++i) {
    const int b = a.node_batch[i];
    int pad = 0;
    for (int q = 0; q < b; ++q) { const int nq = a.voff[a.graph_ptr[q + 1]] - a.voff[a.graph_ptr[q]]; pad += ((nq + 15) & ~15) - nq; }
    const int e0 = a.goff[i], e1 = a.goff[i + 1], c = (e1 - e0 + 31) >> 5, v0 = a.voff[i] + pad;
    for (int j = 0; j < c; ++j) { a.vn_node[v0 + j] = i; a.vn_e0[v0 + j] = min(e0 + 32 * j, e1); }
    if (i == a.graph_ptr[b + 1] - 1) {   // last gather node of its graph: dead virtual nodes up to the tile boundary
      const int nb = a.voff[a.graph_ptr[b + 1]] - a.voff[a.graph_ptr[b]];
      const int padb = ((nb + 15) & ~15) - nb;
      for (int j = 0; j < padb; ++j) { a.vn_node[v0 + c + j] = i; a.vn_e0[v0 + c + j] = e1; }
    }
  }
  if (t == 0) {
    int tot = 0;
    for (int q = 0; q < a.n_graphs; ++q) { const int nq = a.voff[a.graph_ptr[q + 1]] - a.voff[a.graph_ptr[q]]; tot += ((nq + 15) & ~15) - nq; }
    *a.nvn_pad = total + tot;
  }
}
void launch_vn_build_all(const VnListsArgs& L, const VnRowsArgs* rows, int sh_lmax, hipStream_t s) {
  if (L.n <= 0) return;
  hipLaunchKernelGGL(k_vn_lists, dim3(L.n), dim3(1024), 0, s, L);
  VnRowsGroupedArgs R;
  int blocks = 0;
  for (int i = 0; i < L.n; ++i) {
    if (!rows[i].rows) continue;
    VnRowsArgs r = rows[i];
    r.nvn = L.g[i].node_batch ? L.g[i].nvn_pad : L.g[i].voff + L.g[i].gcount;
    r.vn_node = L.g[i].vn_node; r.vn_e0 = L.g[i].vn_e0; r.goff = L.g[i].goff;
    R.first[R.n] = blocks;
    R.g[R.n++] = r;
    blocks += round_up(r.vcap, 16) / 16;
  }
  R.first[R.n] = blocks;
  if (blocks > 0) {
    if (sh_lmax <= 1) hipLaunchKernelGGL((k_vn_rows_grouped<4, 8>), dim3(blocks), dim3(512), 0, s, R);
    else hipLaunchKernelGGL((k_vn_rows_grouped<9, 12>), dim3(blocks), dim3(512), 0, s, R);
  }
  DDMI_CHECK_HIP(hipGetLastError());
}
void launch_vn_build(const int* goff, int gcount, int* cnt_tmp, int* voff, int* vn_node, int* vn_e0, const VnRowsArgs& rows_in,
                     hipStream_t s, const VnPoseTiles* pp) {
  if (gcount <= 0) return;
  hipLaunchKernelGGL(k_vn_count, dim3(cdiv(gcount, 256)), dim3(256), 0, s, goff, gcount, cnt_tmp);
  launch_exclusive_scan(cnt_tmp, voff, gcount, s);
  if (pp) hipLaunchKernelGGL(k_vn_fill_pp, dim3(cdiv(gcount, 256)), dim3(256), 0, s, goff, voff, gcount, pp->node_batch, pp->graph_ptr,
                             pp->n_graphs, vn_node, vn_e0, pp->nvn_pad);
  else hipLaunchKernelGGL(k_vn_fill, dim3(cdiv(gcount, 256)), dim3(256), 0, s, goff, voff, gcount, vn_node, vn_e0);
  if (rows_in.rows) {
    VnRowsArgs r = rows_in;
    r.nvn = pp ? pp->nvn_pad : voff + gcount; r.vn_node = vn_node; r.vn_e0 = vn_e0; r.goff = goff;
    const dim3 grid((unsigned)cdiv(round_up(r.vcap, 16), 8));
    if (r.sh_lmax <= 1) hipLaunchKernelGGL((k_vn_rows<4, 8>), grid, dim3(256), 0, s, r);
    else hipLaunchKernelGGL((k_vn_rows<9, 12>), grid, dim3(256), 0, s, r);
  }
  DDMI_CHECK_HIP(hipGetLastError());
}

// Tile headers of the in-tile pre-reduction (kernels.h, launch_vn_tiles): one workgroup per tile of 16 virtual nodes, thread
// per pair of edge rows.  The representative of a target is the tile's FIRST edge (virtual node, row) that addresses it.
__global__ __launch_bounds__(256) void k_vn_tiles(const int* __restrict__ nvn_p, const float* __restrict__ vrows,
                                                 const int* __restrict__ vn_ne, int* __restrict__ hdr, unsigned char* __restrict__ live) {
  __shared__ int smin, smax, key[FC_TILE_NT];
  const int nvn = *nvn_p, v0 = blockIdx.x * 16, tid = threadIdx.x;
  if (v0 >= nvn) return;
  if (tid == 0) { smin = 0x7fffffff; smax = -0x7fffffff; }
  if (tid < FC_TILE_NT) key[tid] = 0x7fffffff;
  __syncthreads();
  const int* __restrict__ rw = reinterpret_cast<const int*>(vrows);
  int tg[2], ts[2]; bool ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + 256 * i, v = v0 + (e >> 5), r = e & 31;
    ok[i] = v < nvn && r < vn_ne[v];
    tg[i] = ok[i] ? rw[((size_t)v * 32 + r) * 8 + 7] : 0;
    ts[i] = ok[i] ? rw[((size_t)v * 32 + r) * 8 + 5] : 0;
    if (ok[i]) { atomicMin(&smin, tg[i]); atomicMax(&smax, tg[i]); }
  }
  __syncthreads();
  const int t0 = smin, nt = smax - smin + 1;
  const bool pre = smin <= smax && nt <= FC_TILE_NT;
  if (pre) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (ok[i]) atomicMin(&key[tg[i] - t0], tid + 256 * i);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if (ok[i]) live[ts[i]] = (!pre || key[tg[i] - t0] == tid + 256 * i) ? 1 : 0;
  int* __restrict__ h = hdr + (size_t)blockIdx.x * FC_TILE_HDR;
  if (tid < FC_TILE_NT) {
    const int k = key[tid];
    h[4 + tid] = (pre && k != 0x7fffffff) ? rw[((size_t)(v0 + (k >> 5)) * 32 + (k & 31)) * 8 + 5] : -1;
  }
  if (tid == 0) { h[0] = pre ? 1 : 0; h[1] = t0; h[2] = pre ? nt : 0; h[3] = 0; }
}
void launch_vn_tiles(const int* nvn, int vcap, const float* vrows, const int* vn_ne, int* tile_hdr, unsigned char* live, hipStream_t s) {
  if (vcap <= 0) return;
  hipLaunchKernelGGL(k_vn_tiles, dim3(cdiv(vcap, 16)), dim3(256), 0, s, nvn, vrows, vn_ne, tile_hdr, live);
  DDMI_CHECK_HIP(hipGetLastError());
}

static_assert(FC_VN == 16, "k_vn_rows pads the per-edge rows to whole 16-node tiles");

// the kernel instantiations live in their own translation units (k_conv_f32.hip, k_conv_bf.hip, k_conv_l2.hip)
#define FC_EXTERN(MAXD, SHD, MODE, NBK, BF) extern template void launch_conv_fused_k<MAXD, SHD, MODE, NBK, BF>(const FusedConvArgs&, hipStream_t);
FC_EXTERN(3, 4, 1, 4, false) FC_EXTERN(3, 4, 0, 4, false) FC_EXTERN(3, 4, 3, 4, false) FC_EXTERN(3, 4, 4, 4, false)
FC_EXTERN(3, 4, 0, 5, false) FC_EXTERN(3, 4, 3, 5, false) FC_EXTERN(3, 4, 4, 5, false)
FC_EXTERN(3, 4, 0, 4, true) FC_EXTERN(3, 4, 3, 4, true) FC_EXTERN(3, 4, 4, 4, true)
FC_EXTERN(3, 4, 0, 5, true) FC_EXTERN(3, 4, 3, 5, true) FC_EXTERN(3, 4, 4, 5, true)
FC_EXTERN(3, 9, 1, 4, false) FC_EXTERN(3, 9, 0, 4, false) FC_EXTERN(3, 9, 3, 4, false)
FC_EXTERN(5, 9, 1, 4, false) FC_EXTERN(5, 9, 0, 4, false) FC_EXTERN(5, 9, 3, 4, false)
#undef FC_EXTERN
#ifdef DDMI_PROFILING   // phase clocks / workgroup stamps of every kernel TU (profiling builds only)
void fc_prof_report_f32(); void fc_prof_report_bf(); void fc_prof_report_l2(); void fc_prof_report_grp();
void fc_prof_report() { fc_prof_report_f32(); fc_prof_report_bf(); fc_prof_report_l2(); fc_prof_report_grp(); }
#endif

void launch_conv_fused(const FusedConvArgs& a_in, hipStream_t s) {
  if (a_in.vcap <= 0 || a_in.ysplit <= 0) return;
  FusedConvArgs a = a_in;
  a.dbg = ablate_mask();
#ifdef FCV_QUICK   // ISA inspection builds: one instantiation only (hipcc -DFCV_QUICK -save-temps; never linked into the library)
  launch_conv_fused_k<3, 4, 3, 4, true>(a, s);
  return;
#endif
  // In-tile pre-reduction contract: k_reduce_bn trusts ReduceGroup.live (it reads only the flagged rows of the group), and only
  // the MODE 0 / 3 instantiations with l <= 1 rows (PRE_OK in the kernel) honour tile_hdr -- any other route would write one row
  // per edge while the reducer skips most of them.  set_complex keeps the two in step; a launch that breaks it fails here.
  if (a.tile_hdr && (a.generic || a.shared || a.maxd > 3 || a.sh_lmax > 1))
    throw Error(DDMI_ERR_STATE, "k_conv_fused: pre-reduced group routed to a kernel variant without the pre-reducing epilogue");
  // the predicated variant (MODE 1) walks classic 4-slot granules only: a packed / merged granule there would be mis-read
  if (a.generic && a.max_nb > 4) throw Error(DDMI_ERR_ARG, "k_conv_fused: packed granule in a generic layer (weights.cpp builds those layers unpacked)");
  if (a.maxd <= 3 && a.sh_lmax <= 1) {   // the l <= 1 tensor product (FasterTensorProduct structure): static chain shapes, packed granules
    if (a.bf && a.generic) throw Error(DDMI_ERR_ARG, "k_conv_fused: bf16 edge product requested for a generic layer (complex.cpp falls back to f32 there)");
    if (a.generic) launch_conv_fused_k<3, 4, 1, 4>(a, s);
    else if (a.bf && a.max_nb > 4) {
      if (a.dense && a.shared) launch_conv_fused_k<3, 4, 4, 5, true>(a, s);
      else if (a.dense) launch_conv_fused_k<3, 4, 3, 5, true>(a, s);
      else launch_conv_fused_k<3, 4, 0, 5, true>(a, s);
    } else if (a.bf) {
      if (a.dense && a.shared) launch_conv_fused_k<3, 4, 4, 4, true>(a, s);
      else if (a.dense) launch_conv_fused_k<3, 4, 3, 4, true>(a, s);
      else launch_conv_fused_k<3, 4, 0, 4, true>(a, s);
    } else if (a.max_nb > 4) {
      if (a.dense && a.shared) launch_conv_fused_k<3, 4, 4, 5>(a, s);
      else if (a.dense) launch_conv_fused_k<3, 4, 3, 5>(a, s);
      else launch_conv_fused_k<3, 4, 0, 5>(a, s);
    } else {
      if (a.dense && a.shared) launch_conv_fused_k<3, 4, 4, 4>(a, s);
      else if (a.dense) launch_conv_fused_k<3, 4, 3, 4>(a, s);
      else launch_conv_fused_k<3, 4, 0, 4>(a, s);
    }
  } else if (a.bf) {
    throw Error(DDMI_ERR_ARG, "k_conv_fused: bf16 edge product requested with sh_lmax > 1 (complex.cpp falls back to f32 there)");
  } else if (a.maxd <= 3) {
    if (a.generic) launch_conv_fused_k<3, 9, 1, 4>(a, s);
    else if (a.dense) launch_conv_fused_k<3, 9, 3, 4>(a, s);
    else launch_conv_fused_k<3, 9, 0, 4>(a, s);
  } else {
    if (a.generic) launch_conv_fused_k<5, 9, 1, 4>(a, s);
    else if (a.dense) launch_conv_fused_k<5, 9, 3, 4>(a, s);
    else launch_conv_fused_k<5, 9, 0, 4>(a, s);
  }
}

// ---------------------------------------------------------------------- reduce + BN
// One workgroup per target node, 4 waves: wave w sums the message rows b + w, b + w + 4, ... of every group, a lane 4
// consecutive columns (one 16-B streaming load per row and lane, several rows in flight); the four partial rows meet in LDS.
__global__ __launch_bounds__(256) void k_reduce_bn(const ReduceGroup* __restrict__ groups, int n_groups, int nbase,
                                                   int D_in, int D_out, const float* __restrict__ bn_mean,
                                                   const float* __restrict__ bn_scale, const float* __restrict__ bn_bias,
                                                   int residual, const float* __restrict__ X_in,
                                                   float* __restrict__ X_out, int out_stride) {
  __shared__ float red[4][XS + 4];
  const int s = nbase + blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = DDMI_UNIFORM(tid >> 6);
  const bool live = 4 * lane < D_out;      // columns past D_out inside the XS-wide row are never used
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int cnt = 0;
  for (int g = 0; g < n_groups; ++g) {
    const ReduceGroup G = groups[g];
    const int sl = s - G.tbase;
    if (sl < 0 || sl >= G.tcount) continue;
    const int b = G.toff[sl], e = G.toff[sl + 1];
    cnt += e - b;
    if (G.live) {   // pre-reduced group: only the rows flagged live hold (partial) sums; wave w takes the live rows of ordinal w, w + 4, ...
      const float* __restrict__ mp = G.msg + 4 * lane;
      int ord = 0;
      for (int base = b; base < e; base += 64) {
        const int rr = base + lane;
        unsigned long long mask = __ballot(rr < e && G.live[rr] != 0);
        while (mask) {   // up to four of this wave's rows in flight (wave-uniform control flow)
          int rows[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            rows[i] = -1;
            while (mask && rows[i] < 0) {
              const int bit = __builtin_ctzll(mask);
              mask &= mask - 1;
              if ((ord++ & 3) == wave) rows[i] = base + bit;
            }
          }
          float4 v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = (live && rows[i] >= 0) ? nt_load4(mp + (size_t)rows[i] * XS) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (rows[i] >= 0) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
        }
      }
    } else if (live) {
      const float* __restrict__ mp = G.msg + 4 * lane;
      int r = b + wave;
      for (; r + 12 < e; r += 16) {
        const float4 v0 = nt_load4(mp + (size_t)r * XS), v1 = nt_load4(mp + (size_t)(r + 4) * XS);
        const float4 v2 = nt_load4(mp + (size_t)(r + 8) * XS), v3 = nt_load4(mp + (size_t)(r + 12) * XS);
        acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;   // same order as the one-row-at-a-time tail
        acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
        acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
        acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
      }
      for (; r < e; r += 4) {
        const float4 v0 = nt_load4(mp + (size_t)r * XS);
        acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
      }
    }
  }
  if (4 * lane < XS) *reinterpret_cast<float4*>(&red[wave][4 * lane]) = acc;
  __syncthreads();
  const int c = tid;
  if (c >= out_stride) return;
  float v = 0.f;
  if (c < D_out) {
    const float sum = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    v = cnt > 0 ? sum / (float)cnt : 0.f;
    if (bn_scale) v = (v - bn_mean[c]) * bn_scale[c] + bn_bias[c];
    if (residual && c < D_in) v += X_in[(size_t)s * XS + c];
  }
  X_out[(size_t)s * out_stride + c] = v;
}

void launch_reduce_bn(const ReduceGroup* groups_dev, int n_groups, int nbase, int ncount, int D_in, int D_out,
                      const float* bn_mean, const float* bn_scale, const float* bn_bias, int residual,
                      const float* X_in, float* X_out, int out_stride, hipStream_t s) {
  if (ncount <= 0) return;
  hipLaunchKernelGGL(k_reduce_bn, dim3(ncount), dim3(256), 0, s, groups_dev, n_groups, nbase, D_in, D_out, bn_mean,
                     bn_scale, bn_bias, residual, X_in, X_out, out_stride);
  DDMI_CHECK_HIP(hipGetLastError());
}

}  // namespace ddmi
