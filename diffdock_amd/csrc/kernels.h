// Launch wrappers of the gfx950 kernels (definitions in k_*.hip).  All pointers are device
// pointers; `stream` is the caller's HIP stream.
#pragma once
#include "ddmi_common.h"

namespace ddmi {

// ---------------------------------------------------------------- k_gemm.hip
// C[M,N] = act(A[M,K] * W[N,K]^T + bias[N] + rowbias[ridx[m]][N]); row-major, fp32, f32 MFMA.
// m_dev (optional) = device int with the live row count (<= M): row tiles past it exit.
struct GemmArgs {
  const float* A = nullptr; int lda = 0;
  const float* W = nullptr; int ldw = 0;
  const float* bias = nullptr;
  const float* rowbias = nullptr; const int* ridx = nullptr; int ldrb = 0;
  float* C = nullptr; int ldc = 0;
  int M = 0, N = 0, K = 0;
  const int* m_dev = nullptr;
  int act = 0;  // 0 none, 1 relu, 2 tanh
};
void launch_gemm(const GemmArgs& a, hipStream_t s);
constexpr int GEMM_BATCH_MAX = 16;
struct GemmBatch { GemmArgs g[GEMM_BATCH_MAX]; int n = 0; };
void launch_gemm_batch(const GemmBatch& b, hipStream_t s);   // independent problems, one launch

// ---------------------------------------------------------------- k_conv.hip
// Column layout of a contracted node row Y_d[k][n] ("item-major"): for every output block ob and every output
// channel w one ITEM of itemw = 4, 8, .. 64 consecutive columns holding the (path, i) terms that feed that output
// channel, so that after the edge GEMM one lane (or 2/4 neighbouring lanes) owns everything an output needs.
struct ObInfo { int base, itemw, mul, o_off, dout; };       // columns [base, base + mul*itemw)
struct DevPath {    // coupling descriptor of one tensor-product path
  int n_off, mul_out, din, ds, dout, s_off, c_off, o_off;
  int mul_in, i_off, w_off;   // used by the per-edge-weight form (k_tp_apply)
  int g_off;                  // offset of this path's [din][dout] block in the per-edge coupling vector G
};
struct GEntry { int c_idx, s_off, ds, dout; };              // G[g] = sum_j ctab[c_idx + j*dout] * sh[s_off + j]
struct CgItem { int path_begin, path_end, o_off, dout, w; };  // (output block, w) work item of k_tp_apply
struct NcSlot { int x_off, din, comp, mul_in, u_pad, w_pad, wk_off; };   // one column of an item: (path, input component)
constexpr int NC_MAXITEM = 64;
struct NcUnit {     // one (output block, 16-wide w tile) unit of the node contraction: n_w items x itemw columns
  int col_base, itemw, w0, n_w;
  NcSlot slot[NC_MAXITEM];  // slot[s].din == 0 -> padding column (written as 0)
};

// ---- the contracted rows never leave the CU (k_conv_fused).
// A VIRTUAL NODE = one gather node with up to 32 of its edges (nodes with more edges appear several times, nodes
// without edges not at all).  A workgroup owns 16 virtual nodes and walks GRANULES of fused columns: per 8-row k chunk it
// contracts the 16 x rows with the packed second-layer weights into LDS (v_mfma_f32_16x16x4_f32, M = nodes) and
// immediately multiplies the chunk into the edge accumulators (M = edges), then couples with the spherical harmonics and
// writes the message columns of the granule.
// CLASSIC granule: 4 slots (= (path, input component) columns) of one 16-channel tile, column 16*slot + w; an output block
// fed by more than 4 slots takes several granules whose message columns add up.
// PACKED granule (output blocks of <= 10 channels, e.g. the 10 vector channels): ALL slots (<= 8) of the block in one
// granule -- channels 0..7 of slots 2b and 2b+1 share column block b (column 8*slot + w), channels 8, 9 of every slot sit
// in one tail block (column 16*(nb-1) + 2*slot + w - 8): 7 slots x 10 channels take 5 column blocks instead of two
// classic granules with 7, one pass and no add.
constexpr int FC_MAXSLOT = 8;
struct FGran {
  NcSlot slot[FC_MAXSLOT];   // classic: 4 entries, din == 0 -> padding (zero); packed: nslot entries
  int w0, n_w;           // lanes lr < n_w carry output channel w0 + lr (classic)
  int o_off, dout;       // message columns o_off + w*dout + k'
  int g[FC_MAXSLOT];     // slot s couples through gmap[g[s] .. g[s]+dout), -1 = padding
  int accumulate;        // 0: first granule of its (output block, w tile) unit writes, later ones add
  int empty;             // no path reaches this granule: the message columns are zero
  int shape;             // chain-length class: classic 1 = (12,3,3,3), 2 = (3,3,3,3), 3 = (12,-,-,-), 0 = generic; 7 = three 12-step chains of ONE
                         // path, slot s = channel tile 16*s (n_w = 48: the slots feed different output channels, weights.cpp);
                         // packed 4 = 12 | 3x3 | 3x3 (7 slots), 5 = 3x3 | 3x3 (6 slots), 6 = 12 | 3x3 (4 slots): a 12-step chain, then
                         // groups of three 3-step chains (the components of one path, sharing their weights)
  int nlive;             // classic: live (non-padding) slots among 1..3; padding slots trail after the host's sort
  int dup;               // classic: slots whose packed weights coincide (the components of one path share their weights; a padding slot
                         // may borrow any): 0 none, 1 = slots 1..3 use slot 1's fragments, 2 = slots 0..2 use slot 0's, 3 = all use slot 0's
  int nslot;             // live slots (packed), 4 (classic)
  int nb;                // column blocks of the edge product: packed ceil(nslot / 2) + 1, classic 4
};
// per-edge rows of k_conv_fused (k_vn_rows): [vcap rounded up to 16][32][ES] = harmonics | weight | message row, ES = 8 (l <= 1) or 12
struct VnRowsArgs {
  const int* nvn; const int* vn_node; const int* vn_e0; const int* goff;   // filled by launch_vn_build
  const int* arow; const float* nvec; const float* ew; const int* tslot; float sgn; int sh_lmax;
  const int* tgt; int tbase;           // target node of every edge (first node id of the target range)
  int vcap; float* rows; int* vn_ne;   // rows == nullptr: lists only
  int* tile_hdr = nullptr; unsigned char* live = nullptr;   // k_vn_rows_grouped: also the tile headers / live flags of the in-tile pre-reduction (launch_vn_tiles)
};
// tile_per_pose: pad every graph of the batch to whole 16-node tiles (k_vn_fill_pp); nullptr = dense lists
struct VnPoseTiles { const int* node_batch; const int* graph_ptr; int n_graphs; int* nvn_pad; };
void launch_vn_build(const int* goff, int gcount, int* cnt_tmp, int* voff, int* vn_node, int* vn_e0, const VnRowsArgs& rows,
                     hipStream_t s, const VnPoseTiles* pp = nullptr);
// the same for several edge groups in two launches (k_vn_lists: count -> scan -> fill, workgroup = group; k_vn_rows_grouped)
constexpr int VN_GROUPS_MAX = 9;
struct VnListArgs { const int* goff; int gcount; int* voff; int* vn_node; int* vn_e0;
                    const int* node_batch; const int* graph_ptr; int n_graphs; int* nvn_pad; };   // node_batch == nullptr: dense lists
struct VnListsArgs { int n = 0; VnListArgs g[VN_GROUPS_MAX]; };
struct VnRowsGroupedArgs { int n = 0; int first[VN_GROUPS_MAX + 1] = {}; VnRowsArgs g[VN_GROUPS_MAX]; };
static_assert(sizeof(VnRowsGroupedArgs) <= 2048 && sizeof(VnListsArgs) <= 2048, "kernel arguments");
void launch_vn_build_all(const VnListsArgs& lists, const VnRowsArgs* rows /* [lists.n], nvn / vn_node / vn_e0 / goff filled here */, int sh_lmax, hipStream_t s);
// In-tile pre-reduction of the messages (groups whose 16 virtual nodes of a tile send to the same few targets: lig<-rec, where
// 16 residues address the <= 32 atoms of one ligand).  Per tile a header of FC_TILE_HDR ints: [0] = 1 when the tile's targets
// span <= 32 consecutive target rows (else the tile stores one message row per edge as before), [1] = first target row,
// [2] = span, [4 + j] = message row that receives the tile's sum for target [1] + j (the row of the tile's first edge with that
// target; -1: none).  live[message row] = 1 for every row that will be written (k_reduce_bn skips the others).  l <= 1 rows only.
constexpr int FC_TILE_HDR = 36, FC_TILE_NT = 32;
void launch_vn_tiles(const int* nvn, int vcap, const float* vrows, const int* vn_ne, int* tile_hdr, unsigned char* live, hipStream_t s);
void launch_edge_hidden(const int* nvn, int vcap, const int* vn_node, const int* vn_e0, const int* goff, const int* arow,
                        const int* tgt, int tbase, const float* HE, const float* P, const float* Q, int H, int NG8,
                        float* Hb, hipStream_t s, int bf = 0);   // P == Q == nullptr: HE[arow ? arow[e] : e] is the finished hidden row; bf: packed split-bf16 words
void launch_edge_rows(const int* nvn, int vcap, const int* vn_node, const int* vn_e0, const int* goff, const int* arow,
                      const int* tgt, int tbase, const float* HE, const float* P, const float* Q, int H, float* rows, hipStream_t s);
// Hidden rows of the edge MLP from the edge attributes in one pass (ns % 16 == 0): first Linear split over its inputs,
//   h_e = relu(W1e * edge_attr[arow[e]] + P[tgt[e]] + Q[d] (+ rowbias[ridx])),  P = W1s * x_s[:ns], Q = W1d * x_d[:ns] + b1,
// edge-attribute block on the matrix cores, written straight in the A-fragment order of k_conv_fused (replaces the
// per-edge GEMM + k_edge_hidden).
struct EdgeHiddenArgs {
  const int* nvn; int vcap; const int* vn_node; const int* vn_e0; const int* goff;
  const int* arow; const int* tgt; int tbase;
  const float* ea; int ns;            // [rows][ns] edge attribute rows
  const float* W1; int ldw;           // [H][ldw]: columns [0,ns) multiply the edge attributes
  const float* P;                     // [tcount][H], row tgt[e] - tbase
  const float* Q;                     // [gcount][H]
  const float* rowbias; const int* ridx;   // optional per-graph term [B][H], graph of attr row
  int H, NG8;
  int zero_fill;   // write zero fragments for empty row tiles (the dense-row loop multiplies them; the other loops skip them)
  const float* vrows; const int* vn_ne;   // l <= 1: per-edge rows of k_vn_rows (words 6, 7 = attribute row, target row) and edge counts; nullptr = index chain
  float* Hb;
  int bf;          // hidden values as packed split-bf16 words (ddmi_common.h: bf_split2) for the bf16 edge product of k_conv_fused
  int grid;        // workgroups (ddmi_exec_options.hidden_grid); 0 = 2048
};
void launch_edge_hidden_mm(const EdgeHiddenArgs& a, hipStream_t s);
// the hidden rows of several edge groups in one launch (each group into its own Hb; grouped dispatch of the convolution)
constexpr int EH_GROUPS_MAX = 4;
struct EdgeHiddenGroupedArgs {
  int n = 0;
  int first[EH_GROUPS_MAX + 1] = {};   // first workgroup of every group; first[n] = grid size
  EdgeHiddenArgs g[EH_GROUPS_MAX];
};
void launch_edge_hidden_mm_grouped(const EdgeHiddenGroupedArgs& G, hipStream_t s);

struct FusedConvArgs {
  const int* nvn; int vcap;              // live virtual nodes (device) and their capacity (grid size)
  const int* vn_node;                    // [vcap] gather node (local)
  const float* vrows; const int* vn_ne;  // per-edge rows [vcap16][32][ES] and edge counts of the virtual nodes (k_vn_rows)
  const float* X; int gbase;             // node table (stride XS), first gather node
  const float* wpack; int KS, HK;        // packed second layer [HK][KS]
  const float* Hb; int NG8;              // hidden rows in A-fragment order [vcap][2][NG8][64][2], NG8 = ceil(H / 8)
  int sh_lmax;
  const FGran* gran; int ysplit; int gsplit[9];   // blockIdx.y walks granules [gsplit[y], gsplit[y+1])
  const float* cgt;                      // dense coupling rows [granule][FC_MAXSLOT][MAXD][SHD] (host-built, weights.cpp)
  int max_nb;                            // widest granule of the layer, in column blocks (4 classic, 5 = a packed 7-slot granule)
  int maxd;
  int generic;                           // some granule has no static chain shape: predicated kernel variant
  int dense;                             // most virtual nodes hold > 16 edges: multiply both row tiles unconditionally
  int shared;                            // gather nodes with several virtual nodes each: contract the distinct nodes of a tile (4x4x1 MFMA)
  int n_units; short ustart[48];         // first granule of every (output block, w tile) unit: workgroups rotate their visiting order by units
  short ufirst[8], ucount[8];            // units of granule range y: ustart[ufirst[y] .. + ucount[y])
  float* msg;                            // [E][XS]
  const int* tile_hdr = nullptr;         // in-tile pre-reduction headers (launch_vn_tiles); nullptr: one message row per edge
  int bf = 0;                            // edge product on v_mfma_f32_16x16x32_bf16 with split operands (Hb holds packed words); static l <= 1 loops only
  int dbg = 0;
  int prof_slot = 0;                     // profiling builds: edge-group slot of the in-kernel phase clocks
};
#if defined(DDMI_PROFILING)
void fc_prof_report();                   // prints and clears the phase clocks of k_conv_fused (stderr)
#endif
void launch_conv_fused(const FusedConvArgs& a, hipStream_t s);

// Grouped dispatch (round 6; k_conv_grp.hip): the work items (tile, granule range) of SEVERAL edge groups of one interaction layer
// in one grid -- the groups are independent until the joint mean (models/tensor_layers.py:148-231).  Block b belongs to group g
// with first[g] <= b < first[g + 1]; inside the group, block = first[g] + range * ntile[g] + tile (all tiles of range 0, then
// range 1, ... -- the order the 2-D grid of k_conv_fused dispatches, measured better than range-adjacent, profiles/r05_e17_ab.txt).
// Exact-f32 l <= 1 layers with static chain shapes only (mode 0 sparse rows, 3 dense rows, 4 shared-node tiles).
constexpr int FC_GROUPS_MAX = 4;
struct FusedGroupedArgs {
  int n = 0;
  int first[FC_GROUPS_MAX + 1] = {};
  int ntile[FC_GROUPS_MAX] = {};
  int mode[FC_GROUPS_MAX] = {};
  FusedConvArgs g[FC_GROUPS_MAX];
};
static_assert(sizeof(FusedGroupedArgs) <= 3072, "kernel arguments of k_conv_grouped (4 KB limit)");
void launch_conv_grouped(const FusedGroupedArgs& G, hipStream_t s);

struct ReduceGroup { const int* toff; const float* msg; int tbase, tcount; const unsigned char* live = nullptr; };   // live: rows to read (nullptr: all)
// X_out[s] = BN(mean over all groups' incoming messages) + pad(X_in[s]) for s in [nbase, nbase+ncount)
void launch_reduce_bn(const ReduceGroup* groups_dev, int n_groups, int nbase, int ncount, int D_in, int D_out,
                      const float* bn_mean, const float* bn_scale, const float* bn_bias, int residual,
                      const float* X_in, float* X_out, int out_stride, hipStream_t s);

// Fused node update (k_node.hip): launch_reduce_bn for the rows [nbase, nbase + ncount) of a layer AND, from the finished rows,
// the per-node terms of the NEXT layer's first Linear: term t writes out_t[s - base_t][H] = W_t[H][ldw-strided rows] . X_out[s][:ns]
// (+ bias_t) for the nodes s in [base_t, base_t + count_t) -- the P / Q rows k_edge_hidden_mm adds per edge.
constexpr int NU_TERMS_MAX = 8;
struct NodeTerm { const float* W; const float* bias; float* out; int base, count; };
struct NodeUpdateArgs {
  const ReduceGroup* groups; int n_groups, nbase, ncount, D_in, D_out;
  const float *bn_mean, *bn_scale, *bn_bias; int residual;
  const float* X_in; float* X_out;
  int n_terms; NodeTerm term[NU_TERMS_MAX];
  int ns, H, ldw;
  int wpn;   // waves per node: 0 = by node count, 1 = sixteen nodes per workgroup, 4 = four nodes per workgroup (k_reduce_bn's row deal)
};
void launch_node_update(const NodeUpdateArgs& a, hipStream_t s);

// ---------------------------------------------------------------- k_graph.hip
void launch_exclusive_scan(const int* in, int* out, int n, hipStream_t s);  // out[0..n], out[n] = total
void launch_exclusive_scan2(const int* in0, int* out0, int n0, const int* in1, int* out1, int n1, hipStream_t s);   // two scans, one launch
void launch_lig_radius(const float* pos, const int* batch, const int* ptr, int nL, int maxNl, float r, int cap,
                       int* adjrank, int* cnt_g, hipStream_t s);
void launch_ll_count(const int* adjrank, const int* batch, const int* ptr, int nL, int maxNl, const int* bg,
                     const int* bt, int* cnt_g_inout, int* cnt_t, hipStream_t s);
void launch_ll_fill(const float* pos, const int* batch, const int* ptr, int nL, int maxNl, const int* adjrank,
                    const int* goff, const int* toff, const int* bg, const int* bt, int n_bonds, const int* bond_src,
                    const int* bond_dst, const int* bond_grank, const int* bond_trank, float smooth_max,
                    int* tgt, int* tslot, int* featidx, int* ebatch, float* dist, float* nvec, float* ew, hipStream_t s);
void launch_cross_count(const float* lpos, const float* rpos, const int* lbatch, const int* rbatch, const int* lptr,
                        const int* rptr, int nL, int nR, int maxNr, const float* cutoff, float const_cutoff,
                        const int* keep, int* pairrank, int* cnt_l, int* cnt_r, hipStream_t s);
void launch_crop_mask(const float* lpos, const float* rpos, const int* rbatch, const int* lptr, int nR, float cut2, int* keep,
                      hipStream_t s);
void launch_rr_filter(const int* keep, const int* goff, const int* tgt, const int* arow, const int* toff, const int* tlist,
                      const int* gnode, int nL, int nR, int* cnt_g, int* cnt_t, int* goff2, int* toff2, int* tslot_tmp,
                      int* tgt2, int* tslot2, int* arow2, hipStream_t s);
void launch_cross_fill(const float* lpos, const float* rpos, const int* rbatch, const int* lptr, const int* rptr, int nL,
                       int nR, int maxNr, const int* pairrank, const int* offs_l, const int* offs_r, const float* cutoff,
                       float const_cutoff, int smooth, int* g1_tgt, int* g1_tslot, int* g3_tgt, int* g3_tslot,
                       int* pbatch, float* pdist, float* pnvec, float* pew, hipStream_t s,
                       int rbase = -1 /* first global id of the second node type; default nL */);
void launch_cross_cutoff(const float* t_tr, int B, float smin, float smax, float* out, hipStream_t s, int raw = 0);
void launch_tor_radius(const float* pos, const int* ptr, const int* tor_u, const int* tor_v, const int* tor_batch, int nT,
                       float r, int cap, float smooth_max, int* cnt, int* atom, float* dist, float* nvec, float* ew,
                       float* bond_nvec, hipStream_t s);

// ---------------------------------------------------------------- k_embed.hip
void launch_time_embedding(const float* t, int B, const float* freq, int half, float scale, int fourier, float* out, hipStream_t s);
constexpr int TIME_TERMS_MAX = 8;
struct TimeTermsArgs {
  const float* t; int B; const float* freq; int half; float scale; int fourier; float* temb;   // launch_time_embedding's arguments
  int ns, n;                                                                                   // n linear terms temb[sd] -> [ns]
  struct { const float* W; int ldw; const float* bias; float* C; int act; } term[TIME_TERMS_MAX];
  int hid_term; const float* W3; const float* b3; float* out3;   // out3 = W3 . term[hid_term] + b3 (second layer of rec_sigma); W3 == nullptr: none
};
void launch_time_terms(const TimeTermsArgs& a, hipStream_t s);
void launch_lig_node_embed(const int* x, int nL, const float* emb, const int* emb_off, int n_feat, int ns, float* out,
                           hipStream_t s);
void launch_add_rowvec(float* X, int ldx, const float* base, int ldb, const float* vec, int ldv, const int* idx, int rows,
                       int cols, int zero_to, hipStream_t s);
struct EdgeMlpArgs {
  int E = 0; const int* e_dev = nullptr;          // rows (capacity) and optional live count
  const float* dist = nullptr;                    // [E]
  const float* offsets = nullptr; int D = 0; float coeff = 0;  // GaussianSmearing
  const float* feat = nullptr; const int* featidx = nullptr; int nfeat = 0;  // optional extra features (row featidx[e], -1 = zeros)
  const float* W0f = nullptr; int ldw0f = 0;      // [ns][nfeat]
  const float* W0g = nullptr; int ldw0g = 0;      // [ns][D]
  const float* gvec = nullptr; const int* gidx = nullptr;  // [G][ns] per-graph hidden term incl. bias (gidx null -> row 0)
  const float* W1 = nullptr; const float* b1 = nullptr;    // [ns][ns], [ns]
  int ns = 0;
  float* out = nullptr; int ldo = 0;
};
void launch_edge_mlp(const EdgeMlpArgs& a, hipStream_t s);
// vec = pos_dst[dst] - pos[src] (pos_dst = nullptr: same array): static receptor / atom relations
void launch_rec_edge_geom(const float* pos, const int* src, const int* dst, int E, float smooth_max, float* dist,
                          float* nvec, float* ew, hipStream_t s, const float* pos_dst = nullptr);
void launch_add3(float* out, const float* x, int d_in, const float* u1, const float* u2, int rows, int d_out, hipStream_t s);
void launch_concat_rec_input(const float* rec_x, int ldx, const float* emb, int ns, int lm, int nR, float* out,
                             hipStream_t s);

// ---------------------------------------------------------------- k_readout.hip
// read-out prep kernels (round 6): everything of a read-out chain that does not depend on its edge MLP, in one launch
void launch_center_prep(const float* pos, const int* batch, const int* ptr, int nL, int lmax, const float* X, const int* xrow, int ns,
                        float* dist, float* nvec, float* sh, int lds_, float* attr, int lda, hipStream_t s);
void launch_tor_prep(const float* edge_nvec, const float* bond_nvec, int nT, int cap, int lmax, const float* T, int ds, int dts, float* out,
                     const float* X, const int* atom, const int* eu, const int* ev, int ns, float* attr, int lda, hipStream_t s);
void launch_center_edges(const float* pos, const int* batch, const int* ptr, int B, int nL, float* dist, float* nvec,
                         hipStream_t s);
void launch_sh_rows(const float* nvec, float sgn, int E, int lmax, float* sh, int lds, hipStream_t s);
void launch_tor_sh(const float* edge_nvec, const float* bond_nvec, int nT, int cap, int lmax, const float* T, int ds,
                   int dts, float* out, hipStream_t s);
void launch_gather_cols(float* dst, int ldd, int col0, const float* src, int lds_, const int* rowidx, int rows, int cols,
                        const int* rowidx2, hipStream_t s);
struct TpApplyArgs {
  int E; const int* valid_cnt; int cap;   // if valid_cnt: edge e = seg*cap + r is live iff r < valid_cnt[seg]
  const float* Wt; int ldw;               // [E][weight_numel]
  const float* X; const int* xrow;        // gather rows of the node table (stride XS)
  const float* sh; int lds_;              // [E][sh_dim]
  const float* ew;                        // optional per-edge weight
  const DevPath* paths; const float* ctab; const CgItem* items; int n_items;
  int n_paths, z_floats;                  // paths of the layer; sum over them of mul_in * dout (the per-edge scratch of k_tp_apply_edge)
  int form = -1;                          // -1: by launch size; 0 wave per (edge, item), 1 workgroup per edge, 2 thread per (edge, item)
  float* out; int ldo;                    // [E][out_dim]
};
void launch_tp_apply(const TpApplyArgs& a, hipStream_t s);
void launch_segment_mean_bn(const float* rows, int ldr, const int* seg_off, const int* seg_cnt, int cap, int n_seg,
                            int D, const float* bn_mean, const float* bn_scale, const float* bn_bias, float* out,
                            int ldo, hipStream_t s);
struct ScoreHeadArgs {
  int B; const float* gp;                       // [B][12] (or [B][6] odd_parity)
  int odd_parity, scale_by_sigma, ns, ldw0;     // ldw0 = 1 + sigma_embed_dim (row stride of the first Linear)
  const float *tr_w0n, *tr_sig, *tr_w3, *tr_b3;  // w0n [ns] (norm column), sig [B][ns] (incl bias)
  const float *rot_w0n, *rot_sig, *rot_w3, *rot_b3;
  const float *t_tr, *t_rot;
  float tr_smin, tr_smax, rot_smin, rot_smax;
  const float* so3_table; int so3_n;
  float *tr_out, *rot_out;
};
void launch_score_heads(const ScoreHeadArgs& a, hipStream_t s);
// confidence = confidence_predictor(scatter_mean(cat[x[:, :ns], x[:, tail_off : tail_off + n_tail]], batch)) -- cg_model.py:353-366;
// BatchNorm1d folded to scale / shift.  One workgroup per graph.
struct ConfHeadArgs {
  int B; const float* X; const int* lig_ptr;   // ligand rows: the last node table (ldx = XS) or the atom predictor's output
  int ldx, col0;                               // row stride, first column of the leading ns-block
  int ns, tail_off, n_tail;                    // n_tail = 0: scalars only (fewer than 3 layers)
  const float *W0, *b0, *sc0, *sh0, *W1, *b1, *sc1, *sh1, *W2, *b2;
  int n_out; float* out;                       // [B][n_out]
};
void launch_conf_head(const ConfHeadArgs& a, hipStream_t s);
struct TorHeadArgs {
  int nT, ns, in_dim; const float* feat;  // [nT][in_dim] after BN
  const float* W0;                        // [ns][in_dim]
  const float* W3;                        // [ns]
  const int* tor_batch; const float* t_tor; float smin, smax; int scale_by_sigma;
  const float* torus_table; int torus_n;
  float* out;
};
void launch_tor_head(const TorHeadArgs& a, hipStream_t s);

// ---------------------------------------------------------------- k_sample.hip
struct PerturbArgs {
  int B, R; float *tr, *rot, *tor;  // scores in / perturbations out (in place)
  const float *z_tr, *z_rot, *z_tor;  // may be nullptr (zero noise)
  float c_tr_s, c_tr_z, c_rot_s, c_rot_z, c_tor_s, c_tor_z;
  int use_rng; unsigned long long seed; const long long* sample_ids; int step;
};
void launch_perturb(const PerturbArgs& a, hipStream_t s);
// test entry points of the noise generator (ddmi_debug_philox / ddmi_debug_normal)
void launch_debug_philox(const unsigned* ctr, const unsigned* key, int n, unsigned* out, hipStream_t s);
void launch_debug_normal(unsigned long long seed, long long sample0, int n_samples, int step, int n_comp, float* out, hipStream_t s);
// t[0..B) = t_tr, t[B..2B) = t_rot, t[2B..3B) = t_tor: the per-graph time vectors of one step (set_time, utils/diffusion_utils.py:35-57)
constexpr int STEP_TIMES_MAX = 64;
struct StepTimes { int steps; float t[3 * STEP_TIMES_MAX]; };   // (t_tr, t_rot, t_tor) of every step
void launch_fill_times_all(float* t /* [steps][3][B] */, int B, const StepTimes& st, hipStream_t s);
void launch_fill_times(float* t, int B, float t_tr, float t_rot, float t_tor, hipStream_t s);
void launch_modify_conformer(float* pos, int B, int Nl, int R, const int* rot_u, const int* rot_v,
                             const unsigned char* mask_rotate, const float* tr, const float* rot, const float* tor,
                             hipStream_t s);

}  // namespace ddmi
