// Host-side state of a ddmi_model: configuration, weights (host copy keyed like the reference
// state_dict + packed device forms), the static description of the current batch of complexes
// and the device workspace.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "kernels.h"
#include "o3_host.h"

namespace ddmi {

struct HostTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
  int64_t numel() const { int64_t n = 1; for (auto d : shape) n *= d; return n; }
};

// Bump allocator over one hipMalloc'ed slab (freed with the owner).
class DevicePool {
 public:
  ~DevicePool();
  void* alloc_bytes(size_t n);
  template <class T> T* alloc(size_t n) { return reinterpret_cast<T*>(alloc_bytes(n * sizeof(T))); }
  template <class T> T* upload(const std::vector<T>& v) {
    T* p = alloc<T>(v.size() ? v.size() : 1);
    if (!v.empty()) DDMI_CHECK_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return p;
  }
  void release();
  size_t total_bytes() const { return total_; }
 private:
  std::vector<void*> blocks_;
  size_t total_ = 0;
};

struct ConvW {  // one TensorProductConvLayer
  std::string name;
  int G = 1;
  bool faster = false, residual = true, has_bn = true, yform = true;
  Irreps in_irr, sh_irr, out_irr;
  TPTable table;
  int n_edge = 0, H = 0, HK = 0, D_in = 0, D_out = 0, NT = 0, sh_dim = 0, Wn = 0;
  bool depthwise = false; int Wn_dw = 0, n_lin2 = 0;   // depthwise layer: state-dict sizes of the 'uvu' weights and of linear_2 (folded at commit)
  std::vector<float*> W1, b1, W2, b2, wpack;
  int TL = 2;                                                // Linear layers of the per-edge weight MLP (tp_weights_layers)
  std::vector<std::vector<float*>> Wmid, bmid;               // [group][TL - 2] hidden Linear layers H -> H (TL > 2)
  std::vector<float*> W1p, b1p;   // first layer with the hidden units of every block of 16 in the order k_edge_hidden_mm emits them
  int KS = 0;                                                // k-slab size of wpack
  FGran* fgran = nullptr; int n_fgran = 0, HKq = 0; bool fgran_generic = false;          // fused form: granule list, padded hidden-row length
  float* cgt = nullptr; int max_nb = 4;                      // dense coupling rows per granule; widest granule in column blocks
  std::vector<int> fgran_unit;                               // unit id of every granule (split points of the grid)
  int maxd = 1;
  DevPath* paths = nullptr; float* ctab = nullptr; CgItem* items = nullptr; int n_items = 0;
  float *bn_mean = nullptr, *bn_scale = nullptr, *bn_bias = nullptr;
};

struct Mlp2W { float *W0 = nullptr, *b0 = nullptr, *W3 = nullptr, *b3 = nullptr; int in = 0, hid = 0, out = 0; };

struct DebugEntry { const void* ptr; std::vector<int64_t> shape; bool is_int; };

struct Model {
  ddmi_config cfg{};
  int device = 0;
  // ---- weights
  std::vector<std::pair<std::string, std::vector<int64_t>>> spec;  // expected state_dict keys, in order
  std::map<std::string, HostTensor> host_w;
  bool committed = false;
  DevicePool wpool, tpool;
  int ns = 0, sd = 0, D = 0, Dc = 0, nf = 0, lm = 0, H = 0;
  float* lig_emb = nullptr; int* lig_emb_off = nullptr;
  Mlp2W lig_enc;     // additional_features_embedder as (W0 = [ns][ns+sd], b0)
  Mlp2W lig_edge, rec_edge, rec_sigma, cross_edge, center_edge, final_edge, tr_final, rot_final;
  Mlp2W atom_edge, ar_edge, la_edge;            // all_atoms (cross_edge then holds lr_edge_embedding)
  float* atom_emb = nullptr; int* atom_emb_off = nullptr;
  float *conf_W[3] = {}, *conf_b[3] = {}, *conf_bn_scale[2] = {}, *conf_bn_shift[2] = {};   // confidence_predictor
  float *aconf_W[3] = {}, *aconf_b[3] = {};   // atom_confidence_predictor, BatchNorm1d folded into the two hidden Linears
  float* rec_emb = nullptr; float *rec_enc_W = nullptr, *rec_enc_b = nullptr;
  float *off_lig = nullptr, *off_rec = nullptr, *off_cross = nullptr, *off_center = nullptr;
  float coeff_lig = 0, coeff_rec = 0, coeff_cross = 0, coeff_center = 0;
  std::vector<ConvW> rec_emb_layers, lig_emb_layers, conv_layers;
  ConvW final_conv, tor_conv;
  std::vector<ConvW> old_lig, old_rec, old_l2r, old_r2l;   // legacy class: four separate layers per interaction layer
  Mlp2W old_lig_lin, old_rec_lin;                          // OldAtomEncoder.linear (W0/b0 only)
  float *old_lm_W = nullptr, *old_lm_b = nullptr;          // OldAtomEncoder.lm_embedding_layer [ns][1280 + ns]
  float *tor_W0 = nullptr, *tor_W3 = nullptr;
  float* side_Mt = nullptr; int side_K = 0;   // sidechain_predictor as one dense [10][side_K] matrix over a node row (weights.cpp)
  float* tor_T = nullptr; int tor_ds = 0, tor_dts = 0;  // FullTensorProduct(sh, 2e) dense table
  float* time_freq = nullptr;
  std::vector<float> time_freq_host;
  float *so3_table = nullptr, *torus_table = nullptr; int so3_n = 0, torus_n = 0;
  // ---- complex + workspace
  bool has_complex = false;
  hipStream_t side_stream = nullptr;   // ligand-gather edge groups run here, concurrently with the receptor-gather ones
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_cross = nullptr, ev_terms = nullptr;
  bool two_streams = true;
  int layer_overlap = 0;               // 0 joined layers (run_conv), 1 overlapped layer boundaries for chip-filling batches, 2 for every batch
  std::vector<hipEvent_t> ev_pipe;     // run_conv_layers_overlapped: [layer][group launch done / node rows written]
  int fused_shared = 1;     // 1: rec<-lig group contracts the distinct gather nodes of a tile on the 4x4x1 MFMA; 0: per virtual node; 2: every dense group (tests)
  bool fused_tri = true;    // the three light granules of a single-chain 48-channel scalar block as one (exec.merged_granule = 1: separate)
  bool fused_pack = true;   // packed granules for output blocks of <= 10 channels (exec.packed_granules = 1: classic granules only)
  int tp_form = -1;         // read-out tensor product: -1 by launch size; exec.tp_apply forces one form (tests)
  bool fc1_batch = true;    // per-node / per-graph terms of the first Linear of all groups of a layer in one launch (exec.fc1_batch = 1: per group)
  bool fused_mm = true;     // hidden rows straight from the edge attributes (k_edge_hidden_mm); exec.hidden_mm = 1: GEMMs + k_edge_hidden
  int fused_dense = 1;      // branch-free dense-row main loop: 0 never, 1 groups with >= 20 edges per gather node, 2 always
  bool fused_prered = true; // in-tile pre-reduction of the lig<-rec messages (exec.pre_reduce = 1: one message row per edge)
  int fused_ysplit = 0;     // workgroups per 16-virtual-node tile (granule ranges); 0 = spread launches with few tiles over the CUs
  int fused_ysplit_small = 0;   // the same for a small group next to chip-filling ones (ddmi_exec_options.tile_split_small); 0 = automatic
  bool tight_caps = false;      // exec.list_caps = 1: virtual-node list capacities from per-node degree bounds (default: nodes + edges / 32)
  bool time_terms_fused = false; // exec.time_terms = 1: time embedding + its per-graph linear terms + rec_sigma's second layer in one launch (k_time_terms)
  int group_order = 0;          // issue order of a layer's groups on their streams (exec.group_order bits: 1 = side stream reversed, 2 = main stream reversed)
  bool ys_rounds_small = false; // the round model also for the groups of small layers (exec.tile_split_rule = 2; A/B)
  bool ys_rounds = true;        // chip-filling groups: granule-range split from the round model (exec.tile_split_rule = 1: one item per tile, rounds 2-5)
  int n_cus = 256;              // compute units of the device (hipDeviceProp_t::multiProcessorCount)
  int fused_ysplit_last = 0;    // the same for the last chip-filling launch of each stream in a layer (ddmi_exec_options.tile_split_last); 0 = as the others
  int eh_grid = 2048;       // workgroups of k_edge_hidden_mm (ddmi_exec_options.hidden_grid)
  int grouped = 0;          // grouped dispatch of a layer's edge groups (ddmi_exec_options.grouped): 0 / 1 = per-group launches on two streams (default), 2 = grouped wherever supported
  int grouped_split = 0;    // workgroups per tile in grouped launches (exec.grouped_split); 0 = grouped_target / tiles of the layer
  int grouped_target = 1536;
  int node_update_wpn = 0;  // k_node_update: waves per node, 0 = by node count (exec.node_update = 2 / 3 force sixteen / four nodes per workgroup)
  bool node_update = false; // exec.node_update = 1: k_node_update -- a layer's node rows and the next layer's per-node first-Linear terms in one kernel (default: k_reduce_bn + GEMM launches)
  bool vn_merge = true;     // virtual-node lists of a layer's groups in two launches (k_vn_lists, k_vn_rows_grouped); exec.vn_build = 1: one chain per group
  bool tile_per_pose = false;   // tiles of 16 virtual nodes never span two graphs (ddmi_exec_options.tile_per_pose): bit-exact shard invariance
  double crop_cutoff = 0.0;  // > 0: receptor cropped to this distance from the ligand in ddmi_forward (crop_beyond)
  DevicePool cpool;
  struct Cx;  // defined in complex.cpp
  std::shared_ptr<Cx> cx;
  std::map<std::string, DebugEntry> debug;
  // ---- kernel timing
  bool timing = false;
  int timing_level = 0;     // ddmi_set_kernel_timing: 1 = one row per kernel name, 2 = k_conv_fused per edge group, 3 = per (layer, edge group)
  struct TimedPhase { std::string name; double ms = 0; int64_t launches = 0; };
  std::vector<TimedPhase> phases;
  struct EventPair { int phase; hipEvent_t a, b; };
  std::vector<EventPair> pending;     // recorded on the launch stream, resolved by resolve_timings()
  std::vector<hipEvent_t> free_events;
};

// RAII bracket: HIP events on the stream the kernels are launched on (only when timing is enabled)
struct PhaseTimer {
  Model& m; hipStream_t s; int idx = -1; hipEvent_t a{}, b{};
  PhaseTimer(Model& model, const char* name, hipStream_t stream);
  ~PhaseTimer();
};
void resolve_timings(Model& m);

// weights.cpp
void build_weight_spec(Model& m);
void commit_weights(Model& m);
// complex.cpp
void set_complex(Model& m, const ddmi_complex& c, hipStream_t s);
// score mode: tr_out / rot_out / tor_out; confidence mode (cfg.confidence_mode): conf_out [B, num_confidence_outputs] only
void forward(Model& m, const float* lig_pos, const float* t_tr, const float* t_rot, const float* t_tor, float* tr_out,
             float* rot_out, float* tor_out, hipStream_t s, float* conf_out = nullptr, float* atom_conf_out = nullptr);
void modify_conformer(Model& m, float* lig_pos, const float* tr, const float* rot, const float* tor, hipStream_t s);
void sidechain_pred(Model& m, float* out, hipStream_t s);   // model(batch)[3] of the last forward (models/cg_model.py:397-402)
void sample(Model& m, float* lig_pos, const ddmi_sample_cfg& sc, hipStream_t s);
// NaN guard + score / noise combination of step k (utils/sampling.py:117-186) on score arrays, in place
void perturb(Model& m, float* tr, float* rot, float* tor, const ddmi_sample_cfg& sc, int k, hipStream_t s);

}  // namespace ddmi
