/* ddmi.h -- C ABI of the MI355X-native DiffDock score-model sampling path (libddmi.so).
 *
 * The reference (gcorso/DiffDock @ 2024_10_08) is 100 % Python and has no FFI; its
 * boundary for this path is the Python duck type
 *     model = get_model(args, device, t_to_sigma, no_parallel=True)   utils/utils.py:172
 *     tr, rot, tor = model(batch)[:3]                                 utils/sampling.py:116
 *     pos = modify_conformer_batch(pos, batch, tr, rot, tor, mask)    utils/sampling.py:189
 *     data_list, conf = sampling(data_list, model, steps, ...)        utils/sampling.py:69
 * Each entry point below names the reference interface it replaces.  The ctypes binding a
 * maintainer would add is shown in INTEGRATION.md (diffdock_amd/lib.py is that binding).
 *
 * Conventions: all tensor pointers are DEVICE pointers (fp32 / int32 / uint8) unless a
 * parameter is documented as host; they are borrowed for the duration of the call
 * (weights, tables and the static complex description are copied).  Work is enqueued on the
 * caller's HIP stream (pass torch.cuda.current_stream().cuda_stream; NULL = default
 * stream).  Every function returns 0 on success or a negative ddmi_status; the message is
 * available from ddmi_last_error() (thread-local).  A model handle is bound to one device
 * and must not be used from two threads at once; distinct handles are independent.
 */
#ifndef DDMI_H
#define DDMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ddmi_model ddmi_model;
typedef void* ddmi_stream; /* hipStream_t */

enum ddmi_status {
  DDMI_OK = 0,
  DDMI_ERR_ARG = -1,     /* bad argument / inconsistent shapes            */
  DDMI_ERR_STATE = -2,   /* call order (weights / tables / complex unset) */
  DDMI_ERR_HIP = -3,     /* HIP runtime error                             */
  DDMI_ERR_KEY = -4,     /* unknown / missing state_dict key              */
  DDMI_ERR_CAPACITY = -5 /* workspace or neighbour capacity exceeded      */
};

/* Execution options: which of the parity-tested kernel routes a model runs.  Every field 0 = the default a caller should
 * keep; the other values exist for the route-agreement tests (tests/test_gpu_parity.py::test_selectable_kernel_paths_agree_on_the_gpu)
 * and for A/B timing.  Read once at ddmi_create; libddmi.so itself reads NO environment variable -- diffdock_amd/lib.py maps
 * the DDMI_* variables of the test / bench harness onto these fields (INTEGRATION.md has the table). */
typedef struct ddmi_exec_options {
  int32_t streams;          /* 0 = two HIP streams (ligand-gather groups next to residue-gather groups), 1 = one stream            */
  int32_t dense_rows;       /* k_conv_fused dense-row loop: 0 = groups with >= 20 edges per gather node, 1 = never, 2 = always       */
  int32_t shared_tiles;     /* shared-node tiles (4x4x1 contraction): 0 = the rec<-lig group, 1 = never, 2 = every dense group       */
  int32_t packed_granules;  /* 0 = packed granules for output blocks of <= 10 channels, 1 = classic 4-slot granules only             */
  int32_t merged_granule;   /* 0 = the three scalar channel tiles of the first layer as one granule, 1 = separate                    */
  int32_t pre_reduce;       /* 0 = in-tile pre-reduction of the lig<-rec messages, 1 = one message row per edge                      */
  int32_t hidden_mm;        /* 0 = hidden rows straight from the edge attributes (k_edge_hidden_mm), 1 = GEMMs + k_edge_hidden       */
  int32_t fc1_batch;        /* 0 = per-node terms of the first Linear of all groups of a SMALL layer in one launch, 1 = per group    */
  int32_t tile_split;       /* workgroups per 16-virtual-node tile (granule ranges); 0 = automatic                                   */
  int32_t tile_split_small; /* the same for a small group that runs next to chip-filling ones; 0 = automatic                         */
  int32_t hidden_grid;      /* workgroups of k_edge_hidden_mm; 0 = 2048                                                              */
  int32_t tp_apply;         /* read-out tensor product: 0 = by launch size, 1 = wave per pair, 2 = workgroup per edge, 3 = thread    */
  int32_t debug;            /* 1 = print the granule list of every interaction layer to stderr at ddmi_commit_weights (tests)         */
  int32_t tile_per_pose;    /* 1 = the 16-virtual-node tiles of k_conv_fused never span two graphs of the batch (dead virtual nodes
                             * pad every graph to whole tiles): the arithmetic of a pose then does not depend on the poses batched
                             * with it -- a sharded run is BIT-identical to the one-batch run.  0 = dense tiles.     */
  int32_t layer_overlap;    /* interaction layers on the two streams: 0 = joined -- all groups of a layer, then one node update; 1 = node
                             * update split by node type and every group's chain started as soon as the rows it reads exist, for
                             * chip-filling batches; 2 = that for every batch size.  Same kernels and arithmetic (bit-identical scores);
                             * measured neutral at 40 poses and -6 % at 5 (profiles/r05_e11_ab.txt), hence not the default.            */
  int32_t grouped;          /* grouped dispatch of an interaction layer (round 6): per layer ONE launch of the per-node first-Linear terms,
                             * ONE of the hidden rows and ONE k_conv_grouped walking the (edge group, tile, granule range) work items of
                             * every edge group, on one stream -- instead of a hidden-row + k_conv_fused launch per group on two streams.
                             * 0 / 1 = per-group launches (the default: measured 1-2.4 % faster at 5 / 10 / 40 poses because hidden rows overlap
                             * the other stream's convolution, profiles/r06_p2_*), 2 = grouped wherever supported (exact-f32 l <= 1
                             * layers with a two-layer edge MLP; others keep the per-group path).  Bit-identical messages either way.   */
  int32_t grouped_split;    /* workgroups per 16-virtual-node tile in grouped launches (granule ranges), 1..8; 0 = automatic           */
  int32_t vn_build;         /* virtual-node lists + per-edge rows of a layer's edge groups: 0 = two launches for all groups
                             * (k_vn_lists: count -> scan -> fill with a workgroup per group; k_vn_rows_grouped), 1 = a
                             * count -> scan -> fill -> rows chain per group (rounds 2-5)                                              */
  int32_t node_update;      /* 1 (2 / 3: workgroup shape forced to sixteen / four nodes) = k_node_update: the node update of an interaction layer (mean over all groups' messages, BatchNorm,
                             * residual) also produces the NEXT layer's per-node first-Linear terms P / Q from the rows it holds --
                             * no k_gemm_nt_batch launch per layer; 0 = k_reduce_bn + GEMM launches (the default: the fused kernel's
                             * 16-node workgroups stream the message rows at 0.57 instead of 0.40 ms per forward and the GEMMs it
                             * removes ran next to the convolution anyway -- 154.0 against 155.3 poses/s at 40 poses, 107.4 / 108.4
                             * at 5, profiles/r06_p5_*).  CG models with two-layer edge MLPs; node rows bit-identical, P / Q equal up
                             * to the order of an ns-term fp32 sum.                                                                      */
  int32_t tile_split_last;  /* workgroups per tile for the LAST chip-filling k_conv_fused launch of each stream in a layer (its final partial
                             * round of workgroups is the layer's straggler tail); 0 = as tile_split                                   */
  int32_t tile_split_rule;  /* automatic tile_split of a chip-filling group: 0 = cheapest schedule of ceil(tiles x split / CUs) rounds of
                             * (granules per item + prologue) (round 6), 1 = one work item per tile (rounds 2-5)                        */
  int32_t group_order;      /* issue order of a layer's edge groups on their streams: 0 = [lig-lig, rec<-lig] | [lig<-rec, rec-rec];
                             * bit 0 = side stream reversed, bit 1 = main stream reversed (A/B knob)                                   */
  int32_t list_caps;        /* capacity of the virtual-node lists (= grid size of k_conv_fused): 0 = nodes + edges / 32 (up to 2 x the live
                             * count), 1 = per-node degree bounds (one virtual node per residue of an all-pairs cross graph: half the
                             * workgroups of the lig<-rec and rec-rec launches never exist) -- neutral at 5-20 poses, 1.8 % SLOWER at 40
                             * (153.1 against 155.9 poses/s: the empty workgroups pace the dispatch between the two streams,
                             * profiles/r06_p12_*), hence not the default                                                               */
  int32_t time_terms;       /* 1 = k_time_terms: time embedding, its per-graph linear terms and rec_sigma's second layer in one launch
                             * (measured neutral once the cross-graph search runs beside them: profiles/r06_p18_*), 0 = k_time_embedding
                             * -> k_gemm_nt_batch -> k_gemm_nt (the default)                                                           */
} ddmi_exec_options;

/* Hyper-parameters: the keyword arguments get_model passes to CGModel
 * (utils/utils.py:234-276, models/cg_model.py:20-31) and the sigma bounds t_to_sigma reads
 * (utils/diffusion_utils.py:28-32).  Booleans are 0/1.
 * struct_size MUST be sizeof(ddmi_config) of the header the caller was compiled against: the struct is passed by pointer and
 * grows with the library, so ddmi_create rejects a caller built against another layout instead of reading past its struct. */
typedef struct ddmi_config {
  uint32_t struct_size;
  int32_t ns, nv, num_conv_layers, num_prot_emb_layers, sh_lmax;
  int32_t sigma_embed_dim, distance_embed_dim, cross_distance_embed_dim, in_lig_edge_features;
  int32_t lm_embedding_dim; /* 1280 for 'precomputed' ESM2 features, else 0 */
  float lig_max_radius, rec_max_radius, cross_max_distance, center_max_distance;
  int32_t dynamic_max_cross, use_second_order_repr, reduce_pseudoscalars, differentiate_convolutions;
  int32_t embed_also_ligand, batch_norm, smooth_edges, odd_parity, no_torsion, scale_by_sigma;
  int32_t fixed_center_conv;
  float embedding_scale;
  float tr_sigma_min, tr_sigma_max, rot_sigma_min, rot_sigma_max, tor_sigma_min, tor_sigma_max;
  int32_t all_atoms; /* get_model's model_class switch (utils/utils.py:221-224): 1 = AAModel (models/aa_model.py) */
  /* get_model(..., confidence_mode=True): same interaction layers, confidence_predictor read-out (cg_model.py:181-207,
   * 353-366); the model is evaluated with ddmi_confidence instead of ddmi_forward */
  int32_t confidence_mode, num_confidence_outputs;
  /* get_model(..., old=True) (utils/utils.py:180-219): legacy class models/old_cg_model.py (the released DiffDock-L
   * confidence checkpoint, `old_confidence_model: true`).  Score mode (ddmi_forward, old_cg_model.py:293-352) and confidence
   * mode (ddmi_confidence), OldAtomEncoder, sh_lmax = 2. */
  int32_t old_model;
  /* confidence-mode options of the new classes (cg_model.py:184-207, aa_model.py:177-211): per-atom predictor in front of the
   * graph mean (`atom_confidence_loss_weight > 0`, atom_num_confidence_outputs = len(atom_rmsd_classification_cutoff) + 1)
   * and one extra affinity output of confidence_predictor (`affinity_prediction`, parallel = 1) */
  int32_t atom_confidence, atom_num_confidence_outputs, affinity_prediction;
  /* get_timestep_embedding (utils/diffusion_utils.py:129-136): 0 = 'sinusoidal' (embedding_scale multiplies t), 1 = 'fourier'
   * (GaussianFourierProjection, :113-127; its frozen parameter W is the state_dict key `timestep_emb_func.W`) */
  int32_t embedding_type;
  /* FCBlock depth of the per-edge weight MLP of the embedding / interaction layers (models/layers.py:10-17,
   * tensor_layers.py:302-304): 2 (or 0) = Linear, ReLU, Linear; n > 2 adds n - 2 hidden Linear + ReLU (keys fc.3 .. fc.3(n-1)) */
  int32_t tp_weights_layers;
  /* CGModel.sidechain_predictor (models/cg_model.py:173-178,397-402; get_model: sidechain_loss_weight > 0 or backbone_loss_weight > 0,
   * utils/utils.py:274-275): an e3nn o3.Linear on the receptor rows of the last node table -> 4x0e + 2x1e + 4x0o + 2x1o, even and
   * odd halves summed -- the 4th element of the forward tuple.  State-dict key `sidechain_predictor.weight`; read with
   * ddmi_sidechain_pred after ddmi_forward.  CG models only (AAModel asserts it away, models/aa_model.py:38). */
  int32_t sidechain_pred;
  /* TensorProductConvLayer(depthwise=True) in the embedding and interaction layers (models/tensor_layers.py:248-290,324-325;
   * cg_model.py:124,147,168): an e3nn 'uvu' TensorProduct (one per-edge weight per path and input channel) followed by the shared
   * o3.Linear `linear_2`.  Both stages are linear in the weights, so at ddmi_commit_weights the pair is folded into the second
   * layer of the per-edge MLP of an equivalent fully connected tensor product (W2'[(path, u, w)] = W2[(path, u)] * linear_2[u, w];
   * the two e3nn normalisations multiply to the fully connected one) and the layer runs the same kernels.  CG models only. */
  int32_t depthwise_convolution;
  /* ---- execution options (not arguments of the reference's get_model; 0 = default).
   * edge_product: arithmetic of the per-edge product T_e = h_e . Y_d in the interaction layers (k_conv_fused):
   *   0 = v_mfma_f32_16x16x4_f32, an exact fp32 fma chain (the headline route);
   *   1 = split-bf16: both operands carried as bf16 hi + bf16 lo (16 significand bits), the four bf16 products of every
   *       f32 product on v_mfma_f32_16x16x32_bf16 with fp32 accumulation -- a quarter of the matrix-core time of route 0;
   *       applies to the static l <= 1 loops (sh_lmax = 1, ns % 16 == 0), other layers run route 0.  Same 1e-4 parity bar,
   *       reported as its own bench line (dtype "bf16x4-split edge product, f32 accumulate").
   * (The library reads no environment variable: the Python harness maps DDMI_EDGE_PRODUCT onto this field, diffdock_amd/lib.py.) */
  int32_t edge_product;
  ddmi_exec_options exec; /* kernel-route selection, all 0 = defaults (see above) */
} ddmi_config;

/* Static description of one collated batch of complexes = the fields of the PyG Batch the
 * path reads.  Node indices inside edge arrays are batch-global, as PyG
 * collation produces them.  lig_ptr / rec_ptr are HOST arrays [num_graphs+1] of cumulative
 * node counts.  The conformer fields describe ONE graph and are required only by
 * ddmi_modify_conformer / ddmi_sample, which (like the reference's modify_conformer_batch,
 * utils/diffusion_utils.py:60-64) assume all graphs in the batch are copies of one complex. */
typedef struct ddmi_complex {
  int32_t num_graphs, n_lig, n_rec, n_bond_edges, n_rec_edges, n_tor;
  const int32_t* lig_ptr;        /* host [B+1] */
  const int32_t* rec_ptr;        /* host [B+1] */
  const int32_t* lig_x;          /* [n_lig,16] categorical atom features            */
  const int32_t* bond_index;     /* [2,n_bond_edges] ('ligand','lig_bond','ligand') */
  const float* bond_attr;        /* [n_bond_edges, in_lig_edge_features]            */
  const uint8_t* edge_mask;      /* [n_bond_edges] rotatable directed bonds         */
  const float* rec_x;            /* [n_rec, 1+lm_embedding_dim]                     */
  const float* rec_pos;          /* [n_rec,3]                                       */
  const int32_t* rec_edge_index; /* [2,n_rec_edges] ('receptor','rec_contact','receptor') */
  const uint8_t* mask_rotate;    /* [n_tor/B, n_lig/B] or NULL                      */
  /* all_atoms only (models/aa_model.py:275-362): receptor heavy atoms and their two static relations */
  int32_t n_atom, n_atom_edges, n_atom_rec_edges;
  const int32_t* atom_ptr;            /* host [B+1] */
  const int32_t* atom_x;              /* [n_atom,4] categorical atom features                  */
  const float* atom_pos;              /* [n_atom,3]                                            */
  const int32_t* atom_edge_index;     /* [2,n_atom_edges] ('atom','atom_contact','atom')       */
  const int32_t* atom_rec_edge_index; /* [2,n_atom_rec_edges] ('atom','atom_rec_contact','receptor'): row 0 atoms, row 1 residues */
} ddmi_complex;

/* Reverse-diffusion loop parameters = the arguments of sampling() (utils/sampling.py:69-72).
 * Schedules are HOST arrays [inference_steps].  If z_* are NULL the Gaussian draws come
 * from a counter-based generator keyed by (seed, sample_ids[b], step, component), so a run
 * sharded over ranks reproduces the single-rank trajectories sample by sample. */
typedef struct ddmi_sample_cfg {
  int32_t inference_steps;
  const double* tr_schedule;
  const double* rot_schedule;
  const double* tor_schedule;
  int32_t ode, no_random, no_final_step_noise;
  double temp_sampling[3], temp_psi[3], temp_sigma_data[3];
  uint64_t seed;
  const int64_t* sample_ids; /* host [B] or NULL (= 0..B-1) */
  const float* z_tr;         /* device [steps,B,3] or NULL  */
  const float* z_rot;        /* device [steps,B,3] or NULL  */
  const float* z_tor;        /* device [steps,n_tor] or NULL */
  int32_t use_crop;          /* model_args.crop_beyond is not None (utils/sampling.py:104) */
  double crop_beyond;        /* per-step receptor crop at 3*tr_sigma + crop_beyond (utils/sampling.py:107) */
} ddmi_sample_cfg;

/* get_model(args, device, ...) -- utils/utils.py:172.  device = HIP device ordinal. */
int ddmi_create(const ddmi_config* cfg, int device, ddmi_model** out);
void ddmi_destroy(ddmi_model* m);
const char* ddmi_last_error(void);

/* model.load_state_dict(sd, strict=True) -- inference.py:202-203.  One call per tensor with
 * the reference's state_dict key, HOST fp32 data; then ddmi_commit_weights
 * checks that every expected key was provided, pre-packs and uploads. */
int ddmi_set_weight(ddmi_model* m, const char* key, const float* host_data, const int64_t* shape, int ndim);
int ddmi_commit_weights(ddmi_model* m);
/* Number of state_dict keys the configured architecture expects, and the i-th key / shape. */
int ddmi_num_weights(ddmi_model* m);
int ddmi_weight_spec(ddmi_model* m, int i, const char** key, int64_t shape[4], int* ndim);

/* Score-norm tables the reference builds at import: kind 0 = so3._exp_score_norms
 * (utils/so3.py:59, 2000 entries), kind 1 = torus.score_norm_ (utils/torus.py:72-76, 5001
 * entries).  HOST float64. */
int ddmi_set_table(ddmi_model* m, int kind, const double* host_data, int64_t n);
/* Frequencies of the sinusoidal timestep embedding (utils/diffusion_utils.py:99-104),
 * HOST fp32 [sigma_embed_dim/2]; optional -- computed in-library when not supplied. */
int ddmi_set_time_frequencies(ddmi_model* m, const float* host_freq, int64_t n);

/* batch.to(device) + the receptor-side caching of CGModel.embedding (models/cg_model.py:273-295). */
int ddmi_set_complex(ddmi_model* m, const ddmi_complex* c, ddmi_stream stream);

/* tr, rot, tor = model(batch)[:3] -- models/cg_model.py:308-424.
 * lig_pos [n_lig,3]; t_* [B] = batch.complex_t[...]; outputs tr [B,3], rot [B,3], tor [n_tor]. */
int ddmi_forward(ddmi_model* m, const float* lig_pos, const float* t_tr, const float* t_rot, const float* t_tor,
                 float* tr_out, float* rot_out, float* tor_out, ddmi_stream stream);

/* sidechain_pred = model(batch)[3] -- models/cg_model.py:397-402 -- of the LAST ddmi_forward call on this handle:
 * out [n_rec, 10].  Requires ddmi_config.sidechain_pred. */
int ddmi_sidechain_pred(ddmi_model* m, float* out, ddmi_stream stream);

/* confidence, atom_confidence = confidence_model(batch) -- utils/sampling.py:221, models/cg_model.py:353-366 /
 * models/aa_model.py:431-452.  Requires ddmi_config.confidence_mode.  t_* are used raw (sampling() passes 0).
 * conf_out [B, num_confidence_outputs (+ 1 with affinity_prediction)]; atom_conf_out [n_lig, atom_num_confidence_outputs]
 * with ddmi_config.atom_confidence, else NULL (the reference returns zeros there). */
int ddmi_confidence(ddmi_model* m, const float* lig_pos, const float* t_tr, const float* t_rot, const float* t_tor,
                    float* conf_out, float* atom_conf_out, ddmi_stream stream);

/* crop_beyond(graph, cutoff) -- utils/utils.py:388-413 as applied by sampling() before each model call
 * (utils/sampling.py:104-109): subsequent ddmi_forward calls drop the residues farther than `cutoff` from every
 * ligand atom of their graph, with the contact edges touching them.  cutoff <= 0 switches cropping off. */
int ddmi_set_crop_cutoff(ddmi_model* m, float cutoff);

/* modify_conformer_batch -- utils/diffusion_utils.py:60-78.  pos [n_lig,3] updated in place;
 * tr_update [B,3], rot_update [B,3], tor_update [n_tor] or NULL. */
int ddmi_modify_conformer(ddmi_model* m, float* lig_pos, const float* tr_update, const float* rot_update,
                          const float* tor_update, ddmi_stream stream);

/* The hot loop of sampling() -- utils/sampling.py:96-191 -- for one collated batch, entirely
 * on the device (no host synchronisation: the call returns once the steps are enqueued).  lig_pos [n_lig,3] is updated
 * in place. */
int ddmi_sample(ddmi_model* m, float* lig_pos, const ddmi_sample_cfg* cfg, ddmi_stream stream);

/* One step of the score -> perturbation arithmetic of sampling() -- utils/sampling.py:117-186 -- on caller-owned score
 * arrays, in place: the NaN guard (:117-131: if any pose's mean translation score is NaN, NaN -> 0.01 * nanmean|x|,
 * +-inf -> +-that value, per score tensor) followed by  g^2 dt (lambda + T psi / 2) * score + g sqrt(dt (1 + psi)) * z
 * with the coefficients of step `step` of cfg's schedules.  tr [B,3], rot [B,3], tor [n_tor] or NULL.  This is what
 * ddmi_sample applies between ddmi_forward and ddmi_modify_conformer. */
int ddmi_perturb(ddmi_model* m, float* tr, float* rot, float* tor, const ddmi_sample_cfg* cfg, int step,
                 ddmi_stream stream);

/* Introspection for tests and profiling: copy a named internal buffer to the host.
 * ddmi_debug_shape returns the element count and up to 4 dims; names are listed in DESIGN.md. */
int ddmi_debug_shape(ddmi_model* m, const char* name, int64_t shape[4], int* ndim, int* is_int);
int ddmi_debug_read(ddmi_model* m, const char* name, void* host_dst, size_t bytes, ddmi_stream stream);
/* Real-basis Wigner-3j tensor used for weight pre-packing (host double [(2l1+1)(2l2+1)(2l3+1)]). */
int ddmi_wigner_3j(int l1, int l2, int l3, double* host_out);
/* The in-library noise generator of ddmi_sample / ddmi_perturb (noise pointers NULL), exposed for tests: the DEVICE code's
 * Philox4x32-10 block for `n` (counter[4], key[2]) pairs (host arrays in, host uint32 [n][4] out; Random123 known-answer
 * vectors), and the standard-normal draws of samples [sample0, sample0 + n_samples) x components [0, n_comp) at one step, as
 * ddmi_perturb keys them (seed, sample id, step, component) -> device float [n_samples][n_comp].
 * Replaces torch.normal at utils/sampling.py:140-154 when the caller supplies no draws. */
int ddmi_debug_philox(const uint32_t* counters, const uint32_t* keys, int n, uint32_t* host_out);
int ddmi_debug_normal(uint64_t seed, int64_t sample0, int n_samples, int step, int n_comp, float* dev_out, ddmi_stream stream);
/* Name / duration table of the kernels launched by the last ddmi_forward when timing is on.  enabled: 0 = off, 1 = one row per
 * kernel name, 2 = k_conv_fused split per edge group, 3 = per (layer, edge group). */
int ddmi_set_kernel_timing(ddmi_model* m, int enabled);
int ddmi_kernel_timings(ddmi_model* m, int i, const char** name, double* ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* DDMI_H */
