// extern "C" surface of libddmi.so (include/ddmi.h).
#include <cstdlib>
#include <cstring>

#include "model.h"

using namespace ddmi;

struct ddmi_model { Model m; };

static thread_local std::string g_err;

template <class F> static int guard(F&& f) {
  try { f(); return DDMI_OK; }
  catch (const Error& e) { g_err = e.what(); return e.code; }
  catch (const std::exception& e) { g_err = e.what(); return DDMI_ERR_ARG; }
}

extern "C" {

const char* ddmi_last_error(void) { return g_err.c_str(); }

int ddmi_create(const ddmi_config* cfg, int device, ddmi_model** out) {
  return guard([&] {
    DDMI_REQUIRE(cfg && out, DDMI_ERR_ARG, "null argument");
    DDMI_REQUIRE(cfg->struct_size == sizeof(ddmi_config), DDMI_ERR_ARG,
                 "ddmi_config.struct_size != sizeof(ddmi_config) of this library: the caller was built against another include/ddmi.h");
    DDMI_CHECK_HIP(hipSetDevice(device));
    auto* h = new ddmi_model();
    h->m.cfg = *cfg;
    h->m.device = device;
    try { build_weight_spec(h->m); } catch (...) { delete h; throw; }
    {   // execution options (include/ddmi.h, ddmi_exec_options): 0 = default everywhere; the library reads no environment variable
      const ddmi_exec_options& x = h->m.cfg.exec;
      auto in = [&](int v, int hi, const char* name) {
        if (v < 0 || v > hi) { delete h; throw Error(DDMI_ERR_ARG, std::string("ddmi_config.exec.") + name + ": out of range"); }
      };
      if (h->m.cfg.edge_product < 0 || h->m.cfg.edge_product > 1) { delete h; throw Error(DDMI_ERR_ARG, "ddmi_config.edge_product: 0 (f32) or 1 (bf16x4)"); }
      in(x.streams, 1, "streams"); in(x.dense_rows, 2, "dense_rows"); in(x.shared_tiles, 2, "shared_tiles");
      in(x.packed_granules, 1, "packed_granules"); in(x.merged_granule, 1, "merged_granule"); in(x.pre_reduce, 1, "pre_reduce");
      in(x.hidden_mm, 1, "hidden_mm"); in(x.fc1_batch, 1, "fc1_batch"); in(x.tile_split, 8, "tile_split");
      in(x.tile_split_small, 8, "tile_split_small"); in(x.hidden_grid, 1 << 20, "hidden_grid"); in(x.tp_apply, 3, "tp_apply");
      in(x.tile_per_pose, 1, "tile_per_pose"); in(x.layer_overlap, 2, "layer_overlap");
      in(x.grouped, 2, "grouped"); in(x.grouped_split, 8, "grouped_split"); in(x.vn_build, 1, "vn_build"); in(x.node_update, 3, "node_update"); in(x.tile_split_last, 8, "tile_split_last"); in(x.tile_split_rule, 2, "tile_split_rule"); in(x.group_order, 3, "group_order"); in(x.list_caps, 1, "list_caps"); in(x.time_terms, 1, "time_terms");
      h->m.two_streams = x.streams == 0;
      h->m.fused_dense = x.dense_rows == 0 ? 1 : x.dense_rows == 1 ? 0 : 2;
      h->m.fused_shared = x.shared_tiles == 0 ? 1 : x.shared_tiles == 1 ? 0 : 2;
      h->m.fused_pack = x.packed_granules == 0;
      h->m.fused_tri = x.merged_granule == 0;
      h->m.fused_prered = x.pre_reduce == 0;
      h->m.fused_mm = x.hidden_mm == 0;
      h->m.fc1_batch = x.fc1_batch == 0;
      h->m.fused_ysplit = x.tile_split;
      h->m.fused_ysplit_small = x.tile_split_small;
      h->m.eh_grid = x.hidden_grid > 0 ? x.hidden_grid : 2048;
      h->m.tp_form = x.tp_apply == 0 ? -1 : x.tp_apply - 1;   // 0 wave, 1 edge, 2 thread
      h->m.tile_per_pose = x.tile_per_pose != 0;
      h->m.layer_overlap = x.layer_overlap;
      h->m.grouped = x.grouped;
      h->m.grouped_split = x.grouped_split;
      h->m.vn_merge = x.vn_build == 0;
      h->m.node_update = x.node_update >= 1;
      h->m.node_update_wpn = x.node_update == 2 ? 1 : x.node_update == 3 ? 4 : 0;
      h->m.fused_ysplit_last = x.tile_split_last;
      h->m.ys_rounds = x.tile_split_rule != 1;
      h->m.ys_rounds_small = x.tile_split_rule == 2;
      h->m.group_order = x.group_order;
      h->m.tight_caps = x.list_caps == 1;
      h->m.time_terms_fused = x.time_terms == 1;
      {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus >= 16) h->m.n_cus = cus;
      }
    }
    DDMI_CHECK_HIP(hipStreamCreateWithFlags(&h->m.side_stream, hipStreamNonBlocking));
    DDMI_CHECK_HIP(hipEventCreate(&h->m.ev_fork));
    DDMI_CHECK_HIP(hipEventCreate(&h->m.ev_join));
    DDMI_CHECK_HIP(hipEventCreate(&h->m.ev_cross));
    DDMI_CHECK_HIP(hipEventCreate(&h->m.ev_terms));
    *out = h;
  });
}

void ddmi_destroy(ddmi_model* h) {
  if (!h) return;
  Model& m = h->m;
  if (m.side_stream) { (void)hipStreamSynchronize(m.side_stream); (void)hipStreamDestroy(m.side_stream); }
  for (hipEvent_t e : {m.ev_fork, m.ev_join, m.ev_cross, m.ev_terms}) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : m.ev_pipe) (void)hipEventDestroy(e);
  delete h;
}

int ddmi_num_weights(ddmi_model* h) { return h ? (int)h->m.spec.size() : DDMI_ERR_ARG; }

int ddmi_weight_spec(ddmi_model* h, int i, const char** key, int64_t shape[4], int* ndim) {
  return guard([&] {
    DDMI_REQUIRE(h && i >= 0 && i < (int)h->m.spec.size(), DDMI_ERR_ARG, "bad index");
    *key = h->m.spec[i].first.c_str();
    *ndim = (int)h->m.spec[i].second.size();
    for (int d = 0; d < *ndim; ++d) shape[d] = h->m.spec[i].second[d];
  });
}

int ddmi_set_weight(ddmi_model* h, const char* key, const float* data, const int64_t* shape, int ndim) {
  return guard([&] {
    DDMI_REQUIRE(h && key && shape && ndim >= 1 && ndim <= 4, DDMI_ERR_ARG, "bad argument");
    int64_t numel = 1;
    for (int d = 0; d < ndim; ++d) numel *= shape[d];
    DDMI_REQUIRE(data || numel == 0, DDMI_ERR_ARG, "null data");
    Model& m = h->m;
    bool known = false;
    for (auto& kv : m.spec) if (kv.first == key) {
      known = true;
      DDMI_REQUIRE((int)kv.second.size() == ndim && std::equal(shape, shape + ndim, kv.second.begin()), DDMI_ERR_KEY,
                   std::string("shape mismatch for ") + key);
    }
    DDMI_REQUIRE(known, DDMI_ERR_KEY, std::string("unexpected state_dict key: ") + key);
    HostTensor t;
    t.shape.assign(shape, shape + ndim);
    if (numel) t.data.assign(data, data + numel);
    m.host_w[key] = std::move(t);
    m.committed = false;
  });
}

int ddmi_commit_weights(ddmi_model* h) {
  return guard([&] { DDMI_REQUIRE(h, DDMI_ERR_ARG, "null model"); DDMI_CHECK_HIP(hipSetDevice(h->m.device)); commit_weights(h->m); });
}

int ddmi_set_table(ddmi_model* h, int kind, const double* data, int64_t n) {
  return guard([&] {
    DDMI_REQUIRE(h && data && n > 1 && (kind == 0 || kind == 1), DDMI_ERR_ARG, "bad table");
    DDMI_CHECK_HIP(hipSetDevice(h->m.device));
    std::vector<float> f(data, data + n);  // the reference casts the looked-up values with .float()
    float* p = h->m.tpool.upload(f);
    if (kind == 0) { h->m.so3_table = p; h->m.so3_n = (int)n; }
    else { h->m.torus_table = p; h->m.torus_n = (int)n; }
  });
}

int ddmi_set_time_frequencies(ddmi_model* h, const float* freq, int64_t n) {
  return guard([&] {
    DDMI_REQUIRE(h && freq && n == h->m.cfg.sigma_embed_dim / 2, DDMI_ERR_ARG, "need sigma_embed_dim/2 frequencies");
    DDMI_REQUIRE(!h->m.committed, DDMI_ERR_STATE, "set frequencies before ddmi_commit_weights");
    h->m.time_freq_host.assign(freq, freq + n);
  });
}

int ddmi_set_complex(ddmi_model* h, const ddmi_complex* c, ddmi_stream s) {
  return guard([&] {
    DDMI_REQUIRE(h && c, DDMI_ERR_ARG, "null argument");
    DDMI_CHECK_HIP(hipSetDevice(h->m.device));
    set_complex(h->m, *c, (hipStream_t)s);
  });
}

int ddmi_forward(ddmi_model* h, const float* lig_pos, const float* t_tr, const float* t_rot, const float* t_tor,
                 float* tr_out, float* rot_out, float* tor_out, ddmi_stream s) {
  return guard([&] {
    DDMI_REQUIRE(h && lig_pos && t_tr && t_rot && t_tor && tr_out && rot_out, DDMI_ERR_ARG, "null argument");
    DDMI_REQUIRE(!h->m.cfg.confidence_mode, DDMI_ERR_STATE, "confidence models are evaluated with ddmi_confidence");
    DDMI_CHECK_HIP(hipSetDevice(h->m.device));
    forward(h->m, lig_pos, t_tr, t_rot, t_tor, tr_out, rot_out, tor_out, (hipStream_t)s);
  });
}

int ddmi_sidechain_pred(ddmi_model* h, float* out, ddmi_stream s) {
  return guard([&] {
    DDMI_REQUIRE(h && out, DDMI_ERR_ARG, "null argument");
    DDMI_REQUIRE(h->m.cfg.sidechain_pred, DDMI_ERR_STATE, "ddmi_sidechain_pred needs a model created with sidechain_pred");
    DDMI_CHECK_HIP(hipSetDevice(h->m.device));
    sidechain_pred(h->m, out, (hipStream_t)s);
  });
}

int ddmi_confidence(ddmi_model* h, const float* lig_pos, const float* t_tr, const float* t_rot, const float* t_tor,
                    float* conf_out, float* atom_conf_out, ddmi_stream s) {
  return guard([&] {
    DDMI_REQUIRE(h && lig_pos && t_tr && t_rot && t_tor && conf_out, DDMI_ERR_ARG, "null argument");
    DDMI_REQUIRE(h->m.cfg.confidence_mode, DDMI_ERR_STATE, "ddmi_confidence needs a model created with confidence_mode");
    DDMI_REQUIRE(!h->m.cfg.atom_confidence || atom_conf_out, DDMI_ERR_ARG, "atom_confidence models need atom_conf_out");
    DDMI_CHECK_HIP(hipSetDevice(h->m.device));
    forward(h->m, lig_pos, t_tr, t_rot, t_tor, nullptr, nullptr, nullptr, (hipStream_t)s, conf_out, atom_conf_out);
  });
}

int ddmi_set_crop_cutoff(ddmi_model* h, float cutoff) {
  if (!h) return DDMI_ERR_ARG;
  h->m.crop_cutoff = cutoff > 0.f ? cutoff : 0.f;
  return DDMI_OK;
}

int ddmi_modify_conformer(ddmi_model* h, float* lig_pos, const float* tr, const float* rot, const float* tor, ddmi_stream s) {
  return guard([&] {
    DDMI_REQUIRE(h && lig_pos && tr && rot, DDMI_ERR_ARG, "null argument");
    DDMI_CHECK_HIP(hipSetDevice(h->m.device));
    modify_conformer(h->m, lig_pos, tr, rot, tor, (hipStream_t)s);
  });
}

int ddmi_sample(ddmi_model* h, float* lig_pos, const ddmi_sample_cfg* cfg, ddmi_stream s) {
  return guard([&] {
    DDMI_REQUIRE(h && lig_pos && cfg, DDMI_ERR_ARG, "null argument");
    DDMI_CHECK_HIP(hipSetDevice(h->m.device));
    sample(h->m, lig_pos, *cfg, (hipStream_t)s);
  });
}

int ddmi_perturb(ddmi_model* h, float* tr, float* rot, float* tor, const ddmi_sample_cfg* cfg, int step, ddmi_stream s) {
  return guard([&] {
    DDMI_REQUIRE(h && tr && rot && cfg, DDMI_ERR_ARG, "null argument");
    DDMI_CHECK_HIP(hipSetDevice(h->m.device));
    perturb(h->m, tr, rot, tor, *cfg, step, (hipStream_t)s);
  });
}

int ddmi_debug_shape(ddmi_model* h, const char* name, int64_t shape[4], int* ndim, int* is_int) {
  return guard([&] {
    DDMI_REQUIRE(h && name, DDMI_ERR_ARG, "null argument");
    auto it = h->m.debug.find(name);
    DDMI_REQUIRE(it != h->m.debug.end(), DDMI_ERR_KEY, std::string("unknown debug buffer: ") + name);
    *ndim = (int)it->second.shape.size();
    for (int d = 0; d < *ndim && d < 4; ++d) shape[d] = it->second.shape[d];
    *is_int = it->second.is_int;
  });
}

int ddmi_debug_read(ddmi_model* h, const char* name, void* dst, size_t bytes, ddmi_stream s) {
  return guard([&] {
    DDMI_REQUIRE(h && name && dst, DDMI_ERR_ARG, "null argument");
    auto it = h->m.debug.find(name);
    DDMI_REQUIRE(it != h->m.debug.end(), DDMI_ERR_KEY, std::string("unknown debug buffer: ") + name);
    size_t n = 4;
    for (auto d : it->second.shape) n *= (size_t)d;
    DDMI_REQUIRE(bytes <= n, DDMI_ERR_ARG, "read past the end of the buffer");
    DDMI_CHECK_HIP(hipStreamSynchronize((hipStream_t)s));
    DDMI_CHECK_HIP(hipMemcpy(dst, it->second.ptr, bytes, hipMemcpyDeviceToHost));
  });
}

int ddmi_wigner_3j(int l1, int l2, int l3, double* out) {
  return guard([&] {
    DDMI_REQUIRE(out && l1 >= 0 && l2 >= 0 && l3 >= std::abs(l1 - l2) && l3 <= l1 + l2, DDMI_ERR_ARG, "bad l");
    auto w = wigner_3j(l1, l2, l3);
    std::memcpy(out, w.data(), w.size() * sizeof(double));
  });
}

int ddmi_debug_philox(const uint32_t* counters, const uint32_t* keys, int n, uint32_t* host_out) {
  return guard([&] {
    DDMI_REQUIRE(counters && keys && host_out && n > 0, DDMI_ERR_ARG, "null argument");
    struct Buf {   // freed on every exit path, also when a later call throws
      unsigned* p = nullptr;
      explicit Buf(size_t bytes) { DDMI_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&p), bytes)); }
      ~Buf() { if (p) (void)hipFree(p); }
    };
    Buf c((size_t)n * 16), k((size_t)n * 8), o((size_t)n * 16);
    hipStream_t st = nullptr;
    DDMI_CHECK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } sg{st};
    DDMI_CHECK_HIP(hipMemcpyAsync(c.p, counters, (size_t)n * 16, hipMemcpyHostToDevice, st));
    DDMI_CHECK_HIP(hipMemcpyAsync(k.p, keys, (size_t)n * 8, hipMemcpyHostToDevice, st));
    launch_debug_philox(c.p, k.p, n, o.p, st);
    DDMI_CHECK_HIP(hipMemcpyAsync(host_out, o.p, (size_t)n * 16, hipMemcpyDeviceToHost, st));
    DDMI_CHECK_HIP(hipStreamSynchronize(st));
  });
}

int ddmi_debug_normal(uint64_t seed, int64_t sample0, int n_samples, int step, int n_comp, float* dev_out, ddmi_stream s) {
  return guard([&] {
    DDMI_REQUIRE(dev_out && n_samples > 0 && n_comp > 0, DDMI_ERR_ARG, "bad argument");
    launch_debug_normal(seed, sample0, n_samples, step, n_comp, dev_out, (hipStream_t)s);
  });
}

int ddmi_set_kernel_timing(ddmi_model* h, int enabled) {
  if (!h) return DDMI_ERR_ARG;
  resolve_timings(h->m);
#if defined(DDMI_PROFILING)
  (void)hipDeviceSynchronize();
  ddmi::fc_prof_report();   // in-kernel phase clocks / workgroup stamps since the last call (profiling builds only)
#endif
  h->m.timing = enabled != 0;
  h->m.timing_level = enabled;
  if (enabled) h->m.phases.clear();
  return DDMI_OK;
}
int ddmi_kernel_timings(ddmi_model* h, int i, const char** name, double* ms, int64_t* launches) {
  if (!h) return DDMI_ERR_ARG;
  resolve_timings(h->m);
  if (i < 0 || i >= (int)h->m.phases.size()) return DDMI_ERR_ARG;
  *name = h->m.phases[i].name.c_str(); *ms = h->m.phases[i].ms; *launches = h->m.phases[i].launches;
  return DDMI_OK;
}

}  // extern "C"
