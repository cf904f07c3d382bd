// Weight ABI = the reference state_dict keys (models/cg_model.py:85-255,
// models/layers.py:10-67, models/tensor_layers.py:298-307) and their device packing.
#include <algorithm>
#include <cmath>

#include "model.h"

namespace ddmi {

static const int LIG_DIMS[16] = {119, 4, 12, 12, 8, 10, 6, 6, 2, 8, 2, 2, 2, 2, 2, 2};  // datasets/process_mols.py:59-76
static const int REC_DIMS[1] = {38};                                                   // process_mols.py:85-87
static const int ATOM_DIMS[4] = {38, 119, 23, 38};                                     // process_mols.py:78-83 (rec_atom_feature_dims)

DevicePool::~DevicePool() { release(); }
void DevicePool::release() {
  for (void* p : blocks_) (void)hipFree(p);
  blocks_.clear();
  total_ = 0;
}
void* DevicePool::alloc_bytes(size_t n) {
  void* p = nullptr;
  const size_t bytes = round_up((long)(n ? n : 1), 256);
  DDMI_CHECK_HIP(hipMalloc(&p, bytes));
  blocks_.push_back(p);
  total_ += bytes;
  return p;
}

static Irreps layer_irreps(const ddmi_config& c, int i) {
  const int ns = c.ns, nv = c.nv, last = c.reduce_pseudoscalars ? nv : ns;
  i = std::min(i, 3);
  if (c.use_second_order_repr) {
    switch (i) {
      case 0: return make_irreps({{ns, 0, 1}});
      case 1: return make_irreps({{ns, 0, 1}, {nv, 1, -1}, {nv, 2, 1}});
      case 2: return make_irreps({{ns, 0, 1}, {nv, 1, -1}, {nv, 2, 1}, {nv, 1, 1}, {nv, 2, -1}});
      default: return make_irreps({{ns, 0, 1}, {nv, 1, -1}, {nv, 2, 1}, {nv, 1, 1}, {nv, 2, -1}, {last, 0, -1}});
    }
  }
  switch (i) {
    case 0: return make_irreps({{ns, 0, 1}});
    case 1: return make_irreps({{ns, 0, 1}, {nv, 1, -1}});
    case 2: return make_irreps({{ns, 0, 1}, {nv, 1, -1}, {nv, 1, 1}});
    default: return make_irreps({{ns, 0, 1}, {nv, 1, -1}, {nv, 1, 1}, {last, 0, -1}});
  }
}

static int conv_groups(const ddmi_config& c, int l) {
  if (!c.differentiate_convolutions) return 1;
  if (c.all_atoms) return l == c.num_conv_layers - 1 ? 3 : 9;   // models/aa_model.py:157
  return l == c.num_conv_layers - 1 ? 2 : 4;
}

// A depthwise TensorProductConvLayer (models/tensor_layers.py:248-290): instructions (i_in, i_sh, ir_out in ir_in * ir_sh if it
// occurs in the output irreps), in e3nn's creation order, each with mul_in 'uvu' weights; linear_2 = o3.Linear(irreps_mid.sort()
// .simplify(), out): per reached output irrep one [rows, mul_out] slot, rows = the instructions' multiplicities stacked in
// creation order (Irreps.sort is stable), slots in the sorted order of the irreps ((l, p) tuples: odd parity first).
struct DwIns { int i1, i2, io, mul, w_off, row_base, lin_off; };
static std::vector<DwIns> depthwise_instructions(const Irreps& in, const Irreps& sh, const Irreps& out, int* n_tp, int* n_lin) {
  std::vector<DwIns> ins;
  int off = 0;
  std::map<std::pair<int, int>, int> rows;   // (l, p) -> rows so far
  for (int i1 = 0; i1 < (int)in.size(); ++i1)
    for (int i2 = 0; i2 < (int)sh.size(); ++i2)
      for (int l = std::abs(in[i1].l - sh[i2].l); l <= in[i1].l + sh[i2].l; ++l) {
        const int p = in[i1].p * sh[i2].p;
        int io = -1, hits = 0;
        for (int o = 0; o < (int)out.size(); ++o) if (out[o].l == l && out[o].p == p) { io = o; ++hits; }
        if (io < 0) continue;
        if (hits != 1 || sh[i2].mul != 1) throw Error(DDMI_ERR_ARG, "depthwise layer: repeated output irrep / spherical harmonics with multiplicity");
        ins.push_back({i1, i2, io, in[i1].mul, off, rows[{l, p}], 0});
        rows[{l, p}] += in[i1].mul;
        off += in[i1].mul;
      }
  int lin = 0;
  std::map<int, int> lin_off_of;   // output block -> offset of its linear_2 slot
  for (auto& kv : rows) {          // std::map iterates (l, p) in tuple order
    for (int o = 0; o < (int)out.size(); ++o)
      if (out[o].l == kv.first.first && out[o].p == kv.first.second) { lin_off_of[o] = lin; lin += kv.second * out[o].mul; }
  }
  for (auto& q : ins) q.lin_off = lin_off_of[q.io];
  if (n_tp) *n_tp = off;
  if (n_lin) *n_lin = lin;
  return ins;
}

static void init_conv_meta(const ddmi_config& c, ConvW& L, const std::string& name, const Irreps& in, const Irreps& sh,
                           const Irreps& out, int n_edge, int G, bool faster, bool residual, bool yform) {
  L.depthwise = c.depthwise_convolution != 0 && yform;   // embedding / interaction layers; the read-out convolutions stay fully connected
  if (L.depthwise) {   // e3nn semantics (the depthwise branch replaces FasterTensorProduct): the equivalent fully connected table
    faster = false;
    depthwise_instructions(in, sh, out, &L.Wn_dw, &L.n_lin2);
  }
  L.name = name; L.G = G; L.faster = faster; L.residual = residual; L.has_bn = c.batch_norm != 0; L.yform = yform;
  L.in_irr = in; L.sh_irr = sh; L.out_irr = out;
  L.table = faster ? faster_table(in, out) : fctp_table(in, sh, out);
  std::stable_sort(L.table.paths.begin(), L.table.paths.end(),
                   [](const TPPath& a, const TPPath& b) { return a.out_block < b.out_block; });
  L.n_edge = n_edge; L.H = n_edge; L.HK = L.H + 1; 
  L.D_in = irreps_dim(in); L.D_out = irreps_dim(out); L.sh_dim = irreps_dim(sh); L.Wn = L.table.weight_numel;
  // item-major column layout: per output block, per w, one item of (power-of-two quads) x 4 columns
  int nt = 0;
  for (int ob = 0; ob < (int)out.size(); ++ob) {
    int wi = 0;
    for (auto& p : L.table.paths) if (p.out_block == ob) wi += p.din;
    int quads = 1;
    while (quads * 4 < wi) quads *= 2;
    if (quads > 16 && yform) throw Error(DDMI_ERR_ARG, "tensor product with more than 64 terms per output channel");   // (second-order features reach 20-35: several granules of 4 slots whose message columns add up)
    nt = (int)round_up(nt, quads * 4) + out[ob].mul * quads * 4;
  }
  L.NT = nt;
}

void build_weight_spec(Model& m) {
  const ddmi_config& c = m.cfg;
  DDMI_REQUIRE(c.ns > 0 && c.nv >= 0 && c.num_conv_layers >= 1, DDMI_ERR_ARG, "bad ns/nv/num_conv_layers");
  DDMI_REQUIRE(c.sh_lmax == 1 || c.sh_lmax == 2, DDMI_ERR_ARG, "sh_lmax must be 1 or 2");
  DDMI_REQUIRE(c.sigma_embed_dim % 2 == 0 && c.sigma_embed_dim >= 4, DDMI_ERR_ARG, "sigma_embed_dim must be even");
  DDMI_REQUIRE(c.embed_also_ligand || c.num_prot_emb_layers == 0 || c.all_atoms, DDMI_ERR_ARG,
               "embed_also_ligand=False with embedding layers is rejected by the reference's CGModel (cg_model.py:263)");
  DDMI_REQUIRE(irreps_dim(layer_irreps(c, 3)) <= XS, DDMI_ERR_ARG, "irreps wider than the node-table stride");
  m.ns = c.ns; m.sd = c.sigma_embed_dim; m.D = c.distance_embed_dim; m.Dc = c.cross_distance_embed_dim;
  m.nf = c.in_lig_edge_features; m.lm = c.lm_embedding_dim; m.H = 3 * c.ns;
  auto& S = m.spec;
  S.clear();
  const int ns = m.ns, sd = m.sd;
  auto lin = [&](const std::string& n, int fin, int fout, bool bias = true) {
    S.push_back({n + ".weight", {fout, fin}});
    if (bias) S.push_back({n + ".bias", {fout}});
  };
  auto mlp = [&](const std::string& n, int fin, int hid, int fout) { lin(n + ".0", fin, hid); lin(n + ".3", hid, fout); };
  auto encoder = [&](const std::string& n, const int* dims, int nd, int extra) {
    for (int i = 0; i < nd; ++i) S.push_back({n + ".atom_embedding_list." + std::to_string(i) + ".weight", {dims[i], ns}});
    if (extra > 0) lin(n + ".additional_features_embedder", extra + ns, ns);
  };
  auto bn = [&](const std::string& n, const Irreps& irr) {
    int n0 = 0;
    for (auto& b : irr) if (b.l == 0 && b.p == 1) n0 += b.mul;
    const int nf = irreps_num(irr);
    S.push_back({n + ".running_mean", {n0}});
    S.push_back({n + ".running_var", {nf}});
    S.push_back({n + ".weight", {nf}});
    S.push_back({n + ".bias", {n0}});
  };
  auto conv = [&](ConvW& L) {
    for (int g = 0; g < L.G; ++g) {
      const std::string pre = L.G == 1 ? L.name + ".fc" : L.name + ".fc." + std::to_string(g);
      lin(pre + ".0", L.n_edge, L.H);
      for (int j = 1; j + 1 < L.TL; ++j) lin(pre + "." + std::to_string(3 * j), L.H, L.H);   // FCBlock hidden layers (models/layers.py:14-15)
      lin(pre + "." + std::to_string(3 * (L.TL - 1)), L.H, L.depthwise ? L.Wn_dw : L.Wn);
    }
    if (L.depthwise) S.push_back({L.name + ".linear_2.weight", {L.n_lin2}});
    if (L.has_bn) bn(L.name + ".batch_norm", L.out_irr);
  };
  const int TL = c.tp_weights_layers <= 0 ? 2 : c.tp_weights_layers;
  DDMI_REQUIRE(TL >= 2 && TL <= 8, DDMI_ERR_ARG, "tp_weights_layers must be >= 2 (FCBlock asserts it, models/layers.py:12)");
  DDMI_REQUIRE(c.embedding_type == 0 || c.embedding_type == 1, DDMI_ERR_ARG, "embedding_type: 0 sinusoidal, 1 fourier");
  // (ahead of the early returns of the legacy / confidence branches below: init_conv_meta would otherwise build depthwise
  // layers for class families that do not have them)
  DDMI_REQUIRE(!(c.depthwise_convolution && (c.all_atoms || c.old_model)), DDMI_ERR_ARG,
               "depthwise_convolution: CG models of the new class only (AAModel asserts it away, models/aa_model.py:39; get_model(old=True) never passes it)");
  DDMI_REQUIRE(!(c.sidechain_pred && (c.all_atoms || c.old_model)), DDMI_ERR_ARG,
               "sidechain_pred: CG models of the new class only (AAModel asserts it away, models/aa_model.py:38; the legacy class has no such head)");
  const Irreps sh = sh_irreps(c.sh_lmax);
  const bool faster = c.sh_lmax == 1 && !c.use_second_order_repr;
  const int K = c.num_prot_emb_layers, Lc = c.num_conv_layers;
  // score read-outs (cg_model.py:209-255, old_cg_model.py:156-200: the same modules in both class families)
  auto readout_spec = [&](const Irreps& last_out) {
    S.push_back({"center_distance_expansion.offset", {m.D}});
    mlp("center_edge_embedding", m.D + sd, ns, ns);
    const Irreps fout = c.odd_parity ? make_irreps({{1, 1, -1}, {1, 1, 1}}) : make_irreps({{2, 1, -1}, {2, 1, 1}});
    init_conv_meta(c, m.final_conv, "final_conv", last_out, sh, fout, 2 * ns, 1, false, false, false);
    conv(m.final_conv);
    mlp("tr_final_layer", 1 + sd, ns, 1);
    mlp("rot_final_layer", 1 + sd, ns, 1);
    if (!c.no_torsion) {
      mlp("final_edge_embedding", m.D, ns, ns);
      Irreps tor_sh;
      full_tp_dense(sh, make_irreps({{1, 2, 1}}), &tor_sh);
      const Irreps tout = c.odd_parity ? make_irreps({{ns, 0, -1}}) : make_irreps({{ns, 0, -1}, {ns, 0, 1}});
      init_conv_meta(c, m.tor_conv, "tor_bond_conv", last_out, tor_sh, tout, 3 * ns, 1, false, false, false);
      conv(m.tor_conv);
      lin("tor_final_layer.0", c.odd_parity ? ns : 2 * ns, ns, false);
      lin("tor_final_layer.3", ns, 1, false);
    }
  };
  if (c.old_model) {   // models/old_cg_model.py:18-200, models/layers.py:70-118, models/tensor_layers.py:338-380
    DDMI_REQUIRE(c.sh_lmax == 2 && !c.all_atoms && K == 0, DDMI_ERR_ARG,
                 "legacy class: sh_lmax = 2, CG graphs, no embedding layers");
    DDMI_REQUIRE(c.embedding_type == 0, DDMI_ERR_ARG, "legacy class: sinusoidal timestep embedding only");
    auto old_encoder = [&](const std::string& n, const int* dims, int nd, bool lm) {
      for (int i = 0; i < nd; ++i) S.push_back({n + ".atom_embedding_list." + std::to_string(i) + ".weight", {dims[i], ns}});
      lin(n + ".linear", sd, ns);
      if (lm) lin(n + ".lm_embedding_layer", 1280 + ns, ns);
    };
    old_encoder("lig_node_embedding", LIG_DIMS, 16, false);
    mlp("lig_edge_embedding", m.nf + sd + m.D, ns, ns);
    old_encoder("rec_node_embedding", REC_DIMS, 1, m.lm > 0);
    mlp("rec_edge_embedding", sd + m.D, ns, ns);
    mlp("cross_edge_embedding", sd + m.Dc, ns, ns);
    S.push_back({"lig_distance_expansion.offset", {m.D}});
    S.push_back({"rec_distance_expansion.offset", {m.D}});
    S.push_back({"cross_distance_expansion.offset", {m.Dc}});
    ddmi_config co = c;
    co.reduce_pseudoscalars = 0;
    m.rec_emb_layers.clear(); m.lig_emb_layers.clear(); m.conv_layers.clear();
    std::vector<ConvW>* fams[4] = {&m.old_lig, &m.old_rec, &m.old_l2r, &m.old_r2l};
    const char* names[4] = {"lig_conv_layers.", "rec_conv_layers.", "lig_to_rec_conv_layers.", "rec_to_lig_conv_layers."};
    for (int f = 0; f < 4; ++f) {
      fams[f]->assign(Lc, ConvW());
      for (int l = 0; l < Lc; ++l) {
        init_conv_meta(c, (*fams[f])[l], names[f] + std::to_string(l), layer_irreps(co, l), sh, layer_irreps(co, l + 1), 3 * ns, 1,
                       false, false, true);
        conv((*fams[f])[l]);
      }
    }
    if (!c.confidence_mode) {   // score mode of the legacy class: the read-outs of old_cg_model.py:156-200
      readout_spec(layer_irreps(co, Lc));
      return;
    }
    lin("confidence_predictor.0", Lc >= 3 ? 2 * ns : ns, ns);
    lin("confidence_predictor.4", ns, ns);
    lin("confidence_predictor.8", ns, c.affinity_prediction ? 2 : 1);
    for (int i : {1, 5})
      for (const char* k : {".weight", ".bias", ".running_mean", ".running_var"})
        S.push_back({"confidence_predictor." + std::to_string(i) + k, {ns}});
    return;
  }
  encoder("lig_node_embedding", LIG_DIMS, 16, sd);
  mlp("lig_edge_embedding", m.nf + sd + m.D, ns, ns);
  if (c.all_atoms) {   // models/aa_model.py:90-103
    DDMI_REQUIRE(m.D == m.Dc, DDMI_ERR_ARG, "AAModel feeds lig_distance_expansion into la_edge_embedding: distance_embed_dim must equal cross_distance_embed_dim");
    mlp("rec_sigma_embedding", sd, ns, ns);
    encoder("rec_node_embedding", REC_DIMS, 1, m.lm);
    mlp("rec_edge_embedding", m.D, ns, ns);
    encoder("atom_node_embedding", ATOM_DIMS, 4, 0);
    mlp("atom_edge_embedding", m.D, ns, ns);
    mlp("lr_edge_embedding", sd + m.Dc, ns, ns);
    mlp("ar_edge_embedding", m.D, ns, ns);
    mlp("la_edge_embedding", sd + m.Dc, ns, ns);
  } else {
    encoder("rec_node_embedding", REC_DIMS, 1, m.lm);
    mlp("rec_edge_embedding", m.D, ns, ns);
    mlp("rec_sigma_embedding", sd, ns, ns);
    mlp("cross_edge_embedding", sd + m.Dc, ns, ns);
  }
  S.push_back({"lig_distance_expansion.offset", {m.D}});
  S.push_back({"rec_distance_expansion.offset", {m.D}});
  S.push_back({"cross_distance_expansion.offset", {m.Dc}});
  if (c.embedding_type == 1) S.push_back({"timestep_emb_func.W", {sd / 2}});   // GaussianFourierProjection.W (nn.Parameter, requires_grad=False)
  m.rec_emb_layers.assign(K, ConvW());
  m.lig_emb_layers.assign(c.embed_also_ligand ? K : 0, ConvW());
  m.conv_layers.assign(Lc, ConvW());
  for (int i = 0; i < K; ++i) {
    init_conv_meta(c, m.rec_emb_layers[i], "rec_emb_layers." + std::to_string(i), layer_irreps(c, i), sh,
                   layer_irreps(c, i + 1), 3 * ns, (c.all_atoms && c.differentiate_convolutions) ? 4 : 1, faster, true, true);
    m.rec_emb_layers[i].TL = TL;
    conv(m.rec_emb_layers[i]);
  }
  for (int i = 0; i < (int)m.lig_emb_layers.size(); ++i) {
    init_conv_meta(c, m.lig_emb_layers[i], "lig_emb_layers." + std::to_string(i), layer_irreps(c, i), sh,
                   layer_irreps(c, i + 1), 3 * ns, 1, faster, true, true);
    m.lig_emb_layers[i].TL = TL;
    conv(m.lig_emb_layers[i]);
  }
  for (int l = 0; l < Lc; ++l) {
    init_conv_meta(c, m.conv_layers[l], "conv_layers." + std::to_string(l), layer_irreps(c, K + l), sh,
                   layer_irreps(c, K + l + 1), 3 * ns, conv_groups(c, l), faster, true, true);
    m.conv_layers[l].TL = TL;
    conv(m.conv_layers[l]);
  }
  const Irreps last_out = layer_irreps(c, K + Lc);
  if (c.confidence_mode) {   // cg_model.py:181-207: Linear, BatchNorm1d, ReLU, Dropout, Linear, BatchNorm1d, ReLU, Dropout, Linear
    DDMI_REQUIRE(c.num_confidence_outputs >= 1, DDMI_ERR_ARG, "num_confidence_outputs must be >= 1");
    int n_in = K + Lc >= 3 ? ns + (c.reduce_pseudoscalars ? c.nv : ns) : ns;
    auto predictor = [&](const std::string& name, int n_in_, int n_out) {
      lin(name + ".0", n_in_, ns);
      lin(name + ".4", ns, ns);
      lin(name + ".8", ns, n_out);
      for (int i : {1, 5})
        for (const char* k : {".weight", ".bias", ".running_mean", ".running_var"})
          S.push_back({name + "." + std::to_string(i) + k, {ns}});
    };
    if (c.atom_confidence) {   // cg_model.py:184-196: per-atom predictor; its last ns outputs feed the graph mean
      DDMI_REQUIRE(c.atom_num_confidence_outputs >= 1, DDMI_ERR_ARG, "atom_num_confidence_outputs must be >= 1");
      predictor("atom_confidence_predictor", n_in, c.atom_num_confidence_outputs + ns);
      n_in = ns;
    }
    predictor("confidence_predictor", n_in, c.num_confidence_outputs + (c.affinity_prediction ? 1 : 0));
    return;
  }
  if (c.sidechain_pred) {   // models/cg_model.py:173-178: o3.Linear(last_out -> 4x0e + 2x1e + 4x0o + 2x1o), one flat weight vector
    int n = 0;
    for (auto& b : last_out) n += b.mul * ((b.l == 0) ? 4 : (b.l == 1) ? 2 : 0);
    S.push_back({"sidechain_predictor.weight", {n}});
  }
  readout_spec(last_out);
}

// ------------------------------------------------------------------------------ commit
static const HostTensor& W(Model& m, const std::string& k) {
  auto it = m.host_w.find(k);
  DDMI_REQUIRE(it != m.host_w.end(), DDMI_ERR_KEY, "missing state_dict key: " + k);
  return it->second;
}
static float* up(Model& m, const std::string& k) { return m.wpool.upload(W(m, k).data); }

static Mlp2W up_mlp(Model& m, const std::string& n) {
  Mlp2W r;
  const HostTensor &w0 = W(m, n + ".0.weight"), &w3 = W(m, n + ".3.weight");
  r.in = (int)w0.shape[1]; r.hid = (int)w0.shape[0]; r.out = (int)w3.shape[0];
  r.W0 = up(m, n + ".0.weight"); r.b0 = up(m, n + ".0.bias");
  r.W3 = up(m, n + ".3.weight"); r.b3 = up(m, n + ".3.bias");
  return r;
}

static void commit_conv(Model& m, ConvW& L) {
  const int H = L.H, HK = L.HK;
  // (a second load_state_dict on the same handle starts from empty lists: the pool behind the old pointers was released)
  L.W1.clear(); L.b1.clear(); L.W2.clear(); L.b2.clear(); L.wpack.clear(); L.W1p.clear(); L.b1p.clear(); L.Wmid.clear(); L.bmid.clear();
  // coupling tables ---------------------------------------------------------------
  std::vector<DevPath> dp;
  std::vector<float> ctab;
  std::vector<GEntry> gmap;
  int n_off = 0;
  L.maxd = 1;
  for (auto& p : L.table.paths) {
    DevPath d{};
    d.n_off = n_off; d.mul_out = p.mul_out; d.din = p.din; d.ds = p.ds; d.dout = p.dout; d.s_off = p.s_off;
    d.c_off = (int)ctab.size(); d.o_off = p.o_off; d.mul_in = p.mul_in; d.i_off = p.i_off; d.w_off = p.w_off;
    d.g_off = (int)gmap.size();
    for (int i = 0; i < p.din; ++i)
      for (int k = 0; k < p.dout; ++k) gmap.push_back({d.c_off + i * p.ds * p.dout + k, p.s_off, p.ds, p.dout});
    for (double v : p.C) ctab.push_back((float)v);
    dp.push_back(d);
    n_off += p.din * p.mul_out;
    L.maxd = std::max(L.maxd, p.dout);
  }
  std::vector<CgItem> items;
  std::vector<ObInfo> obs;
  std::vector<int> slot_base(L.table.paths.size(), 0);   // column of (path, i=0) inside its item
  int col = 0;
  for (int ob = 0; ob < (int)L.out_irr.size(); ++ob) {
    int pb = -1, pe = -1, wi = 0;
    for (int i = 0; i < (int)L.table.paths.size(); ++i)
      if (L.table.paths[i].out_block == ob) { if (pb < 0) pb = i; pe = i + 1; slot_base[i] = wi; wi += L.table.paths[i].din; }
    if (pb < 0) { pb = pe = 0; }  // an output block no path reaches stays zero
    for (int w = 0; w < L.out_irr[ob].mul; ++w) items.push_back({pb, pe, L.out_irr[ob].off, L.out_irr[ob].d(), w});
    int quads = 1;
    while (quads * 4 < wi) quads *= 2;
    col = (int)round_up(col, quads * 4);
    obs.push_back({col, quads * 4, L.out_irr[ob].mul, L.out_irr[ob].off, L.out_irr[ob].d()});
    col += L.out_irr[ob].mul * quads * 4;
  }
  if (L.yform && col != L.NT) throw Error(DDMI_ERR_ARG, "internal: column layout mismatch");
  L.paths = m.wpool.upload(dp);
  L.ctab = m.wpool.upload(ctab);
  L.items = m.wpool.upload(items);
  L.n_items = (int)items.size();
  // dense layers --------------------------------------------------------------------
  // packed second layer for the node contraction: [k][path][16-w tile][lane 64][step], k = H is the bias row
  std::vector<int> wk_off(L.table.paths.size(), 0);
  int KS = 0;
  for (size_t pi = 0; pi < L.table.paths.size(); ++pi) {
    const TPPath& p = L.table.paths[pi];
    wk_off[pi] = KS;
    KS += (int)round_up(p.mul_in, 4) * (int)round_up(p.mul_out, 16);
  }
  L.KS = KS;
  for (int g = 0; g < L.G; ++g) {
    const std::string pre = L.G == 1 ? L.name + ".fc" : L.name + ".fc." + std::to_string(g);
    L.W1.push_back(up(m, pre + ".0.weight"));
    L.b1.push_back(up(m, pre + ".0.bias"));
    {   // position 4a + i of a block of 16 holds hidden unit 8 (i >> 1) + 2a + (i & 1): a lane of the MFMA first layer (row,
        // quarter a) then ends with k = 8g + 2a + {0, 1} of two 8-k groups -- the float4 of its own fragment lane.  The
        // per-node terms P, Q and the sigma rows come out of their GEMMs in the same order when they use this copy.
      const HostTensor &w1 = W(m, pre + ".0.weight"), &bb = W(m, pre + ".0.bias");
      const int Hh = (int)w1.shape[0], ne = (int)w1.shape[1];
      if (Hh % 16 == 0) {
        std::vector<float> wp((size_t)Hh * ne), bp(Hh);
        for (int pos = 0; pos < Hh; ++pos) {
          const int q = pos & 15, unit = (pos & ~15) + 8 * ((q & 3) >> 1) + 2 * (q >> 2) + (q & 1);
          std::copy(w1.data.begin() + (size_t)unit * ne, w1.data.begin() + (size_t)(unit + 1) * ne, wp.begin() + (size_t)pos * ne);
          bp[pos] = bb.data[unit];
        }
        L.W1p.push_back(m.wpool.upload(wp));
        L.b1p.push_back(m.wpool.upload(bp));
      } else {
        L.W1p.push_back(nullptr);
        L.b1p.push_back(nullptr);
      }
    }
    const std::string last = pre + "." + std::to_string(3 * (L.TL - 1));
    L.Wmid.emplace_back(); L.bmid.emplace_back();
    for (int j = 1; j + 1 < L.TL; ++j) {
      L.Wmid.back().push_back(up(m, pre + "." + std::to_string(3 * j) + ".weight"));
      L.bmid.back().push_back(up(m, pre + "." + std::to_string(3 * j) + ".bias"));
    }
    const HostTensor &w2s = W(m, last + ".weight"), &b2s = W(m, last + ".bias");
    HostTensor w2x, b2x;
    if (L.depthwise) {
      // fold 'uvu' TensorProduct + linear_2 into the last Linear of the edge MLP of the equivalent fully connected layer:
      //   W2'[(path, u, w)][k] = W2[(instruction, u)][k] * linear_2[(row_base + u), w]      (likewise the bias)
      // The e3nn normalisations agree: sqrt(dim_out) (uvu, one instruction per mid block) * rows^-1/2 (linear_2, path_normalization
      // 'element') = sqrt(dim_out / fan_in) of the fully connected product over the same paths (fan_in = rows).
      const HostTensor& l2 = W(m, L.name + ".linear_2.weight");
      const std::vector<DwIns> ins = depthwise_instructions(L.in_irr, L.sh_irr, L.out_irr, nullptr, nullptr);
      w2x.shape = {L.Wn, H}; w2x.data.assign((size_t)L.Wn * H, 0.f);
      b2x.shape = {L.Wn}; b2x.data.assign((size_t)L.Wn, 0.f);
      for (auto& p : L.table.paths) {
        const DwIns* q = nullptr;
        for (auto& c_ : ins)
          if (L.in_irr[c_.i1].off == p.i_off && L.sh_irr[c_.i2].off == p.s_off && c_.io == p.out_block) q = &c_;
        DDMI_REQUIRE(q && q->mul == p.mul_in, DDMI_ERR_ARG, "depthwise layer: path without an instruction");
        for (int u = 0; u < p.mul_in; ++u)
          for (int w = 0; w < p.mul_out; ++w) {
            const float f = l2.data[(size_t)q->lin_off + (size_t)(q->row_base + u) * p.mul_out + w];
            const size_t slot = (size_t)p.w_off + (size_t)u * p.mul_out + w, src = (size_t)q->w_off + u;
            for (int k = 0; k < H; ++k) w2x.data[slot * H + k] = w2s.data[src * H + k] * f;
            b2x.data[slot] = b2s.data[src] * f;
          }
      }
    }
    const HostTensor &w2 = L.depthwise ? w2x : w2s, &b2 = L.depthwise ? b2x : b2s;
    if (!L.yform) {
      L.W2.push_back(m.wpool.upload(w2.data));
      L.b2.push_back(m.wpool.upload(b2.data));
      continue;
    }
    std::vector<float> pack((size_t)HK * KS, 0.f);
    for (size_t pi = 0; pi < L.table.paths.size(); ++pi) {
      const TPPath& p = L.table.paths[pi];
      const int steps = (int)round_up(p.mul_in, 4) / 4;
      for (int k = 0; k < HK; ++k)
        for (int u = 0; u < p.mul_in; ++u)
          for (int w = 0; w < p.mul_out; ++w) {
            const size_t slot = (size_t)p.w_off + (size_t)u * p.mul_out + w;
            // per 16-w tile: chains of whole 4-step pieces [piece][lane = 16*(u%4) + w%16][4], others [lane][step] (k_conv.hip, nc_lane_off)
            const size_t lane = (size_t)(u % 4) * 16 + (size_t)(w % 16), j = (size_t)(u / 4);
            const size_t at = (size_t)(w / 16) * 64 * steps +
                              (steps % 4 == 0 ? (j / 4) * 256 + lane * 4 + (j % 4) : lane * steps + j);
            pack[(size_t)k * KS + wk_off[pi] + at] = k < H ? w2.data[slot * H + k] : b2.data[slot];
          }
    }
    L.wpack.push_back(m.wpool.upload(pack));
  }
  if (L.yform) {
    // work list of the node contraction: (output block, 16-wide w tile) units, heaviest first so the 4 waves balance
    int n_units = 0;   // (output block, 16-wide w tile) units so far
    std::vector<FGran> fg;
    // Packed granules exist only in the static-shape kernel variants: a layer with ANY generic granule (chain shapes outside the
    // static set, H % 16 != 0) runs the predicated variant, which walks classic 4-slot granules only.  So the list is built with
    // packing first and, when a generic granule turns up next to a packed one, once more without.
    for (int attempt = 0; attempt < 2; ++attempt) {
    const bool allow_pack = m.fused_pack && attempt == 0;
    bool any_packed = false;
    n_units = 0;
    fg.clear();
    L.fgran_unit.clear();
    for (int ob = 0; ob < (int)obs.size(); ++ob) {
      const ObInfo& O = obs[ob];
      for (int w0 = 0; w0 < O.mul; w0 += 16) {
        NcUnit U{};
        U.col_base = O.base; U.itemw = O.itemw; U.w0 = w0; U.n_w = std::min(16, O.mul - w0);
        int slot_g[NC_MAXITEM];
        for (int i = 0; i < NC_MAXITEM; ++i) slot_g[i] = -1;
        for (size_t pi = 0; pi < L.table.paths.size(); ++pi) {
          const TPPath& p = L.table.paths[pi];
          if (p.out_block != ob) continue;
          for (int i = 0; i < p.din; ++i) {
            U.slot[slot_base[pi] + i] = {p.i_off, p.din, i, p.mul_in, (int)round_up(p.mul_in, 4),
                                         (int)round_up(p.mul_out, 16), wk_off[pi]};
            slot_g[slot_base[pi] + i] = dp[pi].g_off + i * p.dout;
          }
        }
        // fused form, packed granule: an output block of <= 10 channels fed by one 12-step chain (a scalar path) and / or
        // groups of three 3-step chains (the components of a vector path): every slot of the block in ONE granule
        // (kernels.h, FGran).  Anything else: classic granules, one per quad of item columns.
        bool packed = false;
        if (allow_pack && L.maxd <= 3 && m.cfg.sh_lmax <= 1 && L.H % 16 == 0 && O.mul > 8 && O.mul <= 10 && O.dout == 3) {
          auto chain = [&](int t) { return U.slot[t].din == 0 ? 0 : U.slot[t].u_pad / 4; };
          int c12 = -1, groups[2] = {-1, -1}, ng = 0;
          bool ok = true;
          for (int t = 0; t < O.itemw && ok; ) {
            const NcSlot& S = U.slot[t];
            if (S.din == 0) { ++t; continue; }
            if (S.din == 1 && chain(t) == 12 && c12 < 0) { c12 = t; ++t; continue; }
            if (S.din == 3 && S.comp == 0 && chain(t) == 3 && t + 2 < O.itemw && ng < 2 && U.slot[t + 1].din == 3 &&
                U.slot[t + 2].din == 3 && U.slot[t + 1].comp == 1 && U.slot[t + 2].comp == 2 &&
                U.slot[t + 1].wk_off == S.wk_off && U.slot[t + 2].wk_off == S.wk_off && U.slot[t + 1].x_off == S.x_off &&
                U.slot[t + 2].x_off == S.x_off) { groups[ng++] = t; t += 3; continue; }
            ok = false;
          }
          const int shape = !ok ? 0 : (c12 >= 0 && ng == 2) ? 4 : (c12 < 0 && ng == 2) ? 5 : (c12 >= 0 && ng == 1) ? 6 : 0;
          if (shape) {
            FGran G{};
            G.w0 = 0; G.n_w = U.n_w; G.o_off = O.o_off; G.dout = O.dout; G.shape = shape;
            int ns_ = 0;
            auto put = [&](int t) { G.slot[ns_] = U.slot[t]; G.g[ns_] = slot_g[t]; ++ns_; };
            if (c12 >= 0) put(c12);
            for (int gq = 0; gq < ng; ++gq) for (int i = 0; i < 3; ++i) put(groups[gq] + i);
            for (int t = ns_; t < FC_MAXSLOT; ++t) G.g[t] = -1;
            G.nslot = ns_; G.nb = (ns_ + 1) / 2 + 1;
            fg.push_back(G);
            L.fgran_unit.push_back(n_units);
            packed = any_packed = true;
          }
        }
        for (int q = 0; q < O.itemw / 4 && !packed; ++q) {
          FGran G{};
          G.w0 = w0; G.n_w = U.n_w; G.o_off = O.o_off; G.dout = O.dout; G.accumulate = q > 0; G.empty = 1;
          G.nslot = 4; G.nb = 4;
          for (int sl = 0; sl < FC_MAXSLOT; ++sl) G.g[sl] = -1;
          for (int sl = 0; sl < 4; ++sl) {
            G.slot[sl] = U.slot[4 * q + sl];
            G.g[sl] = slot_g[4 * q + sl];
            if (G.slot[sl].din != 0) G.empty = 0;
          }
          if (G.empty && q > 0) continue;   // padding quad of a wider item: contributes nothing
          // longest chain first: k_conv_fused prefetches 12 weight fragments for slot 0 and 4 for the others
          int orig[4] = {0, 1, 2, 3};
          for (int i = 1; i < 4; ++i)
            for (int j = i; j > 0; --j) {
              auto steps = [&](int t) { return G.slot[t].din == 0 ? 0 : G.slot[t].u_pad; };
              if (steps(j) > steps(j - 1)) {
                std::swap(G.slot[j], G.slot[j - 1]); std::swap(G.g[j], G.g[j - 1]); std::swap(orig[j], orig[j - 1]);
              }
            }
          (void)orig;
          {
            auto st = [&](int t) { return G.slot[t].din == 0 ? 0 : G.slot[t].u_pad / 4; };
            auto fits = [&](int t, int n) { return st(t) == n || st(t) == 0; };
            G.shape = 0;
            if (st(0) == 12 && st(1) == 0 && st(2) == 0 && st(3) == 0) G.shape = 3;
            else if (st(0) == 12 && fits(1, 3) && fits(2, 3) && fits(3, 3)) G.shape = 1;
            else if (st(0) == 3 && fits(1, 3) && fits(2, 3) && fits(3, 3)) G.shape = 2;
          }
          {   // weight sharing between the slots (components of one path): fewer weight requests per chunk in the fused kernel
            auto same = [&](int a_, int b_) { return G.slot[b_].din == 0 || (G.slot[a_].din != 0 && G.slot[a_].wk_off == G.slot[b_].wk_off &&
                                                                                G.slot[a_].u_pad == G.slot[b_].u_pad && G.slot[a_].w_pad == G.slot[b_].w_pad); };
            G.nlive = 0;
            for (int t = 1; t < 4; ++t) if (G.slot[t].din != 0) G.nlive = t;   // pads trail (sorted by chain length)
            G.dup = 0;
            if (G.shape == 1 && G.slot[1].din != 0 && same(1, 2) && same(1, 3)) G.dup = 1;
            if (G.shape == 2 && same(0, 1) && same(0, 2)) G.dup = same(0, 3) ? 3 : 2;
          }
          fg.push_back(G);
          L.fgran_unit.push_back(n_units);
        }
        ++n_units;
      }
    }
    bool generic_now = L.H % 16 != 0;
    for (auto& G : fg) generic_now = generic_now || (!G.empty && G.shape == 0);
    if (!(generic_now && any_packed)) break;
    }
    // Merged granule (shape 7): a 48-channel scalar output block fed by ONE 12-step chain (the first interaction layer: 0e x 0e ->
    // 0e) gives three light granules -- 12 contraction + 8 edge MFMAs per chunk step, too short to cover the request latency of
    // the next chunk.  Their x fragments and hidden rows are the same, so the three 16-channel tiles become the three slots of one
    // granule (36 + 24 MFMAs per step, one hidden-row pass, one epilogue); the slots feed DIFFERENT output channels (16*slot + w).
    bool any_generic = false;   // (a generic granule sends the whole layer to the compiler-scheduled kernel variant, which has no merged form)
    for (auto& G : fg) any_generic = any_generic || (!G.empty && G.shape == 0);
    if (m.fused_tri && !any_generic && L.maxd <= 3 && m.cfg.sh_lmax <= 1 && L.H % 16 == 0) {
      for (size_t i = 0; i + 2 < fg.size(); ++i) {
        auto light = [&](const FGran& G, int w0) {
          return G.shape == 3 && !G.empty && !G.accumulate && G.dout == 1 && G.n_w == 16 && G.w0 == w0 && G.nslot == 4 &&
                 G.o_off == fg[i].o_off && G.slot[0].wk_off == fg[i].slot[0].wk_off && G.slot[0].x_off == fg[i].slot[0].x_off &&
                 G.slot[0].u_pad == 48 && G.g[0] == fg[i].g[0];
        };
        if (!(light(fg[i], 0) && light(fg[i + 1], 16) && light(fg[i + 2], 32))) continue;
        FGran G = fg[i];
        G.shape = 7; G.n_w = 48; G.nslot = 3; G.nb = 3; G.nlive = 2; G.dup = 0;
        G.slot[1] = G.slot[2] = G.slot[0];
        G.g[1] = G.g[2] = G.g[0];
        G.slot[3] = NcSlot{}; G.g[3] = -1;
        fg[i] = G;
        fg.erase(fg.begin() + i + 1, fg.begin() + i + 3);
        L.fgran_unit.erase(L.fgran_unit.begin() + i + 1, L.fgran_unit.begin() + i + 3);
      }
    }
    L.fgran_generic = false;
    for (auto& G : fg) if (!G.empty && G.shape == 0) L.fgran_generic = true;
    if (L.H % 16 != 0) L.fgran_generic = true;   // the static loops walk whole pairs of 8-k groups
    if (L.fgran_generic)
      for (auto& G : fg)
        if (G.shape >= 4) throw Error(DDMI_ERR_ARG, "internal: packed / merged granule in a generic layer (" + L.name + ")");
    L.fgran = m.wpool.upload(fg);
    L.n_fgran = (int)fg.size();
    L.max_nb = 4;
    for (auto& G : fg) L.max_nb = std::max(L.max_nb, G.nb);
    if (m.cfg.exec.debug & 1) {   // granule list of the layer (shape / slots / column blocks), for tests and debugging
      fprintf(stderr, "ddmi granules %s:", L.name.c_str());
      for (auto& G : fg) fprintf(stderr, " [shape %d slots %d nb %d w %d%s]", G.shape, G.nslot, G.nb, G.n_w, G.accumulate ? " acc" : "");
      fprintf(stderr, "\n");
    }
    {   // dense coupling rows of every granule, in the (MAXD, SHD) shape of the kernel instantiation that runs this layer
        // (k_conv.hip, launch_conv_fused): cgt[g][slot][k'][j] = C_path(slot)[comp(slot)][j - s_off][k'], 0 outside the path's sh block
      const int MD = L.maxd <= 3 ? 3 : 5, SD = (L.maxd <= 3 && m.cfg.sh_lmax <= 1) ? 4 : 9;
      std::vector<float> cgt(fg.size() * (size_t)FC_MAXSLOT * MD * SD, 0.f);
      for (size_t gi = 0; gi < fg.size(); ++gi)
        for (int sl = 0; sl < FC_MAXSLOT; ++sl)
          for (int k = 0; k < MD; ++k) {
            const FGran& G = fg[gi];
            if (G.g[sl] < 0 || k >= G.dout) continue;
            const GEntry& E = gmap[G.g[sl] + k];
            for (int j = E.s_off; j < E.s_off + E.ds && j < SD; ++j)
              cgt[((gi * FC_MAXSLOT + sl) * MD + k) * SD + j] = ctab[E.c_idx + (j - E.s_off) * E.dout];
          }
      L.cgt = m.wpool.upload(cgt);
    }
    L.HKq = (int)round_up(L.H, 8);   // hidden width padded to the 8-k groups of the fused kernel
  }
  // batch norm (e3nn.nn.BatchNorm eval, eps 1e-5): per-column mean / scale / bias ---------
  if (L.has_bn) {
    const std::string n = L.name + ".batch_norm";
    const HostTensor &rm = W(m, n + ".running_mean"), &rv = W(m, n + ".running_var"), &w = W(m, n + ".weight"),
                     &b = W(m, n + ".bias");
    std::vector<float> mean(L.D_out, 0.f), scale(L.D_out, 1.f), bias(L.D_out, 0.f);
    int iv = 0, im = 0;
    for (auto& blk : L.out_irr) {
      for (int u = 0; u < blk.mul; ++u) {
        const float sc = w.data[iv + u] / std::sqrt(rv.data[iv + u] + 1e-5f);
        for (int k = 0; k < blk.d(); ++k) {
          const int col = blk.off + u * blk.d() + k;
          scale[col] = sc;
          if (blk.l == 0 && blk.p == 1) { mean[col] = rm.data[im + u]; bias[col] = b.data[im + u]; }
        }
      }
      iv += blk.mul;
      if (blk.l == 0 && blk.p == 1) im += blk.mul;
    }
    L.bn_mean = m.wpool.upload(mean); L.bn_scale = m.wpool.upload(scale); L.bn_bias = m.wpool.upload(bias);
  }
}

// score read-outs of both class families: centre convolution, translation / rotation heads, torsion convolution + head
static void commit_readouts(Model& m) {
  const ddmi_config& c = m.cfg;
  m.center_edge = up_mlp(m, "center_edge_embedding");
  m.tr_final = up_mlp(m, "tr_final_layer");
  m.rot_final = up_mlp(m, "rot_final_layer");
  commit_conv(m, m.final_conv);
  if (!c.no_torsion) {
    m.final_edge = up_mlp(m, "final_edge_embedding");
    commit_conv(m, m.tor_conv);
    m.tor_W0 = up(m, "tor_final_layer.0.weight");
    m.tor_W3 = up(m, "tor_final_layer.3.weight");
    Irreps tsh;
    std::vector<double> T = full_tp_dense(sh_irreps(c.sh_lmax), make_irreps({{1, 2, 1}}), &tsh);
    std::vector<float> Tf(T.begin(), T.end());
    m.tor_T = m.wpool.upload(Tf);
    m.tor_ds = (c.sh_lmax + 1) * (c.sh_lmax + 1);
    m.tor_dts = irreps_dim(tsh);
  }
}

void commit_weights(Model& m) {
  for (auto& kv : m.spec) {
    auto it = m.host_w.find(kv.first);
    DDMI_REQUIRE(it != m.host_w.end(), DDMI_ERR_KEY, "missing state_dict key: " + kv.first);
    DDMI_REQUIRE(it->second.shape == kv.second, DDMI_ERR_KEY, "shape mismatch for " + kv.first);
  }
  m.wpool.release();
  const ddmi_config& c = m.cfg;
  // ligand atom encoder: concatenated embedding tables
  {
    std::vector<float> emb;
    std::vector<int> off;
    int rows = 0;
    for (int i = 0; i < 16; ++i) {
      off.push_back(rows);
      const HostTensor& t = W(m, "lig_node_embedding.atom_embedding_list." + std::to_string(i) + ".weight");
      emb.insert(emb.end(), t.data.begin(), t.data.end());
      rows += LIG_DIMS[i];
    }
    m.lig_emb = m.wpool.upload(emb);
    m.lig_emb_off = m.wpool.upload(off);
    if (!c.old_model) {
      m.lig_enc.W0 = up(m, "lig_node_embedding.additional_features_embedder.weight");
      m.lig_enc.b0 = up(m, "lig_node_embedding.additional_features_embedder.bias");
    }
  }
  m.rec_emb = up(m, "rec_node_embedding.atom_embedding_list.0.weight");
  if (c.old_model) {
    m.old_lig_lin.W0 = up(m, "lig_node_embedding.linear.weight"); m.old_lig_lin.b0 = up(m, "lig_node_embedding.linear.bias");
    m.old_rec_lin.W0 = up(m, "rec_node_embedding.linear.weight"); m.old_rec_lin.b0 = up(m, "rec_node_embedding.linear.bias");
    if (m.lm > 0) { m.old_lm_W = up(m, "rec_node_embedding.lm_embedding_layer.weight"); m.old_lm_b = up(m, "rec_node_embedding.lm_embedding_layer.bias"); }
    m.lig_edge = up_mlp(m, "lig_edge_embedding");
    m.rec_edge = up_mlp(m, "rec_edge_embedding");
    m.cross_edge = up_mlp(m, "cross_edge_embedding");
    if (c.confidence_mode) {
      for (int i = 0; i < 3; ++i) {
        m.conf_W[i] = up(m, "confidence_predictor." + std::to_string(4 * i) + ".weight");
        m.conf_b[i] = up(m, "confidence_predictor." + std::to_string(4 * i) + ".bias");
      }
      for (int i = 0; i < 2; ++i) {
        const std::string n = "confidence_predictor." + std::to_string(4 * i + 1);
        const HostTensor &w = W(m, n + ".weight"), &b = W(m, n + ".bias"), &rm = W(m, n + ".running_mean"), &rv = W(m, n + ".running_var");
        std::vector<float> sc(m.ns), sh(m.ns);
        for (int k = 0; k < m.ns; ++k) { sc[k] = w.data[k] / std::sqrt(rv.data[k] + 1e-5f); sh[k] = b.data[k] - rm.data[k] * sc[k]; }
        m.conf_bn_scale[i] = m.wpool.upload(sc); m.conf_bn_shift[i] = m.wpool.upload(sh);
      }
    }
    auto offs_old = [&](const std::string& k, float*& dev, float& coeff) {
      const HostTensor& t = W(m, k);
      DDMI_REQUIRE(t.data.size() >= 2, DDMI_ERR_ARG, "distance expansion needs >= 2 gaussians");
      dev = m.wpool.upload(t.data);
      const float d = t.data[1] - t.data[0];
      coeff = (float)(-0.5 / ((double)d * (double)d));
    };
    offs_old("lig_distance_expansion.offset", m.off_lig, m.coeff_lig);
    offs_old("rec_distance_expansion.offset", m.off_rec, m.coeff_rec);
    offs_old("cross_distance_expansion.offset", m.off_cross, m.coeff_cross);
    for (auto* fam : {&m.old_lig, &m.old_rec, &m.old_l2r, &m.old_r2l})
      for (auto& L : *fam) commit_conv(m, L);
    if (!c.confidence_mode) {
      offs_old("center_distance_expansion.offset", m.off_center, m.coeff_center);
      commit_readouts(m);
    }
    const int half = m.sd / 2;
    if ((int)m.time_freq_host.size() != half) {
      m.time_freq_host.resize(half);
      const double e = std::log(10000.0) / (half - 1);
      for (int k = 0; k < half; ++k) m.time_freq_host[k] = std::exp((float)((float)k * (float)(-e)));
    }
    m.time_freq = m.wpool.upload(m.time_freq_host);
    m.committed = true;
    return;
  }
  if (m.lm > 0) {
    m.rec_enc_W = up(m, "rec_node_embedding.additional_features_embedder.weight");
    m.rec_enc_b = up(m, "rec_node_embedding.additional_features_embedder.bias");
  }
  m.lig_edge = up_mlp(m, "lig_edge_embedding");
  m.rec_edge = up_mlp(m, "rec_edge_embedding");
  m.rec_sigma = up_mlp(m, "rec_sigma_embedding");
  m.cross_edge = up_mlp(m, c.all_atoms ? "lr_edge_embedding" : "cross_edge_embedding");
  if (c.all_atoms) {
    std::vector<float> emb;
    std::vector<int> off;
    int rows = 0;
    for (int i = 0; i < 4; ++i) {
      off.push_back(rows);
      const HostTensor& t = W(m, "atom_node_embedding.atom_embedding_list." + std::to_string(i) + ".weight");
      emb.insert(emb.end(), t.data.begin(), t.data.end());
      rows += ATOM_DIMS[i];
    }
    m.atom_emb = m.wpool.upload(emb);
    m.atom_emb_off = m.wpool.upload(off);
    m.atom_edge = up_mlp(m, "atom_edge_embedding");
    m.ar_edge = up_mlp(m, "ar_edge_embedding");
    m.la_edge = up_mlp(m, "la_edge_embedding");
  }
  if (c.confidence_mode) {
    for (int i = 0; i < 3; ++i) {
      m.conf_W[i] = up(m, "confidence_predictor." + std::to_string(4 * i) + ".weight");
      m.conf_b[i] = up(m, "confidence_predictor." + std::to_string(4 * i) + ".bias");
    }
    for (int i = 0; i < 2; ++i) {   // BatchNorm1d eval folded to scale / shift
      const std::string n = "confidence_predictor." + std::to_string(4 * i + 1);
      const HostTensor &w = W(m, n + ".weight"), &b = W(m, n + ".bias"), &rm = W(m, n + ".running_mean"), &rv = W(m, n + ".running_var");
      std::vector<float> sc(m.ns), sh(m.ns);
      for (int k = 0; k < m.ns; ++k) {
        sc[k] = w.data[k] / std::sqrt(rv.data[k] + 1e-5f);
        sh[k] = b.data[k] - rm.data[k] * sc[k];
      }
      m.conf_bn_scale[i] = m.wpool.upload(sc); m.conf_bn_shift[i] = m.wpool.upload(sh);
    }
    if (c.atom_confidence) {   // Linear + BatchNorm1d(eval) folded: W' = diag(s) W, b' = s b + shift
      for (int i = 0; i < 3; ++i) {
        const std::string n = "atom_confidence_predictor." + std::to_string(4 * i);
        const HostTensor &w = W(m, n + ".weight"), &b = W(m, n + ".bias");
        std::vector<float> wf = w.data, bf = b.data;
        if (i < 2) {
          const std::string nb = "atom_confidence_predictor." + std::to_string(4 * i + 1);
          const HostTensor &g = W(m, nb + ".weight"), &be = W(m, nb + ".bias"), &rm = W(m, nb + ".running_mean"), &rv = W(m, nb + ".running_var");
          const int cols = (int)w.shape[1];
          for (int r = 0; r < m.ns; ++r) {
            const float sc = g.data[r] / std::sqrt(rv.data[r] + 1e-5f);
            for (int k = 0; k < cols; ++k) wf[(size_t)r * cols + k] *= sc;
            bf[r] = bf[r] * sc + (be.data[r] - rm.data[r] * sc);
          }
        }
        m.aconf_W[i] = m.wpool.upload(wf); m.aconf_b[i] = m.wpool.upload(bf);
      }
    }
  }
  auto offs = [&](const std::string& k, float*& dev, float& coeff) {
    const HostTensor& t = W(m, k);
    DDMI_REQUIRE(t.data.size() >= 2, DDMI_ERR_ARG, "distance expansion needs >= 2 gaussians");
    dev = m.wpool.upload(t.data);
    const float d = t.data[1] - t.data[0];             // GaussianSmearing.coeff, models/layers.py:25
    coeff = (float)(-0.5 / ((double)d * (double)d));
  };
  offs("lig_distance_expansion.offset", m.off_lig, m.coeff_lig);
  offs("rec_distance_expansion.offset", m.off_rec, m.coeff_rec);
  offs("cross_distance_expansion.offset", m.off_cross, m.coeff_cross);
  if (!c.confidence_mode) offs("center_distance_expansion.offset", m.off_center, m.coeff_center);
  for (auto& L : m.rec_emb_layers) commit_conv(m, L);
  for (auto& L : m.lig_emb_layers) commit_conv(m, L);
  for (auto& L : m.conv_layers) commit_conv(m, L);
  if (!c.confidence_mode) commit_readouts(m);
  if (c.sidechain_pred && !c.confidence_mode) {
    // e3nn o3.Linear (e3nn/o3/_linear.py): weight slots [mul_in, mul_out] row-major in the order (i_in, i_out) of the matching
    // irreps; every slot into an output block is scaled by (sum of mul_in over the slots into that block) ** -0.5.  The read-out
    // sums the even half (4x0e + 2x1e, 10 columns) and the odd half (4x0o + 2x1o) of the 20 outputs (cg_model.py:402), so the whole
    // predictor is ONE dense matrix Mt[10][row width] over a node row: column of (w, m) in a half = (l == 0 ? w : 4 + 3 w + m).
    const Irreps in = layer_irreps(c, c.num_prot_emb_layers + c.num_conv_layers);
    const int K = irreps_dim(in);
    const HostTensor& w = W(m, "sidechain_predictor.weight");
    struct OB { int mul, l, p; };
    const OB outs[4] = {{4, 0, 1}, {2, 1, 1}, {4, 0, -1}, {2, 1, -1}};
    std::vector<float> Mt((size_t)10 * K, 0.f);
    int fan[4] = {0, 0, 0, 0};
    for (auto& b : in)
      for (int o = 0; o < 4; ++o) if (b.l == outs[o].l && b.p == outs[o].p) fan[o] += b.mul;
    size_t off = 0;
    for (auto& b : in)
      for (int o = 0; o < 4; ++o) {
        if (b.l != outs[o].l || b.p != outs[o].p) continue;
        const float sc = 1.f / std::sqrt((float)fan[o]);
        const int d = b.d();
        for (int u = 0; u < b.mul; ++u)
          for (int ww = 0; ww < outs[o].mul; ++ww) {
            const float v = w.data[off + (size_t)u * outs[o].mul + ww] * sc;
            for (int mm = 0; mm < d; ++mm) Mt[(size_t)(b.l == 0 ? ww : 4 + 3 * ww + mm) * K + b.off + u * d + mm] += v;
          }
        off += (size_t)b.mul * outs[o].mul;
      }
    DDMI_REQUIRE(off == w.data.size(), DDMI_ERR_KEY, "sidechain_predictor.weight: unexpected size");
    m.side_Mt = m.wpool.upload(Mt);
    m.side_K = K;
  }
  // sinusoidal embedding frequencies (utils/diffusion_utils.py:101-103) unless supplied by the caller; 'fourier': the frozen W
  const int half = m.sd / 2;
  if (c.embedding_type == 1) {
    m.time_freq = up(m, "timestep_emb_func.W");
    m.committed = true;
    return;
  }
  if ((int)m.time_freq_host.size() != half) {
    m.time_freq_host.resize(half);
    const double e = std::log(10000.0) / (half - 1);
    for (int k = 0; k < half; ++k) m.time_freq_host[k] = std::exp((float)((float)k * (float)(-e)));
  }
  m.time_freq = m.wpool.upload(m.time_freq_host);
  m.committed = true;
}

}  // namespace ddmi
