"""Weight ABI of the CG score model = the reference `state_dict` key names and shapes
(derived from models/cg_model.py:85-255, models/layers.py:10-67,
models/tensor_layers.py:298-307).  There are no checkpoints in this environment
(the reference downloads them, inference.py:124-143), so benchmarks and tests use
random-initialised weights of the declared architecture.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Tuple

import torch

from .config import ModelConfig, LIG_FEATURE_DIMS, LM_EMBEDDING_DIM, REC_RESIDUE_FEATURE_DIMS, REC_ATOM_FEATURE_DIMS
from .irreps import parse_irreps, irreps_num, sh_irreps, full_tp_irreps, tp_weight_numel, depthwise_numels


def tor_sh_irreps(cfg: ModelConfig) -> str:
    return full_tp_irreps(sh_irreps(cfg.sh_lmax), "1x2e")


def final_conv_out(cfg: ModelConfig) -> str:
    return "2x1o + 2x1e" if not cfg.odd_parity else "1x1o + 1x1e"


def tor_conv_out(cfg: ModelConfig) -> str:
    return f"{cfg.ns}x0o + {cfg.ns}x0e" if not cfg.odd_parity else f"{cfg.ns}x0o"


def state_dict_spec(cfg: ModelConfig) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """key -> (shape, kind), kind in {linear_w, linear_b, emb, offset, bn_mean, bn_var,
    bn_w, bn_b}.  Only tensors that carry model state are listed (e3nn-internal
    buffers under *.tp.* are not part of the ABI and are ignored by loaders)."""
    ns, sd = cfg.ns, cfg.sigma_embed_dim
    spec: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()

    def lin(name, fin, fout, bias=True):
        spec[f"{name}.weight"] = ((fout, fin), "linear_w")
        if bias:
            spec[f"{name}.bias"] = ((fout,), "linear_b")

    def mlp(name, fin, hid, fout):  # Sequential(Linear, ReLU, Dropout, Linear)
        lin(f"{name}.0", fin, hid)
        lin(f"{name}.3", hid, fout)

    def encoder(name, dims, extra):
        for i, d in enumerate(dims):
            spec[f"{name}.atom_embedding_list.{i}.weight"] = ((d, ns), "emb")
        if extra > 0:
            lin(f"{name}.additional_features_embedder", extra + ns, ns)

    def bn(name, irreps):
        n0 = sum(b.mul for b in parse_irreps(irreps) if b.l == 0 and b.p == 1)
        nf = irreps_num(irreps)
        spec[f"{name}.running_mean"] = ((n0,), "bn_mean")
        spec[f"{name}.running_var"] = ((nf,), "bn_var")
        spec[f"{name}.weight"] = ((nf,), "bn_w")
        spec[f"{name}.bias"] = ((n0,), "bn_b")

    def conv(name, in_irr, sh, out_irr, n_edge, hidden, groups, faster, deep=False):
        W = tp_weight_numel(in_irr, sh, out_irr, faster)
        dw = deep and cfg.depthwise_convolution      # embedding / interaction layers (cg_model.py:124,147,168), not the read-out convolutions
        if dw:   # 'uvu' TensorProduct weights per edge + the shared linear_2 (models/tensor_layers.py:248-290)
            W, n_lin = depthwise_numels(in_irr, sh, out_irr)
            spec[f"{name}.linear_2.weight"] = ((n_lin,), "o3lin")
        for g in range(groups):
            pre = f"{name}.fc" if groups == 1 else f"{name}.fc.{g}"
            tl = cfg.tp_weights_layers if deep else 2   # FCBlock(..., tp_weights_layers, ...) (models/layers.py:10-17): keys 0, 3, .. 3(tl-1)
            lin(f"{pre}.0", n_edge, hidden)
            for j in range(1, tl - 1):
                lin(f"{pre}.{3 * j}", hidden, hidden)
            lin(f"{pre}.{3 * (tl - 1)}", hidden, W)
        if cfg.batch_norm:
            bn(f"{name}.batch_norm", out_irr)

    if cfg.old:
        return _old_spec(cfg, spec, lin, mlp, bn, conv)
    sh = sh_irreps(cfg.sh_lmax)
    encoder("lig_node_embedding", LIG_FEATURE_DIMS, sd)
    mlp("lig_edge_embedding", cfg.in_lig_edge_features + sd + cfg.distance_embed_dim, ns, ns)
    if cfg.all_atoms:   # models/aa_model.py:90-103
        mlp("rec_sigma_embedding", sd, ns, ns)
        encoder("rec_node_embedding", REC_RESIDUE_FEATURE_DIMS, cfg.lm_embedding_dim)
        mlp("rec_edge_embedding", cfg.distance_embed_dim, ns, ns)
        encoder("atom_node_embedding", REC_ATOM_FEATURE_DIMS, 0)
        mlp("atom_edge_embedding", cfg.distance_embed_dim, ns, ns)
        mlp("lr_edge_embedding", sd + cfg.cross_distance_embed_dim, ns, ns)
        mlp("ar_edge_embedding", cfg.distance_embed_dim, ns, ns)
        mlp("la_edge_embedding", sd + cfg.cross_distance_embed_dim, ns, ns)
    else:
        encoder("rec_node_embedding", REC_RESIDUE_FEATURE_DIMS, cfg.lm_embedding_dim)
        mlp("rec_edge_embedding", cfg.distance_embed_dim, ns, ns)
        mlp("rec_sigma_embedding", sd, ns, ns)
        mlp("cross_edge_embedding", sd + cfg.cross_distance_embed_dim, ns, ns)
    spec["lig_distance_expansion.offset"] = ((cfg.distance_embed_dim,), "offset:lig")
    spec["rec_distance_expansion.offset"] = ((cfg.distance_embed_dim,), "offset:rec")
    spec["cross_distance_expansion.offset"] = ((cfg.cross_distance_embed_dim,), "offset:cross")
    if cfg.embedding_type == "fourier":   # GaussianFourierProjection.W (utils/diffusion_utils.py:118-120)
        spec["timestep_emb_func.W"] = ((sd // 2,), "fourier_w")
    K, L = cfg.num_prot_emb_layers, cfg.num_conv_layers
    for i in range(K):
        a, b = cfg.layer_irreps(i)
        conv(f"rec_emb_layers.{i}", a, sh, b, 3 * ns, 3 * ns,
             (4 if cfg.differentiate_convolutions else 1) if cfg.all_atoms else 1, cfg.faster, deep=True)
    if cfg.embed_also_ligand:
        for i in range(K):
            a, b = cfg.layer_irreps(i)
            conv(f"lig_emb_layers.{i}", a, sh, b, 3 * ns, 3 * ns, 1, cfg.faster, deep=True)
    for l in range(L):
        a, b = cfg.layer_irreps(K + l)
        conv(f"conv_layers.{l}", a, sh, b, 3 * ns, 3 * ns, cfg.conv_groups(l), cfg.faster, deep=True)
    last_out = cfg.layer_irreps(K + L - 1)[1]
    if cfg.confidence_mode:   # cg_model.py:181-207: Linear, BatchNorm1d, ReLU, Dropout, Linear, BatchNorm1d, ReLU, Dropout, Linear
        n_in = ns + (cfg.nv if cfg.reduce_pseudoscalars else ns) if K + L >= 3 else ns

        def predictor(name, n_in_, n_out):   # get_model never passes confidence_no_batchnorm: BatchNorm1d is always there
            lin(f"{name}.0", n_in_, ns)
            lin(f"{name}.4", ns, ns)
            lin(f"{name}.8", ns, n_out)
            for i in (1, 5):
                spec[f"{name}.{i}.weight"] = ((ns,), "bn_w")
                spec[f"{name}.{i}.bias"] = ((ns,), "bn_b")
                spec[f"{name}.{i}.running_mean"] = ((ns,), "bn_mean")
                spec[f"{name}.{i}.running_var"] = ((ns,), "bn_var")
        if cfg.atom_confidence:   # cg_model.py:184-196: per-atom predictor, its last ns outputs feed the graph mean
            predictor("atom_confidence_predictor", n_in, cfg.atom_num_confidence_outputs + ns)
            n_in = ns
        predictor("confidence_predictor", n_in, cfg.num_confidence_outputs + (1 if cfg.affinity_prediction else 0))
        return spec
    if cfg.sidechain_pred:   # o3.Linear(last_out -> 4x0e + 2x1e + 4x0o + 2x1o): one flat `weight` (models/cg_model.py:173-178)
        tgt = {(0, 1): 4, (1, 1): 2, (0, -1): 4, (1, -1): 2}
        spec["sidechain_predictor.weight"] = ((sum(x.mul * tgt.get((x.l, x.p), 0) for x in parse_irreps(last_out)),), "o3lin")
    _readout_spec(cfg, spec, lin, mlp, conv, last_out, sh)
    return spec


def _readout_spec(cfg, spec, lin, mlp, conv, last_out, sh):
    """Score read-outs, the same modules in both class families (cg_model.py:209-255, old_cg_model.py:156-200)."""
    ns, sd = cfg.ns, cfg.sigma_embed_dim
    spec["center_distance_expansion.offset"] = ((cfg.distance_embed_dim,), "offset:center")
    mlp("center_edge_embedding", cfg.distance_embed_dim + sd, ns, ns)
    conv("final_conv", last_out, sh, final_conv_out(cfg), 2 * ns, 2 * ns, 1, False)
    mlp("tr_final_layer", 1 + sd, ns, 1)     # Sequential(Linear, Dropout, ReLU, Linear)
    mlp("rot_final_layer", 1 + sd, ns, 1)
    if not cfg.no_torsion:
        mlp("final_edge_embedding", cfg.distance_embed_dim, ns, ns)
        conv("tor_bond_conv", last_out, tor_sh_irreps(cfg), tor_conv_out(cfg), 3 * ns, 3 * ns, 1, False)
        lin("tor_final_layer.0", 2 * ns if not cfg.odd_parity else ns, ns, bias=False)
        lin("tor_final_layer.3", ns, 1, bias=False)


def _old_spec(cfg, spec, lin, mlp, bn, conv):
    """models/old_cg_model.py:18-200 (CGOldModel + OldAtomEncoder + OldTensorProductConvLayer), score or confidence mode."""
    assert cfg.use_old_atom_encoder and cfg.sh_lmax == 2 and not cfg.all_atoms, \
        "legacy class: OldAtomEncoder, sh_lmax = 2 (get_model(old=True) passes no sh_lmax), CG graphs"
    ns, sd = cfg.ns, cfg.sigma_embed_dim

    def old_encoder(name, dims, lm):
        for i, d in enumerate(dims):
            spec[f"{name}.atom_embedding_list.{i}.weight"] = ((d, ns), "emb")
        lin(f"{name}.linear", sd, ns)
        if lm:
            lin(f"{name}.lm_embedding_layer", LM_EMBEDDING_DIM + ns, ns)
    old_encoder("lig_node_embedding", LIG_FEATURE_DIMS, False)
    mlp("lig_edge_embedding", cfg.in_lig_edge_features + sd + cfg.distance_embed_dim, ns, ns)
    old_encoder("rec_node_embedding", REC_RESIDUE_FEATURE_DIMS, cfg.lm_embedding_type is not None)
    mlp("rec_edge_embedding", sd + cfg.distance_embed_dim, ns, ns)
    mlp("cross_edge_embedding", sd + cfg.cross_distance_embed_dim, ns, ns)
    spec["lig_distance_expansion.offset"] = ((cfg.distance_embed_dim,), "offset:lig")
    spec["rec_distance_expansion.offset"] = ((cfg.distance_embed_dim,), "offset:rec")
    spec["cross_distance_expansion.offset"] = ((cfg.cross_distance_embed_dim,), "offset:cross")
    sh = sh_irreps(2)
    old_cfg = cfg.replace(reduce_pseudoscalars=False)
    for fam in ("lig_conv_layers", "rec_conv_layers", "lig_to_rec_conv_layers", "rec_to_lig_conv_layers"):
        for l in range(cfg.num_conv_layers):
            a, b = old_cfg.layer_irreps(l)
            W = tp_weight_numel(a, sh, b, False)
            lin(f"{fam}.{l}.fc.0", 3 * ns, 3 * ns)
            lin(f"{fam}.{l}.fc.3", 3 * ns, W)
            if cfg.batch_norm:
                bn(f"{fam}.{l}.batch_norm", b)
    if not cfg.confidence_mode:
        _readout_spec(old_cfg, spec, lin, mlp, conv, old_cfg.layer_irreps(cfg.num_conv_layers - 1)[1], sh)
        return spec
    lin("confidence_predictor.0", 2 * ns if cfg.num_conv_layers >= 3 else ns, ns)
    lin("confidence_predictor.4", ns, ns)
    lin("confidence_predictor.8", ns, 2 if cfg.affinity_prediction else 1)
    for i in (1, 5):
        spec[f"confidence_predictor.{i}.weight"] = ((ns,), "bn_w")
        spec[f"confidence_predictor.{i}.bias"] = ((ns,), "bn_b")
        spec[f"confidence_predictor.{i}.running_mean"] = ((ns,), "bn_mean")
        spec[f"confidence_predictor.{i}.running_var"] = ((ns,), "bn_var")
    return spec


def gaussian_offsets(cfg: ModelConfig, which: str) -> torch.Tensor:
    stop = {"lig": cfg.lig_max_radius, "rec": cfg.rec_max_radius, "cross": cfg.cross_max_distance,
            "center": cfg.center_max_distance}[which]
    n = cfg.cross_distance_embed_dim if which == "cross" else cfg.distance_embed_dim
    return torch.linspace(0.0, stop, n)


def init_state_dict(cfg: ModelConfig, seed: int = 1234, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Random weights with torch's default init statistics (Linear: U(+-1/sqrt(fan_in)),
    Embedding: xavier-uniform, models/layers.py:52) and perturbed BatchNorm statistics so
    the normalisation is not an identity."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = OrderedDict()

    def U(shape, a):
        return (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1).mul_(a).to(dtype)

    for key, (shape, kind) in state_dict_spec(cfg).items():
        if kind == "linear_w":
            sd[key] = U(shape, 1.0 / math.sqrt(shape[1]))
        elif kind == "linear_b":
            fan_in = state_dict_spec_fan_in(cfg, key)
            sd[key] = U(shape, 1.0 / math.sqrt(fan_in))
        elif kind == "emb":
            sd[key] = U(shape, math.sqrt(6.0 / (shape[0] + shape[1])))
        elif kind.startswith("offset"):
            sd[key] = gaussian_offsets(cfg, kind.split(":")[1]).to(dtype)
        elif kind == "bn_mean":
            sd[key] = U(shape, 0.2)
        elif kind == "bn_var":
            sd[key] = (U(shape, 0.5) + 1.0)
        elif kind == "bn_w":
            sd[key] = (U(shape, 0.5) + 1.0)
        elif kind == "bn_b":
            sd[key] = U(shape, 0.2)
        elif kind == "o3lin":         # e3nn o3.Linear internal weights: torch.randn(weight_numel)
            sd[key] = torch.randn(shape, generator=g, dtype=torch.float64).to(dtype)
        elif kind == "fourier_w":     # torch.randn(embedding_size // 2) * scale
            sd[key] = (torch.randn(shape, generator=g, dtype=torch.float64) * cfg.embedding_scale).to(dtype)
        else:
            raise KeyError(kind)
    return sd


_FAN_CACHE: Dict[int, Dict[str, int]] = {}


def state_dict_spec_fan_in(cfg: ModelConfig, bias_key: str) -> int:
    spec = state_dict_spec(cfg)
    return spec[bias_key[:-len("bias")] + "weight"][0][1]


def gaussian_coeff(offset: torch.Tensor) -> float:
    """GaussianSmearing.coeff (models/layers.py:25)."""
    return -0.5 / float(offset[1] - offset[0]) ** 2
