"""Hyper-parameter contract of the score model.

Mirrors the keyword arguments `get_model` passes to `CGModel`
(reference utils/utils.py:234-276, models/cg_model.py:20-31) plus the noise-schedule
bounds `t_to_sigma` / `sampling` read from the args namespace
(utils/diffusion_utils.py:28-32, utils/sampling.py:133-149).

The real DiffDock-L `model_parameters.yml` is fetched at run time by the reference
(inference.py:124-147) and is not in the repository, so benchmarks use the declared
preset DDL_SYNTH below.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict, replace
import argparse

# vocabulary sizes (reference datasets/process_mols.py:59-87)
LIG_FEATURE_DIMS = (119, 4, 12, 12, 8, 10, 6, 6, 2, 8, 2, 2, 2, 2, 2, 2)
REC_RESIDUE_FEATURE_DIMS = (38,)
REC_ATOM_FEATURE_DIMS = (38, 119, 23, 38)   # amino acid, atomic number, atom_type_2, atom_type_3 (process_mols.py:78-83)
LM_EMBEDDING_DIM = 1280  # 'precomputed' ESM2 embeddings (models/cg_model.py:73-74)


@dataclass(frozen=True)
class ModelConfig:
    # architecture (names as in CGModel.__init__)
    ns: int = 16
    nv: int = 4
    num_conv_layers: int = 2
    num_prot_emb_layers: int = 0
    sh_lmax: int = 2
    sigma_embed_dim: int = 32
    distance_embed_dim: int = 32
    cross_distance_embed_dim: int = 32
    in_lig_edge_features: int = 4
    lig_max_radius: float = 5.0
    rec_max_radius: float = 30.0        # never overridden by get_model
    cross_max_distance: float = 80.0
    center_max_distance: float = 30.0   # never overridden by get_model
    dynamic_max_cross: bool = False
    use_second_order_repr: bool = False
    reduce_pseudoscalars: bool = False
    differentiate_convolutions: bool = True
    tp_weights_layers: int = 2
    embed_also_ligand: bool = True
    batch_norm: bool = True
    smooth_edges: bool = False
    odd_parity: bool = False
    no_torsion: bool = False
    scale_by_sigma: bool = True
    fixed_center_conv: bool = False
    lm_embedding_type: str | None = "precomputed"
    embedding_scale: float = 1000.0     # sinusoidal: multiplies t; fourier: std of the frozen projection W (init only)
    embedding_type: str = "sinusoidal"  # get_timestep_embedding (utils/diffusion_utils.py:129-136): 'sinusoidal' | 'fourier'
    # noise schedule (args.* in the reference)
    tr_sigma_min: float = 0.1
    tr_sigma_max: float = 19.0
    rot_sigma_min: float = 0.03
    rot_sigma_max: float = 1.55
    tor_sigma_min: float = 0.0314
    tor_sigma_max: float = 3.14
    crop_beyond: float | None = None
    all_atoms: bool = False             # AAModel (models/aa_model.py): receptor heavy atoms as a third node type
    # get_model(..., confidence_mode=True) (utils/utils.py:172, models/cg_model.py:181-207,353-366): same interaction
    # layers, read-out = confidence_predictor on the graph-mean scalar ligand features; t is used raw (no t_to_sigma)
    confidence_mode: bool = False
    num_confidence_outputs: int = 1     # len(rmsd_classification_cutoff) + 1 when that is a list
    # per-atom predictor in front of the graph mean (atom_confidence_loss_weight > 0; utils/utils.py:273) and the extra
    # affinity output of confidence_predictor (affinity_prediction, parallel = 1)
    atom_confidence: bool = False
    atom_num_confidence_outputs: int = 1
    affinity_prediction: bool = False
    # receptor features multiplied by 0 at the top of forward (cg_model.py:309-310; needs lm_embedding_type None)
    no_aminoacid_identities: bool = False
    # get_model(..., old=True) (utils/utils.py:180-219): the legacy class models/old_cg_model.py -- what the released DiffDock-L
    # confidence checkpoint is (`old_confidence_model: true`).  Score and confidence mode; always sh_lmax = 2, one confidence output.
    old: bool = False
    use_old_atom_encoder: bool = True
    # CGModel.sidechain_predictor (models/cg_model.py:173-178,397-402): o3.Linear on the receptor rows of the last node table ->
    # 4x0e + 2x1e + 4x0o + 2x1o, even and odd halves summed: the 4th element of the forward tuple, [n_rec, 10]
    # (get_model: sidechain_loss_weight > 0 or backbone_loss_weight > 0, utils/utils.py:274-275)
    sidechain_pred: bool = False
    # TensorProductConvLayer(depthwise=True) in the embedding and interaction layers (models/tensor_layers.py:248-290,324-325;
    # cg_model.py:124,147,168): an e3nn 'uvu' TensorProduct (one weight per path and input channel) followed by o3.Linear
    depthwise_convolution: bool = False
    # execution option (ddmi_config.edge_product, not a reference argument): arithmetic of the per-edge product of the
    # interaction layers -- "f32" (exact fp32 chain, default) | "bf16x4" (split-bf16 operands, fp32 accumulation)
    edge_product: str = "f32"
    # kernel-route selection (ddmi_config.exec = ddmi_exec_options, include/ddmi.h): ((field, value), ...); () = every default
    exec_options: tuple = ()

    # ------------------------------------------------------------------ derived
    @property
    def lm_embedding_dim(self) -> int:
        return LM_EMBEDDING_DIM if self.lm_embedding_type == "precomputed" else 0

    @property
    def faster(self) -> bool:
        # models/cg_model.py:121,144,165
        return self.sh_lmax == 1 and not self.use_second_order_repr

    def irrep_seq(self):
        """models/tensor_layers.py:17-32 (get_irrep_seq)."""
        ns, nv = self.ns, self.nv
        last = nv if self.reduce_pseudoscalars else ns
        if self.use_second_order_repr:
            return [f"{ns}x0e",
                    f"{ns}x0e + {nv}x1o + {nv}x2e",
                    f"{ns}x0e + {nv}x1o + {nv}x2e + {nv}x1e + {nv}x2o",
                    f"{ns}x0e + {nv}x1o + {nv}x2e + {nv}x1e + {nv}x2o + {last}x0o"]
        return [f"{ns}x0e",
                f"{ns}x0e + {nv}x1o",
                f"{ns}x0e + {nv}x1o + {nv}x1e",
                f"{ns}x0e + {nv}x1o + {nv}x1e + {last}x0o"]

    def layer_irreps(self, i):
        """(in, out) irreps strings of layer index i in the emb+conv sequence
        (models/cg_model.py:110-111,154-155)."""
        seq = self.irrep_seq()
        return seq[min(i, len(seq) - 1)], seq[min(i + 1, len(seq) - 1)]

    def conv_groups(self, l) -> int:
        """edge groups of conv layer l (models/cg_model.py:167, models/aa_model.py:157)."""
        if not self.differentiate_convolutions:
            return 1
        if self.all_atoms:
            return 3 if l == self.num_conv_layers - 1 else 9
        return 2 if l == self.num_conv_layers - 1 else 4

    def to_namespace(self) -> argparse.Namespace:
        """Namespace shaped like the reference's args / model_parameters.yml."""
        d = asdict(self)
        d.update(max_radius=self.lig_max_radius, no_batch_norm=not self.batch_norm,
                 no_differentiate_convolutions=not self.differentiate_convolutions,
                 dropout=0.0,
                 esm_embeddings_path="precomputed" if self.lm_embedding_type else None,
                 sidechain_loss_weight=1.0 if self.sidechain_pred else 0.0)
        for k in ('fixed_center_conv', 'lm_embedding_type', 'batch_norm', 'differentiate_convolutions',
                  'lig_max_radius', 'rec_max_radius', 'center_max_distance', 'in_lig_edge_features', 'confidence_mode',
                  'num_confidence_outputs', 'old', 'atom_confidence', 'atom_num_confidence_outputs', 'sidechain_pred'):
            d.pop(k)
        if self.num_confidence_outputs > 1:
            d["rmsd_classification_cutoff"] = [2.0 + i for i in range(self.num_confidence_outputs - 1)]
        if self.atom_confidence:
            d["atom_confidence_loss_weight"] = 1.0
            if self.atom_num_confidence_outputs > 1:
                d["atom_rmsd_classification_cutoff"] = [1.0 + i for i in range(self.atom_num_confidence_outputs - 1)]
        return argparse.Namespace(**d)

    def replace(self, **kw) -> "ModelConfig":
        return replace(self, **kw)


def config_from_args(args) -> ModelConfig:
    """Build a ModelConfig from a reference-style args namespace using the same
    'k in args' fall-backs as get_model (utils/utils.py:174-276)."""
    def has(k):
        return hasattr(args, k)

    def get(k, default):
        return getattr(args, k) if has(k) else default

    lm = None
    for k in ("moad_esm_embeddings_path", "pdbbind_esm_embeddings_path",
              "pdbsidechain_esm_embeddings_path", "esm_embeddings_path"):
        if get(k, None) is not None:
            lm = "precomputed"
    # Arguments the reference's get_model acts on (utils/utils.py:174-276) that the built path does not implement: raise
    # instead of silently computing a different function (most of them leave the weight shapes unchanged).
    unsupported = []
    if has("embedding_type") and args.embedding_type not in ("sinusoidal", "fourier"):
        unsupported.append(f"embedding_type={args.embedding_type!r} (the reference raises too, utils/diffusion_utils.py:135)")
    if get("esm_embeddings_model", None) is not None:
        unsupported.append("esm_embeddings_model (on-the-fly language-model embeddings)")
    if get("parallel", 1) not in (1, None):
        unsupported.append("parallel > 1")
    if get("depthwise_convolution", False) and get("all_atoms", False):
        unsupported.append("depthwise_convolution with all_atoms (AAModel asserts it away, models/aa_model.py:39)")
    sidechain = bool(get("sidechain_loss_weight", 0) and args.sidechain_loss_weight > 0) or \
        bool(get("backbone_loss_weight", 0) and args.backbone_loss_weight > 0)
    if sidechain and get("all_atoms", False):
        unsupported.append("sidechain_pred with all_atoms (AAModel asserts it away, models/aa_model.py:38)")
    if get("include_miscellaneous_atoms", False):
        unsupported.append("include_miscellaneous_atoms")
    if get("tp_weights_layers", 2) < 2:
        unsupported.append("tp_weights_layers < 2 (FCBlock asserts layers >= 2, models/layers.py:12)")
    if unsupported:
        raise NotImplementedError("get_model arguments outside the built path: " + "; ".join(unsupported))
    # norm_by_sigma is stored by the reference classes and never read in forward (cg_model.py:44): accepted, no effect
    if get("num_prot_emb_layers", 0) > 0 and not get("embed_also_ligand", False) and not get("all_atoms", False):
        # CGModel asserts this on every forward (models/cg_model.py:263 "otherwise reimplement padding"); AAModel zero-pads the
        # ligand rows to the width the receptor embedding layers produced (models/aa_model.py:351-357): built
        raise NotImplementedError("num_prot_emb_layers > 0 requires embed_also_ligand (models/cg_model.py:263 asserts it)")
    cut = get("rmsd_classification_cutoff", None)
    acut = get("atom_rmsd_classification_cutoff", None)
    return ModelConfig(
        sidechain_pred=sidechain, depthwise_convolution=bool(get("depthwise_convolution", False)),
        all_atoms=bool(get("all_atoms", False)),
        num_confidence_outputs=len(cut) + 1 if isinstance(cut, list) else 1,
        atom_confidence=get("atom_confidence_loss_weight", 0.0) > 0.0,
        atom_num_confidence_outputs=len(acut) + 1 if isinstance(acut, list) else 1,
        affinity_prediction=bool(get("affinity_prediction", False)),
        no_aminoacid_identities=bool(get("no_aminoacid_identities", False)),
        ns=args.ns, nv=args.nv, num_conv_layers=args.num_conv_layers,
        num_prot_emb_layers=get("num_prot_emb_layers", 0), sh_lmax=get("sh_lmax", 2),
        sigma_embed_dim=args.sigma_embed_dim, distance_embed_dim=args.distance_embed_dim,
        cross_distance_embed_dim=args.cross_distance_embed_dim, lig_max_radius=args.max_radius,
        cross_max_distance=args.cross_max_distance, dynamic_max_cross=args.dynamic_max_cross,
        use_second_order_repr=args.use_second_order_repr,
        reduce_pseudoscalars=get("reduce_pseudoscalars", False),
        differentiate_convolutions=not get("no_differentiate_convolutions", False),
        tp_weights_layers=get("tp_weights_layers", 2), embed_also_ligand=get("embed_also_ligand", False),
        batch_norm=not args.no_batch_norm, smooth_edges=get("smooth_edges", False),
        odd_parity=get("odd_parity", False), no_torsion=args.no_torsion, scale_by_sigma=args.scale_by_sigma,
        fixed_center_conv=(not args.not_fixed_center_conv) if has("not_fixed_center_conv") else False,
        lm_embedding_type=lm,
        embedding_scale=args.embedding_scale if has("embedding_type") else 10000.0,
        embedding_type=get("embedding_type", "sinusoidal"),
        tr_sigma_min=args.tr_sigma_min, tr_sigma_max=args.tr_sigma_max,
        rot_sigma_min=args.rot_sigma_min, rot_sigma_max=args.rot_sigma_max,
        tor_sigma_min=args.tor_sigma_min, tor_sigma_max=args.tor_sigma_max,
        crop_beyond=get("crop_beyond", None))


# Declared benchmark preset: FasterTensorProduct conv layers.
DDL_SYNTH = ModelConfig(
    ns=48, nv=10, num_conv_layers=6, num_prot_emb_layers=0, sh_lmax=1,
    sigma_embed_dim=64, distance_embed_dim=64, cross_distance_embed_dim=64,
    lig_max_radius=5.0, cross_max_distance=80.0, dynamic_max_cross=True,
    use_second_order_repr=False, reduce_pseudoscalars=False, differentiate_convolutions=True,
    tp_weights_layers=2, embed_also_ligand=True, batch_norm=True, smooth_edges=False,
    tr_sigma_min=0.1, tr_sigma_max=19.0, rot_sigma_min=0.03, rot_sigma_max=1.55,
    tor_sigma_min=0.0314, tor_sigma_max=3.14, crop_beyond=None)

# Small preset for CPU-fast parity tests (same structure, all four irrep stages reached).
# tr_sigma_max is lowered so that an UNTRAINED score model does not fling the ligand out of
# cross-graph range during multi-step sampling tests.
TINY = DDL_SYNTH.replace(ns=8, nv=3, num_conv_layers=4, sigma_embed_dim=16, distance_embed_dim=16,
                         cross_distance_embed_dim=16, tr_sigma_max=5.0)
