"""Shared helpers for the test-suite: rebuild inputs from the committed golden fixtures."""
import os

import numpy as np
import torch

from diffdock_amd.config import ModelConfig
from diffdock_amd.hetero import HeteroData

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name):
    return torch.load(os.path.join(GOLDEN, name + ".pt"), weights_only=False)


def tables():
    return (np.load(os.path.join(GOLDEN, "so3_exp_score_norms.npy")),
            np.load(os.path.join(GOLDEN, "torus_score_norm.npy")))


def graph_from_dict(d, pos=None):
    g = HeteroData()
    g["receptor"].x = d["rec_x"]
    g["receptor"].pos = d["rec_pos"]
    g["receptor"].side_chain_vecs = torch.zeros(d["rec_pos"].shape[0], 10)
    g["receptor", "rec_contact", "receptor"].edge_index = d["rec_edge_index"]
    if "atom_x" in d:   # all-atom fixture
        g["atom"].x = d["atom_x"]
        g["atom"].pos = d["atom_pos"]
        g["atom", "atom_contact", "atom"].edge_index = d["atom_edge_index"]
        g["atom", "atom_rec_contact", "receptor"].edge_index = d["atom_rec_edge_index"]
    g["ligand"].x = d["lig_x"]
    g["ligand"].pos = d["lig_pos"] if pos is None else pos
    g["ligand"].edge_mask = d["edge_mask"]
    g["ligand"].mask_rotate = [d["mask_rotate"].numpy()]
    g["ligand", "lig_bond", "ligand"].edge_index = d["bond_index"]
    g["ligand", "lig_bond", "ligand"].edge_attr = d["bond_attr"]
    g.name = "fixture"
    return g


def fixture_case(name):
    fx = load_fixture(name)
    cfg = ModelConfig(**fx["cfg"])
    data_list = [graph_from_dict(fx["graph"], pos=p.clone()) for p in fx["poses"]]
    return fx, cfg, data_list


def oracle_model(cfg, state_dict, so3_t=None, tor_t=None, dtype=torch.float32):
    """The oracle class the configuration selects (get_model's all_atoms switch, utils/utils.py:221-224)."""
    from oracle.aa_model import AAModelOracle
    from oracle.cg_model import CGModelOracle
    if so3_t is None:
        so3_t, tor_t = tables()
    if cfg.old:
        from oracle.old_cg_model import CGOldConfidenceOracle
        return CGOldConfidenceOracle(cfg, state_dict, so3_t, tor_t, dtype)
    return (AAModelOracle if cfg.all_atoms else CGModelOracle)(cfg, state_dict, so3_t, tor_t, dtype)


def rel_err(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def split_draws(draws, steps, B, R):
    """Recorded torch.normal draws of the reference loop -> (z_tr, z_rot, z_tor) tensors
    (utils/sampling.py:140-154 order: tr, rot, tor per step; none on the final step when
    no_final_step_noise)."""
    z_tr, z_rot, z_tor = torch.zeros(steps, B, 3), torch.zeros(steps, B, 3), torch.zeros(steps, B * R)
    it = iter(draws)
    for s in range(steps - 1):
        z_tr[s], z_rot[s], z_tor[s] = next(it), next(it), next(it)
    assert next(it, None) is None
    return z_tr, z_rot, z_tor
