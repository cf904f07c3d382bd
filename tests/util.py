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


def elem_excess(a, b, rtol=1e-4, atol_frac=1e-5):
    """Element-wise parity metric: max over elements of |a - b| / (rtol * |b| + atol), atol = atol_frac * max|b|.
    <= 1 means EVERY element satisfies |a - b| <= 1e-4 * |ref| + atol (north_star: 1e-4 relative fp32); the absolute
    floor (a tenth of the old max-norm bound) only covers elements that are themselves sums with cancellation."""
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    if b.numel() == 0:
        return 0.0
    atol = atol_frac * float(b.abs().max().clamp(min=1e-30))
    return float(((a - b).abs() / (rtol * b.abs() + atol)).max())


def assert_scores_close(mine, ref, names=("tr", "rot", "tor"), rtol=1e-4, atol_frac=1e-5, what=""):
    """Both parity metrics on every score tensor: max-norm relative error and the element-wise bound."""
    for m, r, n in zip(mine, ref, names):
        m = torch.as_tensor(m).detach().cpu()
        assert tuple(m.shape) == tuple(r.shape), (what, n, m.shape, r.shape)
        e_max, e_el = rel_err(m, r), elem_excess(m, r, rtol, atol_frac)
        assert e_max < rtol and e_el <= 1.0, f"{what} {n}: max-norm rel {e_max:.3e}, element-wise excess {e_el:.3f}"


def rmsd(a, b):
    """Per-pose RMSD between two [B, N, 3] coordinate sets (no alignment: same frame)."""
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return ((a - b) ** 2).sum(-1).mean(-1).sqrt()


def checksum(tensors):
    """Fingerprint used by tests/golden/make_golden_fullsize.py for inputs that are regenerated from seeds."""
    items = tensors.items() if isinstance(tensors, dict) else enumerate(tensors)
    s1 = s2 = 0.0
    for _, t in items:
        t = torch.as_tensor(t).double().reshape(-1)
        s1 += float(t.abs().sum())
        s2 += float((t * (1 + torch.arange(t.numel(), dtype=torch.float64) % 7)).sum())
    return [s1, s2]


def graph_tensors(g):
    return [g["receptor"].x, g["receptor"].pos, g["receptor", "receptor"].edge_index, g["ligand"].x, g["ligand"].pos,
            g["ligand"].edge_mask, g["ligand", "ligand"].edge_index, g["ligand", "ligand"].edge_attr,
            torch.from_numpy(np.asarray(g["ligand"].mask_rotate[0]))]


def seeded_case(spec):
    """Inputs of a seeds-only fixture (traj_300_30, fwd_1500_80): configuration, weights, complex and initial poses from
    the deterministic generators, verified against the checksums recorded when the reference was executed."""
    from diffdock_amd.config import DDL_SYNTH
    from diffdock_amd.synth import make_complex, make_pose_list
    from diffdock_amd.weights import init_state_dict
    cfg = DDL_SYNTH.replace(**spec["cfg_replace"])
    sd = init_state_dict(cfg, seed=spec["weight_seed"])
    g = make_complex(seed=spec["complex_seed"], n_res=spec["n_res"], n_lig=spec["n_lig"], all_atoms=cfg.all_atoms)
    dl = make_pose_list(g, spec["n_poses"], tr_sigma_max=cfg.tr_sigma_max, seed=spec["pose_seed"],
                        initial_noise_std_proportion=spec["noise_prop"])
    return cfg, sd, g, dl


def check_seeded_inputs(fx, sd, g):
    for name, got in (("state_dict", checksum(sd)), ("graph", checksum(graph_tensors(g)))):
        want = fx["checks"][name]
        assert all(abs(a - b) <= 1e-9 * max(1.0, abs(b)) for a, b in zip(got, want)), \
            f"seeded {name} differs from the one the reference was executed on (generator drift): {got} vs {want}"


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
