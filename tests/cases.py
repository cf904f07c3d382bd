"""Backend-independent parity cases, run by tests/test_emu_parity.py / tests/test_boundary.py on the CPU emulation build
and by the `-m gpu` tests on the MI355X through the same C ABI.  `make(cfg, sd)` returns a loaded MIScoreModel, `place`
moves a batch to the model's device."""
import copy

import numpy as np
import torch

from diffdock_amd.config import TINY
from diffdock_amd.hetero import HeteroBatch, set_time
from diffdock_amd.synth import make_complex, make_pose_list
from diffdock_amd.weights import init_state_dict
from oracle.conformer import get_t_schedule
from oracle.sampling import nan_guard, perturbations
from util import assert_scores_close, oracle_model, rel_err

TEMP = dict(temp_sampling=[1.17, 2.06, 7.04], temp_psi=[0.73, 0.90, 0.59], temp_sigma_data=[0.93, 0.75, 0.69])


def nan_guard_case(make, place, cfg=TINY):
    """utils/sampling.py:117-131 + :133-186 through ddmi_perturb: scores with NaN / +inf / -inf in some poses against the
    oracle's nan_guard + perturbations; a batch without a NaN mean must pass through the guard untouched (infinities too)."""
    sd = init_state_dict(cfg, seed=3)
    g = make_complex(seed=41, n_res=20, n_lig=9)
    B = 6
    dl = make_pose_list(g, B, tr_sigma_max=cfg.tr_sigma_max, seed=42)
    batch = place(HeteroBatch.from_data_list(dl))
    m = make(cfg, sd)
    R = int(dl[0]["ligand"].edge_mask.sum())
    steps = 5
    s = get_t_schedule(steps)
    gen = torch.Generator().manual_seed(8)
    noise = (torch.randn(steps, B, 3, generator=gen), torch.randn(steps, B, 3, generator=gen), torch.randn(steps, B * R, generator=gen))
    for variant in ("nan", "inf_only", "clean"):
        tr, rot, tor = torch.randn(B, 3, generator=gen), torch.randn(B, 3, generator=gen), torch.randn(B * R, generator=gen)
        if variant == "nan":
            tr[1, 0] = float("nan")
            tr[4] = float("nan")
            rot[1, 2] = float("nan")
            rot[2, 1] = float("inf")          # eps of rot becomes inf (nanmean keeps infinities), as in the reference
            tor[R] = float("nan")
            tor[2 * R + 1] = float("-inf")
        elif variant == "inf_only":           # the mean of pose 3 is inf, not NaN: the guard does not fire
            tr[3, 1] = float("inf")
        for k in (0, steps - 1):
            want_scores = nan_guard(tr.clone(), rot.clone(), tor.clone())
            want = perturbations(cfg, k, steps, (s, s, s), want_scores, (noise[0][k], noise[1][k], noise[2][k]),
                                 no_final_step_noise=True, **TEMP)
            got = m.perturb(batch, tr, rot, tor, k, steps, (s, s, s), noise=noise, no_final_step_noise=True, **TEMP)
            for a, b, name in zip(got, want, ("tr", "rot", "tor")):
                a, b = a.cpu().double(), b.double()
                assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.isinf(a), torch.isinf(b)), (variant, k, name)
                fin = torch.isfinite(b)
                assert torch.equal(torch.sign(a[~fin & ~torch.isnan(b)]), torch.sign(b[~fin & ~torch.isnan(b)]))
                assert (a[fin] - b[fin]).abs().max() <= 1e-6 * b[fin].abs().max(), (variant, k, name)
    # and inside the device loop: a NaN in the initial coordinates of one pose must not leak into the other poses
    dl2 = copy.deepcopy(dl)
    clean = m.sample_batch(place(HeteroBatch.from_data_list(dl2)), 2, (s[:2], s[:2], s[:2]), no_random=True).cpu().reshape(B, -1, 3)
    dl2[2]["ligand"].pos[0, 0] = float("nan")
    dirty = m.sample_batch(place(HeteroBatch.from_data_list(dl2)), 2, (s[:2], s[:2], s[:2]), no_random=True).cpu().reshape(B, -1, 3)
    keep = [i for i in range(B) if i != 2]
    assert torch.isfinite(dirty[keep]).all() and not torch.isfinite(dirty[2]).all()
    return clean, dirty


def neighbour_cap_case(make, place, cfg=TINY):
    """A compact 40-atom ligand: more than 32 atoms within lig_max_radius of most atoms and of every rotatable-bond
    midpoint, so the neighbour caps of radius_graph (32 + self, cg_model.py:477) and of the bond-graph radius search
    (32, cg_model.py:630) bind; which neighbours survive = first by ascending index (torch-cluster's device kernel)."""
    sd = init_state_dict(cfg, seed=5)
    g = make_complex(seed=51, n_res=24, n_lig=40)
    g["ligand"].pos = g["ligand"].pos * 0.33
    dl = make_pose_list(g, 2, tr_sigma_max=cfg.tr_sigma_max, seed=52, initial_noise_std_proportion=0.2)
    batch = HeteroBatch.from_data_list(dl)
    set_time(batch, 0.4, 0.4, 0.4, 2)
    pos = batch["ligand"].pos.reshape(2, 40, 3)
    within = (torch.cdist(pos, pos) < cfg.lig_max_radius).sum(-1)          # includes the atom itself
    assert int((within > 33).sum()) > 20, "the radius-graph cap must bind in this case"
    ref = oracle_model(cfg, sd)(batch, return_intermediates=True)
    m = make(cfg, sd)
    out = m(place(batch))
    assert int(m.debug_buffer("goff_ll")[-1]) == ref[4]["edge_counts"][0]
    n_bond = batch["ligand", "ligand"].edge_index.shape[1]
    uncapped = int((within - 1).sum()) + n_bond
    assert ref[4]["edge_counts"][0] < uncapped                             # edges were really dropped
    assert int(m.debug_buffer("tor_cnt").max()) == 32                      # bond-graph cap reached
    assert_scores_close(out[:3], ref[:3], what="neighbour caps")
    return out


def same_shape_complexes_case(make, place, cfg=TINY):
    """inference.py:224-303 reuses one model over many complexes.  Two DIFFERENT complexes with identical tensor shapes
    (same topology, other residue / atom / bond features), the second collated after the first batch was freed (the
    allocator may hand out the same addresses) and, separately, written INTO the storage of the live batch object: the
    scores must equal those of a fresh handle every time."""
    sd = init_state_dict(cfg, seed=9)
    ga = make_complex(seed=61, n_res=22, n_lig=11)
    other = make_complex(seed=62, n_res=22, n_lig=11)
    gb = ga.clone()
    gb["receptor"].x = other["receptor"].x.clone()
    gb["ligand"].x = ga["ligand"].x.flip(0).clone()
    gb["ligand", "ligand"].edge_attr = ga["ligand", "ligand"].edge_attr.roll(1, dims=1).clone()
    m = make(cfg, sd)

    def run(model, g):
        b = HeteroBatch.from_data_list(make_pose_list(g, 2, tr_sigma_max=cfg.tr_sigma_max, seed=1))
        set_time(b, 0.5, 0.5, 0.5, 2)
        b = place(b)
        return b, [o.cpu().clone() for o in model(b)[:3]]
    batch_a, out_a = run(m, ga)
    del batch_a
    batch_b, out_b = run(m, gb)
    _, fresh_b = run(make(cfg, sd), gb)
    assert all(torch.equal(x, y) for x, y in zip(out_b, fresh_b)), "stale ddmi_set_complex reused for a same-shaped complex"
    assert not torch.equal(out_a[0], out_b[0])
    # in-place overwrite of the static tensors of the live batch object (same object, same addresses)
    src = HeteroBatch.from_data_list(make_pose_list(ga, 2, tr_sigma_max=cfg.tr_sigma_max, seed=1))
    batch_b["receptor"].x.copy_(src["receptor"].x)
    batch_b["ligand"].x.copy_(src["ligand"].x)
    batch_b["ligand", "ligand"].edge_attr.copy_(src["ligand", "ligand"].edge_attr)
    again_a = [o.cpu() for o in m(batch_b)[:3]]
    assert all(torch.equal(x, y) for x, y in zip(again_a, out_a)), "in-place edit of the batch features went undetected"
    m.invalidate_complex()
    assert all(torch.equal(x.cpu(), y) for x, y in zip(m(batch_b)[:3], out_a))
    return out_a, out_b


def config0_case(make, place, cfg, tol_pos=2e-3, steps=4):
    """BASELINE configs[0]: the reference's example complex data/1a0q (416 residues / 23 heavy atoms, read by
    diffdock_amd.io -> tests/golden/1a0q_graph.pt), 4 inference steps x 2 samples: the device loop against the oracle's
    loop on the real geometry (language-model embeddings and RDKit atom features are external inputs: seeded stand-ins)."""
    from oracle.sampling import sampling as oracle_sampling
    from util import graph_from_dict, load_fixture, rmsd
    d = dict(load_fixture("1a0q_graph"))
    rng = np.random.default_rng(3)
    if cfg.lm_embedding_type:
        d["rec_x"] = torch.cat([d["rec_x"], torch.from_numpy((rng.normal(size=(416, cfg.lm_embedding_dim)) * 0.2).astype(np.float32))], 1)
    from diffdock_amd.config import LIG_FEATURE_DIMS
    feats = d["lig_x"].clone()
    for c in range(1, 16):
        feats[:, c] = torch.from_numpy(rng.integers(0, LIG_FEATURE_DIMS[c], size=23))
    d["lig_x"] = feats
    g = graph_from_dict(d)
    sd = init_state_dict(cfg, seed=17)
    B = 2
    dl = make_pose_list(g, B, tr_sigma_max=cfg.tr_sigma_max, seed=5, initial_noise_std_proportion=0.2)
    R = int(d["edge_mask"].sum())
    gen = torch.Generator().manual_seed(4)
    noise = (torch.randn(steps, B, 3, generator=gen), torch.randn(steps, B, 3, generator=gen), torch.randn(steps, B * R, generator=gen))
    s = get_t_schedule(steps)
    batch = HeteroBatch.from_data_list(dl)
    set_time(batch, s[0], s[0], s[0], B)
    om = oracle_model(cfg, sd)
    ref0 = om(batch)[:3]
    m = make(cfg, sd)
    assert_scores_close(m(place(batch))[:3], ref0, what="1a0q step 0")
    ref = oracle_sampling([x.clone() for x in dl], om, steps, cfg, noise, schedules=(s, s, s), batch_size=B,
                          no_final_step_noise=True, **TEMP)
    ref_pos = torch.stack([x["ligand"].pos for x in ref])
    pos = m.sample_batch(place(HeteroBatch.from_data_list(dl)), steps, (s, s, s), noise=noise, no_final_step_noise=True,
                         **TEMP).cpu().reshape(B, -1, 3)
    r = rmsd(pos, ref_pos)
    assert float(r.max()) < tol_pos, r
    return r
