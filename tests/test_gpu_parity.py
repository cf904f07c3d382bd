"""Parity of the gfx950 build on a real MI355X, through the C ABI (libddmi.so via diffdock_amd.model).

Tolerance: BASELINE.json north_star = 1e-4 relative fp32 on the score outputs (tr / rot / tor); the
kernels run exact-fp32 MFMA, differences come from re-association only.  Full-size cases use
size-independent properties: SE(3) equivariance of the scores, batch/shard invariance, determinism."""
import os

import numpy as np
import pytest
import torch

from diffdock_amd.config import DDL_SYNTH
from diffdock_amd.hetero import HeteroBatch, set_time
from diffdock_amd.model import MIScoreModel
from diffdock_amd.synth import make_complex, make_pose_list
from diffdock_amd.weights import init_state_dict
from oracle.cg_model import CGModelOracle
from oracle.conformer import get_t_schedule
from util import assert_scores_close, fixture_case, graph_from_dict, load_fixture, oracle_model, rel_err, split_draws, tables

pytestmark = pytest.mark.gpu
REL = 1e-4
CASES = ["tiny_l1", "tiny_l2", "tiny_l1_1group_emb", "tiny_l2_fixedcenter", "tiny_l2_crop", "tiny_aa_l1", "tiny_aa_l2", "tiny_aa_l2_emb",
         "tiny_noaa", "tiny_2nd", "tiny_aa_2nd",   # tiny_*2nd: use_second_order_repr (2e / 2o node blocks)
         "tiny_fourier", "tiny_tpw3",              # embedding_type='fourier'; tp_weights_layers=3
         "tiny_aa_emb_nolig",                      # AAModel: embedding layers without embed_also_ligand (zero-padded ligand rows)
         "tiny_oddpar", "tiny_aa_oddpar", "tiny_nobn_noscale",   # odd_parity (CG + all-atom); batch_norm off + scale_by_sigma off
         "tiny_sidechain",                                        # sidechain_pred: o3.Linear on the receptor rows, 4th tuple element
         "tiny_depthwise", "tiny_depthwise_l2"]                   # depthwise_convolution: 'uvu' TensorProduct + linear_2 (sh_lmax 1 and 2)


def gpu_model(cfg, sd):
    assert torch.cuda.is_available(), "these tests need the MI355X (run through gpurun with -m gpu)"
    m = MIScoreModel(cfg, device="cuda:0")     # raises DdmiError if libddmi.so is not built: no fallback
    m.load_state_dict(sd)
    m.set_tables(*tables())
    return m


def to_gpu(batch):
    return batch.to("cuda:0")


@pytest.mark.parametrize("name", CASES)
def test_forward_matches_reference_fixture(name):
    fx, cfg, data_list = fixture_case(name)
    m = gpu_model(cfg, fx["state_dict"])
    batch = HeteroBatch.from_data_list(data_list)
    set_time(batch, fx["t"], fx["t"], fx["t"], batch.num_graphs)
    tr, rot, tor, side = m(to_gpu(batch))
    ref = fx["forward"]
    assert tr.is_cuda
    if cfg.sidechain_pred:   # 4th tuple element (models/cg_model.py:397-402): [n_rec, 10]
        assert side.shape == ref["sidechain"].shape and rel_err(side.cpu(), ref["sidechain"]) < REL
    else:
        assert side is None
    assert_scores_close((tr, rot, tor), (ref["tr"], ref["rot"], ref["tor"]))
    if cfg.num_prot_emb_layers == 0:
        for l, ref_nodes in enumerate(ref["conv_out"]):
            mine = torch.from_numpy(m.debug_buffer(f"x{l + 1}"))
            n = ref_nodes.shape[0] if l < len(ref["conv_out"]) - 1 else batch["ligand"].pos.shape[0]
            assert rel_err(mine[:n, :ref_nodes.shape[1]], ref_nodes[:n]) < REL, l


@pytest.mark.parametrize("name", ["tiny_l1", "tiny_l2", "tiny_l2_crop", "tiny_aa_l1", "tiny_aa_l2", "tiny_aa_l2_emb", "tiny_2nd", "tiny_aa_2nd", "tiny_fourier",
                                  "tiny_tpw3", "tiny_aa_emb_nolig", "tiny_oddpar", "tiny_aa_oddpar", "tiny_nobn_noscale"])
def test_device_loop_matches_reference_trajectory(name):
    fx, cfg, data_list = fixture_case(name)
    m = gpu_model(cfg, fx["state_dict"])
    s = fx["sampling"]
    B, R = len(data_list), int(data_list[0]["ligand"].edge_mask.sum())
    noise = split_draws(s["draws"], s["steps"], B, R)
    sched = get_t_schedule(s["steps"])
    pos = m.sample_batch(to_gpu(HeteroBatch.from_data_list(data_list)), s["steps"], (sched, sched, sched), noise=noise,
                         no_final_step_noise=True, crop_beyond=cfg.crop_beyond, **s["temp"])
    assert (pos.cpu().reshape(B, -1, 3) - s["final_pos"]).abs().max() < 2e-3   # Angstrom after 4 chaotic fp32 steps
    if cfg.crop_beyond is not None:
        keep = m.debug_buffer("crop_keep")
        assert 0 < keep.sum() < keep.size


def test_modify_conformer_matches_reference():
    fx, cfg, _ = fixture_case("tiny_l1")
    u = load_fixture("units")
    m = gpu_model(cfg, fx["state_dict"])
    B = u["mc_tr"].shape[0]
    b = to_gpu(HeteroBatch.from_data_list([graph_from_dict(u["mc_graph"]) for _ in range(B)]))
    out = m.modify_conformer_batch(u["mc_pos_in"], b, u["mc_tr"], u["mc_rot"], u["mc_tor"])
    assert (out.cpu() - u["mc_pos_out"]).abs().max() < 5e-5


def synth_batch(cfg, n_res, n_lig, B, seed, t):
    g = make_complex(seed=seed, n_res=n_res, n_lig=n_lig)
    dl = make_pose_list(g, B, tr_sigma_max=cfg.tr_sigma_max, seed=seed + 1, initial_noise_std_proportion=0.6)
    batch = HeteroBatch.from_data_list(dl)
    set_time(batch, t, t, t, B)
    return batch


@pytest.mark.parametrize("lmax,t", [(1, 0.8), (2, 0.15)])
def test_ddl_synth_forward_matches_oracle(lmax, t):
    """The declared benchmark architecture (ns=48, nv=10, 6 layers, 64-d embeddings) on a 300-residue /
    30-atom complex, 2 poses (the oracle materialises [E, 6928] weights, so it stays small)."""
    cfg = DDL_SYNTH.replace(sh_lmax=lmax)
    sd = init_state_dict(cfg, seed=1234)
    batch = synth_batch(cfg, 300, 30, 2, seed=3, t=t)
    so3_t, tor_t = tables()
    tr, rot, tor, _ = CGModelOracle(cfg, sd, so3_t, tor_t)(batch)
    m = gpu_model(cfg, sd)
    tr2, rot2, tor2, _ = m(to_gpu(batch))
    assert_scores_close((tr2, rot2, tor2), (tr, rot, tor))


@pytest.fixture(scope="module")
def width48_case():
    """Benchmark-width model on a small complex: inputs, oracle scores and the scores of the default kernel route."""
    cfg = DDL_SYNTH
    sd = init_state_dict(cfg, seed=1234)
    batch = synth_batch(cfg, 100, 40, 2, seed=5, t=0.6)     # 40-atom ligand: receptor residues with two virtual nodes
    so3_t, tor_t = tables()
    ref = CGModelOracle(cfg, sd, so3_t, tor_t)(batch)[:3]
    base = [o.cpu() for o in gpu_model(cfg, sd)(to_gpu(batch))[:3]]
    return cfg, sd, batch, ref, base


@pytest.mark.parametrize("env", [{"DDMI_FUSED_PACK": "0"}, {"DDMI_FUSED_DENSE": "0"},
                                 {"DDMI_FUSED_DENSE": "2"}, {"DDMI_FUSED_MM": "0"}, {"DDMI_STREAMS": "1"}, {"DDMI_FUSED_YS": "3"},
                                 {"DDMI_FUSED_SHARED": "0"}, {"DDMI_FUSED_SHARED": "2", "DDMI_FUSED_DENSE": "2"}, {"DDMI_FC1_BATCH": "0"}, {"DDMI_FUSED_TRI": "0"}, {"DDMI_FUSED_PRERED": "0"},
                                 {"DDMI_GROUPED": "1"}, {"DDMI_GROUPED": "2"}, {"DDMI_GROUPED": "2", "DDMI_GROUPED_YS": "3"},
                                 {"DDMI_GROUPED": "2", "DDMI_FUSED_PRERED": "0", "DDMI_FUSED_SHARED": "0"},
                                 {"DDMI_NODE_UPDATE": "1"}, {"DDMI_VN_BUILD": "1"}, {"DDMI_LIST_CAPS": "1"}, {"DDMI_YS_RULE": "1"}, {"DDMI_TIME_TERMS": "1"}, {"DDMI_GROUP_ORDER": "3", "DDMI_FUSED_YS_LAST": "2"}, {"DDMI_NODE_UPDATE": "1", "DDMI_VN_BUILD": "1", "DDMI_GROUPED": "2"}],
                         ids=lambda e: ",".join(f"{k[5:]}={v}" for k, v in e.items()))
def test_selectable_kernel_paths_agree_on_the_gpu(env, width48_case, monkeypatch):
    """Every selectable route of an edge group (classic instead of packed granules for the 10-channel vector blocks,
    sparse- / dense-row loop, GEMM first layer, one stream, granule-range splits,
    the rec<-lig group per virtual node instead of per distinct gather node / every group through the shared-node kernel,
    per-group launches on two streams / one grouped launch per layer, the fused node update instead of k_reduce_bn + first-Linear GEMM
    launches, one virtual-node list chain per group instead of the merged build)
    against the default route and the oracle at the benchmark width.  The routes are fields of ddmi_config.exec (the library
    reads no environment variable); diffdock_amd/lib.py maps these harness variables onto them when a model handle is created,
    so each handle is built under its own environment."""
    cfg, sd, batch, ref, base = width48_case
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    m = gpu_model(cfg, sd)
    m.set_kernel_timing(True)
    out = m(to_gpu(batch))[:3]
    launched = m.kernel_timings()
    assert "k_conv_fused" in launched
    for o, b, r in zip(out, base, ref):
        assert rel_err(o.cpu(), r) < REL and rel_err(o.cpu(), b) < 1e-5


@pytest.mark.parametrize("name", ["tiny_conf_l2", "tiny_conf_aa_l1", "tiny_conf_atom"])
def test_confidence_mode_matches_reference_fixture(name):
    fx, cfg, data_list = fixture_case(name)
    m = MIScoreModel(cfg, device="cuda:0")
    m.load_state_dict(fx["state_dict"])
    batch = HeteroBatch.from_data_list(data_list)
    set_time(batch, fx["t"], fx["t"], fx["t"], batch.num_graphs)
    conf, atom_conf = m(to_gpu(batch))
    assert conf.is_cuda and rel_err(conf.cpu(), fx["forward"]["confidence"]) < REL
    if cfg.atom_confidence:
        assert rel_err(atom_conf.cpu(), fx["forward"]["atom_confidence"]) < REL
    else:
        assert not atom_conf.any()


@pytest.mark.parametrize("name", ["tiny_oldconf", "tiny_oldconf_2l"])
def test_legacy_confidence_class_matches_reference_fixture(name):
    fx, cfg, data_list = fixture_case(name)
    m = MIScoreModel(cfg, device="cuda:0")
    m.load_state_dict(fx["state_dict"])
    batch = HeteroBatch.from_data_list(data_list)
    set_time(batch, fx["t"], fx["t"], fx["t"], batch.num_graphs)
    conf = m(to_gpu(batch))
    assert conf.is_cuda and rel_err(conf.cpu(), fx["forward"]["confidence"]) < REL


def test_legacy_class_score_mode_matches_reference_fixture():
    """get_model(old=True) in score mode (old_cg_model.py:293-352): forward 3-tuple and the device loop."""
    fx, cfg, data_list = fixture_case("tiny_oldscore")
    m = gpu_model(cfg, fx["state_dict"])
    batch = HeteroBatch.from_data_list(data_list)
    set_time(batch, fx["t"], fx["t"], fx["t"], batch.num_graphs)
    out = m(to_gpu(batch))
    assert len(out) == 3
    for mine, key in zip(out, ("tr", "rot", "tor")):
        assert rel_err(mine.cpu(), fx["forward"][key]) < REL, key
    s = fx["sampling"]
    B, R = len(data_list), int(data_list[0]["ligand"].edge_mask.sum())
    sched = get_t_schedule(s["steps"])
    pos = m.sample_batch(to_gpu(HeteroBatch.from_data_list(data_list)), s["steps"], (sched, sched, sched),
                         noise=split_draws(s["draws"], s["steps"], B, R), no_final_step_noise=True, **s["temp"])
    assert (pos.cpu().reshape(B, -1, 3) - s["final_pos"]).abs().max() < 2e-3


def test_legacy_confidence_class_full_width_matches_oracle():
    """The legacy confidence class at DiffDock-L-like widths (ns=24, nv=6, 5 layers, sh_lmax=2) on a 200-residue / 28-atom
    complex, 3 poses, t = 0 (cross cutoff 20 A) -- static-shape and generic fused kernels, load mode, against the oracle."""
    cfg = DDL_SYNTH.replace(old=True, confidence_mode=True, sh_lmax=2, ns=24, nv=6, num_conv_layers=5, sigma_embed_dim=32,
                            distance_embed_dim=32, cross_distance_embed_dim=32, embed_also_ligand=False)
    sd = init_state_dict(cfg, seed=5)
    g = make_complex(seed=21, n_res=200, n_lig=28)
    dl = make_pose_list(g, 3, tr_sigma_max=cfg.tr_sigma_max, seed=22, initial_noise_std_proportion=0.2)
    batch = HeteroBatch.from_data_list(dl)
    set_time(batch, 0, 0, 0, 3)
    ref = oracle_model(cfg, sd)(batch)
    m = MIScoreModel(cfg, device="cuda:0")
    m.load_state_dict(sd)
    assert rel_err(m(to_gpu(batch)).cpu(), ref) < REL


def test_all_atom_ddl_width_matches_oracle():
    """AAModel (models/aa_model.py) at the DDL-synth channel widths: 120 residues / ~900 receptor atoms / 24 ligand atoms,
    2 poses started inside the pocket (so that the ligand<->atom radius graph is populated), nine edge groups per layer,
    against the oracle (which materialises the per-edge weights, hence the small complex)."""
    cfg = DDL_SYNTH.replace(all_atoms=True, num_conv_layers=4, lm_embedding_type=None)
    sd = init_state_dict(cfg, seed=77)
    g = make_complex(seed=12, n_res=120, n_lig=24, lm_dim=0, all_atoms=True)
    dl = make_pose_list(g, 2, tr_sigma_max=cfg.tr_sigma_max, seed=13, initial_noise_std_proportion=0.05)
    batch = HeteroBatch.from_data_list(dl)
    set_time(batch, 0.4, 0.4, 0.4, 2)
    tr, rot, tor, _, inter = oracle_model(cfg, sd)(batch, return_intermediates=True)
    m = gpu_model(cfg, sd)
    tr2, rot2, tor2, _ = m(to_gpu(batch))
    assert inter["edge_counts"][2] > 100 and int(m.debug_buffer("offs_la_l")[-1]) == inter["edge_counts"][2]
    assert_scores_close((tr2, rot2, tor2), (tr, rot, tor))
    for l in range(cfg.num_conv_layers - 1):   # all node rows: ligand, residues, atoms
        mine = torch.from_numpy(m.debug_buffer(f"x{l + 1}"))
        ref = inter[f"node_attr{l + 1}"]
        assert rel_err(mine[:, :ref.shape[1]], ref) < REL, l


def test_all_atom_bench_size_complex_matches_oracle():
    """AAModel on the complex of the all-atom bench line (300 residues / ~2250 receptor atoms / 30 ligand atoms, all six interaction
    layers at the DDL-synth widths), 2 poses instead of 40 so that the oracle -- which materialises the per-edge weights -- stays
    within seconds: scores and every node table against the oracle.  (The 40-pose run itself is covered by determinism /
    equivariance / shard invariance below; the per-pose arithmetic does not depend on the batch size up to tile composition.)"""
    cfg = DDL_SYNTH.replace(all_atoms=True, lm_embedding_type=None, dynamic_max_cross=False, cross_max_distance=80.0)
    sd = init_state_dict(cfg, seed=99)
    g = make_complex(seed=14, n_res=300, n_lig=30, lm_dim=0, all_atoms=True)
    dl = make_pose_list(g, 2, tr_sigma_max=cfg.tr_sigma_max, seed=15, initial_noise_std_proportion=0.05)
    batch = HeteroBatch.from_data_list(dl)
    set_time(batch, 0.5, 0.5, 0.5, 2)
    tr, rot, tor, _, inter = oracle_model(cfg, sd)(batch, return_intermediates=True)
    m = gpu_model(cfg, sd)
    tr2, rot2, tor2, _ = m(to_gpu(batch))
    assert batch["atom"].pos.shape[0] > 2 * 2000 and inter["edge_counts"][2] > 100
    assert_scores_close((tr2, rot2, tor2), (tr, rot, tor))
    for l in range(cfg.num_conv_layers - 1):
        mine = torch.from_numpy(m.debug_buffer(f"x{l + 1}"))
        ref = inter[f"node_attr{l + 1}"]
        assert rel_err(mine[:, :ref.shape[1]], ref) < REL, l


def test_full_size_properties():
    """BASELINE configs[2] shape (40 poses x 300 residues x 30 atoms): equivariance, shard invariance,
    run-to-run determinism of the device path."""
    # fixed_center_conv=True: with the reference's default a pose's score depends on its index in the batch
    # (cg_model.py:371-374 indexes the ligand table by graph id), which would make shard invariance meaningless
    cfg = DDL_SYNTH.replace(fixed_center_conv=True)
    sd = init_state_dict(cfg, seed=1234)
    m = gpu_model(cfg, sd)
    B = 40
    batch = synth_batch(cfg, 300, 30, B, seed=4, t=0.5)
    tr, rot, tor, _ = m(to_gpu(batch))
    tr_b, rot_b, tor_b, _ = m(to_gpu(batch))
    assert torch.equal(tr, tr_b) and torch.equal(rot, rot_b) and torch.equal(tor, tor_b)      # deterministic
    assert torch.isfinite(tr).all() and torch.isfinite(rot).all() and torch.isfinite(tor).all()
    # global rotation + translation of the whole complex: tr is a vector (1o+1e summed -> rotates), rot likewise
    q = torch.tensor([0.3, -0.5, 0.7, 0.4], dtype=torch.float64)
    q = q / q.norm()
    w, x, y, z = q
    Rm = torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                       [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                       [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]).float()
    shift = torch.tensor([[3.0, -2.0, 5.0]])
    rb = synth_batch(cfg, 300, 30, B, seed=4, t=0.5)
    rb["ligand"].pos = rb["ligand"].pos @ Rm.T + shift
    rb["receptor"].pos = rb["receptor"].pos @ Rm.T + shift
    tr_r, rot_r, tor_r, _ = m(to_gpu(rb))
    assert rel_err(tr_r.cpu(), tr.cpu() @ Rm.T) < 2e-3 and rel_err(rot_r.cpu(), rot.cpu() @ Rm.T) < 2e-3
    assert rel_err(tor_r.cpu(), tor.cpu()) < 2e-3
    # shard invariance: poses 0..19 / 20..39 evaluated alone equal their rows in the batch of 40
    g = make_complex(seed=4, n_res=300, n_lig=30)
    dl = make_pose_list(g, B, tr_sigma_max=cfg.tr_sigma_max, seed=5, initial_noise_std_proportion=0.6)
    R = tor.numel() // B
    for lo in (0, 20):
        sb = HeteroBatch.from_data_list(dl[lo:lo + 20])
        set_time(sb, 0.5, 0.5, 0.5, 20)
        tr_s, rot_s, tor_s, _ = m(to_gpu(sb))
        assert rel_err(tr_s.cpu(), tr[lo:lo + 20].cpu()) < 1e-5
        assert rel_err(tor_s.cpu(), tor[lo * R:(lo + 20) * R].cpu()) < 1e-5


def test_all_atom_full_size_properties():
    """The all-atom model at the BASELINE configs[2] shape (40 poses x 300 residues x ~2250 receptor atoms x 30 ligand atoms,
    0.7 M atom-atom edges): determinism, SE(3) equivariance of the scores, and shard invariance -- no oracle at this size."""
    cfg = DDL_SYNTH.replace(all_atoms=True, fixed_center_conv=True, dynamic_max_cross=False, cross_max_distance=80.0)
    sd = init_state_dict(cfg, seed=99)
    m = gpu_model(cfg, sd)
    B = 40
    g = make_complex(seed=14, n_res=300, n_lig=30, all_atoms=True)
    dl = make_pose_list(g, B, tr_sigma_max=cfg.tr_sigma_max, seed=15, initial_noise_std_proportion=0.2)

    def run(lst, rot=None, shift=None):
        b = HeteroBatch.from_data_list(lst)
        if rot is not None:
            for nt in ("ligand", "receptor", "atom"):
                b[nt].pos = b[nt].pos @ rot.T + shift
        set_time(b, 0.5, 0.5, 0.5, len(lst))
        return m(to_gpu(b))[:3]
    tr, rot, tor = run(dl)
    tr_b, rot_b, tor_b = run(dl)
    assert torch.equal(tr, tr_b) and torch.equal(rot, rot_b) and torch.equal(tor, tor_b)
    assert torch.isfinite(tr).all() and torch.isfinite(rot).all() and torch.isfinite(tor).all()
    assert int(m.debug_buffer("offs_la_l")[-1]) > 0
    q = torch.tensor([0.2, 0.6, -0.3, 0.7], dtype=torch.float64)
    q = q / q.norm()
    w, x, y, z = q
    Rm = torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                       [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                       [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]).float()
    tr_r, rot_r, tor_r = run(dl, Rm, torch.tensor([[-4.0, 1.0, 2.5]]))
    assert rel_err(tr_r.cpu(), tr.cpu() @ Rm.T) < 2e-3 and rel_err(rot_r.cpu(), rot.cpu() @ Rm.T) < 2e-3
    assert rel_err(tor_r.cpu(), tor.cpu()) < 2e-3
    R = tor.numel() // B
    tr_s, rot_s, tor_s = run(dl[10:20])
    assert rel_err(tr_s.cpu(), tr[10:20].cpu()) < 1e-5 and rel_err(tor_s.cpu(), tor[10 * R:20 * R].cpu()) < 1e-5


def test_sharded_sampling_is_sample_invariant():
    """Counter-based noise keyed by sample id: sampling 8 poses at once == sampling them as 2 shards."""
    cfg = DDL_SYNTH.replace(num_conv_layers=3, tr_sigma_max=5.0, fixed_center_conv=True)
    sd = init_state_dict(cfg, seed=7)
    m = gpu_model(cfg, sd)
    g = make_complex(seed=9, n_res=120, n_lig=20)
    dl = make_pose_list(g, 8, tr_sigma_max=cfg.tr_sigma_max, seed=10, initial_noise_std_proportion=0.5)
    sched = get_t_schedule(5)
    full = m.sample_batch(to_gpu(HeteroBatch.from_data_list(dl)), 5, (sched, sched, sched), seed=123,
                          sample_ids=list(range(8)), no_final_step_noise=True).cpu().reshape(8, -1, 3)
    parts = []
    for lo in (0, 4):
        parts.append(m.sample_batch(to_gpu(HeteroBatch.from_data_list(dl[lo:lo + 4])), 5, (sched, sched, sched), seed=123,
                                    sample_ids=list(range(lo, lo + 4)), no_final_step_noise=True).cpu().reshape(4, -1, 3))
    dev = float((torch.cat(parts) - full).abs().max())
    print(f"sharded vs joint sampling, 5 steps: max |dx| = {dev:.3e} A")
    # not bit-identical: which 16 virtual nodes share a tile depends on the batch, and two things inside a tile depend on its
    # composition -- whether it takes the shared-node (4x4x1) contraction (<= 4 distinct gather nodes) and the order in which
    # the pre-reduction adds message rows of one target; both are re-associations of fp32 sums
    assert dev < 1e-4    # measured 7.6e-6 A after 5 steps (poses travel ~10 A): fp32 rounding level, 100x below the old 1e-3 bound


def test_ddl_synth_cropped_forward_matches_oracle():
    """crop_beyond at benchmark scale: one forward of the cropped batch (oracle crops the PyG-style batch on the host,
    the library masks residues and re-compacts the contact graph on the device)."""
    import copy
    from oracle.sampling import crop_beyond
    cfg = DDL_SYNTH
    sd = init_state_dict(cfg, seed=1234)
    g = make_complex(seed=6, n_res=300, n_lig=30)
    dl = make_pose_list(g, 2, tr_sigma_max=cfg.tr_sigma_max, seed=7, initial_noise_std_proportion=0.4)
    cutoff = 14.0
    cropped = [crop_beyond(copy.deepcopy(d), cutoff) for d in dl]
    assert all(0 < c["receptor"].pos.shape[0] < 300 for c in cropped)
    ob = HeteroBatch.from_data_list(cropped)
    set_time(ob, 0.2, 0.2, 0.2, 2)
    so3_t, tor_t = tables()
    tr, rot, tor, _ = CGModelOracle(cfg, sd, so3_t, tor_t)(ob)
    m = gpu_model(cfg, sd)
    batch = HeteroBatch.from_data_list(dl)
    set_time(batch, 0.2, 0.2, 0.2, 2)
    m.set_crop_cutoff(cutoff)
    tr2, rot2, tor2, _ = m(to_gpu(batch))
    m.set_crop_cutoff(None)
    assert int(m.debug_buffer("crop_keep").sum()) == sum(c["receptor"].pos.shape[0] for c in cropped)
    assert_scores_close((tr2, rot2, tor2), (tr, rot, tor))


def test_large_pocket_stress_config():
    """BASELINE configs[4]: 1500-residue receptor / 80-atom ligand, 40 poses, every ligand-atom x residue pair a cross
    edge (static 80 A cutoff): 4.8 M cross edges per direction.  No oracle at this size (it would need ~0.2 TB for the
    per-edge weights): finite outputs, exact edge count, determinism, and agreement of pose 0..1 with the same two
    poses evaluated alone (fixed_center_conv makes scores batch-independent)."""
    cfg = DDL_SYNTH.replace(dynamic_max_cross=False, cross_max_distance=80.0, fixed_center_conv=True)
    sd = init_state_dict(cfg, seed=1234)
    m = gpu_model(cfg, sd)
    B = 40
    g = make_complex(seed=8, n_res=1500, n_lig=80)
    dl = make_pose_list(g, B, tr_sigma_max=cfg.tr_sigma_max, seed=9, initial_noise_std_proportion=0.1)
    batch = HeteroBatch.from_data_list(dl)
    set_time(batch, 0.5, 0.5, 0.5, B)
    tr, rot, tor, _ = m(to_gpu(batch))
    assert torch.isfinite(tr).all() and torch.isfinite(rot).all() and torch.isfinite(tor).all()
    d = torch.cdist(batch["ligand"].pos.cpu().reshape(B, 80, 3), g["receptor"].pos.cpu()[None].expand(B, -1, -1))
    assert int(m.debug_buffer("offs_l")[-1]) == int((d < 80.0).sum())
    tr_b, rot_b, tor_b, _ = m(to_gpu(batch))
    assert torch.equal(tr, tr_b) and torch.equal(tor, tor_b)
    sb = HeteroBatch.from_data_list(dl[:2])
    set_time(sb, 0.5, 0.5, 0.5, 2)
    tr_s, rot_s, tor_s, _ = m(to_gpu(sb))
    R = tor.numel() // B
    assert rel_err(tr_s.cpu(), tr[:2].cpu()) < 1e-5 and rel_err(tor_s.cpu(), tor[:2 * R].cpu()) < 1e-5


def test_sampling_crops_confidence_graphs_like_the_reference():
    """sampling(..., confidence_model_args.crop_beyond) (utils/sampling.py:213-217): the confidence graphs are cropped around
    the FINAL poses (18 A here: 18 / 4 / 40 of 40 residues survive for the three poses) before the confidence model runs.
    Fixture = the reference's own sampling() + crop_beyond + both model classes executed (make_golden.py conf_crop)."""
    import argparse
    import copy
    from diffdock_amd.sampling import sampling
    fx = load_fixture("conf_crop")
    fs, cfg, data_list = fixture_case("tiny_l1")
    fc, ccfg, _ = fixture_case("tiny_conf_l2")
    score = gpu_model(cfg, fs["state_dict"])
    conf_model = MIScoreModel(ccfg, device="cuda:0")
    conf_model.load_state_dict(fc["state_dict"])
    B, R = len(data_list), int(data_list[0]["ligand"].edge_mask.sum())
    sched = get_t_schedule(fx["steps"])
    cargs = argparse.Namespace(crop_beyond=fx["crop_beyond"], all_atoms=False)
    out, conf = sampling(copy.deepcopy(data_list), score, fx["steps"], sched, sched, sched, model_args=cfg, confidence_model=conf_model,
                         confidence_data_list=copy.deepcopy(data_list), confidence_model_args=cargs, batch_size=B,
                         no_final_step_noise=True, noise=split_draws(fx["draws"], fx["steps"], B, R), device="cuda:0")
    pos = torch.stack([d["ligand"].pos.cpu() for d in out])
    assert (pos - fx["final_pos"]).abs().max() < 2e-3
    assert conf.shape == fx["confidence"].shape and rel_err(conf.cpu(), fx["confidence"]) < 1e-4


def test_tile_per_pose_shards_are_bit_identical_at_full_size():
    """ddmi_exec_options.tile_per_pose: BASELINE configs[2] / configs[3] shapes -- the 40-pose batch against its
    20 + 20 and 8 x 5 shards, torch.equal on every score, and the 5-step device loop against its shards position for position
    (noise keyed by global sample id).  fixed_center_conv: with the reference's default a pose's score depends on its graph id."""
    cfg = DDL_SYNTH.replace(fixed_center_conv=True, exec_options=(("tile_per_pose", 1),))
    sd = init_state_dict(cfg, seed=1234)
    m = gpu_model(cfg, sd)
    B = 40
    g = make_complex(seed=4, n_res=300, n_lig=30)
    dl = make_pose_list(g, B, tr_sigma_max=cfg.tr_sigma_max, seed=5, initial_noise_std_proportion=0.6)

    def run(lst):
        b = HeteroBatch.from_data_list(lst)
        set_time(b, 0.5, 0.5, 0.5, len(lst))
        return [o.clone() for o in m(to_gpu(b))[:3]]
    tr, rot, tor = run(dl)
    R = tor.numel() // B
    for n in (20, 5):
        for lo in range(0, B, n):
            tr_s, rot_s, tor_s = run(dl[lo:lo + n])
            assert torch.equal(tr_s, tr[lo:lo + n]) and torch.equal(rot_s, rot[lo:lo + n]) and torch.equal(tor_s, tor[lo * R:(lo + n) * R]), (n, lo)
    # same function as the dense-tile default, at rounding distance
    d = gpu_model(cfg.replace(exec_options=()), sd)
    b = HeteroBatch.from_data_list(dl)
    set_time(b, 0.5, 0.5, 0.5, B)
    for x, y in zip(d(to_gpu(b))[:3], (tr, rot, tor)):
        assert rel_err(x.cpu(), y.cpu()) < 1e-5
    sched = get_t_schedule(5)
    full = m.sample_batch(to_gpu(HeteroBatch.from_data_list(dl[:10])), 5, (sched, sched, sched), seed=123, sample_ids=list(range(10)),
                          no_final_step_noise=True).reshape(10, -1, 3)
    for lo in (0, 5):
        part = m.sample_batch(to_gpu(HeteroBatch.from_data_list(dl[lo:lo + 5])), 5, (sched, sched, sched), seed=123,
                              sample_ids=list(range(lo, lo + 5)), no_final_step_noise=True).reshape(5, -1, 3)
        assert torch.equal(part, full[lo:lo + 5]), lo


def test_layer_overlap_is_bit_identical_at_full_size():
    """ddmi_exec_options.layer_overlap: BASELINE configs[2] shapes (40 poses), the overlapped layer boundaries (1: the rule for chip-filling
    batches, 2: forced) against the joined form (all groups of a layer, then one node update) -- same kernels and arguments in another launch order on
    two streams, so every score and the 5-step device loop (per-step crop included) are equal bit for bit.  A missing dependency
    between the streams (a chain started before its rows were written) shows up here as a difference; repeated to give a race a
    second chance."""
    sd = init_state_dict(DDL_SYNTH, seed=1234)
    B = 40
    g = make_complex(seed=4, n_res=300, n_lig=30)
    dl = make_pose_list(g, B, tr_sigma_max=DDL_SYNTH.tr_sigma_max, seed=5, initial_noise_std_proportion=0.6)
    sched = get_t_schedule(5)
    res = {}
    for mode in (0, 1, 2):   # joined | overlapped by the size rule (on at this size) | always
        m = gpu_model(DDL_SYNTH.replace(exec_options=(("layer_overlap", mode),)), sd)
        outs = []
        for rep in range(3):
            b = HeteroBatch.from_data_list(dl)
            set_time(b, 0.5, 0.5, 0.5, B)
            outs.append([o.clone() for o in m(to_gpu(b))[:3]])
        for o in outs[1:]:
            assert all(torch.equal(x, y) for x, y in zip(o, outs[0])), mode
        traj = m.sample_batch(to_gpu(HeteroBatch.from_data_list(dl)), 5, (sched, sched, sched), seed=123, sample_ids=list(range(B)),
                              no_final_step_noise=True, crop_beyond=20.0).clone()
        res[mode] = (outs[0], traj)
    for mode in (1, 2):
        assert all(torch.equal(x, y) for x, y in zip(res[mode][0], res[0][0])), mode
        assert torch.equal(res[mode][1], res[0][1]), mode


def test_grouped_dispatch_is_bit_identical_at_full_size():
    """ddmi_exec_options.grouped: BASELINE configs[2] shapes (40 poses) and one GPU's share of configs[3] (5 poses) -- one
    k_conv_grouped launch per interaction layer (2, also with a forced granule-range split) against one k_conv_fused launch per
    (layer, edge group) on two streams (1): the same device code and arguments per work item, so every score and the 5-step
    device loop (per-step crop included) are equal bit for bit; the launch counts say which route ran."""
    sd = init_state_dict(DDL_SYNTH, seed=1234)
    g = make_complex(seed=4, n_res=300, n_lig=30)
    sched = get_t_schedule(5)
    for B in (40, 5):
        dl = make_pose_list(g, B, tr_sigma_max=DDL_SYNTH.tr_sigma_max, seed=5, initial_noise_std_proportion=0.6)
        res = {}
        for key, opts in (("per_group", (("grouped", 1),)), ("grouped", (("grouped", 2),)), ("grouped_ys3", (("grouped", 2), ("grouped_split", 3)))):
            m = gpu_model(DDL_SYNTH.replace(exec_options=opts), sd)
            outs = []
            for rep in range(2):
                b = HeteroBatch.from_data_list(dl)
                set_time(b, 0.5, 0.5, 0.5, B)
                if rep == 1:
                    m.set_kernel_timing(True)
                outs.append([o.clone() for o in m(to_gpu(b))[:3]])
            n_launch = m.kernel_timings()["k_conv_fused"][1]
            m.set_kernel_timing(False)
            # (the grouped kernel is exact-f32 only: under the split-bf16 edge product the layers keep their per-group launches)
            bf = os.environ.get("DDMI_EDGE_PRODUCT", "f32") != "f32"
            assert n_launch == (22 if key == "per_group" or bf else 6), (key, n_launch)
            assert all(torch.equal(x, y) for x, y in zip(outs[1], outs[0])), key
            traj = m.sample_batch(to_gpu(HeteroBatch.from_data_list(dl)), 5, (sched, sched, sched), seed=123, sample_ids=list(range(B)),
                                  no_final_step_noise=True, crop_beyond=20.0).clone()
            res[key] = (outs[0], traj)
        for key in ("grouped", "grouped_ys3"):
            assert all(torch.equal(x, y) for x, y in zip(res[key][0], res["per_group"][0])), (B, key)
            assert torch.equal(res[key][1], res["per_group"][1]), (B, key)
