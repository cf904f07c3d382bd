"""CPU run of the UNMODIFIED kernel sources (diffdock_amd/csrc/*.hip) under tests/hipemu against the
oracle and the reference-executed fixtures: graph construction, CSR layouts, MFMA fragment layouts, the
node-contracted tensor-product layer, read-outs, pose update and the device-resident step loop.
(GPU numerics and performance are covered by the `-m gpu` tests.)"""
import os
import subprocess

import numpy as np
import pytest
import torch

from diffdock_amd.hetero import HeteroBatch, set_time
from diffdock_amd.model import MIScoreModel
from oracle.cg_model import CGModelOracle
from oracle.conformer import get_t_schedule
from util import assert_scores_close, fixture_case, graph_from_dict, load_fixture, oracle_model, rel_err, split_draws, tables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "hipemu", "libddmi_emu.so")
CASES = ["tiny_l1", "tiny_l2", "tiny_l1_1group_emb", "tiny_l2_fixedcenter", "tiny_aa_l1", "tiny_aa_l2", "tiny_aa_l2_emb",   # tiny_aa_*: AAModel
         "tiny_noaa", "tiny_2nd", "tiny_aa_2nd",   # tiny_*2nd: use_second_order_repr (2e / 2o node blocks)
         "tiny_fourier", "tiny_tpw3",              # embedding_type='fourier'; tp_weights_layers=3
         "tiny_aa_emb_nolig",                      # AAModel: embedding layers without embed_also_ligand (zero-padded ligand rows)
         "tiny_oddpar", "tiny_aa_oddpar", "tiny_nobn_noscale",   # odd_parity (CG + all-atom); batch_norm off + scale_by_sigma off
         "tiny_sidechain",                                        # sidechain_pred: o3.Linear on the receptor rows, 4th tuple element
         "tiny_depthwise", "tiny_depthwise_l2"]                   # depthwise_convolution: 'uvu' TensorProduct + linear_2 (sh_lmax 1 and 2)


@pytest.fixture(scope="session")
def emu_lib():
    r = subprocess.run(["make", "-j8", "-C", os.path.join(ROOT, "diffdock_amd", "csrc"), "emu"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return EMU


def make_model(cfg, sd, emu_lib):
    m = MIScoreModel(cfg, device="cpu", lib_path=emu_lib)
    m.load_state_dict(sd)
    m.set_tables(*tables())
    return m


@pytest.mark.parametrize("name", CASES)
def test_forward_matches_reference_fixture(name, emu_lib):
    fx, cfg, data_list = fixture_case(name)
    m = make_model(cfg, fx["state_dict"], emu_lib)
    batch = HeteroBatch.from_data_list(data_list)
    set_time(batch, fx["t"], fx["t"], fx["t"], batch.num_graphs)
    tr, rot, tor, side = m(batch)
    ref = fx["forward"]
    if cfg.sidechain_pred:   # 4th tuple element (models/cg_model.py:397-402): [n_rec, 10]
        assert side.shape == ref["sidechain"].shape and rel_err(side, ref["sidechain"]) < 1e-4
    else:
        assert side is None
    assert_scores_close((tr, rot, tor), (ref["tr"], ref["rot"], ref["tor"]))
    if cfg.num_prot_emb_layers == 0:   # per-layer node tables (ligand + receptor rows)
        for l, ref_nodes in enumerate(ref["conv_out"]):
            mine = torch.from_numpy(m.debug_buffer(f"x{l + 1}"))
            n = ref_nodes.shape[0] if l < len(ref["conv_out"]) - 1 else batch["ligand"].pos.shape[0]
            assert rel_err(mine[:n, :ref_nodes.shape[1]], ref_nodes[:n]) < 1e-4, l
    # edge counts of the graphs built on the device against the oracle's graph builders
    so3_t, tor_t = tables()
    inter = oracle_model(cfg, fx["state_dict"], so3_t, tor_t)(batch, return_intermediates=True)[4]
    assert int(m.debug_buffer("goff_ll")[-1]) == inter["edge_counts"][0]
    assert int(m.debug_buffer("offs_l")[-1]) == inter["edge_counts"][1] == int(m.debug_buffer("offs_r")[-1])
    if cfg.all_atoms:   # ligand <-> atom radius graph built on the device, and really populated in the fixture
        assert int(m.debug_buffer("offs_la_l")[-1]) == inter["edge_counts"][2] == int(m.debug_buffer("offs_la_a")[-1]) > 0


@pytest.mark.parametrize("name", ["tiny_l1", "tiny_l2", "tiny_l2_crop", "tiny_aa_l1", "tiny_aa_l2", "tiny_aa_l2_emb", "tiny_2nd", "tiny_aa_2nd", "tiny_fourier",
                                  "tiny_tpw3", "tiny_aa_emb_nolig", "tiny_oddpar", "tiny_aa_oddpar", "tiny_nobn_noscale"])
def test_device_loop_matches_reference_trajectory(name, emu_lib):
    fx, cfg, data_list = fixture_case(name)
    m = make_model(cfg, fx["state_dict"], emu_lib)
    s = fx["sampling"]
    B, R = len(data_list), int(data_list[0]["ligand"].edge_mask.sum())
    noise = split_draws(s["draws"], s["steps"], B, R)
    sched = get_t_schedule(s["steps"])
    pos = m.sample_batch(HeteroBatch.from_data_list(data_list), s["steps"], (sched, sched, sched), noise=noise,
                         no_final_step_noise=True, crop_beyond=cfg.crop_beyond, **s["temp"])
    assert (pos.reshape(B, -1, 3) - s["final_pos"]).abs().max() < 2e-3   # Angstrom, 4 chaotic fp32 steps
    if cfg.crop_beyond is not None:   # the last step really cropped: fewer residues kept than present
        keep = m.debug_buffer("crop_keep")
        assert 0 < keep.sum() < keep.size


def test_modify_conformer_matches_reference(emu_lib):
    fx, cfg, _ = fixture_case("tiny_l1")
    u = load_fixture("units")
    m = make_model(cfg, fx["state_dict"], emu_lib)
    B = u["mc_tr"].shape[0]
    b = HeteroBatch.from_data_list([graph_from_dict(u["mc_graph"]) for _ in range(B)])
    out = m.modify_conformer_batch(u["mc_pos_in"], b, u["mc_tr"], u["mc_rot"], u["mc_tor"])
    assert (out - u["mc_pos_out"]).abs().max() < 5e-5
    rigid = m.modify_conformer_batch(u["mc_pos_in"], b, u["mc_tr"], u["mc_rot"], None)
    from oracle.conformer import modify_conformer_batch
    ref = modify_conformer_batch(u["mc_pos_in"], B, None, None, u["mc_tr"], u["mc_rot"], None)
    assert (rigid - ref).abs().max() < 5e-5


def test_no_cross_edges_and_ragged_batch(emu_lib):
    """Edge cases the reference handles (with e3nn tensor products): a ligand out of cross-graph range
    (empty cross groups) and a batch of two DIFFERENT complexes (ragged sizes)."""
    from diffdock_amd.config import TINY
    from diffdock_amd.synth import make_complex
    from diffdock_amd.weights import init_state_dict
    cfg = TINY.replace(sh_lmax=2)
    sd = init_state_dict(cfg, seed=5)
    g1, g2 = make_complex(seed=11, n_res=24, n_lig=9), make_complex(seed=12, n_res=31, n_lig=13)
    g1["ligand"].pos = g1["ligand"].pos + torch.tensor([[500.0, 0.0, 0.0]])   # far away: no cross edges for graph 0
    batch = HeteroBatch.from_data_list([g1, g2])
    set_time(batch, 0.4, 0.4, 0.4, 2)
    so3_t, tor_t = tables()
    ref = CGModelOracle(cfg, sd, so3_t, tor_t)(batch)
    m = make_model(cfg, sd, emu_lib)
    out = m(batch)
    for a, b in zip(out[:3], ref[:3]):
        assert rel_err(a, b) < 1e-4
    g2["ligand"].pos = g2["ligand"].pos + torch.tensor([[0.0, 700.0, 0.0]])    # now NO cross edges at all
    batch = HeteroBatch.from_data_list([g1, g2])
    set_time(batch, 0.4, 0.4, 0.4, 2)
    ref = CGModelOracle(cfg, sd, so3_t, tor_t)(batch)
    out = make_model(cfg, sd, emu_lib)(batch)
    assert int(m.debug_buffer("offs_l")[-1]) >= 0
    for a, b in zip(out[:3], ref[:3]):
        assert rel_err(a, b) < 1e-4


@pytest.mark.parametrize("no_torsion", [False, True])
def test_rigid_ligand_and_no_torsion_early_out(no_torsion, emu_lib):
    """cg_model.py:404: a ligand without rotatable bonds, or `no_torsion`, returns (tr, rot, empty(0), None); the device loop
    then runs the rigid-body update only (modify_conformer_batch with zero torsions)."""
    from diffdock_amd.config import TINY
    from diffdock_amd.synth import make_complex, make_pose_list
    from diffdock_amd.weights import init_state_dict
    cfg = TINY.replace(no_torsion=no_torsion)
    sd = init_state_dict(cfg, seed=5)
    g = make_complex(seed=11, n_res=20, n_lig=8)
    if not no_torsion:
        g["ligand"].edge_mask = torch.zeros_like(g["ligand"].edge_mask)
        g["ligand"].mask_rotate = [g["ligand"].mask_rotate[0][:0]]
    dl = make_pose_list(g, 2, tr_sigma_max=cfg.tr_sigma_max, seed=3, no_torsion=no_torsion)
    b = HeteroBatch.from_data_list(dl)
    set_time(b, 0.5, 0.5, 0.5, 2)
    so3_t, tor_t = tables()
    ref = CGModelOracle(cfg, sd, so3_t, tor_t)(b)
    m = make_model(cfg, sd, emu_lib)
    tr, rot, tor, none = m(b)
    assert none is None and ref[3] is None and tor.shape == (0,) and ref[2].shape == (0,)
    assert rel_err(tr, ref[0]) < 1e-4 and rel_err(rot, ref[1]) < 1e-4
    sched = get_t_schedule(3)
    start = b["ligand"].pos.clone()
    pos = m.sample_batch(b, 3, (sched, sched, sched), seed=1, no_final_step_noise=True)
    assert pos.shape == start.shape and torch.isfinite(pos).all()
    # rigid motion only: the intramolecular distances of every pose are those of the start conformer
    for k in range(2):
        a0, a1 = start.reshape(2, -1, 3)[k], pos.reshape(2, -1, 3)[k]
        assert (torch.cdist(a0, a0) - torch.cdist(a1, a1)).abs().max() < 1e-3


@pytest.mark.parametrize("n_res,n_lig,B", [(20, 8, 1), (3, 4, 2), (40, 2, 3)])
def test_degenerate_sizes(n_res, n_lig, B, emu_lib):
    """A batch of one pose, a 3-residue receptor (fewer neighbours than the 24-nearest graph asks for), a 2-atom ligand:
    forward against the oracle, and the device loop stays finite."""
    from diffdock_amd.config import TINY
    from diffdock_amd.synth import make_complex, make_pose_list
    from diffdock_amd.weights import init_state_dict
    cfg = TINY
    sd = init_state_dict(cfg, seed=5)
    g = make_complex(seed=11, n_res=n_res, n_lig=n_lig)
    b = HeteroBatch.from_data_list(make_pose_list(g, B, tr_sigma_max=cfg.tr_sigma_max, seed=3))
    set_time(b, 0.5, 0.5, 0.5, B)
    so3_t, tor_t = tables()
    ref = CGModelOracle(cfg, sd, so3_t, tor_t)(b)
    m = make_model(cfg, sd, emu_lib)
    out = m(b)
    for o, r in zip(out[:3], ref[:3]):
        assert o.shape == r.shape and (r.numel() == 0 or rel_err(o, r) < 1e-4)
    sched = get_t_schedule(3)
    pos = m.sample_batch(b, 3, (sched, sched, sched), seed=1, no_final_step_noise=True)
    assert pos.shape == b["ligand"].pos.shape and torch.isfinite(pos).all()


def test_errors_are_python_exceptions(emu_lib):
    from diffdock_amd.lib import DdmiError
    fx, cfg, data_list = fixture_case("tiny_l1")
    m = MIScoreModel(cfg, device="cpu", lib_path=emu_lib)
    sd = dict(fx["state_dict"])
    k = next(iter(sd))
    with pytest.raises(RuntimeError):
        m.load_state_dict({kk: v for kk, v in sd.items() if kk != k})           # strict: missing key
    with pytest.raises(RuntimeError):
        m.load_state_dict({**sd, "bogus.weight": torch.zeros(1)})              # strict: unexpected key
    bad = dict(sd)
    bad[k] = torch.zeros(3, 3)
    with pytest.raises(DdmiError):
        m.load_state_dict(bad)                                                  # shape mismatch
    assert set(m.expected_keys()) == set(sd)
    # a C caller that asks for class families the reference cannot build is refused at ddmi_create, whichever branch of the weight
    # spec it would take (the checks sit ahead of the early returns of the legacy / confidence branches)
    import ctypes as C
    from diffdock_amd import lib as L
    lib = L.load(emu_lib)
    for kw in (dict(old=True, depthwise_convolution=True, sh_lmax=2), dict(old=True, sidechain_pred=True, sh_lmax=2),
               dict(all_atoms=True, depthwise_convolution=True), dict(all_atoms=True, confidence_mode=True, depthwise_convolution=True)):
        c = L.make_config(cfg.replace(**kw))
        h = C.c_void_p()
        assert lib.ddmi_create(C.byref(c), 0, C.byref(h)) != 0, kw
        assert b"CG models of the new class only" in lib.ddmi_last_error()


def test_crop_with_embedding_layers_matches_oracle(emu_lib):
    """crop_beyond + receptor embedding layers: the reference re-embeds the CROPPED receptor each step."""
    from oracle.sampling import sampling as oracle_sampling
    fx, cfg, data_list = fixture_case("tiny_l1_1group_emb")
    cfg = cfg.replace(crop_beyond=9.0)
    so3_t, tor_t = tables()
    B, R = len(data_list), int(data_list[0]["ligand"].edge_mask.sum())
    g = torch.Generator().manual_seed(1)
    steps = 3
    noise = (torch.randn(steps, B, 3, generator=g), torch.randn(steps, B, 3, generator=g), torch.randn(steps, B * R, generator=g))
    ref = oracle_sampling([d.clone() for d in data_list], CGModelOracle(cfg, fx["state_dict"], so3_t, tor_t), steps, cfg, noise,
                          batch_size=B, no_final_step_noise=True)
    ref = torch.stack([d["ligand"].pos for d in ref])
    m = make_model(cfg, fx["state_dict"], emu_lib)
    sched = get_t_schedule(steps)
    pos = m.sample_batch(HeteroBatch.from_data_list(data_list), steps, (sched, sched, sched), noise=noise,
                         no_final_step_noise=True, crop_beyond=cfg.crop_beyond)
    keep = m.debug_buffer("crop_keep")
    assert 0 < keep.sum() < keep.size
    assert (pos.reshape(B, -1, 3) - ref).abs().max() < 2e-3

@pytest.mark.parametrize("lmax", [1, 2])
def test_fused_conv_full_width_matches_oracle(lmax, emu_lib, monkeypatch):
    """DDL-synth channel widths (ns=48, nv=10) on a small complex: the statically-shaped main loop of k_conv_fused
    (chain shapes (12,3,3,3)/(3,3,3,3)/(12,-,-,-) and the packed 12|3x3 granule of the second layer), the generic variant at
    sh_lmax=2, receptor residues with more than 32 ligand neighbours (two virtual nodes per residue), against the oracle, with
    the dense-row and the sparse-row loop."""
    from dataclasses import replace
    from diffdock_amd.config import DDL_SYNTH
    from diffdock_amd.synth import make_complex, make_pose_list
    from diffdock_amd.weights import init_state_dict
    cfg = replace(DDL_SYNTH, num_conv_layers=2, sh_lmax=lmax, lm_embedding_type=None, dynamic_max_cross=False,
                  cross_max_distance=80.0, tr_sigma_max=5.0)
    sd = init_state_dict(cfg, seed=3)
    g = make_complex(seed=1, n_res=12, n_lig=40, lm_dim=0)
    dl = make_pose_list(g, 2, tr_sigma_max=cfg.tr_sigma_max, seed=5, initial_noise_std_proportion=0.3)
    b = HeteroBatch.from_data_list(dl)
    set_time(b, 0.6, 0.6, 0.6, b.num_graphs)
    so3_t, tor_t = tables()
    ref = CGModelOracle(cfg, sd, so3_t, tor_t)(b)[:3]
    outs = {}
    for dense in ("1", "0"):          # dense-row and sparse-row loops (lmax 2: the generic, compiler-scheduled variant both times)
        monkeypatch.setenv("DDMI_FUSED_DENSE", dense)
        m = make_model(cfg, sd, emu_lib)
        m.set_kernel_timing(True)
        outs[dense] = m(b)[:3]
        assert "k_conv_fused" in m.kernel_timings()
        assert int(m.debug_buffer("vn_off_cross")[-1]) == 2 * b["receptor"].pos.shape[0]   # 40 neighbours -> 2 virtual nodes
        for o, r in zip(outs[dense], ref):
            assert rel_err(o, r) < 1e-4
    for a_, b_ in zip(outs["1"], outs["0"]):
        assert rel_err(a_, b_) < 1e-5
    if lmax == 1:   # the merged first-layer granule (three scalar channel tiles in one) against the separate granules
        monkeypatch.setenv("DDMI_FUSED_TRI", "0")
        sep = make_model(cfg, sd, emu_lib)(b)[:3]
        for a_, b_ in zip(outs["0"], sep):
            assert rel_err(a_, b_) < 1e-5


def test_packed_granules_match_oracle_and_classic_granules(emu_lib, monkeypatch, capfd):
    """Four interaction layers at the DDL-synth widths: from the third layer on the 10-channel vector blocks are fed by 6 or 7
    (path, component) slots and run as ONE packed granule each (12 | 3x3 | 3x3 in 5 column blocks, 3x3 | 3x3 in 4; the second
    layer's 12 | 3x3 in 3) instead of two classic granules whose message columns add up.  Against the oracle and against
    the classic granules (DDMI_FUSED_PACK=0)."""
    from dataclasses import replace
    from diffdock_amd.config import DDL_SYNTH
    from diffdock_amd.synth import make_complex, make_pose_list
    from diffdock_amd.weights import init_state_dict
    cfg = replace(DDL_SYNTH, num_conv_layers=4, lm_embedding_type=None, dynamic_max_cross=False, cross_max_distance=80.0, tr_sigma_max=5.0)
    sd = init_state_dict(cfg, seed=3)
    g = make_complex(seed=1, n_res=12, n_lig=20, lm_dim=0)
    b = HeteroBatch.from_data_list(make_pose_list(g, 2, tr_sigma_max=cfg.tr_sigma_max, seed=5, initial_noise_std_proportion=0.3))
    set_time(b, 0.6, 0.6, 0.6, b.num_graphs)
    ref = CGModelOracle(cfg, sd, *tables())(b)[:3]
    outs = {}
    monkeypatch.setenv("DDMI_DEBUG_GRAN", "1")
    for pack in ("1", "0"):
        monkeypatch.setenv("DDMI_FUSED_PACK", pack)
        capfd.readouterr()
        m = make_model(cfg, sd, emu_lib)
        listing = capfd.readouterr().err
        assert ("shape 4 slots 7 nb 5" in listing) == (pack == "1") and ("shape 5 slots 6 nb 4" in listing) == (pack == "1")
        assert ("shape 6 slots 4 nb 3" in listing) == (pack == "1") and (" acc]" in listing) == (pack == "0")
        assert "conv_layers.0: [shape 7 slots 3 nb 3 w 48] [shape 3" in listing   # first layer: the three scalar tiles as one granule
        outs[pack] = m(b)[:3]
        for o, r in zip(outs[pack], ref):
            assert rel_err(o, r) < 1e-4
    for a_, b_ in zip(outs["1"], outs["0"]):
        assert rel_err(a_, b_) < 1e-5


@pytest.mark.parametrize("ns", [16, 32])
def test_packing_is_dropped_in_layers_with_generic_granules(ns, emu_lib, monkeypatch, capfd):
    """ns = 16 / 32 with nv = 10: the 4- / 8-step scalar chains are outside the static shape set, so every layer with a scalar
    input path runs the predicated kernel variant, which walks classic 4-slot granules only.  Such a layer must not contain a
    packed granule (round-3 defect: its slots 4..6 were dropped, 1 % error in the 1e block); packing on / off must agree."""
    from dataclasses import replace
    from diffdock_amd.config import DDL_SYNTH
    from diffdock_amd.synth import make_complex, make_pose_list
    from diffdock_amd.weights import init_state_dict
    cfg = replace(DDL_SYNTH, ns=ns, nv=10, num_conv_layers=4, lm_embedding_type=None, dynamic_max_cross=False, cross_max_distance=80.0,
                  tr_sigma_max=5.0)
    sd = init_state_dict(cfg, seed=3)
    g = make_complex(seed=1, n_res=10, n_lig=12, lm_dim=0)
    b = HeteroBatch.from_data_list(make_pose_list(g, 2, tr_sigma_max=cfg.tr_sigma_max, seed=5, initial_noise_std_proportion=0.3))
    set_time(b, 0.6, 0.6, 0.6, b.num_graphs)
    ref = CGModelOracle(cfg, sd, *tables())(b)[:3]
    outs = {}
    monkeypatch.setenv("DDMI_DEBUG_GRAN", "1")
    for pack in ("1", "0"):
        monkeypatch.setenv("DDMI_FUSED_PACK", pack)
        capfd.readouterr()
        m = make_model(cfg, sd, emu_lib)
        for line in capfd.readouterr().err.splitlines():
            if line.startswith("ddmi granules") and "[shape 0 " in line:
                assert not any(f"[shape {s} " in line for s in (4, 5, 6, 7)), line
        outs[pack] = m(b)[:3]
        for o, r in zip(outs[pack], ref):
            assert rel_err(o, r) < 1e-5
    for a_, b_ in zip(outs["1"], outs["0"]):
        assert rel_err(a_, b_) < 1e-5


def test_shared_node_contraction_matches_oracle(emu_lib, monkeypatch):
    """Shared-node tiles of k_conv_fused (MODE 4: the x tile holds the distinct gather nodes of the 16 virtual nodes, classic
    granules contract them on the 4x4x1 MFMA, packed granules read their rows through the slot map), forced onto EVERY edge
    group (DDMI_FUSED_SHARED=2 with dense rows): tiles with 16 distinct nodes (four passes), tiles that mix nodes with one and
    several virtual nodes, the bias row, against the oracle and against the per-virtual-node form."""
    from dataclasses import replace
    from diffdock_amd.config import DDL_SYNTH
    from diffdock_amd.synth import make_complex, make_pose_list
    from diffdock_amd.weights import init_state_dict
    cfg = replace(DDL_SYNTH, num_conv_layers=4, lm_embedding_type=None, dynamic_max_cross=False, cross_max_distance=80.0, tr_sigma_max=5.0)
    sd = init_state_dict(cfg, seed=3)
    g = make_complex(seed=2, n_res=75, n_lig=7, lm_dim=0)     # 75 receptor neighbours per ligand atom: 32 + 32 + 11 edges
    b = HeteroBatch.from_data_list(make_pose_list(g, 2, tr_sigma_max=cfg.tr_sigma_max, seed=5, initial_noise_std_proportion=0.3))
    set_time(b, 0.6, 0.6, 0.6, b.num_graphs)
    ref = CGModelOracle(cfg, sd, *tables())(b)[:3]
    monkeypatch.setenv("DDMI_FUSED_DENSE", "2")
    outs = {}
    for shared in ("2", "1", "0"):
        monkeypatch.setenv("DDMI_FUSED_SHARED", shared)
        m = make_model(cfg, sd, emu_lib)
        outs[shared] = m(b)[:3]
        for o, r in zip(outs[shared], ref):
            assert rel_err(o, r) < 1e-4, shared
    for k in ("2", "1"):
        for a_, b_ in zip(outs[k], outs["0"]):
            assert rel_err(a_, b_) < 1e-5


def test_split_bf16_edge_product_matches_f32_route_and_oracle(emu_lib, monkeypatch):
    """ddmi_config.edge_product = 1 ("bf16x4"): the per-edge product of the static l <= 1 loops on v_mfma_f32_16x16x32_bf16 with
    both operands split into bf16 hi + bf16 lo (hidden rows and contracted chunks as packed words, four bf16 products per f32
    product, f32 accumulation).  DDL-synth widths, four layers (classic, merged and packed granules), shared-node tiles for the
    rec<-lig group, dense and sparse row loops: same 1e-4 bar against the oracle as the f32 route, and within 2e-5 of it."""
    from dataclasses import replace
    from diffdock_amd.config import DDL_SYNTH
    from diffdock_amd.synth import make_complex, make_pose_list
    from diffdock_amd.weights import init_state_dict
    from util import elem_excess
    cfg = replace(DDL_SYNTH, num_conv_layers=4, lm_embedding_type=None, dynamic_max_cross=False, cross_max_distance=80.0, tr_sigma_max=5.0)
    sd = init_state_dict(cfg, seed=3)
    g = make_complex(seed=2, n_res=75, n_lig=7, lm_dim=0)     # 75 receptor neighbours per ligand atom: shared-node tiles; 7 <= 16: sparse rows
    b = HeteroBatch.from_data_list(make_pose_list(g, 2, tr_sigma_max=cfg.tr_sigma_max, seed=5, initial_noise_std_proportion=0.3))
    set_time(b, 0.6, 0.6, 0.6, b.num_graphs)
    ref = CGModelOracle(cfg, sd, *tables())(b)[:3]
    f32 = make_model(cfg, sd, emu_lib)(b)[:3]
    bf = make_model(cfg.replace(edge_product="bf16x4"), sd, emu_lib)(b)[:3]
    for o, f, r in zip(bf, f32, ref):
        assert rel_err(o, r) < 1e-4 and elem_excess(o, r) <= 1.0
        assert 0 < rel_err(o, f) < 2e-5      # another arithmetic route (not bit-equal), at fp32-level distance
    monkeypatch.setenv("DDMI_EDGE_PRODUCT", "bf16x4")          # the environment override of the test harness selects the same route
    env = make_model(cfg, sd, emu_lib)(b)[:3]
    for o, e in zip(bf, env):
        assert torch.equal(o, e)
    monkeypatch.setenv("DDMI_EDGE_PRODUCT", "fp8")
    from diffdock_amd.lib import DdmiError
    with pytest.raises(DdmiError):
        make_model(cfg, sd, emu_lib)


def test_in_tile_pre_reduction_of_lig_rec_messages(emu_lib, monkeypatch):
    """lig<-rec group: the 16 residues of a tile send to the same <= 32 ligand atoms, so a tile sums its message rows per target in
    LDS (per wave, then the eight partial sums in wave order) and ONE row per (tile, target) leaves it; k_reduce_bn reads only the
    rows flagged live (tensor_layers.py:144,220-221: the scatter-mean itself is unchanged -- counts are the true edge counts).
    3 poses x 12 residues x 20 atoms: tiles 0 and 1 straddle two poses (targets span 40 rows: one row per edge as before), tile 2
    is pre-reduced.  Against the oracle and against the per-edge route (DDMI_FUSED_PRERED=0)."""
    from dataclasses import replace
    from diffdock_amd.config import DDL_SYNTH
    from diffdock_amd.synth import make_complex, make_pose_list
    from diffdock_amd.weights import init_state_dict
    cfg = replace(DDL_SYNTH, num_conv_layers=4, lm_embedding_type=None, dynamic_max_cross=False, cross_max_distance=80.0, tr_sigma_max=5.0)
    sd = init_state_dict(cfg, seed=3)
    g = make_complex(seed=1, n_res=12, n_lig=20, lm_dim=0)
    b = HeteroBatch.from_data_list(make_pose_list(g, 3, tr_sigma_max=cfg.tr_sigma_max, seed=5, initial_noise_std_proportion=0.3))
    set_time(b, 0.6, 0.6, 0.6, b.num_graphs)
    ref = CGModelOracle(cfg, sd, *tables())(b)[:3]
    outs = {}
    for pre in ("1", "0"):
        monkeypatch.setenv("DDMI_FUSED_PRERED", pre)
        m = make_model(cfg, sd, emu_lib)
        outs[pre] = m(b)[:3]
        if pre == "1":
            hdr = m.debug_buffer("prered_tile_hdr")
            assert hdr[:3, 0].tolist() == [0, 0, 1] and hdr[2, 1:3].tolist() == [40, 20]   # (mode, first target row, span)
            assert (hdr[2, 4:24] >= 0).all() and (hdr[2, 24:36] == -1).all()            # one message row per target of the tile
        for o, r in zip(outs[pre], ref):
            assert rel_err(o, r) < 1e-4
    for a_, b_ in zip(outs["1"], outs["0"]):
        assert rel_err(a_, b_) < 1e-5


@pytest.mark.parametrize("name", ["tiny_l1", "tiny_l2"])
def test_readout_tensor_product_forms_agree(name, emu_lib, monkeypatch):
    """final_conv / tor_bond_conv in the direct (per-edge-weight) form: the wave-per-item, thread-per-item and
    workgroup-per-edge kernels (k_readout.hip; picked by launch size in production, forced here) against the reference fixture."""
    fx, cfg, data_list = fixture_case(name)
    batch = HeteroBatch.from_data_list(data_list)
    set_time(batch, fx["t"], fx["t"], fx["t"], batch.num_graphs)
    ref = fx["forward"]
    for form in ("edge", "thread", "wave"):
        monkeypatch.setenv("DDMI_TP_APPLY", form)
        m = make_model(cfg, fx["state_dict"], emu_lib)
        tr, rot, tor, _ = m(batch)
        assert_scores_close((tr, rot, tor), (ref["tr"], ref["rot"], ref["tor"]), what=form)


def test_ligand_atoms_with_many_receptor_neighbours(emu_lib):
    """Ligand-gather groups through the fused kernel with several virtual nodes per ligand atom (70 receptor neighbours ->
    32 + 32 + 6 edges: the node term is repeated per virtual node, the last one is a sparse tile) at a width the MFMA first
    layer and the dense-row loop accept (ns = 16), against the oracle."""
    from diffdock_amd.config import TINY
    from diffdock_amd.synth import make_complex, make_pose_list
    from diffdock_amd.weights import init_state_dict
    cfg = TINY.replace(ns=16, nv=4, sh_lmax=1, num_conv_layers=3, dynamic_max_cross=False, cross_max_distance=200.0,
                       lm_embedding_type=None)
    sd = init_state_dict(cfg, seed=9)
    g = make_complex(seed=21, n_res=70, n_lig=5, lm_dim=0)
    b = HeteroBatch.from_data_list(make_pose_list(g, 2, tr_sigma_max=cfg.tr_sigma_max, seed=4))
    set_time(b, 0.5, 0.5, 0.5, 2)
    so3_t, tor_t = tables()
    ref = CGModelOracle(cfg, sd, so3_t, tor_t)(b)[:3]
    m = make_model(cfg, sd, emu_lib)
    m.set_kernel_timing(True)
    out = m(b)[:3]
    launched = m.kernel_timings()
    assert "k_conv_fused" in launched
    assert int(m.debug_buffer("vn_off_rl")[-1]) == 3 * b["ligand"].pos.shape[0]     # ceil(70 / 32) virtual nodes per ligand atom
    assert int(m.debug_buffer("vn_off_cross")[-1]) == b["receptor"].pos.shape[0]    # 5 ligand neighbours: one sparse tile each
    for o, r in zip(out, ref):
        assert rel_err(o, r) < 1e-4


@pytest.mark.parametrize("name", ["tiny_conf_l2", "tiny_conf_aa_l1", "tiny_conf_atom"])
def test_confidence_mode_matches_reference_fixture(name, emu_lib):
    """get_model(..., confidence_mode=True) for CGModel / AAModel (cg_model.py:353-366), fixture from the reference."""
    fx, cfg, data_list = fixture_case(name)
    m = MIScoreModel(cfg, device="cpu", lib_path=emu_lib)
    m.load_state_dict(fx["state_dict"])
    batch = HeteroBatch.from_data_list(data_list)
    set_time(batch, fx["t"], fx["t"], fx["t"], batch.num_graphs)
    conf, atom_conf = m(batch)
    ref = fx["forward"]
    assert conf.shape == ref["confidence"].shape and rel_err(conf, ref["confidence"]) < 1e-4
    assert atom_conf.shape == ref["atom_confidence"].shape
    if cfg.atom_confidence:   # per-atom predictor in front of the graph mean (+ the affinity column of the graph predictor)
        assert rel_err(atom_conf, ref["atom_confidence"]) < 1e-4
    else:
        assert not atom_conf.any()


@pytest.mark.parametrize("name", ["tiny_oldconf", "tiny_oldconf_2l"])
def test_legacy_confidence_class_matches_reference_fixture(name, emu_lib, monkeypatch):
    """models/old_cg_model.py in confidence mode (get_model(old=True)): OldAtomEncoder, four separately normalised layers per
    interaction layer, the swapped [edge, gather, target] input of the lig->rec layer."""
    fx, cfg, data_list = fixture_case(name)
    batch = HeteroBatch.from_data_list(data_list)
    set_time(batch, fx["t"], fx["t"], fx["t"], batch.num_graphs)
    m = MIScoreModel(cfg, device="cpu", lib_path=emu_lib)
    m.load_state_dict(fx["state_dict"])
    conf = m(batch)
    assert torch.is_tensor(conf) and conf.shape == fx["forward"]["confidence"].shape
    assert rel_err(conf, fx["forward"]["confidence"]) < 1e-4


def test_legacy_class_score_mode_matches_reference_fixture(emu_lib, monkeypatch):
    """models/old_cg_model.py in score mode (get_model(old=True, confidence_mode=False)): 3-tuple scores and the device loop
    against the reference-executed fixture."""
    fx, cfg, data_list = fixture_case("tiny_oldscore")
    s = fx["sampling"]
    B, R = len(data_list), int(data_list[0]["ligand"].edge_mask.sum())
    m = make_model(cfg, fx["state_dict"], emu_lib)
    batch = HeteroBatch.from_data_list(data_list)
    set_time(batch, fx["t"], fx["t"], fx["t"], batch.num_graphs)
    out = m(batch)
    assert len(out) == 3
    for mine, key in zip(out, ("tr", "rot", "tor")):
        assert mine.shape == fx["forward"][key].shape and rel_err(mine, fx["forward"][key]) < 1e-4, key
    sched = get_t_schedule(s["steps"])
    pos = m.sample_batch(HeteroBatch.from_data_list(data_list), s["steps"], (sched, sched, sched),
                         noise=split_draws(s["draws"], s["steps"], B, R), no_final_step_noise=True, **s["temp"])
    assert (pos.reshape(B, -1, 3) - s["final_pos"]).abs().max() < 2e-3


def test_sampling_calls_confidence_model(emu_lib):
    """sampling(..., confidence_model=...) (utils/sampling.py:208-231): confidences of the final poses, both with separate
    confidence graphs (t = 0) and on the sampling batch itself (last step's t), against the oracle on the returned poses."""
    import copy
    from diffdock_amd.sampling import sampling
    fx, cfg, data_list = fixture_case("tiny_l2")
    fc, ccfg, _ = fixture_case("tiny_conf_l2")
    score = make_model(cfg, fx["state_dict"], emu_lib)
    conf_model = MIScoreModel(ccfg, device="cpu", lib_path=emu_lib)
    conf_model.load_state_dict(fc["state_dict"])
    sched = get_t_schedule(3)
    for own_graphs, t_conf in ((True, 0.0), (False, float(sched[-1]))):
        out, conf = sampling(copy.deepcopy(data_list), score, 3, sched, sched, sched, model_args=cfg, confidence_model=conf_model,
                             confidence_data_list=copy.deepcopy(data_list) if own_graphs else None, batch_size=2,
                             no_final_step_noise=True, seed=3)
        b = HeteroBatch.from_data_list(out)
        set_time(b, t_conf, t_conf, t_conf, b.num_graphs)
        assert conf.shape == (len(data_list), ccfg.num_confidence_outputs)
        assert rel_err(conf, oracle_model(ccfg, fc["state_dict"])(b)[0]) < 1e-4


def test_sampling_crops_confidence_graphs_like_the_reference(emu_lib):
    """sampling(..., confidence_model_args.crop_beyond) (utils/sampling.py:213-217): the confidence graphs are cropped around
    the FINAL poses (18 A here: 18 / 4 / 40 of 40 residues survive for the three poses) before the confidence model runs.
    Fixture = the reference's own sampling() + crop_beyond + both model classes executed (make_golden.py conf_crop)."""
    import argparse
    import copy
    from diffdock_amd.sampling import sampling
    fx = load_fixture("conf_crop")
    fs, cfg, data_list = fixture_case("tiny_l1")
    fc, ccfg, _ = fixture_case("tiny_conf_l2")
    score = make_model(cfg, fs["state_dict"], emu_lib)
    conf_model = MIScoreModel(ccfg, device="cpu", lib_path=emu_lib)
    conf_model.load_state_dict(fc["state_dict"])
    B, R = len(data_list), int(data_list[0]["ligand"].edge_mask.sum())
    sched = get_t_schedule(fx["steps"])
    cargs = argparse.Namespace(crop_beyond=fx["crop_beyond"], all_atoms=False)
    out, conf = sampling(copy.deepcopy(data_list), score, fx["steps"], sched, sched, sched, model_args=cfg, confidence_model=conf_model,
                         confidence_data_list=copy.deepcopy(data_list), confidence_model_args=cargs, batch_size=B,
                         no_final_step_noise=True, noise=split_draws(fx["draws"], fx["steps"], B, R))
    pos = torch.stack([d["ligand"].pos.cpu() for d in out])
    assert (pos - fx["final_pos"]).abs().max() < 2e-3
    assert conf.shape == fx["confidence"].shape and rel_err(conf.cpu(), fx["confidence"]) < 1e-4


def test_sampling_scores_the_other_poses_when_one_confidence_graph_is_fully_cropped(emu_lib):
    """One pose farther than crop_beyond from every residue of ITS confidence graph (here: that graph's receptor moved 1000 A away)
    must not cost the whole batch its confidence scores: that pose gets -1000 (what nan_to_num gives a failed pose,
    utils/sampling.py:229), the others the scores they get without it; only a batch with no scorable pose raises."""
    import argparse
    import copy
    from diffdock_amd.sampling import sampling
    fx = load_fixture("conf_crop")
    fs, cfg, data_list = fixture_case("tiny_l1")
    fc, ccfg, _ = fixture_case("tiny_conf_l2")
    score = make_model(cfg, fs["state_dict"], emu_lib)
    conf_model = MIScoreModel(ccfg, device="cpu", lib_path=emu_lib)
    conf_model.load_state_dict(fc["state_dict"])
    B, R = len(data_list), int(data_list[0]["ligand"].edge_mask.sum())
    sched = get_t_schedule(fx["steps"])
    cargs = argparse.Namespace(crop_beyond=fx["crop_beyond"], all_atoms=False)

    def run(conf_list):
        return sampling(copy.deepcopy(data_list), score, fx["steps"], sched, sched, sched, model_args=cfg, confidence_model=conf_model,
                        confidence_data_list=conf_list, confidence_model_args=cargs, batch_size=B, no_final_step_noise=True,
                        noise=split_draws(fx["draws"], fx["steps"], B, R))[1]
    full = run(copy.deepcopy(data_list))
    far = copy.deepcopy(data_list)
    far[1]["receptor"].pos = far[1]["receptor"].pos + 1000.0
    conf = run(far)
    assert conf.shape == full.shape
    assert bool((conf[1] == -1000).all())
    keep = [i for i in range(B) if i != 1]
    assert rel_err(conf[keep].cpu(), full[keep].cpu()) < 1e-4
    for g in far:
        g["receptor"].pos = g["receptor"].pos + 1000.0
    with pytest.raises(ValueError, match="every pose"):
        run(far)


def test_sidechain_pred_under_a_device_crop_and_after_other_passes(emu_lib):
    """sidechain_pred with crop_beyond: the reference crops the graph first and returns rows for the KEPT residues only
    (utils/utils.py:388-413, models/cg_model.py:397-402); the library's node table still holds every residue, so
    MIScoreModel.__call__ compacts the rows through the device's crop mask.  And ddmi_sidechain_pred belongs to the ddmi_forward
    directly before it: after a sampling loop on the same handle it raises instead of reading that pass's table."""
    import copy
    from diffdock_amd.lib import DdmiError
    from oracle.sampling import crop_beyond
    fs, cfg, data_list = fixture_case("tiny_sidechain")
    m = make_model(cfg, fs["state_dict"], emu_lib)
    d = torch.cdist(data_list[0]["ligand"].pos, data_list[0]["receptor"].pos).min(0).values
    cutoff = float(d.sort().values[len(d) // 2]) + 1e-3      # about half of the residues of pose 0 survive
    cropped = [crop_beyond(copy.deepcopy(g), cutoff) for g in data_list]
    n_keep = sum(int(c["receptor"].pos.shape[0]) for c in cropped)
    assert 0 < n_keep < sum(int(g["receptor"].pos.shape[0]) for g in data_list)
    ob = HeteroBatch.from_data_list(cropped)
    set_time(ob, 0.4, 0.4, 0.4, ob.num_graphs)
    ref = oracle_model(cfg, fs["state_dict"])(ob)
    batch = HeteroBatch.from_data_list(data_list)
    set_time(batch, 0.4, 0.4, 0.4, batch.num_graphs)
    m.set_crop_cutoff(cutoff)
    out = m(batch)
    m.set_crop_cutoff(None)
    assert out[3].shape == ref[3].shape == (n_keep, 10)
    assert rel_err(out[3], ref[3]) < 1e-4
    for o, r in zip(out[:3], ref[:3]):
        assert rel_err(o, r) < 1e-4
    # a sampling loop in between: the table of its last step is not what ddmi_sidechain_pred may read
    sched = get_t_schedule(2)
    m.sample_batch(batch, 2, (sched, sched, sched), seed=1, sample_ids=list(range(batch.num_graphs)), no_final_step_noise=True)
    side = torch.empty(int(batch["receptor"].pos.shape[0]), 10)
    with pytest.raises(DdmiError):
        from diffdock_amd import lib as _l
        _l.check(m.lib, m.lib.ddmi_sidechain_pred(m._h, side.data_ptr(), None))


def test_all_atom_ragged_batch_and_empty_ligand_atom_group(emu_lib):
    """AAModel on a batch of two DIFFERENT complexes (residue / atom / ligand counts differ), first with one ligand out of
    reach of every receptor atom, then with the ligand<->atom group completely empty (the reference's FasterTensorProduct
    cannot run that; sh_lmax = 2 can), against the oracle."""
    from diffdock_amd.config import TINY
    from diffdock_amd.synth import make_complex, make_pose_list
    from diffdock_amd.weights import init_state_dict
    cfg = TINY.replace(all_atoms=True, sh_lmax=2, num_conv_layers=3, dynamic_max_cross=False, cross_max_distance=60.0)
    sd = init_state_dict(cfg, seed=2)
    g1 = make_complex(seed=31, n_res=14, n_lig=7, all_atoms=True, atoms_per_res=(2, 5))
    g2 = make_complex(seed=32, n_res=19, n_lig=11, all_atoms=True, atoms_per_res=(2, 5))
    d1 = make_pose_list(g1, 1, tr_sigma_max=5.0, seed=1, initial_noise_std_proportion=0.05)[0]
    d2 = make_pose_list(g2, 1, tr_sigma_max=5.0, seed=2, initial_noise_std_proportion=0.05)[0]
    d2["ligand"].pos = d2["ligand"].pos + torch.tensor([30.0, 0.0, 0.0])
    m = make_model(cfg, sd, emu_lib)
    for empty in (False, True):
        if empty:
            d1["ligand"].pos = d1["ligand"].pos + torch.tensor([0.0, 40.0, 0.0])
        batch = HeteroBatch.from_data_list([d1, d2])
        set_time(batch, 0.5, 0.5, 0.5, 2)
        ref = oracle_model(cfg, sd)(batch, return_intermediates=True)
        assert (ref[4]["edge_counts"][2] == 0) == empty
        out = m(batch)
        assert int(m.debug_buffer("offs_la_l")[-1]) == ref[4]["edge_counts"][2]
        for o, r in zip(out[:3], ref[:3]):
            assert rel_err(o, r) < 1e-4


def test_tile_per_pose_makes_shards_bit_identical(emu_lib):
    """ddmi_exec_options.tile_per_pose: every graph of the batch padded to whole 16-virtual-node tiles, so that which virtual
    nodes share a tile of k_conv_fused -- hence whether the tile takes the shared-node (4x4x1) contraction and in which order
    the pre-reduction adds the rows of a target -- depends on the pose alone.  A pose evaluated alone, in a batch of 2 and in
    a batch of 3 gives BIT-identical scores; with dense tiles (the default) the same comparison is only
    rounding-level close.  DDL-synth widths; 21 residues x 12 atoms: 21 virtual nodes per pose in the residue-gather groups, so
    without the option tiles straddle poses.  Also one step of the device loop, position for position."""
    from dataclasses import replace
    from diffdock_amd.config import DDL_SYNTH
    from diffdock_amd.synth import make_complex, make_pose_list
    from diffdock_amd.weights import init_state_dict
    cfg = replace(DDL_SYNTH, num_conv_layers=3, lm_embedding_type=None, dynamic_max_cross=False, cross_max_distance=80.0, tr_sigma_max=5.0,
                  fixed_center_conv=True)     # (the default centre convolution indexes the ligand table by graph id, cg_model.py:371-374)
    sd = init_state_dict(cfg, seed=3)
    g = make_complex(seed=2, n_res=21, n_lig=12, lm_dim=0)
    dl = make_pose_list(g, 3, tr_sigma_max=cfg.tr_sigma_max, seed=5, initial_noise_std_proportion=0.3)
    assert int(dl[0]["ligand"].edge_mask.sum()) > 0

    def scores(model, lst):
        b = HeteroBatch.from_data_list(lst)
        set_time(b, 0.6, 0.6, 0.6, b.num_graphs)
        return [o.clone() for o in model(b)[:3]]
    pp = make_model(cfg.replace(exec_options=(("tile_per_pose", 1),)), sd, emu_lib)
    full = scores(pp, dl)
    R = full[2].numel() // 3
    for lo, hi in ((0, 1), (1, 3)):
        part = scores(pp, dl[lo:hi])
        assert torch.equal(part[0], full[0][lo:hi]) and torch.equal(part[1], full[1][lo:hi]) and torch.equal(part[2], full[2][lo * R:hi * R])
    # same function as the default route (dense tiles), at rounding distance
    dense = scores(make_model(cfg, sd, emu_lib), dl)
    for a_, b_ in zip(full, dense):
        assert rel_err(a_, b_) < 1e-5
    b3 = HeteroBatch.from_data_list(dl)
    set_time(b3, 0.6, 0.6, 0.6, 3)
    for o, r in zip(full, CGModelOracle(cfg, sd, *tables())(b3)[:3]):
        assert rel_err(o, r) < 1e-4
    # device loop: counter-based noise keyed by sample id + per-pose tiles -> identical trajectories
    sched = get_t_schedule(1)
    pos = pp.sample_batch(HeteroBatch.from_data_list(dl), 1, (sched, sched, sched), seed=11, sample_ids=[0, 1, 2], no_final_step_noise=True).reshape(3, -1, 3)
    one = pp.sample_batch(HeteroBatch.from_data_list(dl[1:2]), 1, (sched, sched, sched), seed=11, sample_ids=[1], no_final_step_noise=True).reshape(1, -1, 3)
    assert torch.equal(one[0], pos[1])


def test_layer_overlap_is_bit_identical(emu_lib):
    """ddmi_exec_options.layer_overlap: the interaction layers with the node update split by node type and every group's chain
    started as soon as its rows exist (run_conv_layers_overlapped) launch the SAME kernels
    with the SAME arguments as the joined form (run_conv) -- only the order differs, so the scores are bit-identical.  Forced on
    (2) against off (0) on a small batch: forward scores with sidechain_pred (the last layer then reduces every row) and one step
    of the device loop with the per-step receptor crop (its own reduce-group list).  (The emulator runs streams in launch order:
    this covers the data flow; the stream dependencies are covered on the GPU, test_layer_overlap_is_bit_identical_at_full_size.)"""
    from dataclasses import replace
    from diffdock_amd.config import DDL_SYNTH
    from diffdock_amd.synth import make_complex, make_pose_list
    from diffdock_amd.weights import init_state_dict
    g = make_complex(seed=4, n_res=16, n_lig=10, lm_dim=0)
    dl = make_pose_list(g, 2, tr_sigma_max=5.0, seed=6, initial_noise_std_proportion=0.3)
    sched = get_t_schedule(1)
    cfg = replace(DDL_SYNTH, num_conv_layers=3, lm_embedding_type=None, dynamic_max_cross=False, cross_max_distance=80.0, tr_sigma_min=0.1,
                  tr_sigma_max=0.5, sidechain_pred=True)     # (per-step crop radius = 3 sigma_tr(t) + crop_beyond = 4.5 A at t = 1)
    sd = init_state_dict(cfg, seed=3)
    outs, traj = [], []
    for mode in (0, 2):
        m = make_model(cfg.replace(exec_options=(("layer_overlap", mode),)), sd, emu_lib)
        b = HeteroBatch.from_data_list(dl)
        set_time(b, 0.6, 0.6, 0.6, b.num_graphs)
        outs.append([o.clone() for o in m(b)])
        traj.append(m.sample_batch(HeteroBatch.from_data_list(dl), 1, (sched, sched, sched), seed=11, sample_ids=[0, 1],
                                   no_final_step_noise=True, crop_beyond=3.0).clone())
        keep = m.debug_buffer("crop_keep")
        assert 0 < keep.sum() < keep.size
    assert len(outs[0]) == len(outs[1]) == 4
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)
    assert torch.equal(traj[0], traj[1])


def test_grouped_dispatch_is_bit_identical(emu_lib):
    """ddmi_exec_options.grouped (round 6): per interaction layer ONE launch of the first-Linear node terms, ONE of the hidden rows
    (each edge group into its own buffer) and ONE k_conv_grouped walking the (edge group, tile, granule range) work items of every
    group -- the same device code and arguments per work item as the per-group k_conv_fused launches, so the scores are
    bit-identical: grouped (2) against per-group launches (1), with two granule-range splits, on the CG model (4 groups, sidechain
    rows, one step of the device loop under the per-step crop) and the all-atom model (9 groups = three grouped launches)."""
    from dataclasses import replace
    from diffdock_amd.config import DDL_SYNTH
    from diffdock_amd.synth import make_complex, make_pose_list
    from diffdock_amd.weights import init_state_dict
    sched = get_t_schedule(1)
    for all_atoms in (False, True):
        g = make_complex(seed=4, n_res=16, n_lig=10, lm_dim=0, all_atoms=all_atoms, **({"atoms_per_res": (2, 4)} if all_atoms else {}))
        dl = make_pose_list(g, 2, tr_sigma_max=5.0, seed=6, initial_noise_std_proportion=0.3)
        cfg = replace(DDL_SYNTH, num_conv_layers=2 if all_atoms else 3, lm_embedding_type=None, dynamic_max_cross=False, cross_max_distance=80.0,
                      tr_sigma_min=0.1, tr_sigma_max=0.5, sidechain_pred=not all_atoms, all_atoms=all_atoms)
        sd = init_state_dict(cfg, seed=3)
        outs, traj, launches = [], [], []
        # (a forced granule-range split of the grouped launch: GPU route test, test_grouped_dispatch_is_bit_identical_at_full_size)
        for opts in ((("grouped", 1),), (("grouped", 2), ("grouped_split", 0 if all_atoms else 3))):
            m = make_model(cfg.replace(exec_options=opts), sd, emu_lib)
            b = HeteroBatch.from_data_list(dl)
            set_time(b, 0.6, 0.6, 0.6, b.num_graphs)
            m.set_kernel_timing(True)
            outs.append([o.clone() for o in m(b) if o is not None])
            launches.append(m.kernel_timings()["k_conv_fused"][1])
            m.set_kernel_timing(False)
            traj.append(None if all_atoms else m.sample_batch(HeteroBatch.from_data_list(dl), 1, (sched, sched, sched), seed=11, sample_ids=[0, 1],
                                                              no_final_step_noise=True, crop_beyond=3.0).clone())
        # per-group: one launch per (layer, group); grouped: one per layer (all-atom: nine groups in chunks of four)
        assert launches[0] == (9 + 3 if all_atoms else 4 + 4 + 2)
        assert all(n == (3 + 1 if all_atoms else 3) for n in launches[1:])
        for k in range(1, len(outs)):
            assert len(outs[0]) == len(outs[k])
            for a_, b_ in zip(outs[0], outs[k]):
                assert torch.equal(a_, b_)
            assert all_atoms or torch.equal(traj[0], traj[k])


def test_fused_node_update_matches_separate_launches(emu_lib):
    """ddmi_exec_options.node_update = 1: k_node_update (a layer's node rows AND the next layer's per-node first-Linear terms P / Q in
    one kernel, the per-graph sigma terms of every layer from one batched launch) against k_reduce_bn + k_gemm_nt_batch launches:
    the node tables are the same sums in the same order -- layer 1's table, which no fused P / Q has touched yet, is bit-identical --
    and the scores agree at rounding level (P / Q take a 48-term fp32 sum in another order) and with the oracle.  With the per-step
    crop (its own reduce-group list), sidechain rows (the last layer reduces every row), a ragged batch of two complexes."""
    from dataclasses import replace
    from diffdock_amd.config import DDL_SYNTH
    from diffdock_amd.synth import make_complex, make_pose_list
    from diffdock_amd.weights import init_state_dict
    cfg = replace(DDL_SYNTH, num_conv_layers=3, lm_embedding_type=None, dynamic_max_cross=False, cross_max_distance=80.0, tr_sigma_min=0.1,
                  tr_sigma_max=0.5, sidechain_pred=True)
    sd = init_state_dict(cfg, seed=3)
    g1 = make_complex(seed=4, n_res=19, n_lig=10, lm_dim=0)
    g2 = make_complex(seed=5, n_res=13, n_lig=7, lm_dim=0)
    dl = make_pose_list(g1, 2, tr_sigma_max=5.0, seed=6, initial_noise_std_proportion=0.3) + make_pose_list(g2, 1, tr_sigma_max=5.0, seed=7, initial_noise_std_proportion=0.3)
    sched = get_t_schedule(1)
    res = {}
    # (fused = four waves per node at this size; fused16 = the sixteen-nodes-per-workgroup shape of chip-filling batches; fused + grouped
    # dispatch: GPU route test)
    for key, opts in (("separate", ()), ("fused", (("node_update", 1),)), ("fused16", (("node_update", 2),))):
        m = make_model(cfg.replace(exec_options=opts), sd, emu_lib)
        b = HeteroBatch.from_data_list(dl)
        set_time(b, 0.6, 0.6, 0.6, b.num_graphs)
        m.set_kernel_timing(True)
        out = [o.clone() for o in m(b)]
        timers = m.kernel_timings()
        m.set_kernel_timing(False)
        x1 = torch.from_numpy(m.debug_buffer("x1").copy())
        cropped, traj = None, None
        if key != "fused16":
            m.set_crop_cutoff(6.0)
            cropped = [o.clone() for o in m(b)]
            m.set_crop_cutoff(None)
            traj = m.sample_batch(HeteroBatch.from_data_list(dl[:2]), 1, (sched, sched, sched), seed=11, sample_ids=[0, 1], no_final_step_noise=True).clone()
        res[key] = (out, x1, cropped, traj, timers)
    # launches of the first-Linear GEMMs per forward: per layer and group before, the first layer's batch + the sigma batch now
    assert res["fused"][4]["conv_fc1_gemms"][1] == 2
    assert res["separate"][4]["conv_fc1_gemms"][1] > 2
    assert torch.equal(res["fused16"][0][0], res["fused"][0][0]) and torch.equal(res["fused16"][0][2], res["fused"][0][2])   # same sums, same MFMA chains
    for key in ("fused", "fused16"):
        assert torch.equal(res[key][1], res["separate"][1])          # first interaction layer's node table
        for a_, b_ in zip(res[key][0], res["separate"][0]):
            assert rel_err(a_, b_) < 1e-5
        if res[key][2] is not None:
            for a_, b_ in zip(res[key][2], res["separate"][2]):
                assert a_.shape == b_.shape and rel_err(a_, b_) < 1e-5
            assert (res[key][3] - res["separate"][3]).abs().max() < 1e-4
    b = HeteroBatch.from_data_list(dl)
    set_time(b, 0.6, 0.6, 0.6, b.num_graphs)
    ref = CGModelOracle(cfg, sd, *tables())(b)
    for o, r in zip(res["fused"][0][:3], ref[:3]):
        assert rel_err(o, r) < 1e-4
