"""The oracle against fixtures produced by EXECUTING the reference's python
(tests/golden/make_golden.py).  These pin the restated wiring of CGModel.forward,
TensorProductConvLayer / tp_scatter_*, FasterTensorProduct, the pose update and the
reverse-diffusion loop; the tiny_aa_* cases pin AAModel.forward (models/aa_model.py) the same way."""
import numpy as np
import pytest
import torch

from diffdock_amd.hetero import HeteroBatch, set_time
from oracle import conformer as oc
from oracle.cg_model import CGModelOracle
from oracle.layers import faster_tensor_product, gaussian_smearing
from oracle.sampling import sampling
from util import fixture_case, load_fixture, oracle_model, rel_err, split_draws, tables, graph_from_dict

CASES = ["tiny_l1", "tiny_l2", "tiny_l1_1group_emb", "tiny_l2_fixedcenter", "tiny_l2_crop", "tiny_aa_l1", "tiny_aa_l2", "tiny_aa_l2_emb",
         "tiny_noaa", "tiny_2nd", "tiny_aa_2nd",   # tiny_*2nd: use_second_order_repr (2e / 2o node blocks)
         "tiny_fourier", "tiny_tpw3",              # embedding_type='fourier'; tp_weights_layers=3
         "tiny_aa_emb_nolig",                      # AAModel: embedding layers without embed_also_ligand (zero-padded ligand rows)
         "tiny_oddpar", "tiny_aa_oddpar", "tiny_nobn_noscale",   # odd_parity (CG + all-atom); batch_norm off + scale_by_sigma off
         "tiny_sidechain",                                        # sidechain_pred: o3.Linear on the receptor rows, 4th tuple element
         "tiny_depthwise", "tiny_depthwise_l2"]                   # depthwise_convolution: 'uvu' TensorProduct + linear_2 (sh_lmax 1 and 2)


@pytest.mark.parametrize("name", CASES)
def test_forward_matches_reference(name):
    fx, cfg, data_list = fixture_case(name)
    so3_t, tor_t = tables()
    model = oracle_model(cfg, fx["state_dict"], so3_t, tor_t)
    batch = HeteroBatch.from_data_list(data_list)
    set_time(batch, fx["t"], fx["t"], fx["t"], batch.num_graphs)
    tr, rot, tor, side, inter = model(batch, return_intermediates=True)
    ref = fx["forward"]
    if cfg.sidechain_pred:   # models/cg_model.py:397-402
        assert side.shape == ref["sidechain"].shape == (batch["receptor"].pos.shape[0], 10) and rel_err(side, ref["sidechain"]) < 2e-5
    else:
        assert side is None
    for l, ref_nodes in enumerate(ref["conv_out"]):
        mine = inter[f"node_attr{l + 1}"]
        assert rel_err(mine, ref_nodes) < 2e-5, (l, rel_err(mine, ref_nodes))
    assert rel_err(tr, ref["tr"]) < 2e-5
    assert rel_err(rot, ref["rot"]) < 2e-5
    assert rel_err(tor, ref["tor"]) < 2e-5


@pytest.mark.parametrize("name", ["tiny_conf_l2", "tiny_conf_aa_l1", "tiny_conf_atom"])
def test_confidence_matches_reference(name):
    """get_model(..., confidence_mode=True) executed by the reference (CGModel / AAModel, cg_model.py:353-366)."""
    fx, cfg, data_list = fixture_case(name)
    model = oracle_model(cfg, fx["state_dict"])
    batch = HeteroBatch.from_data_list(data_list)
    set_time(batch, fx["t"], fx["t"], fx["t"], batch.num_graphs)
    conf, atom_conf, inter = model(batch, return_intermediates=True)
    ref = fx["forward"]
    for l, ref_nodes in enumerate(ref["conv_out"]):
        assert rel_err(inter[f"node_attr{l + 1}"], ref_nodes) < 2e-5, l
    assert conf.shape == ref["confidence"].shape and rel_err(conf, ref["confidence"]) < 2e-5
    if cfg.atom_confidence:   # per-atom predictor (+ affinity column), cg_model.py:184-207,357-360
        assert atom_conf.shape == ref["atom_confidence"].shape == (batch["ligand"].pos.shape[0], cfg.atom_num_confidence_outputs)
        assert rel_err(atom_conf, ref["atom_confidence"]) < 2e-5
    else:
        assert torch.equal(atom_conf, ref["atom_confidence"])


@pytest.mark.parametrize("name", ["tiny_oldconf", "tiny_oldconf_2l"])
def test_legacy_confidence_matches_reference(name):
    """get_model(..., old=True, confidence_mode=True) executed by the reference: models/old_cg_model.py CGOldModel with
    OldAtomEncoder and OldTensorProductConvLayer -- the class of the released DiffDock-L confidence checkpoint."""
    fx, cfg, data_list = fixture_case(name)
    batch = HeteroBatch.from_data_list(data_list)
    set_time(batch, fx["t"], fx["t"], fx["t"], batch.num_graphs)
    conf = oracle_model(cfg, fx["state_dict"])(batch)
    assert conf.shape == fx["forward"]["confidence"].shape and rel_err(conf, fx["forward"]["confidence"]) < 2e-5


def test_legacy_score_mode_matches_reference():
    """get_model(..., old=True) in score mode executed by the reference: the legacy encoder / four-layer interaction blocks
    followed by the read-outs of old_cg_model.py:293-352; forward scores and the reference sampling() trajectory."""
    fx, cfg, data_list = fixture_case("tiny_oldscore")
    so3_t, tor_t = tables()
    model = oracle_model(cfg, fx["state_dict"], so3_t, tor_t)
    batch = HeteroBatch.from_data_list(data_list)
    set_time(batch, fx["t"], fx["t"], fx["t"], batch.num_graphs)
    out = model(batch)
    assert len(out) == 3
    for mine, key in zip(out, ("tr", "rot", "tor")):
        assert mine.shape == fx["forward"][key].shape and rel_err(mine, fx["forward"][key]) < 2e-5, key
    s = fx["sampling"]
    B, R = len(data_list), int(data_list[0]["ligand"].edge_mask.sum())
    noise = split_draws(s["draws"], s["steps"], B, R)
    res = sampling(data_list, model, s["steps"], cfg, noise, batch_size=B, no_final_step_noise=True, **s["temp"])
    final = torch.stack([d["ligand"].pos for d in res])
    assert (final - s["final_pos"]).abs().max() < 2e-3


@pytest.mark.parametrize("name", CASES)
def test_sampling_matches_reference(name):
    fx, cfg, data_list = fixture_case(name)
    so3_t, tor_t = tables()
    model = oracle_model(cfg, fx["state_dict"], so3_t, tor_t)
    s = fx["sampling"]
    B = len(data_list)
    R = int(data_list[0]["ligand"].edge_mask.sum())
    noise = split_draws(s["draws"], s["steps"], B, R)
    out = sampling(data_list, model, s["steps"], cfg, noise, batch_size=B, no_final_step_noise=True, **s["temp"])
    final = torch.stack([d["ligand"].pos for d in out])
    # 4 chaotic steps in fp32: compare in Angstrom against the reference trajectory end-points
    assert (final - s["final_pos"]).abs().max() < 2e-3, (final - s["final_pos"]).abs().max()


def test_unit_fixtures():
    u = load_fixture("units")
    assert torch.allclose(oc.axis_angle_to_matrix(u["axis_angle"]), u["rot_mats"], atol=1e-6)
    R, t = oc.kabsch_batch(u["kabsch_A"], u["kabsch_B"])
    assert torch.allclose(R, u["kabsch_R"], atol=1e-5) and torch.allclose(t, u["kabsch_t"], atol=1e-5)
    g = graph_from_dict(u["mc_graph"])
    B = u["mc_tr"].shape[0]
    ei = g["ligand", "ligand"].edge_index
    rot_edges = ei.T[g["ligand"].edge_mask]
    out = oc.modify_conformer_batch(u["mc_pos_in"], B, rot_edges, torch.from_numpy(g["ligand"].mask_rotate[0]),
                                    u["mc_tr"], u["mc_rot"], u["mc_tor"])
    assert torch.allclose(out, u["mc_pos_out"], atol=2e-5), (out - u["mc_pos_out"]).abs().max()
    irr = "8x0e + 3x1o + 3x1e + 8x0o"
    assert torch.allclose(faster_tensor_product(irr, irr, u["ftp_x"], u["ftp_sh"], u["ftp_w"]), u["ftp_out"], atol=1e-5)
    assert torch.allclose(oc.sinusoidal_embedding(1000.0 * u["sin_t"], 16), u["sin_emb"], atol=1e-6)
    assert torch.allclose(gaussian_smearing(torch.linspace(0, 5, 16), u["gs_dist"]), u["gs_out"], atol=1e-7)
    assert np.allclose(oc.get_t_schedule(20), u["t_schedule_20"].numpy())


def test_crop_beyond_matches_reference():
    """utils/utils.py:388-413 executed by the reference on one complex (fixture) vs the restated crop."""
    from oracle.sampling import crop_beyond
    fx = load_fixture("crop_beyond")
    g = graph_from_dict(fx["graph"])
    n_before = g["receptor"].pos.shape[0]
    crop_beyond(g, fx["cutoff"])
    assert 0 < g["receptor"].pos.shape[0] < n_before      # the fixture really crops
    assert torch.equal(g["receptor"].pos, fx["rec_pos"])
    assert torch.equal(g["receptor", "receptor"].edge_index, fx["rec_edge_index"])
