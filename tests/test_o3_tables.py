"""Product-side coupling tables (diffdock_amd/o3.py) against the oracle's tensor products."""
import numpy as np
import pytest
import torch

from diffdock_amd import o3
from diffdock_amd.irreps import full_tp_irreps, irreps_dim, sh_irreps, tp_weight_numel
from oracle import e3nn_lite as e3
from oracle.layers import faster_tensor_product


def eval_table(table, out_dim, x, sh, w):
    E = x.shape[0]
    out = np.zeros((E, out_dim))
    for p in table:
        X = x[:, p.i_off:p.i_off + p.mul_in * p.di].reshape(E, p.mul_in, p.di)
        S = sh[:, p.s_off:p.s_off + p.ds]
        W = w[:, p.w_off:p.w_off + p.mul_in * p.mul_out].reshape(E, p.mul_in, p.mul_out)
        o = np.einsum("euw,eui,ej,ijk->ewk", W, X, S, p.C)
        out[:, p.o_off:p.o_off + p.mul_out * p.do] += o.reshape(E, -1)
    return out


def test_w3j_matches_oracle():
    for ls in [(0, 0, 0), (1, 1, 0), (1, 1, 1), (1, 1, 2), (1, 2, 1), (2, 2, 2), (1, 2, 3), (2, 2, 4), (2, 1, 3)]:
        assert np.allclose(o3.wigner_3j(*ls), e3.wigner_3j(*ls).numpy(), atol=1e-13)


@pytest.mark.parametrize("irr_in,irr_out", [
    ("6x0e", "6x0e + 3x1o"), ("6x0e + 3x1o", "6x0e + 3x1o + 3x1e"),
    ("6x0e + 3x1o + 3x1e", "6x0e + 3x1o + 3x1e + 6x0o"),
    ("6x0e + 3x1o + 3x1e + 6x0o", "6x0e + 3x1o + 3x1e + 6x0o"),
    ("6x0e + 3x1o + 3x1e + 2x0o", "6x0e + 3x1o + 3x1e + 2x0o")])
def test_faster_table(irr_in, irr_out):
    rng = np.random.default_rng(0)
    table, numel = o3.faster_path_table(irr_in, irr_out)
    assert numel == tp_weight_numel(irr_in, sh_irreps(1), irr_out, True)
    E = 4
    x, sh, w = rng.normal(size=(E, irreps_dim(irr_in))), rng.normal(size=(E, 4)), rng.normal(size=(E, numel))
    ref = faster_tensor_product(irr_in, irr_out, torch.from_numpy(x), torch.from_numpy(sh), torch.from_numpy(w)).numpy()
    assert np.allclose(eval_table(table, irreps_dim(irr_out), x, sh, w), ref, atol=1e-12)


@pytest.mark.parametrize("lmax", [1, 2])
@pytest.mark.parametrize("irr_out", ["6x0e + 3x1o + 3x1e + 6x0o", "2x1o + 2x1e"])
def test_fctp_table(lmax, irr_out):
    rng = np.random.default_rng(1)
    irr_in = "6x0e + 3x1o + 3x1e + 6x0o"
    shs = sh_irreps(lmax)
    table, numel = o3.fctp_path_table(irr_in, shs, irr_out)
    tp = e3.FullyConnectedTensorProduct(irr_in, shs, irr_out)
    assert numel == tp.weight_numel
    E = 4
    x, sh, w = rng.normal(size=(E, irreps_dim(irr_in))), rng.normal(size=(E, irreps_dim(shs))), rng.normal(size=(E, numel))
    ref = tp(torch.from_numpy(x), torch.from_numpy(sh), torch.from_numpy(w)).numpy()
    assert np.allclose(eval_table(table, irreps_dim(irr_out), x, sh, w), ref, atol=1e-12)


@pytest.mark.parametrize("lmax", [1, 2])
def test_tor_conv_table_and_full_tp(lmax):
    rng = np.random.default_rng(2)
    shs = sh_irreps(lmax)
    tor_sh = full_tp_irreps(shs, "1x2e")
    ftp = e3.FullTensorProduct(e3.Irreps.spherical_harmonics(lmax), "2e")
    assert str(e3.Irreps(tor_sh)) == str(ftp.irreps_out)
    T = o3.full_tp_table(shs, "1x2e")
    a, b = rng.normal(size=(5, irreps_dim(shs))), rng.normal(size=(5, 5))
    assert np.allclose(np.einsum("ei,ej,ijk->ek", a, b, T), ftp(torch.from_numpy(a), torch.from_numpy(b)).numpy(), atol=1e-12)
    irr_in, irr_out = "6x0e + 3x1o + 3x1e + 6x0o", "6x0o + 6x0e"
    table, numel = o3.fctp_path_table(irr_in, tor_sh, irr_out)
    tp = e3.FullyConnectedTensorProduct(irr_in, ftp.irreps_out, irr_out)
    assert numel == tp.weight_numel
    x, sh, w = rng.normal(size=(3, 30)), rng.normal(size=(3, irreps_dim(tor_sh))), rng.normal(size=(3, numel))
    ref = tp(torch.from_numpy(x), torch.from_numpy(sh), torch.from_numpy(w)).numpy()
    assert np.allclose(eval_table(table, 12, x, sh, w), ref, atol=1e-12)


def test_survey_weight_numels():
    """W of the DDL-synth layers quoted in SURVEY.md 8a (a13, a15)."""
    from diffdock_amd.config import DDL_SYNTH as c
    Ws = [tp_weight_numel(*c.layer_irreps(l), True) if False else tp_weight_numel(c.layer_irreps(l)[0], sh_irreps(1), c.layer_irreps(l)[1], True)
          for l in range(6)]
    assert Ws == [2784, 3464, 4144, 6928, 6928, 6928]
    last = c.layer_irreps(5)[1]
    assert tp_weight_numel(last, sh_irreps(1), "2x1o + 2x1e", False) == 272
    assert tp_weight_numel(last, full_tp_irreps(sh_irreps(1), "1x2e"), "48x0o + 48x0e", False) == 960
    assert tp_weight_numel(last, sh_irreps(2), last, False) == 7128
    assert tp_weight_numel(last, full_tp_irreps(sh_irreps(2), "1x2e"), "48x0o + 48x0e", False) == 6528
