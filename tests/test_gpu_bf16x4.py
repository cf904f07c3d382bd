"""The split-bf16 edge product (ddmi_config.edge_product = 1, bench.py --edge-product bf16x4) on the MI355X, selected through
the CONFIG (cfg.edge_product), so that the driver's plain `pytest -m gpu` run covers the route its secondary bench line uses:
the bench batch itself, the 20-step teacher-forced run, the 1500 / 80 forward -- all against fixtures produced by executing the
reference (tests/golden/make_golden_fullsize.py) at north_star's 1e-4 bound -- and the route next to the exact-f32 one at the
benchmark width.  Operands carry 16 significand bits (bf16 hi + bf16 lo), products are exact, accumulation is fp32: the route is a
SECONDARY line with its own dtype, never the headline."""
import pytest
import torch

from diffdock_amd.config import DDL_SYNTH
from diffdock_amd.hetero import HeteroBatch, set_time
from diffdock_amd.model import MIScoreModel
from diffdock_amd.weights import init_state_dict
from oracle.cg_model import CGModelOracle
from oracle.conformer import get_t_schedule
from test_gpu_parity import synth_batch
from util import assert_scores_close, check_seeded_inputs, elem_excess, load_fixture, rel_err, seeded_case, tables

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def bf_model(cfg, sd):
    m = MIScoreModel(cfg.replace(edge_product="bf16x4"), device=DEV)
    m.load_state_dict(sd)
    m.set_tables(*tables())
    return m


def test_bf16x4_bench_batch_forward_matches_reference_execution():
    fx = load_fixture("fwd_300_30_b40")
    cfg, sd, g, dl = seeded_case(fx["spec"])
    check_seeded_inputs(fx, sd, g)
    B, t = fx["spec"]["n_poses"], fx["spec"]["t"]
    m = bf_model(cfg, sd)
    batch = HeteroBatch.from_data_list(dl)
    set_time(batch, t, t, t, B)
    out = m(batch.to(DEV))[:3]
    assert int(m.debug_buffer("offs_l")[-1]) == B * 30 * 300
    assert_scores_close(out, (fx["tr"], fx["rot"], fx["tor"]), what="bf16x4 300/30 x 40")
    worst = max(elem_excess(o.cpu(), fx[n]) for o, n in zip(out, ("tr", "rot", "tor")))
    print("bf16x4 bench batch: worst element-wise excess (1.0 = at the 1e-4 bound):", worst)
    rows = fx["lig_rows_idx"]
    for l, ref_rows in enumerate(fx["lig_rows"]):
        mine = torch.from_numpy(m.debug_buffer(f"x{l + 1}"))[rows, :ref_rows.shape[1]]
        assert rel_err(mine, ref_rows) < 1e-4 and elem_excess(mine, ref_rows) <= 1.0, l


def test_bf16x4_twenty_step_teacher_forced_scores_match_reference_execution():
    fx = load_fixture("traj_300_30")
    cfg, sd, g, dl = seeded_case(fx["spec"])
    check_seeded_inputs(fx, sd, g)
    m = bf_model(cfg, sd)
    B, steps = fx["spec"]["n_poses"], fx["spec"]["steps"]
    sched = get_t_schedule(steps)
    batch = HeteroBatch.from_data_list(dl).to(DEV)
    worst = {"tr": 0.0, "rot": 0.0, "tor": 0.0}
    for k, rec in enumerate(fx["steps"]):
        set_time(batch, sched[k], sched[k], sched[k], B, device=DEV)
        batch["ligand"].pos = rec["pos_in"].to(DEV)
        out = m(batch)[:3]
        assert_scores_close(out, (rec["tr"], rec["rot"], rec["tor"]), what=f"bf16x4 step {k}")
        for o, n in zip(out, ("tr", "rot", "tor")):
            worst[n] = max(worst[n], elem_excess(o.cpu(), rec[n]))
    print("bf16x4 teacher-forced worst element-wise excess (1.0 = at the 1e-4 bound):", worst)


def test_bf16x4_large_pocket_forward_matches_reference_execution():
    fx = load_fixture("fwd_1500_80")
    cfg, sd, g, dl = seeded_case(fx["spec"])
    check_seeded_inputs(fx, sd, g)
    B, t = fx["spec"]["n_poses"], fx["spec"]["t"]
    m = bf_model(cfg, sd)
    batch = HeteroBatch.from_data_list(dl)
    set_time(batch, t, t, t, B)
    out = m(batch.to(DEV))[:3]
    assert_scores_close(out, (fx["tr"], fx["rot"], fx["tor"]), what="bf16x4 1500/80")


def test_bf16x4_route_next_to_the_f32_route_at_benchmark_width(monkeypatch):
    """Same inputs through both arithmetic routes: both within 1e-4 of the oracle, different from each other (the route is really
    another arithmetic) at fp32-rounding distance."""
    monkeypatch.delenv("DDMI_EDGE_PRODUCT", raising=False)     # (the harness variable would turn the default route into bf16x4 too)
    cfg = DDL_SYNTH
    sd = init_state_dict(cfg, seed=1234)
    batch = synth_batch(cfg, 100, 40, 2, seed=5, t=0.6)
    ref = CGModelOracle(cfg, sd, *tables())(batch)[:3]
    f32 = MIScoreModel(cfg, device=DEV)
    f32.load_state_dict(sd)
    f32.set_tables(*tables())
    a = [o.cpu() for o in f32(batch.to(DEV))[:3]]
    b = [o.cpu() for o in bf_model(cfg, sd)(batch.to(DEV))[:3]]
    for x, y, r in zip(a, b, ref):
        assert rel_err(y, r) < 1e-4 and elem_excess(y, r) <= 1.0
        assert 0 < rel_err(y, x) < 5e-5
