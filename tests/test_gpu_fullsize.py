"""Loop-level and full-size parity on the MI355X against fixtures produced by EXECUTING the reference
(tests/golden/make_golden_fullsize.py): BASELINE.json configs[1] shape (DDL-synth score model, 300 residues / 30 atoms,
20 steps x 10 poses) and configs[4] shape (1500 residues / 80 atoms, every pair a cross edge).

Tolerances.  Scores: north_star's 1e-4 relative fp32, asserted element-wise (|a - b| <= 1e-4 |ref| + 1e-5 max|ref|) and in
max-norm.  Free-running trajectories: 20 steps of a chaotic map in fp32 amplify rounding, so the yardstick is the
trajectory's own sensitivity -- the distance between the reference's float32 run and the same run in float64 (stored in
the fixture): the device path must stay within 3x that distance (floor 5e-4 A) of the float32 reference."""
import numpy as np
import pytest
import torch

import cases
from diffdock_amd.hetero import HeteroBatch, set_time
from diffdock_amd.model import MIScoreModel
from diffdock_amd.sampling import sampling
from oracle.conformer import get_t_schedule
from util import (assert_scores_close, check_seeded_inputs, elem_excess, load_fixture, rel_err, rmsd, seeded_case, split_draws,
                  tables)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def gpu_model(cfg, sd):
    m = MIScoreModel(cfg, device=DEV)     # raises DdmiError if libddmi.so is not built: no fallback
    m.load_state_dict(sd)
    m.set_tables(*tables())
    return m


def place(batch):
    return batch.to(DEV)


@pytest.fixture(scope="module")
def traj():
    fx = load_fixture("traj_300_30")
    cfg, sd, g, dl = seeded_case(fx["spec"])
    check_seeded_inputs(fx, sd, g)
    assert torch.equal(torch.stack([d["ligand"].pos for d in dl]), fx["init_pos"])
    return fx, cfg, dl, gpu_model(cfg, sd)


def test_twenty_step_teacher_forced_scores_match_reference_execution(traj):
    """Every one of the 20 model calls of the reference's sampling() run, re-evaluated on the device at the ligand
    positions the reference's model saw (teacher forcing: no error accumulates across steps)."""
    fx, cfg, dl, m = traj
    B, steps = fx["spec"]["n_poses"], fx["spec"]["steps"]
    assert len(fx["steps"]) == steps
    sched = get_t_schedule(steps)
    batch = place(HeteroBatch.from_data_list(dl))
    worst = {"tr": 0.0, "rot": 0.0, "tor": 0.0}
    for k, rec in enumerate(fx["steps"]):
        set_time(batch, sched[k], sched[k], sched[k], B, device=DEV)
        batch["ligand"].pos = rec["pos_in"].to(DEV)
        out = m(batch)[:3]
        assert_scores_close(out, (rec["tr"], rec["rot"], rec["tor"]), what=f"step {k}")
        for o, n in zip(out, ("tr", "rot", "tor")):
            worst[n] = max(worst[n], elem_excess(o.cpu(), rec[n]))
    print("teacher-forced worst element-wise excess (1.0 = at the 1e-4 bound):", worst)


def test_twenty_step_free_running_rmsd_to_reference_execution(traj):
    fx, cfg, dl, m = traj
    B, steps = fx["spec"]["n_poses"], fx["spec"]["steps"]
    R = int(dl[0]["ligand"].edge_mask.sum())
    noise = split_draws(fx["draws"], steps, B, R)
    sched = get_t_schedule(steps)
    yard = rmsd(fx["final_pos_f64"], fx["final_pos"])            # float32 reference vs float64 oracle, same draws
    pos = m.sample_batch(place(HeteroBatch.from_data_list(dl)), steps, (sched, sched, sched), noise=noise,
                         no_final_step_noise=True, **fx["temp"]).cpu().reshape(B, -1, 3)
    r = rmsd(pos, fx["final_pos"])
    moved = rmsd(fx["final_pos"], fx["init_pos"])
    print("final-pose RMSD to the reference run per pose [A]:", [f"{x:.2e}" for x in r.tolist()])
    print("rounding yardstick (reference f32 vs f64) [A]:   ", [f"{x:.2e}" for x in yard.tolist()],
          " pose displacement over the run [A]:", [f"{x:.1f}" for x in moved.tolist()])
    assert torch.isfinite(pos).all()
    assert float(r.max()) <= max(3.0 * float(yard.max()), 5e-4)
    # the step-wise route (model(batch) + ddmi_perturb + ddmi_modify_conformer per step, the reference's loop structure)
    dl2 = [d.clone() for d in dl]
    out, _ = sampling(dl2, m, steps, sched, sched, sched, device=DEV, model_args=cfg, batch_size=B, no_final_step_noise=True,
                      noise=noise, native_loop=False, **fx["temp"])
    pos2 = torch.stack([d["ligand"].pos.cpu() for d in out])
    assert float(rmsd(pos2, pos).max()) <= max(3.0 * float(yard.max()), 1e-3)


def test_all_atom_twenty_step_trajectory_matches_reference_execution():
    """models/aa_model.py through the reference's own sampling(): 20 steps x 4 poses, 100 residues with their heavy atoms
    (~750 atom nodes), 30-atom ligand, DDL-synth widths with sh_lmax = 2 (the reference FasterTensorProduct cannot run a step
    in which no ligand atom has a receptor atom within reach).  Teacher-forced scores of every step within the 1e-4 bound, the
    free-running final poses within 3x the trajectory's own float32 / float64 rounding yardstick."""
    fx = load_fixture("traj_aa_100_30")
    cfg, sd, g, dl = seeded_case(fx["spec"])
    check_seeded_inputs(fx, sd, g)
    assert cfg.all_atoms and g["atom"].pos.shape[0] > 500
    assert torch.equal(torch.stack([d["ligand"].pos for d in dl]), fx["init_pos"])
    m = gpu_model(cfg, sd)
    B, steps = fx["spec"]["n_poses"], fx["spec"]["steps"]
    sched = get_t_schedule(steps)
    batch = place(HeteroBatch.from_data_list(dl))
    worst = {"tr": 0.0, "rot": 0.0, "tor": 0.0}
    for k, rec in enumerate(fx["steps"]):
        set_time(batch, sched[k], sched[k], sched[k], B, device=DEV)
        batch["ligand"].pos = rec["pos_in"].to(DEV)
        out = m(batch)[:3]
        assert_scores_close(out, (rec["tr"], rec["rot"], rec["tor"]), what=f"all-atom step {k}")
        for o, n in zip(out, ("tr", "rot", "tor")):
            worst[n] = max(worst[n], elem_excess(o.cpu(), rec[n]))
    print("all-atom teacher-forced worst element-wise excess (1.0 = at the 1e-4 bound):", worst)
    R = int(dl[0]["ligand"].edge_mask.sum())
    noise = split_draws(fx["draws"], steps, B, R)
    yard = rmsd(fx["final_pos_f64"], fx["final_pos"])
    pos = m.sample_batch(place(HeteroBatch.from_data_list(dl)), steps, (sched, sched, sched), noise=noise,
                         no_final_step_noise=True, **fx["temp"]).cpu().reshape(B, -1, 3)
    r = rmsd(pos, fx["final_pos"])
    print("all-atom final-pose RMSD to the reference run [A]:", [f"{x:.2e}" for x in r.tolist()],
          " yardstick:", [f"{x:.2e}" for x in yard.tolist()])
    assert torch.isfinite(pos).all() and float(r.max()) <= max(3.0 * float(yard.max()), 5e-4)


def test_large_pocket_forward_matches_reference_execution():
    """BASELINE configs[4] shape: 1500 residues / 80 atoms, 120 000 cross edges per pose and direction, through the
    reference's own CGModel.forward (its per-edge weights [E, 6928] materialised on the host, 2 poses)."""
    fx = load_fixture("fwd_1500_80")
    cfg, sd, g, dl = seeded_case(fx["spec"])
    check_seeded_inputs(fx, sd, g)
    B, t = fx["spec"]["n_poses"], fx["spec"]["t"]
    m = gpu_model(cfg, sd)
    batch = HeteroBatch.from_data_list(dl)
    set_time(batch, t, t, t, B)
    out = m(place(batch))[:3]
    assert int(m.debug_buffer("offs_l")[-1]) == B * 80 * 1500
    assert_scores_close(out, (fx["tr"], fx["rot"], fx["tor"]), what="1500/80")
    nl = B * 80
    for l, ref_rows in enumerate(fx["lig_rows"]):             # ligand rows of every interaction layer
        mine = torch.from_numpy(m.debug_buffer(f"x{l + 1}"))[:nl, :ref_rows.shape[1]]
        assert rel_err(mine, ref_rows) < 1e-4 and elem_excess(mine, ref_rows) <= 1.0, l


def test_bench_batch_forward_matches_reference_execution():
    """bench.py's EXACT default batch (BASELINE configs[2]: 40 poses of the 300-residue / 30-atom complex, bench.py's seeds,
    static 80 A cross cutoff, the default batch-index-dependent centre convolution of models/cg_model.py:371-374) through
    the reference's own CGModel.forward: scores of all 40 poses and the ligand rows of every interaction layer for the first
    and the last pose."""
    fx = load_fixture("fwd_300_30_b40")
    cfg, sd, g, dl = seeded_case(fx["spec"])
    check_seeded_inputs(fx, sd, g)
    B, t = fx["spec"]["n_poses"], fx["spec"]["t"]
    assert B == 40 and not cfg.fixed_center_conv
    m = gpu_model(cfg, sd)
    batch = HeteroBatch.from_data_list(dl)
    set_time(batch, t, t, t, B)
    out = m(place(batch))[:3]
    assert int(m.debug_buffer("offs_l")[-1]) == B * 30 * 300
    assert_scores_close(out, (fx["tr"], fx["rot"], fx["tor"]), what="300/30 x 40")
    rows = fx["lig_rows_idx"]
    for l, ref_rows in enumerate(fx["lig_rows"]):
        mine = torch.from_numpy(m.debug_buffer(f"x{l + 1}"))[rows, :ref_rows.shape[1]]
        assert rel_err(mine, ref_rows) < 1e-4 and elem_excess(mine, ref_rows) <= 1.0, l


def test_nan_guard_on_the_device():
    cases.nan_guard_case(gpu_model, place)


def test_neighbour_caps_on_the_device():
    cases.neighbour_cap_case(gpu_model, place)


def test_same_shaped_complexes_through_the_caching_allocator():
    cases.same_shape_complexes_case(gpu_model, place)


def test_configs0_plumbing_1a0q_full_width():
    """BASELINE configs[0]: data/1a0q geometry (416 residues / 23 atoms), 4 steps x 2 samples, DDL-synth width."""
    from diffdock_amd.config import DDL_SYNTH
    r = cases.config0_case(gpu_model, place, DDL_SYNTH)
    print("1a0q 4-step RMSD to the oracle loop [A]:", r.tolist())


@pytest.mark.parametrize("n_res,n_lig,lmax", [(150, 20, 1), (500, 45, 1), (500, 45, 2)])
def test_mix_shapes_forward_matches_oracle(n_res, n_lig, lmax):
    """The corner shapes of bench.py's PDBBind-like mix (Nr in {150, 300, 500} x Nl in {20, 30, 45}) against the
    oracle, all pairs connected: 20-atom ligands leave the 32-row edge tiles of a residue 37 % empty, 45-atom ligands give
    every residue two virtual nodes (32 + 13 edges); with sh_lmax = 2 the e3nn-style tensor product (generic granules)."""
    from diffdock_amd.config import DDL_SYNTH
    from diffdock_amd.synth import make_complex, make_pose_list
    from diffdock_amd.weights import init_state_dict
    from oracle.cg_model import CGModelOracle
    cfg = DDL_SYNTH.replace(dynamic_max_cross=False, cross_max_distance=80.0, sh_lmax=lmax)
    sd = init_state_dict(cfg, seed=1234)
    g = make_complex(seed=10 + n_res // 100, n_res=n_res, n_lig=n_lig)
    dl = make_pose_list(g, 2, tr_sigma_max=cfg.tr_sigma_max, seed=3, initial_noise_std_proportion=0.3)
    batch = HeteroBatch.from_data_list(dl)
    set_time(batch, 0.45, 0.45, 0.45, 2)
    ref = CGModelOracle(cfg, sd, *tables())(batch)[:3]
    m = gpu_model(cfg, sd)
    out = m(place(batch))[:3]
    assert int(m.debug_buffer("offs_l")[-1]) == 2 * n_res * n_lig
    assert_scores_close(out, ref, what=f"{n_res}/{n_lig} lmax {lmax}")
