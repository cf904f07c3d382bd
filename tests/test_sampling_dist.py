"""Host logic of the sampler and the pose-sharded multi-process path (gloo, world size 2, CPU; the kernels run
under tests/hipemu).  The same code runs on RCCL when the backend is "nccl"."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from diffdock_amd.dist import shard_bounds
from diffdock_amd.hetero import HeteroBatch
from diffdock_amd.model import MIScoreModel
from diffdock_amd.sampling import sampling, step_coefficients
from oracle.conformer import get_t_schedule
from util import fixture_case, split_draws, tables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "hipemu", "libddmi_emu.so")


@pytest.fixture(scope="module")
def emu_lib():
    r = subprocess.run(["make", "-j8", "-C", os.path.join(ROOT, "diffdock_amd", "csrc"), "emu"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return EMU


def test_shard_bounds_cover_everything():
    for n in (1, 5, 40, 41):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_step_coefficients_match_oracle_perturbations():
    from oracle.sampling import perturbations
    _, cfg, _ = fixture_case("tiny_l1")
    s = get_t_schedule(5)
    temp = dict(temp_sampling=[1.17, 2.06, 7.04], temp_psi=[0.73, 0.90, 0.59], temp_sigma_data=[0.93, 0.75, 0.69])
    g = torch.Generator().manual_seed(0)
    sc = (torch.randn(3, 3, generator=g), torch.randn(3, 3, generator=g), torch.randn(6, generator=g))
    z = (torch.randn(3, 3, generator=g), torch.randn(3, 3, generator=g), torch.randn(6, generator=g))
    for kw in (dict(), temp, dict(ode=True), dict(no_final_step_noise=True)):
        for t_idx in (0, 4):
            ref = perturbations(cfg, t_idx, 5, (s, s, s), sc, z, **kw)
            co = step_coefficients(cfg, t_idx, 5, (s, s, s), **kw)
            for r, (a, b), x, zz in zip(ref, co, sc, z):
                assert torch.allclose(r.float(), np.float32(a) * x + np.float32(b) * zz, rtol=1e-5, atol=1e-7)


def test_sampling_signature_native_and_stepwise(emu_lib):
    fx, cfg, data_list = fixture_case("tiny_l1")
    s = fx["sampling"]
    B, R = len(data_list), int(data_list[0]["ligand"].edge_mask.sum())
    noise = split_draws(s["draws"], s["steps"], B, R)
    sched = get_t_schedule(s["steps"])
    outs = []
    for native in (True, False):
        m = MIScoreModel(cfg, device="cpu", lib_path=emu_lib)
        m.load_state_dict(fx["state_dict"])
        m.set_tables(*tables())
        dl = [d.clone() for d in data_list]
        # batch_size >= N: with the reference's default (non-"fixed") centre convolution a pose's score depends on its
        # position in the batch (cg_model.py:371-374 indexes the ligand table by graph id), so the fixture, generated with
        # all poses in one batch, is only reproduced by the same batching
        out, conf = sampling(dl, m, s["steps"], sched, sched, sched, "cpu", None, cfg, batch_size=8, noise=noise,
                             no_final_step_noise=True, native_loop=native, **s["temp"])
        assert conf is None
        outs.append(torch.stack([d["ligand"].pos for d in out]))
    assert (outs[0] - s["final_pos"]).abs().max() < 2e-3
    assert (outs[1] - s["final_pos"]).abs().max() < 2e-3


def test_sampling_with_crop_beyond_native_and_stepwise(emu_lib):
    """model_args.crop_beyond (utils/sampling.py:104-109) through both loops against the reference trajectory."""
    fx, cfg, data_list = fixture_case("tiny_l2_crop")
    s = fx["sampling"]
    B, R = len(data_list), int(data_list[0]["ligand"].edge_mask.sum())
    noise = split_draws(s["draws"], s["steps"], B, R)
    sched = get_t_schedule(s["steps"])
    for native in (True, False):
        m = MIScoreModel(cfg, device="cpu", lib_path=emu_lib)
        m.load_state_dict(fx["state_dict"])
        m.set_tables(*tables())
        out, _ = sampling([d.clone() for d in data_list], m, s["steps"], sched, sched, sched, "cpu", None, cfg, batch_size=8,
                          noise=noise, no_final_step_noise=True, native_loop=native)
        assert (torch.stack([d["ligand"].pos for d in out]) - s["final_pos"]).abs().max() < 2e-3


WORKER = r"""
import os, sys, torch, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch.distributed as dist
from diffdock_amd.dist import sample_sharded
from diffdock_amd.model import MIScoreModel
from diffdock_amd.sampling import sampling
from util import fixture_case, tables
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
fx, cfg, data_list = fixture_case("tiny_l2")
cfg = cfg.replace(fixed_center_conv=True)   # batch-composition independent scores (see test above)
data_list = [data_list[i % len(data_list)].clone() for i in range(5)]
m = MIScoreModel(cfg, device="cpu", lib_path={emu!r})
m.load_state_dict(fx["state_dict"]); m.set_tables(*tables())
sched = np.linspace(1, 0, 3)[:-1]
pos = sample_sharded(data_list, m, 2, (sched, sched, sched), sampling, device="cpu", model_args=cfg, seed=5,
                     batch_size=8, no_final_step_noise=True)
if dist.get_rank() == 0:
    torch.save(pos, {out!r})
dist.destroy_process_group()
"""


def test_two_rank_gloo_run_reproduces_single_rank(emu_lib, tmp_path):
    """5 poses on 2 ranks (uneven blocks of 3 and 2, padded for the one all_gather) == the same 5 poses on one rank:
    per-sample noise streams are keyed by the global sample index."""
    out = str(tmp_path / "pos.pt")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, emu=emu_lib, out=out, port=29611))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(2)]
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e.decode()[-2000:]
    multi = torch.load(out)
    fx, cfg, data_list = fixture_case("tiny_l2")
    cfg = cfg.replace(fixed_center_conv=True)
    data_list = [data_list[i % len(data_list)].clone() for i in range(5)]
    m = MIScoreModel(cfg, device="cpu", lib_path=emu_lib)
    m.load_state_dict(fx["state_dict"])
    m.set_tables(*tables())
    sched = np.linspace(1, 0, 3)[:-1]
    single, _ = sampling(data_list, m, 2, sched, sched, sched, "cpu", None, cfg, seed=5, batch_size=8, no_final_step_noise=True)
    single = torch.stack([d["ligand"].pos for d in single])
    assert multi.shape == single.shape
    assert (multi - single).abs().max() < 1e-4
