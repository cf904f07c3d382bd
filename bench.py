#!/usr/bin/env python
"""poses/sec of the reverse-diffusion sampling loop (20 inference steps x 40 samples, DDL-synth score model).

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N>1 launched through
torch.distributed.run, one rank per GPU.  A "step" = one complete pass of the hot path over one batch:
all 20 reverse-diffusion steps of the rank's 40 poses of one synthetic 300-residue / 30-atom complex
(BASELINE.json configs[2]), i.e. 20 score-model forwards + 20 pose updates, entirely on the device
(ddmi_sample).  Inputs (graph, weights, tables, initial poses) are resident in HBM before the timed
region.  Each rank owns 40 independent poses (weak scaling: N GPUs sample 40*N poses); the only
collective is one RCCL all_gather of the final coordinates per step, as the reference's sampler would
hand them to the confidence model.

Cross graph: the untrained (random-weight) score model cannot keep the ligand in the pocket, and with the
reference's dynamic cutoff 3*sigma_tr+20 A the ligand would drift out of range and the cross graph would
thin out, shrinking the work.  The bench therefore pins the cross graph at its upper bound
(dynamic_max_cross=False, cutoff = cross_max_distance = 80 A: every ligand-atom x residue pair is an
edge, as SURVEY.md 8 assumes: 1.02 M edges per interaction layer at 40 poses) and reports the measured
edge count in `config`.

Extra objects on the JSON line: `roofline` (dominant kernel, HIP-event timed inside the timed region) and
`cpu_baseline` (the oracle = pure-torch restatement of the reference, timed on this host's cores on a
bounded sample: 2 poses x the first 3 of the 20 steps of the same complex, extrapolated linearly).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from diffdock_amd.config import DDL_SYNTH  # noqa: E402
from diffdock_amd.hetero import HeteroBatch, set_time  # noqa: E402
from diffdock_amd.model import MIScoreModel  # noqa: E402
from diffdock_amd.synth import make_complex, make_pose_list  # noqa: E402
from diffdock_amd.tables import default_tables  # noqa: E402
from diffdock_amd.weights import init_state_dict  # noqa: E402

INFERENCE_STEPS, SAMPLES, N_RES, N_LIG = 20, 40, 300, 30
HBM_PEAK_GBS, MFMA_F32_PEAK_TFLOPS = 8000.0, 157.3     # /opt/skills/guides/MI355X_MICROARCH.md
TEMP = dict(temp_sampling=[1.170050527854316, 2.06391612594481, 7.044261621607846],       # default_inference_args.yaml
            temp_psi=[0.727287304570729, 0.9022615585677628, 0.5946212391366862],
            temp_sigma_data=[0.9299802531572672, 0.7464326999906034, 0.6943254174849822])


def bench_cfg():
    return DDL_SYNTH.replace(dynamic_max_cross=False, cross_max_distance=80.0)


def t_schedule(steps):
    return np.linspace(1, 0, steps + 1)[:-1]      # get_t_schedule('expbeta', alpha=beta=1), diffusion_utils.py:138-142


def conv_work(cfg, nL, nR, e_ll, e_lr, e_rr, fused=True, fused_lig=True):
    """Algorithmic flops / bytes of every (layer, edge group) launch of the convolution kernels of one forward pass
    (DESIGN.md section 4).  NT = columns of a contracted node row (sum over paths of din*mul_out), K = 3ns + 1.
    Groups that run k_conv_fused keep the contracted rows in LDS: their HBM bytes are the x rows, the hidden rows and the
    messages, and their node term is counted once per gather NODE (the kernel repeats it per 32-edge virtual node of a
    ligand atom -- that repetition is not algorithmic work).  With DDMI_FUSED_LIG < 3 the ligand-gather groups run
    k_node_contract + k_edge_conv with the rows Y in HBM."""
    from diffdock_amd.irreps import parse_irreps, sh_irreps
    from diffdock_amd.o3 import faster_path_table, fctp_path_table
    out = []
    HK = 3 * cfg.ns + 1
    L = cfg.num_conv_layers
    for l in range(L):
        a, b = cfg.layer_irreps(cfg.num_prot_emb_layers + l)
        table, _ = faster_path_table(a, b) if cfg.faster else fctp_path_table(a, sh_irreps(cfg.sh_lmax), b)
        NT = sum(p.di * p.mul_out for p in table)
        mac_node = sum(p.mul_in * p.mul_out * p.di for p in table)
        d_in, d_out = sum(x.dim for x in parse_irreps(a)), sum(x.dim for x in parse_irreps(b))
        # (gather nodes, target nodes, edges, gather side is receptor)
        groups = [(nL, nL, e_ll, False), (nR, nL, e_lr, True), (nR, nR, e_rr, True), (nL, nR, e_lr, False)]
        for gcount, tcount, E, rec_gather in (groups if l < L - 1 else groups[:2]):
            H = 3 * cfg.ns
            node_flops, edge_flops = 2.0 * gcount * HK * mac_node, 2.0 * E * HK * NT
            if fused and (rec_gather or fused_lig):
                out.append({"k_conv_fused": {"flops": node_flops + edge_flops,
                                             "bytes": gcount * d_in * 4.0 + E * (H * 4.0 + d_out * 4.0)}})
            else:
                out.append({"k_edge_conv": {"flops": edge_flops,
                                            "bytes": gcount * HK * NT * 4.0 + E * (H * 4.0 + d_out * 4.0) + (gcount + tcount) * H * 4.0},
                            "k_node_contract": {"flops": node_flops, "bytes": gcount * (HK * NT * 4.0 + d_in * 4.0)}})
    return out


def cpu_baseline(cfg, sd, so3_t, tor_t, g):
    """Oracle (restated reference, pure torch fp32, all host cores) on a bounded sample of the same workload."""
    from oracle.cg_model import CGModelOracle
    from oracle.sampling import sampling as oracle_sampling
    threads = min(os.cpu_count(), 64)     # the oracle's small einsums do not scale past a few dozen threads
    torch.set_num_threads(threads)
    n_s, n_steps = 2, 3
    dl = make_pose_list(g, n_s, tr_sigma_max=cfg.tr_sigma_max, seed=77, initial_noise_std_proportion=0.3)
    model = CGModelOracle(cfg, sd, so3_t, tor_t)
    R = int(dl[0]["ligand"].edge_mask.sum())
    gen = torch.Generator().manual_seed(0)
    noise = (torch.randn(INFERENCE_STEPS, n_s, 3, generator=gen), torch.randn(INFERENCE_STEPS, n_s, 3, generator=gen),
             torch.randn(INFERENCE_STEPS, n_s * R, generator=gen))
    s = t_schedule(INFERENCE_STEPS)
    t0 = time.time()
    # first n_steps of the 20-step schedule (largest cross cutoffs = the same all-pairs graph as the GPU run)
    oracle_sampling(dl, model, n_steps, cfg, noise, schedules=(s, s, s), batch_size=n_s, **TEMP)
    dt = time.time() - t0
    poses_per_s = n_s / (dt / n_steps * INFERENCE_STEPS)
    return {"value": poses_per_s, "unit": "poses/s", "cores": threads, "kind": "port",
            "sample": f"{n_s} poses x {n_steps} of {INFERENCE_STEPS} steps, same complex and weights, "
                      f"extrapolated linearly; torch {torch.__version__}, {dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--samples", type=int, default=SAMPLES, help="poses per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lib", default=None, help="path of an alternative libddmi build (kernel A/B experiments)")
    ap.add_argument("--all-atoms", action="store_true",
                    help="secondary workload: the all-atom score model (models/aa_model.py), ~7.5 receptor atoms per residue")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for --gpus N"
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)

    cfg = bench_cfg()
    if args.all_atoms:
        cfg = cfg.replace(all_atoms=True)
    sd = init_state_dict(cfg, seed=1234)
    so3_t, tor_t = default_tables()
    model = MIScoreModel(cfg, device=str(dev), lib_path=args.lib)
    model.load_state_dict(sd)
    model.set_tables(so3_t, tor_t)
    g = make_complex(seed=0, n_res=N_RES, n_lig=N_LIG, all_atoms=args.all_atoms)
    B = args.samples
    dl = make_pose_list(g, B, tr_sigma_max=cfg.tr_sigma_max, seed=1000 + rank, initial_noise_std_proportion=0.3)
    batch = HeteroBatch.from_data_list(dl).to(dev)
    sched = t_schedule(INFERENCE_STEPS)
    ids = list(range(rank * B, (rank + 1) * B))
    gathered = [torch.empty(B * N_LIG, 3, device=dev) for _ in range(world)] if world > 1 else None

    def one_step(seed):
        pos = model.sample_batch(batch, INFERENCE_STEPS, (sched, sched, sched), seed=seed, sample_ids=ids,
                                 no_final_step_noise=True, **TEMP)
        if world > 1:
            dist.all_gather(gathered, pos)
        return pos

    for w in range(args.warmup):
        one_step(w)
    model.set_kernel_timing(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        pos = one_step(100 + k)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    timings = model.kernel_timings()
    model.set_kernel_timing(False)
    assert os.environ.get("DDMI_BENCH_NOCHECK") or torch.isfinite(pos).all()   # NOCHECK: timing-only ablation builds

    if rank == 0:
        # edges actually processed by the last forward of the run
        e_ll = int(model.debug_buffer("goff_ll")[-1])
        e_lr = int(model.debug_buffer("offs_l")[-1])
        e_rr = int(model.debug_buffer("rr_goff")[-1])
        L = cfg.num_conv_layers
        edges_per_layer = e_ll + 2 * e_lr + e_rr
        # ---- roofline of the dominant kernel (by HIP-event time inside the timed region)
        kern = {k: v for k, v in timings.items() if k.startswith("k_") or k == "conv_fc1_gemms"}
        dom = max(kern, key=lambda k: kern[k][0]) if kern else None
        roof = None
        if dom in ("k_edge_conv", "k_node_contract", "k_conv_fused") and not args.all_atoms:
            ms, n = kern[dom]
            avg_s = ms / max(n, 1) * 1e-3
            work = [w[dom] for w in conv_work(cfg, B * N_LIG, B * N_RES, e_ll, e_lr, e_rr, fused="k_conv_fused" in kern,
                                              fused_lig="k_edge_conv" not in kern and "k_conv_fused_load" not in kern)
                    if dom in w]
            launches_per_forward = len(work)
            flops = sum(w["flops"] for w in work) / launches_per_forward
            bytes_ = sum(w["bytes"] for w in work) / launches_per_forward
            ridge = MFMA_F32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
            if flops / bytes_ > ridge:
                ach = flops / avg_s / 1e12
                roof = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / MFMA_F32_PEAK_TFLOPS, "traffic": None}
            else:
                ach = bytes_ / avg_s / 1e9
                roof = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "traffic": None}
            roof.update({"avg_launch_ms": avg_s * 1e3, "launches": n, "launches_per_forward": launches_per_forward,
                         "alg_flops_per_launch": flops, "alg_bytes_per_launch": bytes_,
                         "concurrent_streams": 1 if os.environ.get("DDMI_STREAMS") == "1" else 2,
                         "alg_definition": "mean over the launches of this kernel in one forward -- all four edge groups of every "
                                           "layer, 22 launches (exact f32 on v_mfma_f32_16x16x4_f32, "
                                           "peak = dense f32 MFMA); k_conv_fused: 2*145*sum(mul_in*mul_out*din) flop per gather node "
                                           "+ 2*145*NT flop per edge, bytes = x rows + 576 B hidden row + 624 B message per edge; "
                                           "k_edge_conv: 2*145*NT flop/edge, bytes = contracted rows Y (145*NT*4 B per gather node) + "
                                           "hidden + message rows; k_node_contract: node flops, Y written once.  With 2 streams the "
                                           "ligand-gather launches run concurrently with the receptor-gather ones, so the launch "
                                           "durations overlap (their sum exceeds the wall time); DDMI_STREAMS=1 serialises them"})
            if roof["concurrent_streams"] == 2 and dom == "k_conv_fused" and world == 1:
                # the same kernel timed with the launches serialised on ONE stream (untimed extra pass, second handle):
                # with two streams the launch durations overlap, so the figures above understate the kernel alone
                os.environ["DDMI_STREAMS"] = "1"
                m1 = MIScoreModel(cfg, device=str(dev), lib_path=args.lib)
                del os.environ["DDMI_STREAMS"]
                m1.load_state_dict(sd)
                m1.set_tables(so3_t, tor_t)
                m1.sample_batch(batch, INFERENCE_STEPS, (sched, sched, sched), seed=7, sample_ids=ids, no_final_step_noise=True, **TEMP)
                m1.set_kernel_timing(True)
                m1.sample_batch(batch, INFERENCE_STEPS, (sched, sched, sched), seed=8, sample_ids=ids, no_final_step_noise=True, **TEMP)
                torch.cuda.synchronize()
                ms1, n1 = m1.kernel_timings()[dom]
                ach1 = flops / (ms1 / max(n1, 1) * 1e-3) / 1e12
                roof["serialised"] = {"avg_launch_ms": ms1 / max(n1, 1), "launches": n1, "achieved": ach1,
                                      "frac": ach1 / MFMA_F32_PEAK_TFLOPS}
                del m1
        cpu = None if (args.no_cpu_baseline or args.all_atoms or world > 1) else cpu_baseline(cfg, sd, so3_t, tor_t, g)   # N = 1 only
        out = {
            "metric": "poses/sec (20 steps x 40 samples, DiffDock-L score model)" if not args.all_atoms else
                      "poses/sec (20 steps x 40 samples, all-atom score model -- secondary workload)",
            "value": world * B * args.steps / dt,
            "unit": "poses/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: DDL-synth score model (ns=48 nv=10 6 layers sh_lmax=1), "
                                   f"{INFERENCE_STEPS} steps x {B} poses/GPU, synthetic {N_RES}-residue receptor / "
                                   f"{N_LIG}-atom ligand, cross graph pinned at its upper bound (static 80 A cutoff), "
                                   f"low-temperature SDE, random-init weights",
                       "poses_per_gpu": B, "inference_steps": INFERENCE_STEPS, "edges_per_layer": edges_per_layer,
                       "edges": {"lig_lig": e_ll, "cross_each_direction": e_lr, "rec_rec": e_rr,
                                 **({"lig_atom_each_direction": int(model.debug_buffer("offs_la_l")[-1]),
                                     "atom_atom": int(model.debug_buffer("aa_goff")[-1]),
                                     "atom_rec_each_direction": int(model.debug_buffer("ar_goff")[-1]),
                                     "atoms": int(batch["atom"].pos.shape[0])} if args.all_atoms else {})},
                       "parallelism": f"pose-sharded x{world}, 1 all_gather/step" if world > 1 else "single GPU"},
            "roofline": roof, "cpu_baseline": cpu,
            "phase_ms_per_forward": {k: v[0] / max(args.steps * INFERENCE_STEPS, 1) for k, v in timings.items()},
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
