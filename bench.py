#!/usr/bin/env python
"""poses/sec of the reverse-diffusion sampling loop (20 inference steps x 40 samples, DDL-synth score model).

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N>1 launched through
torch.distributed.run, one rank per GPU.  A "step" = one complete pass of the hot path over one batch:
all 20 reverse-diffusion steps of the rank's poses of every complex of the workload, i.e. 20 score-model
forwards + 20 pose updates per complex, entirely on the device (ddmi_sample).  Inputs (graphs, weights, tables,
initial poses) are resident in HBM before the timed region.

Workloads (--config; BASELINE.json `configs`, default = the one `metric` is quoted on):
  configs2 (default)  20 steps x 40 poses of one synthetic 300-residue / 30-atom complex
  configs1            20 steps x 10 poses of the same complex
  mix                 SURVEY 8d's PDBBind-test-like mix: the nine complexes Nr in {150,300,500} x Nl in {20,30,45},
                      40 poses each, sampled one complex after the other (one model handle per complex, all resident)
  configs4            large-pocket stress: 1500 residues / 80 atoms, 40 poses, 4.8 M cross edges per direction

Multi-GPU (--scaling): "strong" (default) = BASELINE configs[3] / north_star: the SAME 40 poses sharded in contiguous blocks
over the ranks (5 per GPU at 8); "weak" = every rank samples its own 40 poses of the complex.  At N = 1 both are the headline
run.  Either way the only collective is one RCCL all_gather of the final coordinates per step, as the reference's sampler
would hand them to the confidence model.

Timing: the K timed steps run with the library's per-kernel HIP-event timers OFF (plain production path); the per-phase
milliseconds and the roofline objects come from one extra, untimed step with the timers on (and, for the dominant kernel, one
more on a single stream: with two streams the launch durations overlap).

Cross graph: the untrained (random-weight) score model cannot keep the ligand in the pocket, and with the
reference's dynamic cutoff 3*sigma_tr+20 A the ligand would drift out of range and the cross graph would
thin out, shrinking the work.  The bench therefore pins the cross graph at its upper bound
(dynamic_max_cross=False, cutoff = cross_max_distance = 80 A: every ligand-atom x residue pair is an
edge, as SURVEY.md 8 assumes: 1.02 M edges per interaction layer at 40 poses) and reports the measured
edge count in `config`.

Extra objects on the JSON line: `roofline` (dominant kernel, HIP-event timed inside the timed region) and
`cpu_baseline` (bounded CPU sample of the same workload: the reference's own sampling() + CGModel executed under the
third-party stand-ins of tests/golden/make_golden.py when /root/reference is present -- kind "reference-executed" --
else the oracle = pure-torch restatement, kind "port", which is what the GPU box can run).
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
os.environ.setdefault("DDMI_HARNESS", "1")     # the DDMI_* route variables of tools/*.sh are harness knobs (diffdock_amd/lib.py)

from diffdock_amd.config import DDL_SYNTH  # noqa: E402
from diffdock_amd.dist import shard_bounds  # noqa: E402
from diffdock_amd.hetero import HeteroBatch  # noqa: E402
from diffdock_amd.model import MIScoreModel  # noqa: E402
from diffdock_amd.synth import make_complex, make_pose_list  # noqa: E402
from diffdock_amd.tables import default_tables  # noqa: E402
from diffdock_amd.weights import init_state_dict  # noqa: E402

INFERENCE_STEPS, SAMPLES = 20, 40
HBM_PEAK_GBS, MFMA_F32_PEAK_TFLOPS = 8000.0, 157.3     # /opt/skills/guides/MI355X_MICROARCH.md
TEMP = dict(temp_sampling=[1.170050527854316, 2.06391612594481, 7.044261621607846],       # default_inference_args.yaml
            temp_psi=[0.727287304570729, 0.9022615585677628, 0.5946212391366862],
            temp_sigma_data=[0.9299802531572672, 0.7464326999906034, 0.6943254174849822])
# (n_res, n_lig, complex seed) of every complex of a workload
WORKLOADS = {
    "configs2": dict(complexes=[(300, 30, 0)], samples=40, label="BASELINE configs[2]"),
    "configs1": dict(complexes=[(300, 30, 0)], samples=10, label="BASELINE configs[1]"),
    "mix": dict(complexes=[(nr, nl, 10 + 3 * i + j) for i, nr in enumerate((150, 300, 500)) for j, nl in enumerate((20, 30, 45))],
                samples=40, label="SURVEY 8d PDBBind-test-like mix (9 complexes, Nr in {150,300,500} x Nl in {20,30,45})"),
    "configs4": dict(complexes=[(1500, 80, 8)], samples=40, label="BASELINE configs[4] large-pocket stress"),
}
# last committed PMC pass for the dominant kernel (tools/round_profile.sh, separate --pmc passes, gfx950 corrections applied)
TRAFFIC_RECORD = os.path.join(ROOT, "profiles", "traffic_latest.json")
KERNEL_SOURCE = os.path.join(ROOT, "diffdock_amd", "csrc", "k_conv_tile.h")     # the device code of k_conv_fused / k_conv_grouped


def traffic_from_record(rec, kernel_source=KERNEL_SOURCE):
    """(bytes per launch, source note, L2 hit rate) of a counter record -- only when the record was collected on the kernel
    source that is in the tree now (sha256 of k_conv_tile.h stored by tools/traffic_json.py): counters of an older kernel are not
    replayed next to a newer one."""
    import hashlib
    want = rec.get("kernel_source_sha256")
    have = hashlib.sha256(open(kernel_source, "rb").read()).hexdigest() if os.path.exists(kernel_source) else None
    if want is None or want != have:
        return None, f"stale: {rec.get('source')} was collected on another k_conv_tile.h (re-run tools/round_profile.sh)", None
    return rec["bytes_per_launch"], rec["source"], rec.get("l2_hit_rate")


def scatter_traffic_from_record(rec):
    sc = rec.get("scatter") if traffic_from_record(rec)[0] is not None else None
    return sc["bytes_per_launch"] if sc else None


def bench_cfg():
    return DDL_SYNTH.replace(dynamic_max_cross=False, cross_max_distance=80.0)


def t_schedule(steps):
    return np.linspace(1, 0, steps + 1)[:-1]      # get_t_schedule('expbeta', alpha=beta=1), diffusion_utils.py:138-142


def conv_work(cfg, nL, nR, e_ll, e_lr, e_rr):
    """Algorithmic flops / bytes of every (layer, edge group) launch of the convolution kernels of one forward pass
    (DESIGN.md section 4).  NT = columns of a contracted node row (sum over paths of din*mul_out), K = 3ns + 1.
    k_conv_fused keeps the contracted rows in LDS: its HBM bytes are the x rows, the hidden rows and the messages, and the node
    term is counted once per gather NODE (tiles with more than four distinct nodes repeat it per 32-edge virtual node -- that
    repetition is not algorithmic work).  `ref_flops` = the same launch priced by SURVEY 8(d)'s formula
    for the REFERENCE's association (per edge 2*(3ns*3ns + 3ns*W) + 6W, W = weight_numel): the re-association removes
    13x of it, so a fraction of peak computed with it would exceed 1."""
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
        W = sum(p.mul_in * p.mul_out for p in table)             # weight_numel of the layer
        d_in, d_out = sum(x.dim for x in parse_irreps(a)), sum(x.dim for x in parse_irreps(b))
        # (gather nodes, target nodes, edges, gather side is receptor)
        groups = [(nL, nL, e_ll, False), (nR, nL, e_lr, True), (nR, nR, e_rr, True), (nL, nR, e_lr, False)]
        for gi, (gcount, tcount, E, rec_gather) in enumerate(groups if l < L - 1 else groups[:2]):
            H = 3 * cfg.ns
            node_flops, edge_flops = 2.0 * gcount * HK * mac_node, 2.0 * E * HK * NT
            ref_flops = E * (2.0 * (H * H + H * W) + 6.0 * W)
            out.append({"k_conv_fused": {"flops": node_flops + edge_flops, "ref_flops": ref_flops, "group": gi,
                                         "bytes": gcount * d_in * 4.0 + E * (H * 4.0 + d_out * 4.0)}})
    return out


def copy_poses(dl):
    import copy
    return copy.deepcopy(dl)


def cpu_baseline(cfg, sd, so3_t, tor_t, g, full=False):
    """Bounded CPU sample of the same workload on this host's cores: the first steps of the 20-step schedule (largest cross
    cutoffs = the same all-pairs graph as the GPU run) for a few poses, extrapolated linearly to 20 steps."""
    threads = min(os.cpu_count(), 64)     # the small einsums of the per-edge tensor product do not scale past a few dozen threads
    torch.set_num_threads(threads)
    s = t_schedule(INFERENCE_STEPS)
    big = g["receptor"].pos.shape[0] * g["ligand"].pos.shape[0] > 50000
    if os.path.isdir("/root/reference") and os.path.exists(os.path.join(ROOT, "tests", "golden", "make_golden.py")):
        # the reference's own python: utils/sampling.sampling + models/cg_model.CGModel (FasterTensorProduct conv layers)
        import copy
        from functools import partial
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        from make_golden import install_stubs
        cwd = os.getcwd()
        scratch = os.path.join(ROOT, ".scratch", "tables")
        os.makedirs(scratch, exist_ok=True)
        os.chdir(scratch)                  # utils/so3.py / utils/torus.py cache their tables in the working directory
        try:
            install_stubs()
            from models import tensor_layers
            tensor_layers.FasterTensorProduct.irreps_out = property(lambda self: self.out_irreps)   # reference defect, make_golden.py
            from utils.utils import get_model
            from utils.diffusion_utils import t_to_sigma as t_to_sigma_compl
            from utils import sampling as ref_sampling
            args = cfg.to_namespace()
            t_to_sigma = partial(t_to_sigma_compl, args=args)
            model = get_model(args, torch.device("cpu"), t_to_sigma=t_to_sigma, no_parallel=True)
            model.load_state_dict(sd, strict=False)
            model.eval()
            n_s, n_steps = (1, 2) if big else (4, 5)
            dl = make_pose_list(g, n_s, tr_sigma_max=cfg.tr_sigma_max, seed=77, initial_noise_std_proportion=0.3)
            torch.manual_seed(0)
            t0 = time.time()
            ref_sampling.sampling(copy.deepcopy(dl), model, n_steps, s, s, s, torch.device("cpu"), t_to_sigma, args,
                                  batch_size=n_s, no_final_step_noise=False, **TEMP)
            dt = time.time() - t0
        finally:
            os.chdir(cwd)
        kind = "reference-executed"
    else:
        from oracle.cg_model import CGModelOracle
        from oracle.sampling import sampling as oracle_sampling
        n_s, n_steps = (1, 2) if big else (2, 3)
        dl = make_pose_list(g, n_s, tr_sigma_max=cfg.tr_sigma_max, seed=77, initial_noise_std_proportion=0.3)
        model = CGModelOracle(cfg, sd, so3_t, tor_t)
        R = int(dl[0]["ligand"].edge_mask.sum())
        gen = torch.Generator().manual_seed(0)
        noise = (torch.randn(INFERENCE_STEPS, n_s, 3, generator=gen), torch.randn(INFERENCE_STEPS, n_s, 3, generator=gen),
                 torch.randn(INFERENCE_STEPS, n_s * R, generator=gen))
        # the per-edge einsums of the restated tensor product stop scaling after a few dozen threads: time the sample at 16 threads
        # and at every core (<= 64), report the faster one as the baseline and both in `thread_scaling`
        scaling = {}
        for th in ([threads] if big or threads <= 16 else [16, threads]):
            torch.set_num_threads(th)
            t0 = time.time()
            oracle_sampling(copy_poses(dl), model, n_steps, cfg, noise, schedules=(s, s, s), batch_size=n_s, **TEMP)
            scaling[th] = time.time() - t0
        threads = min(scaling, key=scaling.get)
        dt = scaling[threads]
        thread_scaling = {str(th): round(n_s / (t / n_steps * INFERENCE_STEPS), 5) for th, t in scaling.items()}
        kind = "port"
    poses_per_s = n_s / (dt / n_steps * INFERENCE_STEPS)
    return {"value": poses_per_s, "unit": "poses/s", "cores": threads, "kind": kind,
            **({"thread_scaling_poses_per_s": thread_scaling} if kind == "port" else {}),
            "sample": f"{n_s} poses x the first {n_steps} of {INFERENCE_STEPS} steps of the "
                      f"{g['receptor'].pos.shape[0]}-residue / {g['ligand'].pos.shape[0]}-atom complex, same weights, extrapolated "
                      f"linearly to {INFERENCE_STEPS} steps; torch {torch.__version__}, {threads} threads, {dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="configs2", choices=sorted(WORKLOADS))
    ap.add_argument("--samples", type=int, default=None, help="poses per complex (per GPU with weak scaling); default: the workload's")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="multi-GPU: strong (default) = the same 40 poses sharded over the GPUs (BASELINE configs[3]); weak = 40 poses per GPU")
    ap.add_argument("--no-serialised-pass", action="store_true",
                    help="skip the extra one-stream pass behind roofline.serialised (kernel-trace profiles: one launch regime only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-small-batch", action="store_true", help="skip the extra 5- and 10-pose runs behind the `small_batch` key of the default line")
    ap.add_argument("--lib", default=None, help="path of an alternative libddmi build (kernel A/B experiments)")
    ap.add_argument("--edge-product", default="f32", choices=["f32", "bf16x4"],
                    help="arithmetic of the per-edge product of the interaction layers (ddmi_config.edge_product): f32 = exact fp32 chain "
                         "(the headline), bf16x4 = split-bf16 operands on the bf16 matrix pipe, fp32 accumulation (secondary line, its own dtype)")
    ap.add_argument("--pose-shards", type=int, default=1,
                    help="split the poses of a complex on ONE GPU into K independent batches, each with its own model handle and HIP "
                         "stream, enqueued back to back: the layer-boundary phases of one shard (reduce, per-node terms, hidden rows -- "
                         "kernels that cannot fill the chip) overlap the convolution kernels of the others.  Poses are independent "
                         "trajectories (utils/sampling.py:80,91-93); noise is keyed by global sample id, so the poses are the one-batch poses")
    ap.add_argument("--tile-per-pose", dest="tile_per_pose", action="store_true", default=None,
                    help="ddmi_exec_options.tile_per_pose: tiles of k_conv_fused never span two poses -> a pose's arithmetic does not depend on "
                         "its neighbours in the batch (bit-exact shard invariance).  Default: ON for --gpus N > 1 (measured cost 1.9 - 2.6 %: 148.2 vs "
                         "151.1 and 150.9 vs 154.9 poses/s at 40 poses, profiles/r05_v1 / r05_v2_bench_tile_per_pose.json), off at N = 1")
    ap.add_argument("--no-tile-per-pose", dest="tile_per_pose", action="store_false")
    ap.add_argument("--layer-overlap", choices=["joined", "auto", "always"], default="joined",
                    help="ddmi_exec_options.layer_overlap: joined = all groups of a layer, then one node update (the default), auto = layer "
                         "boundaries overlapped for chip-filling batches, always = for every batch size (neutral at 40 poses, profiles/r05_e11_ab.txt)")
    ap.add_argument("--fixed-center-conv", action="store_true",
                    help="build the model with fixed_center_conv (models/cg_model.py:371-374: the default indexes the ligand table by graph id, "
                         "so a pose's score depends on its position in the batch -- in the reference too)")
    ap.add_argument("--verify-shards", dest="verify_shards", action="store_true", default=None,
                    help="multi-rank strong runs (default ON there): after the timed region sample once more with a fixed seed, gather, and let "
                         "rank 0 compare the gathered poses with the same blocks sampled on its own GPU and -- fixed_center_conv, a second model "
                         "handle when the timed model is not -- with all poses in ONE batch (readiness check of the sharded path; key `shard_check`)")
    ap.add_argument("--no-verify-shards", dest="verify_shards", action="store_false")
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
        if os.environ.get("DDMI_BENCH_SHARE_GPU"):     # test hook: all ranks on cuda:0 over gloo (exercises the multi-rank code path on a 1-GPU box)
            local_rank = 0
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)

    wl = WORKLOADS[args.config]
    cfg = bench_cfg()
    if args.all_atoms:
        cfg = cfg.replace(all_atoms=True)
    if args.edge_product != "f32":
        cfg = cfg.replace(edge_product=args.edge_product)
    if args.fixed_center_conv:
        cfg = cfg.replace(fixed_center_conv=True)
    if args.tile_per_pose is None:
        args.tile_per_pose = world > 1          # sharded runs: every rank's poses as the one-batch run would compute them
    if args.tile_per_pose:
        cfg = cfg.replace(exec_options=tuple(cfg.exec_options) + (("tile_per_pose", 1),))
    if args.layer_overlap != "joined":
        cfg = cfg.replace(exec_options=tuple(cfg.exec_options) + (("layer_overlap", {"auto": 1, "always": 2}[args.layer_overlap]),))
    sd = init_state_dict(cfg, seed=1234)
    so3_t, tor_t = default_tables()
    S = args.samples or wl["samples"]              # poses per complex
    strong = args.scaling == "strong" and world > 1
    if args.verify_shards is None:
        args.verify_shards = strong
    lo, hi = shard_bounds(S, rank, world) if strong else (0, S)
    B = hi - lo                                    # poses of each complex on this rank
    sched = t_schedule(INFERENCE_STEPS)
    K = max(1, min(args.pose_shards, B)) if B > 0 else 1

    def new_model():
        mdl = MIScoreModel(cfg, device=str(dev), lib_path=args.lib)
        mdl.load_state_dict(sd)
        mdl.set_tables(so3_t, tor_t)
        return mdl

    def make_shards(dl, ids, models=None):
        """K contiguous blocks of the rank's poses: (model handle, collated batch, sample ids, stream) each"""
        out = []
        for k in range(K):
            a, b = shard_bounds(len(dl), k, K)
            if b > a:
                out.append(dict(model=models[k] if models else new_model(), batch=HeteroBatch.from_data_list(dl[a:b]).to(dev), ids=ids[a:b],
                                stream=torch.cuda.Stream(device=dev) if K > 1 else None, n=b - a))
        return out

    jobs = []
    for (n_res, n_lig, cseed) in wl["complexes"]:
        g = make_complex(seed=cseed, n_res=n_res, n_lig=n_lig, all_atoms=args.all_atoms)
        # strong: all ranks draw the same S initial poses and keep their block; weak: every rank its own S poses
        dl = make_pose_list(g, S, tr_sigma_max=cfg.tr_sigma_max, seed=1000 + (0 if strong else rank), initial_noise_std_proportion=0.3)
        dl = dl[lo:hi]
        ids = list(range(lo, hi)) if strong else list(range(rank * S, (rank + 1) * S))
        shards = make_shards(dl, ids) if B > 0 else []
        if not shards:
            shards = [dict(model=new_model(), batch=None, ids=[], stream=None, n=0)]
        cap = max(shard_bounds(S, r, world)[1] - shard_bounds(S, r, world)[0] for r in range(world)) if strong else S
        gathered = [torch.empty(cap * n_lig, 3, device=dev) for _ in range(world)] if world > 1 else None
        jobs.append(dict(model=shards[0]["model"], shards=shards, g=g, batch=shards[0]["batch"], ids=ids, n_res=n_res, n_lig=n_lig,
                         gathered=gathered, cap=cap))

    def sample_job(j, seed):
        """the 20-step loop of every pose shard of one complex; shards on their own streams, enqueued back to back"""
        live = [sh for sh in j["shards"] if sh["batch"] is not None]
        if not live:
            return torch.zeros(0, 3, device=dev)
        if len(live) == 1 and live[0]["stream"] is None:
            sh = live[0]
            return sh["model"].sample_batch(sh["batch"], INFERENCE_STEPS, (sched, sched, sched), seed=seed, sample_ids=sh["ids"],
                                            no_final_step_noise=True, **TEMP)
        cur = torch.cuda.current_stream(dev)
        outs = []
        for sh in live:
            sh["stream"].wait_stream(cur)
            with torch.cuda.stream(sh["stream"]):
                outs.append(sh["model"].sample_batch(sh["batch"], INFERENCE_STEPS, (sched, sched, sched), seed=seed, sample_ids=sh["ids"],
                                                     no_final_step_noise=True, **TEMP))
        for sh, o in zip(live, outs):
            cur.wait_stream(sh["stream"])
            o.record_stream(cur)
        return torch.cat(outs)

    gather_mode = os.environ.get("DDMI_BENCH_GATHER", "drain")      # drain (default) | enqueue (the collective behind the loop, no host sync) | host
    trace_on = bool(os.environ.get("DDMI_BENCH_TRACE"))
    trace_sync = os.environ.get("DDMI_BENCH_TRACE") == "2"      # 2: also drain the GPU behind the enqueue (changes what the all_gather waits for)
    t_origin = time.perf_counter()

    def trace(what):
        """DDMI_BENCH_TRACE=1: per-rank wall clocks of the multi-rank step (stderr); `sync` entries drain the GPU first"""
        if trace_on:
            print(f"[trace rank {rank}] {time.perf_counter() - t_origin:10.4f} s  {what}", file=sys.stderr, flush=True)

    def one_step(seed, jobs_=None):
        last = None
        for j in (jobs if jobs_ is None else jobs_):
            trace(f"seed {seed}: enqueue sample_job ({len(j['ids'])} poses)")
            pos = sample_job(j, seed)
            trace("  sample_job enqueued")
            if trace_sync:
                torch.cuda.synchronize()
                trace("  sample_job drained (sync)")
            if world > 1:     # one all_gather per complex: the final coordinates of every pose on every rank
                buf = pos
                if pos.shape[0] != j["cap"] * j["n_lig"]:
                    buf = torch.zeros(j["cap"] * j["n_lig"], 3, device=dev)
                    buf[:pos.shape[0]] = pos
                if gather_mode == "drain":      # the rank's 20-step loop drained before the collective is entered
                    torch.cuda.current_stream(dev).synchronize()
                    trace("  loop drained")
                if gather_mode == "host":       # test hook: the collective on host tensors (no device-side stream of the backend involved)
                    hb = buf.cpu()
                    hl = [torch.empty_like(hb) for _ in range(world)]
                    dist.all_gather(hl, hb)
                    for dst, src in zip(j["gathered"], hl):
                        dst.copy_(src)
                else:
                    dist.all_gather(j["gathered"], buf)
                trace("  all_gather returned")
            last = pos
        return last

    def timed(steps, jobs_=None):
        """K steps bracketed by barrier + synchronize on both sides; MAX over the ranks."""
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = None
        for k in range(steps):
            last = one_step(100 + k, jobs_)
        if world > 1:
            dist.barrier()
            trace("barrier behind the timed steps returned")
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt_], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_ = float(tmax)
        return dt_, last

    for w in range(args.warmup):
        one_step(w)
    dt, pos = timed(args.steps)           # the timed region: per-kernel timers off
    weak_extra = None
    if strong:
        # extra key: the weak-scaling figure (every rank its own S poses of each complex), one warm-up + one timed step
        wjobs = []
        for j in jobs:
            dlw = make_pose_list(j["g"], S, tr_sigma_max=cfg.tr_sigma_max, seed=1000 + rank, initial_noise_std_proportion=0.3)
            widx = list(range(rank * S, (rank + 1) * S))
            wsh = [dict(model=j["model"], batch=HeteroBatch.from_data_list(dlw).to(dev), ids=widx, stream=None, n=S)]
            wjobs.append(dict(j, shards=wsh, batch=wsh[0]["batch"], ids=widx, cap=S,
                              gathered=[torch.empty(S * j["n_lig"], 3, device=dev) for _ in range(world)]))
        one_step(7, wjobs)
        dtw, _ = timed(1, wjobs)
        weak_extra = {"value": world * S * len(jobs) / dtw, "unit": "poses/s", "poses_per_gpu": S * len(jobs), "steps": 1,
                      "note": "weak scaling: every rank samples its own poses (not the BASELINE configs[3] partition)"}
        del wjobs
    shard_check = None
    if strong and args.verify_shards:
        # readiness check of the sharded path, behind the timed region (utils/sampling.py:80,91-93: poses are independent trajectories,
        # noise keyed by global sample id):
        #  (1) every rank's gathered block against the SAME block sampled by rank 0 as its own batch, with the bench's model;
        #  (2) the gathered poses against all S poses sampled in ONE batch on rank 0 -- with a model built with fixed_center_conv:
        #      the reference's default centre convolution (models/cg_model.py:371-374) indexes the ligand table by GRAPH id, so a
        #      pose's score depends on its position in the batch, in the reference too, and a batch of 40 and eight batches of 5
        #      are different functions under it (3.5-4 A apart after 20 steps).  When the bench's own model is not "fixed", a second
        #      handle is.  tile_per_pose (the default here) makes (2) bit-exact.
        def gathered_blocks(j):
            return torch.cat([j["gathered"][r][:(shard_bounds(S, r, world)[1] - shard_bounds(S, r, world)[0]) * j["n_lig"]] for r in range(world)])
        one_step(4242)
        torch.cuda.synchronize()
        worst, blocks = 0.0, []
        if rank == 0:
            mdl = jobs[0]["model"]
            for j in jobs:
                dlf = make_pose_list(j["g"], S, tr_sigma_max=cfg.tr_sigma_max, seed=1000, initial_noise_std_proportion=0.3)
                for r in range(world):
                    b0, b1 = shard_bounds(S, r, world)
                    if b1 <= b0:
                        continue
                    got = j["gathered"][r][:(b1 - b0) * j["n_lig"]].reshape(b1 - b0, j["n_lig"], 3)
                    mine = mdl.sample_batch(HeteroBatch.from_data_list(dlf[b0:b1]).to(dev), INFERENCE_STEPS, (sched, sched, sched), seed=4242,
                                            sample_ids=list(range(b0, b1)), no_final_step_noise=True, **TEMP).reshape(b1 - b0, j["n_lig"], 3)
                    worst = max(worst, float((got - mine).abs().max()))
                    blocks.append([b0, b1])
                mdl.invalidate_complex()
        # (2): every rank samples its block once more with the fixed-centre model, one more all_gather, rank 0 samples the one batch
        cfg_fix = cfg if cfg.fixed_center_conv else cfg.replace(fixed_center_conv=True)
        worst_one = 0.0
        for j in jobs:
            mfix = j["model"] if cfg.fixed_center_conv else MIScoreModel(cfg_fix, device=str(dev), lib_path=args.lib)
            if mfix is not j["model"]:
                mfix.load_state_dict(sd)          # (fixed_center_conv changes no weight)
                mfix.set_tables(so3_t, tor_t)
            jf = dict(j, model=mfix, shards=[dict(sh, model=mfix) for sh in j["shards"]])
            one_step(4242, [jf])
            torch.cuda.synchronize()
            if rank == 0:
                dlf = make_pose_list(j["g"], S, tr_sigma_max=cfg.tr_sigma_max, seed=1000, initial_noise_std_proportion=0.3)
                full = mfix.sample_batch(HeteroBatch.from_data_list(dlf).to(dev), INFERENCE_STEPS, (sched, sched, sched), seed=4242,
                                         sample_ids=list(range(S)), no_final_step_noise=True, **TEMP).reshape(S, j["n_lig"], 3)
                worst_one = max(worst_one, float((gathered_blocks(j).reshape(S, j["n_lig"], 3) - full).abs().max()))
            mfix.invalidate_complex()
            j["model"].invalidate_complex()
        if rank == 0:
            shard_check = {"max_abs_diff_vs_same_block_on_rank0_angstrom": worst, "max_abs_diff_vs_one_batch_angstrom": worst_one,
                           "one_batch_model": "fixed_center_conv" + ("" if cfg.fixed_center_conv else " (second handle: the timed model keeps the reference default)"),
                           "blocks": blocks[:world], "tolerance_angstrom": 1e-3, "ok": bool(worst <= 1e-3 and worst_one <= 1e-3),
                           "note": "20-step final ligand coordinates gathered from the ranks vs (1) the same blocks sampled on rank 0 with the timed "
                                   "model, (2) all poses in one batch with fixed_center_conv; per-sample Philox streams keyed by global sample id"}
    small_batch = None
    if world == 1 and args.config == "configs2" and args.samples is None and not args.all_atoms and not args.no_small_batch:
        # the batch ONE GPU runs when configs[3] shards the 40 poses over 8 / 4 GPUs: one warm-up + one measured 20-step run each,
        # outside the timed region of the headline (own model handle; reported so that the driver's record shows the strong-scaling share)
        small_batch = {}
        j0 = jobs[0]
        for nb in (5, 10):
            dls = make_pose_list(j0["g"], nb, tr_sigma_max=cfg.tr_sigma_max, seed=1000, initial_noise_std_proportion=0.3)
            mb = new_model()
            bb = HeteroBatch.from_data_list(dls).to(dev)
            run = lambda seed: mb.sample_batch(bb, INFERENCE_STEPS, (sched, sched, sched), seed=seed, sample_ids=list(range(nb)), no_final_step_noise=True, **TEMP)
            run(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(2):
                run(100 + k)
            torch.cuda.synchronize()
            dts = (time.perf_counter() - t0) / 2
            e_ll, e_lr, e_rr = int(mb.debug_buffer("goff_ll")[-1]), int(mb.debug_buffer("offs_l")[-1]), int(mb.debug_buffer("rr_goff")[-1])
            fl = sum(w["k_conv_fused"]["flops"] for w in conv_work(cfg, nb * j0["n_lig"], nb * j0["n_res"], e_ll, e_lr, e_rr))
            small_batch[f"{nb}_poses"] = {"value": nb / dts, "unit": "poses/s", "ms_per_forward": dts / INFERENCE_STEPS * 1e3,
                                          "roofline_frac_wall": fl * INFERENCE_STEPS / dts / 1e12 / MFMA_F32_PEAK_TFLOPS}
            del mb, bb
        small_batch["note"] = ("one GPU's share of BASELINE configs[3] at 8 / 4 GPUs (5 / 10 poses of the same complex), 2 measured 20-step runs each "
                               "after one warm-up, outside the headline's timed region; roofline_frac_wall as roofline.frac")
    # one more, untimed step with the per-kernel HIP-event timers on: phase table and kernel-level roofline figures
    all_models = [sh["model"] for j in jobs for sh in j["shards"]]
    for mdl in all_models:
        mdl.set_kernel_timing(True)
    one_step(999)
    torch.cuda.synchronize()
    timings = {}
    for mdl in all_models:
        for k, (ms, n) in mdl.kernel_timings().items():
            a = timings.get(k, (0.0, 0))
            timings[k] = (a[0] + ms, a[1] + n)
        mdl.set_kernel_timing(False)
    timed_steps = 1                       # steps behind `timings`
    assert os.environ.get("DDMI_BENCH_NOCHECK") or torch.isfinite(pos).all()   # NOCHECK: timing-only ablation builds

    if rank == 0:
        # edges actually processed by the last forward of the run, per complex
        edges, work = [], []
        for j in jobs:
            tot = dict(lig_lig=0, cross_each_direction=0, rec_rec=0)
            extra = dict(lig_atom_each_direction=0, atom_atom=0, atom_rec_each_direction=0, atoms=0)
            for sh in j["shards"]:
                if sh["batch"] is None:
                    continue
                m = sh["model"]
                e_ll, e_lr, e_rr = int(m.debug_buffer("goff_ll")[-1]), int(m.debug_buffer("offs_l")[-1]), int(m.debug_buffer("rr_goff")[-1])
                tot["lig_lig"] += e_ll; tot["cross_each_direction"] += e_lr; tot["rec_rec"] += e_rr
                if args.all_atoms:
                    extra["lig_atom_each_direction"] += int(m.debug_buffer("offs_la_l")[-1]); extra["atom_atom"] += int(m.debug_buffer("aa_goff")[-1])
                    extra["atom_rec_each_direction"] += int(m.debug_buffer("ar_goff")[-1]); extra["atoms"] += int(sh["batch"]["atom"].pos.shape[0])
                work += conv_work(cfg, sh["n"] * j["n_lig"], sh["n"] * j["n_res"], e_ll, e_lr, e_rr)     # one entry per launch
            edges.append(dict(n_res=j["n_res"], n_lig=j["n_lig"], **tot, **(extra if args.all_atoms else {})))
        n_forwards = timed_steps * INFERENCE_STEPS * len(jobs)                  # forwards behind `timings`
        n_forwards_timed = args.steps * INFERENCE_STEPS * len(jobs)             # forwards inside the timed region
        # ---- roofline of the dominant kernel.  Headline `frac` = the kernel's algorithmic flops of all forwards of the TIMED
        # region / the region's wall clock / peak: the one figure a driver's own clock can check.  The kernel-level views (launch
        # durations from the extra timed-with-events step: overlapping with two streams; serialised on one stream) are sub-objects.
        kern = {k: v for k, v in timings.items() if k.startswith("k_") or k == "conv_fc1_gemms"}
        dom = max(kern, key=lambda k: kern[k][0]) if kern else None
        roof, roof_scatter = None, None
        if dom == "k_conv_fused" and not args.all_atoms:
            ms, n = kern[dom]
            avg_s = ms / max(n, 1) * 1e-3
            w_dom = [w[dom] for w in work if dom in w]    # one entry per (layer, edge group) of one forward of every complex
            # launches of this kernel in one forward of every complex, as MEASURED by the event pass: 22 per complex with one
            # k_conv_fused launch per (layer, edge group), 6 with the grouped dispatch (one k_conv_grouped launch per layer)
            n_launch = max(1, round(n / max(n_forwards // len(jobs), 1)))
            grouped_launches = n_launch < len(w_dom)
            flops = sum(w["flops"] for w in w_dom) / n_launch
            ref_flops = sum(w["ref_flops"] for w in w_dom) / n_launch
            bytes_ = sum(w["bytes"] for w in w_dom) / n_launch
            traffic, traffic_src, l2_hit = None, None, None
            if os.path.exists(TRAFFIC_RECORD):        # counter bytes are collected by a separate rocprofv3 --pmc pass (tools/round_profile.sh)
                rec = json.load(open(TRAFFIC_RECORD))
                if rec.get("kernel") == dom and rec.get("config") == args.config and args.samples is None and world == 1:
                    traffic, traffic_src, l2_hit = traffic_from_record(rec)
            conv_flops_fwd = sum(w["flops"] for w in w_dom) / len(jobs)                     # per forward of one complex
            wall_ach = conv_flops_fwd * n_forwards_timed / dt / 1e12                         # TFLOP/s over the timed region
            fwd_ms = timings.get("forward_total", (0.0, 0))[0] / max(n_forwards, 1)
            streams = 1 if os.environ.get("DDMI_STREAMS") == "1" else 2
            roof = {"kernel": dom, "bound": "mfma", "edge_product": args.edge_product, "achieved": wall_ach, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": wall_ach / MFMA_F32_PEAK_TFLOPS, "traffic": traffic,
                    "frac_definition": "algorithmic flops of this kernel in every forward of the timed region / wall clock of the timed "
                                       "region (all kernels, both streams) / dense f32 MFMA peak",
                    "traffic_source": traffic_src, "traffic_unit": "bytes per launch (FETCH_SIZE x 2 per the gfx950 correction + WRITE_SIZE)",
                    "l2_hit_rate": l2_hit,
                    "launches_per_forward": n_launch // len(jobs), "alg_flops_per_launch": flops, "alg_bytes_per_launch": bytes_,
                    "alg_flops_reference_assoc": ref_flops,
                    "forward_ms_timed_region": dt / n_forwards_timed * 1e3, "forward_ms_event_pass": fwd_ms,
                    "dispatch": "grouped: one k_conv_grouped launch per interaction layer walks the work items of all its edge groups"
                                if grouped_launches else "one k_conv_fused launch per (layer, edge group)",
                    ("kernel_grouped" if grouped_launches else "kernel_two_streams" if streams == 2 else "kernel_one_stream"): {
                        "avg_launch_ms": avg_s * 1e3, "launches": n, "achieved": flops / avg_s / 1e12,
                        "frac": flops / avg_s / 1e12 / MFMA_F32_PEAK_TFLOPS,
                        "note": "HIP events around every launch of one extra untimed step" +
                                ("; grouped launches run one after the other on one stream: the durations do not overlap" if grouped_launches else
                                 "; the two streams run launches concurrently, so these durations overlap (their sum exceeds the forward)"
                                 if streams == 2 else "")},
                    "alg_definition": "mean over the launches of this kernel in one forward -- all four edge groups of every "
                                      "layer, per launch as dispatched (exact f32 on v_mfma_f32_16x16x4_f32, peak = dense f32 MFMA); "
                                      "k_conv_fused: 2*145*sum(mul_in*mul_out*din) flop per gather node "
                                      "+ 2*145*NT flop per edge (the re-associated contraction, DESIGN 2: 13x fewer flops than the "
                                      "reference's association, which alg_flops_reference_assoc prices by SURVEY 8d's formula), "
                                      "bytes = x rows + 576 B hidden row + 624 B message per edge"}
            if streams == 2 and dom == "k_conv_fused" and world == 1 and len(jobs) == 1 and not args.no_serialised_pass:
                # the same kernel timed with the launches serialised on ONE stream (untimed extra pass, second handle)
                j = jobs[0]
                os.environ["DDMI_STREAMS"] = "1"
                m1 = MIScoreModel(cfg, device=str(dev), lib_path=args.lib)
                del os.environ["DDMI_STREAMS"]
                m1.load_state_dict(sd)
                m1.set_tables(so3_t, tor_t)
                sh0 = j["shards"][0]      # (with --pose-shards: the first shard's batch -- the launches the roofline's per-launch work describes)
                m1.sample_batch(sh0["batch"], INFERENCE_STEPS, (sched, sched, sched), seed=7, sample_ids=sh0["ids"], no_final_step_noise=True, **TEMP)
                m1.set_kernel_timing(True, level=2)       # one row per edge group: k_conv_fused:g0 .. g3
                m1.sample_batch(sh0["batch"], INFERENCE_STEPS, (sched, sched, sched), seed=8, sample_ids=sh0["ids"], no_final_step_noise=True, **TEMP)
                torch.cuda.synchronize()
                t1 = m1.kernel_timings()
                rows = {k: v for k, v in t1.items() if k.startswith(dom)}
                ms1, n1 = sum(v[0] for v in rows.values()), sum(v[1] for v in rows.values())
                flops1 = sum(w["flops"] for w in w_dom) / len(w_dom)        # per (layer, edge group) launch of this pass
                ach1 = flops1 / (ms1 / max(n1, 1) * 1e-3) / 1e12
                names = ["lig-lig", "lig<-rec", "rec-rec", "rec<-lig"]
                per_group = {}
                for gi in range(4):
                    r = rows.get(f"{dom}:g{gi}")
                    fl = sum(w["flops"] for w in w_dom if w.get("group") == gi) / max(K, 1) / len(jobs)   # per forward of the first shard
                    if r and r[0] > 0:
                        per_group[names[gi]] = {"ms_per_forward": r[0] / INFERENCE_STEPS, "launches_per_forward": r[1] // INFERENCE_STEPS,
                                                "frac": fl * INFERENCE_STEPS / (r[0] * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS}
                roof["serialised"] = {"dispatch": "one k_conv_fused launch per (layer, edge group) on one stream (level-2 kernel timers: per-group rows)",
                                      "avg_launch_ms": ms1 / max(n1, 1), "launches": n1, "achieved": ach1,
                                      "frac": ach1 / MFMA_F32_PEAK_TFLOPS,
                                      "forward_ms": t1["forward_total"][0] / max(t1["forward_total"][1], 1),
                                      "per_group": per_group}
                if "k_reduce_bn" in t1:
                    timings.setdefault("k_reduce_bn_serialised", t1["k_reduce_bn"])
                del m1
        if "k_reduce_bn" in timings and not args.all_atoms:
            # the scatter stage of the path (north_star: HBM GB/s on the gather / scatter): k_reduce_bn reads every message row once
            # (624 B per edge, fp32), adds the residual row and writes the node row
            L_ = cfg.num_conv_layers
            sc_bytes = 0.0
            prered_rows = []
            for e, j in zip(edges, jobs):
                nl_, nr_ = B * j["n_lig"], B * j["n_res"]
                # in-tile pre-reduction (round 4): of the lig<-rec group only ONE message row per (tile, target) is written and
                # read back -- count those rows from the tile headers of the last forward instead of one row per edge
                lr_rows = e["cross_each_direction"]
                try:
                    hdr = j["model"].debug_buffer("prered_tile_hdr")
                    try:
                        nvn = int(j["model"].debug_buffer("vn_count_cross")[0])     # (tile_per_pose: the padded list)
                    except Exception:
                        nvn = int(j["model"].debug_buffer("vn_off_cross")[-1])
                    nt_ = (nvn + 15) // 16
                    hdr = hdr[:nt_]
                    pre = hdr[:, 0] != 0
                    ne_tile = j["model"].debug_buffer("vn_ne_cross")[:nt_ * 16].reshape(nt_, 16).sum(1)
                    # pre-reducing tiles write one row per target, the others (a tile that spans two poses) one per edge
                    lr_rows = int((hdr[pre, 4:] >= 0).sum()) + int(ne_tile[~pre].sum())
                    prered_rows.append(lr_rows)
                except Exception:
                    pass
                for l in range(L_):
                    a_, b_ = cfg.layer_irreps(cfg.num_prot_emb_layers + l)
                    from diffdock_amd.irreps import parse_irreps
                    d_in, d_out = sum(x.dim for x in parse_irreps(a_)), sum(x.dim for x in parse_irreps(b_))
                    full = l < L_ - 1
                    n_edges = e["lig_lig"] + lr_rows + (e["cross_each_direction"] + e["rec_rec"] if full else 0)
                    n_nodes = nl_ + (nr_ if full else 0)
                    sc_bytes += 4.0 * (n_edges * d_out + n_nodes * (d_in + d_out))
            ms_s, n_s = timings.get("k_reduce_bn_serialised", timings["k_reduce_bn"])
            per_launch_s = ms_s / max(n_s, 1) * 1e-3
            per_launch_b = sc_bytes / (L_ * len(jobs))
            sc_traffic = None
            if os.path.exists(TRAFFIC_RECORD) and args.config == "configs2" and args.samples is None and world == 1 and K == 1:
                sc_traffic = scatter_traffic_from_record(json.load(open(TRAFFIC_RECORD)))
            roof_scatter = {"kernel": "k_reduce_bn", "bound": "hbm", "achieved": per_launch_b / per_launch_s / 1e9, "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": per_launch_b / per_launch_s / 1e9 / HBM_PEAK_GBS, "traffic": sc_traffic,
                            "traffic_unit": "bytes per launch from the same PMC passes as roofline.traffic (FETCH_SIZE x 2 + WRITE_SIZE)",
                            "avg_launch_ms": per_launch_s * 1e3, "alg_bytes_per_launch": per_launch_b,
                            "lig_rec_message_rows": prered_rows or None,
                            "alg_definition": "per interaction layer: 4 B x (D_out per message row read -- one per edge, for the pre-reduced "
                                              "lig<-rec group one per (tile, target) -- + D_in + D_out per "
                                              "target node); mean over the layers; launch durations from HIP events"
                                              + (" (one stream)" if "k_reduce_bn_serialised" in timings else "")}
            timings.pop("k_reduce_bn_serialised", None)
        cpu = None
        if not (args.no_cpu_baseline or args.all_atoms or world > 1):     # N = 1 only
            cpu = cpu_baseline(cfg, sd, so3_t, tor_t, jobs[len(jobs) // 2]["g"])
            if len(jobs) > 1:
                cpu["sample"] += " (the middle complex of the mix stands for all nine)"
            # the reference's OWN python (utils/sampling.sampling + models/cg_model.CGModel under the third-party stand-ins) cannot
            # travel to the GPU box; it was timed once on the 8-core build container (profiles/r02_cpu_reference_executed.json)
            rec_path = os.path.join(ROOT, "profiles", "r02_cpu_reference_executed.json")
            if cpu["kind"] == "port" and os.path.exists(rec_path) and args.config in ("configs2", "configs1"):
                rec = json.load(open(rec_path))
                cpu["reference_executed_build_box"] = {"value": rec["poses_per_s"], "unit": "poses/s", "cores": rec["cores"], "kind": "reference",
                                                       "sample": f"{rec['poses']} poses x {rec['steps']} steps, {rec['wall_s']:.0f} s wall, {rec['host']}",
                                                       "source": "profiles/r02_cpu_reference_executed.json (recorded, not re-timed in this run)"}
        (n_res0, n_lig0, _) = wl["complexes"][0]
        shape = f"synthetic {n_res0}-residue receptor / {n_lig0}-atom ligand" if len(jobs) == 1 else f"{len(jobs)} synthetic complexes"
        total_poses = (S if strong else world * S) * len(jobs)
        out = {
            "metric": "poses/sec (20 steps x 40 samples, DiffDock-L score model)" if not args.all_atoms else
                      "poses/sec (20 steps x 40 samples, all-atom score model -- secondary workload)",
            "value": total_poses * args.steps / dt,
            "unit": "poses/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32" if args.edge_product == "f32" else
                     "bf16x4-split edge product (operands as bf16 hi + bf16 lo, the four bf16 products of every f32 product on "
                     "v_mfma_f32_16x16x32_bf16, f32 accumulate); node contraction and everything else f32 -- SECONDARY line, the headline is dtype f32",
            "data": "synthetic",
            "config": {"workload": f"{wl['label']}: DDL-synth score model (ns=48 nv=10 6 layers sh_lmax=1), "
                                   f"{INFERENCE_STEPS} steps x {S} poses per complex, {shape}, cross graph pinned at its upper bound "
                                   f"(static 80 A cutoff), low-temperature SDE, random-init weights",
                       "name": args.config, "poses_per_complex": S, "poses_per_gpu": B * len(jobs), "inference_steps": INFERENCE_STEPS,
                       "complexes": len(jobs), "edges": edges[0] if len(edges) == 1 else edges, "tile_per_pose": bool(args.tile_per_pose),
                       # models/cg_model.py:371-374: the default centre convolution indexes the ligand table by GRAPH id, so a pose's score
                       # depends on its position in the batch (in the reference too): a sharded run reproduces the one-batch poses only with "fixed"
                       "center_conv": "fixed" if cfg.fixed_center_conv else "batch-indexed (reference default)",
                       "edges_per_layer": sum(e["lig_lig"] + 2 * e["cross_each_direction"] + e["rec_rec"] for e in edges),
                       "parallelism": ("single GPU" if world == 1 else
                                       f"strong: the {S} poses of a complex sharded in blocks over {world} GPUs, 1 all_gather per complex" if strong else
                                       f"weak: {S} poses per GPU x {world} GPUs (pose-sharded, 1 all_gather per complex)")},
            "roofline": roof, "roofline_scatter": roof_scatter, "cpu_baseline": cpu, "small_batch": small_batch, "weak_scaling": weak_extra, "shard_check": shard_check,
            "phase_ms_per_forward": {k: v[0] / max(n_forwards, 1) for k, v in timings.items()},
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
