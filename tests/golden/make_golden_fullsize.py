"""Reference-EXECUTED fixtures at the BASELINE.json shapes (run in THIS container; the GPU box has no reference):

    python tests/golden/make_golden_fullsize.py [traj] [big] [randpos]

Same harness as make_golden.py (the reference's own models/cg_model.py, models/tensor_layers.py, utils/sampling.py ...
run unmodified on top of stand-ins for the absent third-party wheels).  The inputs of these cases are too large to
commit (the DDL-synth state_dict alone is ~100 MB), so a fixture stores the SEEDS of the deterministic generators
(diffdock_amd.synth / diffdock_amd.weights.init_state_dict: numpy default_rng + a torch CPU generator) together with
input checksums, and the reference's outputs:

  traj_300_30.pt    utils/sampling.sampling() for 20 steps x NP poses of the 300-residue / 30-atom complex with the
                    DDL-synth score model, low-temperature SDE of default_inference_args.yaml: the recorded torch.normal
                    draws, per step the ligand positions the model saw and the tr / rot / tor scores it returned
                    (forward hook), the final poses -- plus the oracle run in float64 on the same draws, whose distance
                    from the float32 reference run is the trajectory's own sensitivity to rounding (the yardstick for
                    free-running parity).  Wall time of the reference run is written to profiles/ as the
                    reference-executed CPU baseline of this container.
  fwd_1500_80.pt    one CGModel.forward of BASELINE configs[4] (1500 residues / 80 atoms, every pair a cross edge), NB poses.
  fwd_300_30_b40.pt one CGModel.forward at bench.py's EXACT default batch (BASELINE configs[2]: 40 poses of the 300-residue /
                    30-atom complex, seeds of bench.py, static 80 A cross cutoff, default = batch-index-dependent centre
                    convolution models/cg_model.py:371-374).
  randpos.pt        utils/sampling.randomize_position with its scipy / numpy / torch draws recorded.
"""
import copy
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

TEMP = dict(temp_sampling=[1.170050527854316, 2.06391612594481, 7.044261621607846],       # default_inference_args.yaml
            temp_psi=[0.727287304570729, 0.9022615585677628, 0.5946212391366862],
            temp_sigma_data=[0.9299802531572672, 0.7464326999906034, 0.6943254174849822])


def checksum(tensors):
    """Order-independent fingerprint of a dict / list of tensors (float64 sum of |x| and of x * (1 + index mod 7))."""
    items = tensors.items() if isinstance(tensors, dict) else enumerate(tensors)
    s1 = s2 = 0.0
    for _, t in items:
        t = torch.as_tensor(t).double().reshape(-1)
        s1 += float(t.abs().sum())
        s2 += float((t * (1 + torch.arange(t.numel(), dtype=torch.float64) % 7)).sum())
    return [s1, s2]


def graph_tensors(g):
    return [g["receptor"].x, g["receptor"].pos, g["receptor", "receptor"].edge_index, g["ligand"].x, g["ligand"].pos,
            g["ligand"].edge_mask, g["ligand", "ligand"].edge_index, g["ligand", "ligand"].edge_attr,
            torch.from_numpy(np.asarray(g["ligand"].mask_rotate[0]))]


def setup_reference():
    from make_golden import install_stubs
    scratch = os.path.join(ROOT, ".scratch", "tables")
    os.makedirs(scratch, exist_ok=True)
    os.chdir(scratch)
    install_stubs()
    np.random.seed(0)
    from models import tensor_layers
    tensor_layers.FasterTensorProduct.irreps_out = property(lambda self: self.out_irreps)   # reference defect, see make_golden.py


def ref_model(cfg, sd):
    from functools import partial
    from utils.utils import get_model
    from utils.diffusion_utils import t_to_sigma as t_to_sigma_compl
    args = cfg.to_namespace()
    if cfg.fixed_center_conv:
        args.not_fixed_center_conv = False
    t_to_sigma = partial(t_to_sigma_compl, args=args)
    model = get_model(args, torch.device("cpu"), t_to_sigma=t_to_sigma, no_parallel=True)
    model.load_state_dict(sd, strict=False)
    model.eval()
    return model, args, t_to_sigma


def case_traj(n_poses=4, steps=20, spec_over=None, out="traj_300_30.pt", profile_json=True):
    from utils import sampling as ref_sampling
    from utils.diffusion_utils import get_t_schedule
    from diffdock_amd.config import DDL_SYNTH
    from diffdock_amd.synth import make_complex, make_pose_list
    from diffdock_amd.weights import init_state_dict
    from oracle.sampling import sampling as oracle_sampling
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import oracle_model, split_draws, tables
    spec = dict(cfg_replace=dict(dynamic_max_cross=False, cross_max_distance=80.0), n_res=300, n_lig=30, complex_seed=0,
                pose_seed=1000, noise_prop=0.3, weight_seed=1234, n_poses=n_poses, steps=steps)
    spec.update(spec_over or {})
    cfg = DDL_SYNTH.replace(**spec["cfg_replace"])
    sd = init_state_dict(cfg, seed=spec["weight_seed"])
    g = make_complex(seed=spec["complex_seed"], n_res=spec["n_res"], n_lig=spec["n_lig"], all_atoms=cfg.all_atoms)
    dl = make_pose_list(g, n_poses, tr_sigma_max=cfg.tr_sigma_max, seed=spec["pose_seed"],
                        initial_noise_std_proportion=spec["noise_prop"])
    model, args, t_to_sigma = ref_model(cfg, sd)
    rec = []
    hook = model.register_forward_hook(lambda m, inp, out: rec.append(dict(
        pos_in=inp[0]["ligand"].pos.detach().clone(), tr=out[0].detach().clone(), rot=out[1].detach().clone(),
        tor=out[2].detach().clone())))
    draws = []
    real_normal = torch.normal

    def rec_normal(*a, **kw):
        z = real_normal(*a, **kw)
        draws.append(z.clone())
        return z
    torch.manual_seed(11)
    ref_sampling.torch.normal = rec_normal
    sched = get_t_schedule("expbeta", steps)
    torch.set_num_threads(os.cpu_count())
    t0 = time.time()
    try:
        out_list, _ = ref_sampling.sampling(copy.deepcopy(dl), model, steps, sched, sched, sched, torch.device("cpu"), t_to_sigma,
                                            args, batch_size=n_poses, no_final_step_noise=True, **TEMP)
    finally:
        ref_sampling.torch.normal = real_normal
        hook.remove()
    wall = time.time() - t0
    final = torch.stack([d["ligand"].pos for d in out_list])
    print(f"reference sampling: {n_poses} poses x {steps} steps in {wall:.1f} s on {os.cpu_count()} cores", flush=True)
    fx = dict(spec=spec, temp=TEMP, draws=draws, steps=[{k: v for k, v in r.items()} for r in rec], final_pos=final,
              init_pos=torch.stack([d["ligand"].pos for d in dl]),
              checks=dict(state_dict=checksum(sd), graph=checksum(graph_tensors(g))),
              reference_wall_s=wall, reference_cores=os.cpu_count(), torch=torch.__version__)
    torch.save(fx, os.path.join(HERE, out))     # reference part first; the float64 yardstick is added below
    if profile_json:
      with open(os.path.join(ROOT, "profiles", "r02_cpu_reference_executed.json"), "w") as f:
        json.dump({"what": "reference utils/sampling.sampling + models/cg_model.CGModel (FasterTensorProduct conv layers) executed "
                           "under tests/golden/make_golden.py's third-party stand-ins, DDL-synth width, 300 residues / 30 atoms, "
                           "all-pairs cross graph", "poses": n_poses, "steps": steps, "wall_s": wall,
                   "poses_per_s": n_poses / wall, "cores": os.cpu_count(), "threads": torch.get_num_threads(),
                   "torch": torch.__version__, "host": "build container (not the GPU box)"}, f, indent=1)
    # the same trajectory in float64 (oracle, same draws): how far rounding alone moves the final poses
    R = int(dl[0]["ligand"].edge_mask.sum())
    noise = split_draws(draws, steps, n_poses, R)
    so3_t, tor_t = tables()
    o64 = oracle_model(cfg, sd, so3_t, tor_t, dtype=torch.float64)
    dl64 = copy.deepcopy(dl)
    for d in dl64:
        d["ligand"].pos = d["ligand"].pos.double()
        d["receptor"].pos = d["receptor"].pos.double()
        if cfg.all_atoms:
            d["atom"].pos = d["atom"].pos.double()
    n64 = tuple(z.double() for z in noise)
    rec64 = []
    out64 = oracle_sampling(dl64, o64, steps, cfg, n64, schedules=(sched, sched, sched), batch_size=n_poses,
                            no_final_step_noise=True, record=rec64, **TEMP)
    final64 = torch.stack([d["ligand"].pos for d in out64])
    rmsd64 = ((final64 - final.double()) ** 2).sum(-1).mean(-1).sqrt()
    print("final-pose RMSD float32 reference vs float64 oracle:", rmsd64.tolist(), flush=True)
    fx.update(final_pos_f64=final64, steps_f64=[dict(pos_in=r["pos_in"], tr=r["tr"], rot=r["rot"], tor=r["tor"]) for r in rec64])
    torch.save(fx, os.path.join(HERE, out))


def case_big(n_poses=1, spec=None, out="fwd_1500_80.pt"):
    from utils.diffusion_utils import set_time
    from diffdock_amd.config import DDL_SYNTH
    from diffdock_amd.hetero import HeteroBatch
    from diffdock_amd.synth import make_complex, make_pose_list
    from diffdock_amd.weights import init_state_dict
    spec = spec or dict(cfg_replace=dict(dynamic_max_cross=False, cross_max_distance=80.0), n_res=1500, n_lig=80, complex_seed=8,
                        pose_seed=9, noise_prop=0.1, weight_seed=1234, n_poses=n_poses, t=0.5)
    n_poses = spec["n_poses"]
    cfg = DDL_SYNTH.replace(**spec["cfg_replace"])
    sd = init_state_dict(cfg, seed=spec["weight_seed"])
    g = make_complex(seed=spec["complex_seed"], n_res=spec["n_res"], n_lig=spec["n_lig"])
    dl = make_pose_list(g, n_poses, tr_sigma_max=cfg.tr_sigma_max, seed=spec["pose_seed"],
                        initial_noise_std_proportion=spec["noise_prop"])
    model, args, _ = ref_model(cfg, sd)
    batch = HeteroBatch.from_data_list(copy.deepcopy(dl))
    set_time(batch, None, spec["t"], spec["t"], spec["t"], n_poses, False, torch.device("cpu"))
    layer_out = []
    nl = n_poses * spec["n_lig"]
    hooks = [l.register_forward_hook(lambda m, i, o: layer_out.append(o[:nl].detach().clone())) for l in model.conv_layers]
    t0 = time.time()
    with torch.no_grad():
        tr, rot, tor = model(batch)[:3]
    for h in hooks:
        h.remove()
    print(f"reference forward {spec['n_res']}/{spec['n_lig']} x {n_poses}: {time.time() - t0:.1f} s", flush=True)
    torch.save(dict(spec=spec, tr=tr, rot=rot, tor=tor, lig_rows=layer_out,
                    checks=dict(state_dict=checksum(sd), graph=checksum(graph_tensors(g))), torch=torch.__version__),
               os.path.join(HERE, out))
    print("tr", tr.tolist(), "tor[:4]", tor[:4].tolist())


def case_randpos():
    """utils/sampling.py:16-58 with every random draw recorded (np.random.uniform, scipy Rotation.random, torch.normal)."""
    from utils import sampling as ref_sampling
    from diffdock_amd.synth import make_complex
    from make_golden import graph_to_dict
    g = make_complex(seed=31, n_res=50, n_lig=16)
    out = {}
    for tag, prop in (("prop", 1.46), ("sigma", -0.5)):
        dl = [copy.deepcopy(g) for _ in range(3)]
        rec = dict(torsion=[], rotation=[], tr=[])
        real_uniform, real_normal, real_R = np.random.uniform, torch.normal, ref_sampling.R

        class RecR:
            @staticmethod
            def random(*a, **kw):
                r = real_R.random(*a, **kw)
                rec["rotation"].append(torch.from_numpy(r.as_matrix()).clone())
                return r

        def rec_uniform(*a, **kw):
            u = real_uniform(*a, **kw)
            rec["torsion"].append(np.array(u, copy=True))
            return u

        def rec_normal(*a, **kw):
            z = real_normal(*a, **kw)
            rec["tr"].append(z.clone())
            return z
        np.random.seed(5)
        torch.manual_seed(6)
        ref_sampling.np.random.uniform = rec_uniform
        ref_sampling.torch.normal = rec_normal
        ref_sampling.R = RecR
        try:
            ref_sampling.randomize_position(dl, False, False, 19.0, initial_noise_std_proportion=prop)
        finally:
            ref_sampling.np.random.uniform = real_uniform
            ref_sampling.torch.normal = real_normal
            ref_sampling.R = real_R
        out[tag] = dict(prop=prop, draws=rec, pos=torch.stack([d["ligand"].pos for d in dl]))
    torch.save(dict(graph=graph_to_dict(g), tr_sigma_max=19.0, cases=out), os.path.join(HERE, "randpos.pt"))
    print("randpos", out["prop"]["pos"][0, :2].tolist())


if __name__ == "__main__":
    which = sys.argv[1:] or ["randpos", "traj", "big"]
    setup_reference()
    if "randpos" in which:
        case_randpos()
    if "traj" in which:
        case_traj(n_poses=int(os.environ.get("TRAJ_POSES", 4)))
    if "aatraj" in which:   # all-atom model (models/aa_model.py): 20-step trajectory, 100 residues with their heavy atoms, 30-atom ligand;
                            # sh_lmax = 2: the reference FasterTensorProduct cannot run a step in which no ligand atom has a receptor atom in reach
        case_traj(n_poses=int(os.environ.get("TRAJ_POSES", 4)), out="traj_aa_100_30.pt", profile_json=False,
                  spec_over=dict(cfg_replace=dict(dynamic_max_cross=False, cross_max_distance=80.0, all_atoms=True, sh_lmax=2), n_res=100, n_lig=30,
                                 complex_seed=3, pose_seed=1003))
    if "big" in which:
        case_big(n_poses=int(os.environ.get("BIG_POSES", 1)))
    if "b40" in which:   # bench.py's default workload: same seeds (complex 0, poses 1000, weights 1234), 40 poses, one forward
        case_big(spec=dict(cfg_replace=dict(dynamic_max_cross=False, cross_max_distance=80.0), n_res=300, n_lig=30, complex_seed=0,
                           pose_seed=1000, noise_prop=0.3, weight_seed=1234, n_poses=40, t=0.7), out="fwd_300_30_b40.pt")
        fx = torch.load(os.path.join(HERE, "fwd_300_30_b40.pt"), weights_only=False)     # keep the ligand rows of the first and the last pose only
        rows = torch.cat([torch.arange(0, 30), torch.arange(39 * 30, 40 * 30)])
        fx["lig_rows_idx"] = rows
        fx["lig_rows"] = [r[rows].clone() for r in fx["lig_rows"]]
        torch.save(fx, os.path.join(HERE, "fwd_300_30_b40.pt"))
