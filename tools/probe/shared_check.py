"""GPU check of the shared-node contraction route: DDMI_FUSED_SHARED = 2 / 1 against 0 on a few complexes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from dataclasses import replace
import torch
from diffdock_amd.config import DDL_SYNTH
from diffdock_amd.hetero import HeteroBatch, set_time
from diffdock_amd.model import MIScoreModel
from diffdock_amd.synth import make_complex, make_pose_list
from diffdock_amd.weights import init_state_dict
from util import rel_err, tables

def run(cfg, sd, b, env):
    for k, v in env.items(): os.environ[k] = v
    m = MIScoreModel(cfg, device="cuda:0"); m.load_state_dict(sd); m.set_tables(*tables())
    out = [o.cpu() for o in m(b.to("cuda:0"))[:3]]
    nodes = [torch.from_numpy(m.debug_buffer(f"x{l + 1}")) for l in range(cfg.num_conv_layers)]
    return out, nodes

for (nres, nlig, layers, dense, cutoff) in [(75, 7, 4, "2", 80.0), (300, 30, 2, "1", 80.0), (300, 30, 2, "1", None), (40, 20, 3, "2", 80.0)]:
    cfg = replace(DDL_SYNTH, num_conv_layers=layers, lm_embedding_type=None, tr_sigma_max=5.0)
    if cutoff: cfg = replace(cfg, dynamic_max_cross=False, cross_max_distance=cutoff)
    sd = init_state_dict(cfg, seed=3)
    g = make_complex(seed=2, n_res=nres, n_lig=nlig, lm_dim=0)
    b = HeteroBatch.from_data_list(make_pose_list(g, 2, tr_sigma_max=cfg.tr_sigma_max, seed=5, initial_noise_std_proportion=0.3))
    set_time(b, 0.6, 0.6, 0.6, b.num_graphs)
    os.environ["DDMI_FUSED_DENSE"] = dense
    ref, refn = run(cfg, sd, b, {"DDMI_FUSED_SHARED": "0"})
    for sh in ("1", "2"):
        out, nodes = run(cfg, sd, b, {"DDMI_FUSED_SHARED": sh})
        print(nres, nlig, "dense", dense, "cutoff", cutoff, "shared", sh, "scores", [f"{rel_err(o, r):.2e}" for o, r in zip(out, ref)],
              "layers", [f"{rel_err(x, y):.2e}" for x, y in zip(nodes, refn)], flush=True)
