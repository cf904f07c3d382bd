"""Host enqueue time vs GPU time of one ddmi_sample call (20 steps): is the step loop launch-bound?  (GPU box)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
from diffdock_amd.hetero import HeteroBatch
from diffdock_amd.model import MIScoreModel
from diffdock_amd.synth import make_complex, make_pose_list
from diffdock_amd.tables import default_tables
from diffdock_amd.weights import init_state_dict
cfg = B.bench_cfg(); sd = init_state_dict(cfg, seed=1234)
m = MIScoreModel(cfg, device="cuda:0"); m.load_state_dict(sd); m.set_tables(*default_tables())
g = make_complex(seed=0, n_res=300, n_lig=30)
sched = B.t_schedule(20)
for nb in (5, 10, 40):
    dl = make_pose_list(g, nb, tr_sigma_max=cfg.tr_sigma_max, seed=1000, initial_noise_std_proportion=0.3)
    batch = HeteroBatch.from_data_list(dl).to("cuda:0")
    m.sample_batch(batch, 20, (sched,) * 3, seed=1, no_final_step_noise=True, **B.TEMP); torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.sample_batch(batch, 20, (sched,) * 3, seed=2, no_final_step_noise=True, **B.TEMP)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"B={nb:3d}: host enqueue of 20 steps {1e3*(t1-t0):7.1f} ms, until done {1e3*(t2-t0):7.1f} ms  ({(t1-t0)/(t2-t0):.0%} of the wall time)")
