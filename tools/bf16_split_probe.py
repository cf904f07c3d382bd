"""Accuracy probe for a split-bf16 main loop, on the oracle: the two large contractions of every
interaction layer (second fc layer, per-edge tensor-product contraction) evaluated as sums of bf16 x bf16 products with
float32 accumulation -- 3 terms (hi*hi + hi*lo + lo*hi) or 6 terms (three-way split) -- against the float64 oracle."""
import sys, time
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle.layers as L
from diffdock_amd.config import DDL_SYNTH
from diffdock_amd.hetero import HeteroBatch, set_time
from diffdock_amd.synth import make_complex, make_pose_list
from diffdock_amd.weights import init_state_dict
from oracle.cg_model import CGModelOracle
from util import elem_excess, rel_err, tables

MODE = {"terms": 0}

def split(x, n):
    parts = []
    r = x
    for _ in range(n):
        p = r.to(torch.bfloat16).to(torch.float32)
        parts.append(p)
        r = r - p
    return parts

def split_mm(f, a, b):
    """f(a, b) bilinear; terms 3: a = hi+lo, b = hi+lo without lo*lo; 6: three-way split, terms with index sum <= 2."""
    t = MODE["terms"]
    if t == 0 or a.dtype != torch.float32:
        return f(a, b)
    n = 2 if t == 3 else 3
    A, B = split(a, n), split(b, n)
    out = None
    for i in range(n):
        for j in range(n):
            if i + j <= n - 1:
                v = f(A[i], B[j])
                out = v if out is None else out + v
    return out

orig_linear, orig_einsum = L.linear, torch.einsum
def linear(sd, name, x):
    if ".fc." in name and name.endswith(".3") and "conv_layers" in name:
        w = sd[name + ".weight"]
        return split_mm(lambda a, b: a @ b.t(), x, w) + sd[name + ".bias"]
    return orig_linear(sd, name, x)
L.linear = linear
class E:
    @staticmethod
    def einsum(eq, a, b):
        if eq in ("eu,euw->ew", "eum,euw->ewm"):
            return split_mm(lambda p, q: orig_einsum(eq, p, q), a, b)
        return orig_einsum(eq, a, b)
L.torch = type("T", (), {**{k: getattr(torch, k) for k in dir(torch) if not k.startswith("__")}, "einsum": E.einsum})

cfg = DDL_SYNTH.replace(dynamic_max_cross=False, cross_max_distance=80.0)
sd = init_state_dict(cfg, seed=3)
sd64 = {k: v.double() if v.is_floating_point() else v for k, v in sd.items()}
g = make_complex(seed=1, n_res=300, n_lig=30)
dl = make_pose_list(g, 2, tr_sigma_max=cfg.tr_sigma_max, seed=5)
so3_t, tor_t = tables()
for t in (0.9, 0.3, 0.05):
    b = HeteroBatch.from_data_list(dl)
    set_time(b, t, t, t, b.num_graphs)
    MODE["terms"] = 0
    ref64 = CGModelOracle(cfg, sd, so3_t, tor_t, torch.float64)(b)[:3]
    res = {}
    for terms in (0, 3, 6):
        MODE["terms"] = terms
        t0 = time.time()
        out = CGModelOracle(cfg, sd, so3_t, tor_t)(b)[:3]
        res[terms] = [(elem_excess(o, r), rel_err(o, r)) for o, r in zip(out, ref64)]
        print(f"t={t} terms={terms}: " + "  ".join(f"{n}: excess {e:.3f} rel {r:.2e}" for n, (e, r) in zip(("tr", "rot", "tor"), res[terms])), f"({time.time()-t0:.0f}s)", flush=True)
