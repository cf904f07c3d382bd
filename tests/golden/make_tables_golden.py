"""Generate tests/golden/{so3_exp_score_norms,torus_score_norm}.npy by IMPORTING the
reference's own utils/so3.py and utils/torus.py (both need only numpy/scipy/torch/tqdm).

Run from the repo root in THIS container (the reference is absent on the GPU box):
    python tests/golden/make_tables_golden.py
utils/torus.py draws its Monte-Carlo samples from the unseeded global numpy RNG at import
(utils/torus.py:72-76); np.random.seed(0) is set immediately before the import so the
fixture is reproducible.  Both modules cache .npy files in the CWD, so we chdir into a
scratch directory first.
"""
import os, sys, time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
scratch = os.path.join(ROOT, ".scratch", "tables")
os.makedirs(scratch, exist_ok=True)
os.chdir(scratch)
sys.path.insert(0, "/root/reference")

t0 = time.time()
from utils import so3  # noqa: E402
np.save(os.path.join(HERE, "so3_exp_score_norms.npy"), so3._exp_score_norms)
print("so3 done", time.time() - t0, so3._exp_score_norms.shape)

t0 = time.time()
np.random.seed(0)
from utils import torus  # noqa: E402
np.save(os.path.join(HERE, "torus_score_norm.npy"), torus.score_norm_)
print("torus done", time.time() - t0, torus.score_norm_.shape)
