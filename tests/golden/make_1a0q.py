"""tests/golden/1a0q_graph.pt: the reference's example complex data/1a0q (416 residues, 23 heavy atoms; BASELINE configs[0])
read by diffdock_amd.io in THIS container (the GPU box has no /root/reference), stored as plain arrays:

    python tests/golden/make_1a0q.py

tests/test_io.py re-parses the files when /root/reference is present and checks the arrays; the configs[0] plumbing tests
(4 steps x 2 samples, device loop vs oracle) run from the fixture."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from diffdock_amd.io import complex_graph  # noqa: E402
from make_golden import graph_to_dict  # noqa: E402

D = "/root/reference/data/1a0q"
g = complex_graph(f"{D}/1a0q_protein_processed.pdb", f"{D}/1a0q_ligand.sdf", lm_dim=0, name="1a0q")
d = graph_to_dict(g)
d["original_center"] = g.original_center
torch.save(d, os.path.join(HERE, "1a0q_graph.pt"))
print({k: tuple(v.shape) for k, v in d.items()})
