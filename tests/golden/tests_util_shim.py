"""make_golden.py helper: rebuild a committed fixture's inputs (the same code as tests/util.fixture_case, importable from here)."""
import os
import sys

import torch


def fixture_case_shim(golden_dir, name):
    sys.path.insert(0, os.path.join(os.path.dirname(golden_dir)))
    from util import graph_from_dict
    from diffdock_amd.config import ModelConfig
    fx = torch.load(os.path.join(golden_dir, name + ".pt"), weights_only=False)
    cfg = ModelConfig(**fx["cfg"])
    return fx, cfg, [graph_from_dict(fx["graph"], pos=p.clone()) for p in fx["poses"]]
