import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# the DDMI_* route variables the tests set are honoured by diffdock_amd/lib.py only under the harness switch
os.environ.setdefault("DDMI_HARNESS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests are skipped, not failed, on a box without a GPU (a plain `pytest` then stays green on the CPU).
    On a GPU box nothing is skipped, and the HIP library must load: diffdock_amd.lib raises if libddmi.so is missing."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (no GPU visible)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
