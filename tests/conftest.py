import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    """Read-only view of one tests/golden/*.npz file."""

    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    def __contains__(self, k):
        return k in self._z.files

    def np(self, k):
        return self._z[k]

    def t(self, k):
        return torch.from_numpy(self._z[k].copy())  # copy(): C-contiguous, keeps 0-d

    def keys(self, prefix=""):
        return [k for k in self._z.files if k.startswith(prefix)]

    def state_dict(self, prefix="sd/"):
        return {k[len(prefix):]: self.t(k) for k in self.keys(prefix)}

    def levels(self, name):
        n = int(self._z[f"{name}/n_levels"])
        es = [self.t(f"{name}/e{l}") for l in range(n)]
        ids = [self.t(f"{name}/ids{l}") for l in range(n - 1)]
        return es, ids


@pytest.fixture(scope="session")
def graphs():
    return Golden("graphs")


def load_golden(name):
    return Golden(name)


def rel_err(a, b):
    """max|a-b| / max|b|  -- the 'relative fp32' figure quoted in BASELINE.json."""
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
