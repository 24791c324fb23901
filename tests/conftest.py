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


class KinkMargin:
    """Smallest |pre-activation| reaching any nn.ReLU of `model` inside the `with` block.

    ReLU gradients are discontinuous where a pre-activation crosses zero; two correct fp32 implementations
    with different summation orders may disagree on the sign of a pre-activation that is within round-off of
    0 (observed: one such element flips ~1e-2 of a weight gradient).  Gradient-parity tests therefore use
    data whose pre-activations all stay >= ~1e-6 away from the kink (tests/golden/make_golden.py does the
    same for the fixtures and records the margin)."""

    def __init__(self, model):
        self.model, self.min, self._h = model, float("inf"), []

    def __enter__(self):
        def hook(_m, inp, _out):
            self.min = min(self.min, float(inp[0].detach().abs().min()))
        self._h = [m.register_forward_hook(hook) for m in self.model.modules() if isinstance(m, torch.nn.ReLU)]
        return self

    def __exit__(self, *a):
        for h in self._h:
            h.remove()


def pick_seed(build, first=0, tries=200, margin=1.5e-6):
    """build(seed) -> (model, run) where run() executes the oracle forward; returns the first seed whose
    forward keeps every ReLU input `margin` away from zero (else the best of `tries`)."""
    best = (first, -1.0)
    for seed in range(first, first + tries):
        model, run = build(seed)
        with KinkMargin(model) as km, torch.no_grad():
            run()
        if km.min > margin:
            return seed
        if km.min > best[1]:
            best = (seed, km.min)
    return best[0]
