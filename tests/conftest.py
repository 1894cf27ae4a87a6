import json
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


def load_golden(name):
    """tests/golden/<name>.npz -> (dict of torch tensors, meta dict)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = json.loads(bytes(z["meta"]).decode())
    out = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
    return out, meta


def params_of(g, prefix="param."):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


def golden_names(kind):
    names = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz"))
    return [n for n in names if n.startswith(kind)]


def rel_err(a, b):
    """||a-b|| / ||b|| in float64 (0 when both are all-zero)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    den = b.norm().item()
    num = (a - b).norm().item()
    if den == 0.0:
        return num
    return num / den


@pytest.fixture(scope="session")
def cuda0():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
