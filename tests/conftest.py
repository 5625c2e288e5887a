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
    """npz -> dict of torch tensors (uint16 arrays are bf16 bit patterns)."""
    out = {}
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        for k in z.files:
            a = z[k]
            if a.dtype == np.uint16:
                out[k] = torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)
            elif a.ndim == 0:
                out[k] = a.item()
            else:
                out[k] = torch.from_numpy(a.copy())
    return out


@pytest.fixture(scope="session")
def golden():
    return load_golden


def bf16_ulp(x):
    """Spacing of bf16 at |x| (8 significant bits)."""
    x = x.abs().float().clamp_min(2.0 ** -126)
    return torch.exp2(torch.floor(torch.log2(x)) - 7)
