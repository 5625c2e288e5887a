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
    # the CPU oracle: torch's intra-op pool at the 256 reported hardware threads of the GPU pool's hosts is ~80x slower than at 32
    # (measured: one VAE decoder chunk 56 s vs 0.7 s), the containers' usable cores being far fewer
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))


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


# ---- measured parity margins --------------------------------------------------------------------------------------------
# Composite GPU tests (whole forwards, trajectories) gate on statistics (fraction inside rtol 1e-3 / atol 1e-4, max error,
# rms against the no-rounding truth run). Every such test reports what it MEASURED through `record_margin`; at the end of a GPU
# session the numbers are written to gpurun_out/parity_margins.json (copied to profiles/rNN_parity_margins.json), so that the
# thresholds in tests/test_gpu_parity.py can be ratcheted to measured x 1.5 instead of being guessed.
_MARGINS = {}


def record_margin(name, **values):
    _MARGINS[name] = {k: (float(v) if isinstance(v, (int, float)) or hasattr(v, "item") else v) for k, v in values.items()}


def pytest_sessionfinish(session, exitstatus):
    if not _MARGINS:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "parity_margins.json")
    old = {}
    if os.path.exists(path):
        try:
            old = json.load(open(path))
        except Exception:
            old = {}
    old.update(_MARGINS)
    with open(path, "w") as f:
        json.dump(old, f, indent=1, sort_keys=True)
