"""umT5-XXL text encoder (24 layers, dim 4096, 64 heads, ffn 10240; random-init weights) on the HIP kernels: time per prompt.
Usage: python tools/t5_bench.py [n_tokens ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from univid_amd import _lib  # noqa: E402
from univid_amd.wan.t5 import umt5_xxl_encoder  # noqa: E402


def main():
    _lib.init()
    dev = "cuda"
    with torch.device(dev):
        m = umt5_xxl_encoder().to(torch.bfloat16)
    m = m.eval().requires_grad_(False)
    g = torch.Generator(device=dev).manual_seed(0)
    for p in m.parameters():
        if p.dim() >= 2:
            p.copy_((torch.randn(p.shape, device=dev, generator=g, dtype=torch.float32) * (0.5 / p.shape[-1] ** 0.5)).to(p.dtype))
    n_params = sum(p.numel() for p in m.parameters())
    print(f"umT5-XXL encoder: {n_params / 1e9:.2f} B parameters")
    for n in [int(a) for a in sys.argv[1:]] or [16, 77, 512]:
        ids = torch.randint(1, 256384, (n,), generator=torch.Generator().manual_seed(n))
        out = m.encode(ids)
        torch.cuda.synchronize()
        assert out.shape == (n, 4096) and torch.isfinite(out.float()).all()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            out = m.encode(ids)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        flops = 24 * (2 * n * 4096 * 4096 * 4 + 2 * n * 4096 * 10240 * 3 + 4 * n * n * 4096)
        print(f"n={n:4d}: {dt * 1e3:8.2f} ms per prompt, {flops / dt / 1e12:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
