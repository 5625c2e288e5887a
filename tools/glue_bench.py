"""Micro-benchmark of the HBM-bound DiT glue kernels at the bench shapes (GB/s = algorithmic bytes / time)."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univid_amd import _lib
from univid_amd.wan.model import rope_params, _freqs_device

_lib.init()
dev = "cuda"
C, D, L, B = 3072, 128, 11440, 2
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B * L, C, device=dev, generator=g)
h = torch.empty(B * L, C, device=dev, dtype=torch.bfloat16)
tab = torch.randn(2, 6 * C, device=dev, generator=g)
tid = (torch.arange(B * L, device=dev) >= L).to(torch.int32)
w, b = torch.randn(C, device=dev, generator=g), torch.randn(C, device=dev, generator=g)
q = torch.randn(B * L, C, device=dev, generator=g).to(torch.bfloat16)
d_ = D
freqs = torch.cat([rope_params(1024, d_ - 4 * (d_ // 6)), rope_params(1024, 2 * (d_ // 6)), rope_params(1024, 2 * (d_ // 6))], dim=1)
fr = _freqs_device(freqs, torch.device(dev))


def timeit(fn, nbytes, name, iters=20, rounds=5):
    ts = []
    for r in range(rounds + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1) / iters)
    t = statistics.median(ts)
    print(f"{name:44s} {t * 1e3:8.1f} us  {nbytes / t / 1e6:8.1f} GB/s", flush=True)


M = B * L
timeit(lambda: _lib.layernorm_mod(x, h, M, C, 1e-6, mode=1, tab=tab, shift_off=0, scale_off=C, tid=tid), M * C * 6, "layernorm_mod AdaLN (f32 -> bf16), 22880 rows")
timeit(lambda: _lib.layernorm_mod(x, h, M, C, 1e-6, mode=2, w=w, b=b), M * C * 6, "layernorm_mod affine (norm3), 22880 rows")
timeit(lambda: _lib.rmsnorm_rope(q[:L], q[:L], w, L, C, D, 1e-6, fr, (13, 22, 40)), L * C * 4, "rmsnorm_rope with RoPE, 11440 rows")
timeit(lambda: _lib.rmsnorm_rope(q, q, w, M, C, D, 1e-6), M * C * 4, "rmsnorm_rope no RoPE (cross q), 22880 rows")
k = torch.randn(B * L, C, device=dev, generator=g).to(torch.bfloat16)
timeit(lambda: _lib.rmsnorm_rope_qk(q, k, w, b, M, L, C, D, 1e-6, fr, (13, 22, 40)), 2 * M * C * 4, "rmsnorm_rope_qk (q+k, 2 samples, RoPE), 4 x 11440 rows")
