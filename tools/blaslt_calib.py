"""Calibration only (developer tool): what the vendor library (hipBLASLt through torch.matmul / F.linear) reaches on the DiT's GEMM
shapes on this device, next to uv_gemm_bf16_nt with the plain bf16 epilogue. Random operands, interleaved rounds."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univid_amd import _lib
from univid_amd._lib import EPI_BF16
_lib.init()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
for M, N, K in ((22880, 3072, 3072), (22880, 14336, 3072), (22880, 3072, 14336), (8192, 8192, 8192)):
    A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    W = ((torch.rand(N, K, device=dev, generator=g) * 2 - 1) * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = {"hipblaslt": [], "uv": []}
    for r in range(5):
        for name in ("hipblaslt", "uv"):
            fn = (lambda: torch.nn.functional.linear(A, W)) if name == "hipblaslt" else (lambda: _lib.gemm_bf16(A, W, None, out, EPI_BF16))
            for _ in range(3):
                fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(8):
                fn()
            e.record(); torch.cuda.synchronize()
            res[name].append(s.elapsed_time(e) / 8)
    f = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: " + "  ".join(f"{k}: {statistics.median(v) * 1e3:.1f} us {f / statistics.median(v) / 1e9:.0f} TF/s" for k, v in res.items()))
