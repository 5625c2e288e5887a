"""Race screen for the ping-pong GEMM schedules: the persistent kernel (tile_cfg 0 / 18), the one-tile-per-workgroup schedule
(tile_cfg 7 / 8) against the 4-phase reference schedule (tile_cfg 14) and the 16-wave kernel (tile_cfg 5), FULL output, bit for bit (all three accumulate in the same K order),
repeated while the chip is busy. Any LDS hazard (a fragment read before its DMA landed, a half-tile restaged too early)
shows up as a mismatching tile."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univid_amd import _lib
from univid_amd._lib import EPI_BF16, EPI_BF16_T
_lib.init()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(3)
bad = 0
for (M, N, K) in [(22880, 3072, 3072), (22880, 14336, 3072), (22880, 3072, 14336), (11440, 3072, 3072), (4000, 1024, 256), (2300, 2048, 640)]:
    A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    W = ((torch.rand(N, K, device=dev, generator=g) * 2 - 1) * 0.05).to(torch.bfloat16)
    ref = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    _lib.gemm_bf16(A, W, None, ref, EPI_BF16, tile_cfg=5)
    for rep in range(int(os.environ.get("REPS", 6))):
        for cfg in (7, 14, 8, 0) + ((18,) if M >= 8192 else ()):
            out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
            _lib.gemm_bf16(A, W, None, out, EPI_BF16, tile_cfg=cfg)
            if not torch.equal(out, ref):
                d = (out.float() - ref.float()).abs()
                bad += 1
                print(f"MISMATCH M={M} N={N} K={K} cfg={cfg} rep={rep}: {int((d > 0).sum())} elements, max {float(d.max()):.3e}", flush=True)
    print(f"M={M} N={N} K={K}: done", flush=True)
print("race screen:", "FAILED" if bad else "clean")
sys.exit(1 if bad else 0)
