"""Same-process A/B of the persistent GEMM's tile-walk group height (UV_GEMM_GM, developer knob): how many 256-row tiles tall the
column groups are that an XCD's workgroups walk (L2 footprint: GM activation panels + 32 / GM weight panels per XCD at a time)."""
import os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univid_amd import _lib
from univid_amd._lib import EPI_BF16, EPI_GELU_BF16
_lib.init()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
for name, M, N, K, epi in (("q", 22880, 3072, 3072, EPI_BF16), ("ffn.0", 22880, 14336, 3072, EPI_GELU_BF16), ("ffn.2-shape", 22880, 3072, 14336, EPI_BF16)):
    A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    W = ((torch.rand(N, K, device=dev, generator=g) * 2 - 1) * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = {gm: [] for gm in (2, 4, 8, 16, 32)}
    for r in range(5):
        for gm in res:
            _lib.set_option(_lib.OPT_GEMM_GM, int(gm))
            for _ in range(2):
                _lib.gemm_bf16(A, W, None, out, epi)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(6):
                _lib.gemm_bf16(A, W, None, out, epi)
            e.record(); torch.cuda.synchronize()
            res[gm].append(s.elapsed_time(e) / 6)
    f = 2.0 * M * N * K
    print(f"{name} {M}x{N}x{K}: " + "  ".join(f"GM={gm}: {statistics.median(v)*1e3:.0f} us {f/statistics.median(v)/1e9:.0f} TF/s" for gm, v in res.items()), flush=True)
