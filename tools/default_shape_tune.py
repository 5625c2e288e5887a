"""Re-tuning sweep at UniVid's default workload (121 frames 704 x 1280: L = 27 280 tokens, CFG pair stacked = 54 560 rows), same process,
interleaved rounds, random data (developer tool): (1) the 12- / 8-unit block cut of flash_attn_fwd12_kernel (uv_set_option
UV_OPT_ATTN_CUT: n8 eight-unit blocks per head; 0 = the list-scheduling model's own choice), (2) the tile-walk group height of the
persistent GEMM (UV_OPT_GEMM_GM) on the three DiT GEMM shapes at 54 560 rows. Results do not depend on either switch (tile / block order only).
    python3 tools/default_shape_tune.py [L] [rounds]"""
import math, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univid_amd import _lib
from univid_amd._lib import EPI_BF16, EPI_GELU_BF16
_lib.init()
dev = "cuda"
L = int(sys.argv[1]) if len(sys.argv) > 1 else 27280
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
H, D, B = 24, 128, 2
C = H * D
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(B * L, C, device=dev, generator=g).to(torch.bfloat16)
k = torch.randn(B * L, C, device=dev, generator=g).to(torch.bfloat16)
vt = torch.randn(C, (B - 1) * L + (L + 63) // 64 * 64, device=dev, generator=g).to(torch.bfloat16)
out = torch.empty(B * L, C, device=dev, dtype=torch.bfloat16)
nwu = (L + 31) // 32


def timed(fn, n):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


cuts = [0] + [n8 + 1 for n8 in range(0, min(nwu // 8, 40) + 1)]
res = {c: [] for c in cuts}
ref = None
for r in range(rounds):
    for c in cuts:
        _lib.set_option(_lib.OPT_ATTN_CUT, c)
        res[c].append(timed(lambda: _lib.flash_attn(q, k, vt, out, L, L, H, D, 1 / math.sqrt(D), batch=B), 3))
        if r == 0:
            h = out.view(torch.int16).sum(dtype=torch.int64).item()
            ref = h if ref is None else ref
            assert h == ref, "the cut changed the result"
_lib.set_option(_lib.OPT_ATTN_CUT, 0)
fl = B * 4.0 * L * L * C
print(f"self-attention L={L} B={B}: nwu={nwu} units per head")
for c in cuts:
    m = statistics.median(res[c])
    n8 = c - 1
    n12 = 0 if c == 0 else max(0, -(-(nwu - 8 * n8) // 12))
    tag = "model's choice" if c == 0 else f"{n12} x 12 + {n8} x 8"
    print(f"  cut {tag:>18}: {m:8.3f} ms  {fl / m / 1e9:7.1f} TF/s", flush=True)

M = B * L
for name, N, K, epi in (("q", 3072, 3072, EPI_BF16), ("ffn.0", 14336, 3072, EPI_GELU_BF16), ("ffn.2-shape", 3072, 14336, EPI_BF16)):
    A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    W = ((torch.rand(N, K, device=dev, generator=g) * 2 - 1) * 0.05).to(torch.bfloat16)
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    gms = (0, 2, 4, 8, 16)
    rs = {gm: [] for gm in gms}
    for r in range(rounds):
        for gm in gms:
            _lib.set_option(_lib.OPT_GEMM_GM, gm)
            rs[gm].append(timed(lambda: _lib.gemm_bf16(A, W, None, o, epi), 4))
    _lib.set_option(_lib.OPT_GEMM_GM, 0)
    f = 2.0 * M * N * K
    print(f"{name} {M}x{N}x{K}: " + "  ".join(f"GM={'auto' if gm == 0 else gm}: {statistics.median(v) * 1e3:.0f} us {f / statistics.median(v) / 1e9:.0f} TF/s" for gm, v in rs.items()), flush=True)
    del A, W, o
