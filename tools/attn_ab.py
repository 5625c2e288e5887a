"""Same-process A/B of the two long-sequence self-attention kernels (developer tool): flash_attn_fwd12_kernel vs flash_attn_pw4_kernel.
Checks that the outputs are BIT-IDENTICAL (same per-query arithmetic by construction) on several shapes, then times interleaved rounds
at the bench shape on random data. env: L, B, ROUNDS, N."""
import hashlib, math, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univid_amd import _lib
_lib.init()
dev = "cuda"; BF16 = torch.bfloat16
H, D = 24, 128
C = H * D


import ctypes
_diag_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "diag", "libuv_diag.so")
if not os.path.exists(_diag_path):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "diag"))
    import build_diag
    build_diag.build()
_diag = ctypes.CDLL(_diag_path)
_P, _L, _I = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
_diag.uv_diag_flash_attn_pw4.argtypes = [_P, _L, _P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _I, ctypes.c_float, _P]


def run(kind, q, k, vt, out, L, Lk, B):
    """fwd12 = the product's kernel through the C ABI; pw4 = the diagnostic kernel of tools/diag/libuv_diag.so"""
    if kind == "pw4":
        rc = _diag.uv_diag_flash_attn_pw4(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), vt.data_ptr(), vt.stride(0), out.data_ptr(),
                                          out.stride(0), B, L, Lk, H, D, 1 / math.sqrt(D), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    else:
        _lib.flash_attn(q, k, vt, out, L, Lk, H, D, 1 / math.sqrt(D), batch=B)


def make(L, Lk, B, seed, scale=1.0, spike=False):
    g = torch.Generator(device=dev).manual_seed(seed)
    q = (torch.randn(B * L, C, device=dev, generator=g) * scale).to(BF16)
    k = (torch.randn(B * Lk, C, device=dev, generator=g) * scale).to(BF16)
    if spike:   # forces the deferred-maximum rescale late in the key sequence for some query rows
        k[Lk // 2 + 37] *= 6.0
        k[Lk - 5] *= 9.0
    vt = torch.randn(C, (B - 1) * Lk + (Lk + 63) // 64 * 64, device=dev, generator=g).to(BF16)
    return q, k, vt


ok = True
for (L, Lk, B, scale, spike) in ((2048, 2048, 1, 1.0, False), (2100, 2100, 1, 1.0, False), (2304, 4000, 2, 1.0, False), (3000, 2999 // 8 * 8, 2, 2.0, True),
                                 (11440, 11440, 2, 1.0, False), (11440, 11440, 1, 3.0, True)):
    q, k, vt = make(L, Lk, B, 1 + L, scale, spike)
    o1 = torch.empty(B * L, C, dtype=BF16, device=dev); o2 = torch.full_like(o1, float("nan"))
    run("fwd12", q, k, vt, o1, L, Lk, B)
    run("pw4", q, k, vt, o2, L, Lk, B)
    torch.cuda.synchronize()
    same = torch.equal(o1.view(torch.int16), o2.view(torch.int16))
    nbad = (o1.view(torch.int16) != o2.view(torch.int16)).sum().item()
    d = (o1.float() - o2.float()).abs().max().item()
    fin = torch.isfinite(o2.float()).all().item()
    print(f"L={L} Lk={Lk} B={B} scale={scale} spike={spike}: bit-identical={same} mismatching elements={nbad} max|diff|={d:.3e} finite={fin}", flush=True)
    ok &= same
    if not same:
        bad = (o1.view(torch.int16) != o2.view(torch.int16)).nonzero()
        rows = torch.unique(bad[:, 0]); cols = torch.unique(bad[:, 1])
        print("   first bad rows", rows[:16].tolist(), "n rows", rows.numel(), " first bad cols", cols[:16].tolist(), "n cols", cols.numel())
        r256 = torch.bincount((rows % L) % 256, minlength=256).view(8, 32).sum(1).tolist()
        print("   bad rows by 32-row block inside a 256-row workgroup (wave w block X = 2w+X):", r256)
        c128 = torch.bincount(cols % 128, minlength=128).view(4, 32).sum(1).tolist()
        print("   bad cols by 32-wide d tile:", c128, " NaNs:", torch.isnan(o2.float()).sum().item())
        rel = ((o1.float() - o2.float()).abs() / (o1.float().abs() + 1e-3))
        print("   median |rel diff| over mismatches:", rel[o1.view(torch.int16) != o2.view(torch.int16)].median().item())

L = int(os.environ.get("L", 11440)); B = int(os.environ.get("B", 2))
q, k, vt = make(L, L, B, 7)
out = torch.empty(B * L, C, dtype=BF16, device=dev)
n = int(os.environ.get("N", 10)); rounds = int(os.environ.get("ROUNDS", 5))
res = {"fwd12": [], "pw4": []}
for r in range(rounds):
    for kind in res:
        for _ in range(3):
            run(kind, q, k, vt, out, L, L, B)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            run(kind, q, k, vt, out, L, L, B)
        e.record(); torch.cuda.synchronize()
        res[kind].append(s.elapsed_time(e) / n)
fl = 4.0 * B * L * L * C
for kind, v in res.items():
    print(f"{kind}: median {statistics.median(v):.3f} ms  min {min(v):.3f} ms  {fl / statistics.median(v) / 1e9:.1f} TFLOP/s  rounds {[round(x, 3) for x in v]}")
print("ALL BIT-IDENTICAL" if ok else "MISMATCH")
