"""Same kernels, same shapes, different operand DATA (developer tool): how much of the attention / GEMM time is the chip's
power management (MI355X_MICROARCH.md 'DVFS give-back' item 1) rather than the instruction schedule. All variants run the same
instruction stream and move the same bytes; interleaved rounds in one process on one device."""
import math, os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univid_amd import _lib
from univid_amd._lib import EPI_BF16
_lib.init()
dev = "cuda"; BF16 = torch.bfloat16
H, D, L, B = 24, 128, 11440, 2
C = H * D
torch.manual_seed(0)
KINDS = {"randn": lambda *s: torch.randn(*s, device=dev), "zeros": lambda *s: torch.zeros(*s, device=dev),
         "ones": lambda *s: torch.ones(*s, device=dev), "+-1": lambda *s: torch.sign(torch.randn(*s, device=dev)),
         "randn*0.01": lambda *s: torch.randn(*s, device=dev) * 0.01}


def timed(fns, iters, rounds=4):
    res = {k: [] for k in fns}
    for rnd in range(rounds + 1):
        for k, fn in fns.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                fn()
            e.record(); torch.cuda.synchronize()
            if rnd:
                res[k].append(s.elapsed_time(e) / iters)
    return {k: statistics.median(v) for k, v in res.items()}


out = torch.empty(B * L, C, dtype=BF16, device=dev)
data = {k: (f(B * L, C).to(BF16), f(B * L, C).to(BF16), f(C, (B - 1) * L + (L + 63) // 64 * 64).to(BF16)) for k, f in KINDS.items()}
t = timed({k: (lambda q=q, kk=kk, vt=vt: _lib.flash_attn(q, kk, vt, out, L, L, H, D, 1 / math.sqrt(D), batch=B)) for k, (q, kk, vt) in data.items()}, 20)
fl = 4.0 * B * L * L * C
print(f"self-attention, batch {B}, L = {L} ({_lib.attn_kernel_name(L, L, D, B)})")
for k, v in t.items():
    print(f"  {k:12s} {v:7.4f} ms  {fl / v / 1e9:7.1f} TFLOP/s  x{v / t['zeros']:.3f} of zeros")
del data
for name, M, N, K in (("ffn.0 shape", 2 * L, 14336, 3072), ("q shape", 2 * L, 3072, 3072)):
    o = torch.empty(M, N, dtype=BF16, device=dev)
    bias = torch.zeros(N, dtype=BF16, device=dev)
    ops = {k: (f(M, K).to(BF16), f(N, K).to(BF16)) for k, f in KINDS.items()}
    t = timed({k: (lambda a=a, w=w: _lib.gemm_bf16(a, w, bias, o, EPI_BF16)) for k, (a, w) in ops.items()}, 6)
    print(f"GEMM {name} {M} x {N} x {K} (bf16 out)")
    for k, v in t.items():
        print(f"  {k:12s} {v * 1e3:7.1f} us  {2.0 * M * N * K / v / 1e9:7.1f} TFLOP/s  x{v / t['zeros']:.3f} of zeros")
    del ops, o
