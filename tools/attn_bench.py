"""Attention-only micro benchmark (developer tool): full-size self-attention launches for profiling."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univid_amd import _lib
_lib.init()
dev = "cuda"; BF16 = torch.bfloat16
L, H, D = int(os.environ.get("L", 11440)), 24, 128
B = int(os.environ.get("B", 1))     # stacked samples (the CFG pair is B=2)
C = H * D
torch.manual_seed(0)
Lk = int(os.environ.get("LK", L))
q = torch.randn(B * L, C, device=dev).to(BF16); k = torch.randn(B * Lk, C, device=dev).to(BF16)
vt = torch.randn(C, (B - 1) * Lk + (Lk + 63) // 64 * 64, device=dev).to(BF16)
out = torch.empty(B * L, C, dtype=BF16, device=dev)
n = int(os.environ.get("N", 5))
for _ in range(n):
    _lib.flash_attn(q, k, vt, out, L, Lk, H, D, 1 / math.sqrt(D), batch=B)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(n):
    _lib.flash_attn(q, k, vt, out, L, Lk, H, D, 1 / math.sqrt(D), batch=B)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / n
import hashlib
print(f"attention B{B} Lq{L} Lk{Lk}: {ms:.3f} ms {4*B*L*Lk*C/ms/1e9:.1f} TFLOP/s  kernel={_lib.attn_kernel_name(L, Lk, D, B)}  out sha={hashlib.sha1(out.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]}")

