"""Diagnostic (round 4): does the attention kernel's output store write more bytes than it should? Runs the batch-2 self-attention launch
twice - output row stride 3072 (the LDS-transposed 16-byte store path) and 3076 (the direct 8-byte store path, taken when ldo % 8 != 0) -
for `rocprofv3 --pmc WRITE_SIZE` / FETCH_SIZE passes; the two dispatches differ in Grid-independent order: first 3072, then 3076."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from univid_amd import _lib
_lib.init()
dev = "cuda"; BF16 = torch.bfloat16
L, H, D, B = int(os.environ.get("L", 11440)), 24, 128, 2
C = H * D
torch.manual_seed(0)
q = torch.randn(B * L, C, device=dev).to(BF16); k = torch.randn(B * L, C, device=dev).to(BF16)
vt = torch.randn(C, (B - 1) * L + (L + 63) // 64 * 64, device=dev).to(BF16)
for ld in (3072, 3076, 3072, 3076):
    buf = torch.zeros(B * L, ld, dtype=BF16, device=dev)
    out = buf[:, :C]
    torch.cuda.synchronize()
    _lib.flash_attn(q, k, vt, out, L, L, H, D, 1 / math.sqrt(D), batch=B)
    torch.cuda.synchronize()
    print(ld, float(out.float().abs().mean()))
