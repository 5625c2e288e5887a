"""How far the fused QK-RMSNorm + RoPE kernel is from the oracle in bf16 ulps, over thousands of full-size rows (a tool behind the gate of
test_full_size_glue_kernels_sampled_rows_vs_oracle): the kernel and the oracle sum the row's squares in different fp32 orders, so the
intermediate bf16 rounding of the normalised value (WanRMSNorm's type_as, model.py:79-96) flips by one ulp on a few elements per
100 000, and the rotation turns such a flip into up to two ulps of the output.   python3 tests/manual/rope_ulp_stats.py [rows]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import wan_dit
from univid_amd import _lib
from univid_amd.wan.model import _freqs_device
_lib.init()
dev, BF16 = "cuda", torch.bfloat16
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
Ls, B, C, H = 11440, 2, 3072, 24
grid = (13, 22, 40)
M = B * Ls
g = torch.Generator(device=dev).manual_seed(17)
gc = torch.Generator().manual_seed(5)
rows = torch.unique(torch.randint(0, M, (n,), generator=gc))
q = (torch.randn(M, C, device=dev, generator=g) * 1.5).to(BF16)
k = (torch.randn(M, C, device=dev, generator=g) * 1.5).to(BF16)
wq, wk = torch.randn(C, device=dev, generator=g) * 0.1 + 1, torch.randn(C, device=dev, generator=g) * 0.1 + 1
q_in, k_in = q[rows].cpu(), k[rows].cpu()
freqs = wan_dit.rope_table(C // H)
_lib.rmsnorm_rope_qk(q, k, wq, wk, M, Ls, C, C // H, 1e-6, _freqs_device(freqs, torch.device(dev)), grid)
pos = (rows % Ls)
c = C // H // 2
fa, fb, fc = freqs.split([c - 2 * (c // 3), c // 3, c // 3], dim=1)
f_ = pos // (grid[1] * grid[2]); rem = pos % (grid[1] * grid[2]); h_ = rem // grid[2]; w_ = rem % grid[2]
fi = torch.cat([fa[f_], fb[h_], fc[w_]], dim=1).unsqueeze(1)               # [rows, 1, 64] complex128


def ulp(x):
    return torch.pow(2.0, torch.floor(torch.log2(x.abs().clamp_min(1e-30))) - 7)


for got, xin, wt, nm in ((q[rows], q_in, wq.cpu(), "q"), (k[rows], k_in, wk.cpu(), "k")):
    y = wan_dit.rms_norm(xin.unsqueeze(0), wt, 1e-6).view(len(rows), H, C // H)
    xi = torch.view_as_complex(y.to(torch.float64).reshape(len(rows), H, -1, 2))
    ref = torch.view_as_real(xi * fi).flatten(2).float().reshape(len(rows), C).to(BF16).float()
    gotf = got.float().cpu()
    d = (gotf - ref).abs()
    u = d / ulp(torch.maximum(ref.abs(), gotf.abs()))
    tot = d.numel()
    print(f"{nm}: {len(rows)} rows, {tot} elements: bit-identical {float((d == 0).float().mean()):.6f}; > 1 ulp: {int((u > 1.001).sum())} "
          f"({float((u > 1.001).float().mean()):.2e}); > 2 ulp: {int((u > 2.001).sum())}; max {float(u.max()):.2f} ulp", flush=True)
