"""GEMM micro-benchmark on the DiT's shapes: every tile config, interleaved rounds in ONE process, random operands.

    python tools/gemm_bench.py [--cfgs 0,5,6,7] [--rounds 5] [--check]

Prints median/min TFLOP/s per (shape, config). `--check` compares each config with an fp64 reference on a row sample.
"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univid_amd import _lib  # noqa: E402
from univid_amd._lib import EPI_BF16, EPI_GELU_BF16, EPI_GATE_RESID_F32, EPI_BF16_T, EPI_F32_FROM_BF16, EPI_RESID_F32  # noqa: E402

L2 = 22880  # cond+uncond stacked tokens at 704x1280x49 (2 x 11440)
SHAPES = [
    ("q/k (bf16)", L2, 3072, 3072, EPI_BF16),
    ("v (bf16^T)", L2, 3072, 3072, EPI_BF16_T),
    ("o (gate+resid f32)", L2, 3072, 3072, EPI_GATE_RESID_F32),
    ("ffn.0 (gelu)", L2, 14336, 3072, EPI_GELU_BF16),
    ("ffn.2 (gate+resid f32)", L2, 3072, 14336, EPI_GATE_RESID_F32),
    ("single sample q", 11440, 3072, 3072, EPI_BF16),
    ("ctx k/v", 1024, 3072, 3072, EPI_BF16),
    ("ffn.0 without gelu", L2, 14336, 3072, EPI_BF16),
    ("tail strip K=3072", 1120, 3072, 3072, EPI_BF16),
    ("tail strip K=14336", 1120, 3072, 14336, EPI_GATE_RESID_F32),
    ("o-shape, f32 write only", L2, 3072, 3072, EPI_F32_FROM_BF16),
    ("o-shape, f32 resid", L2, 3072, 3072, EPI_RESID_F32),
    ("8192^3", 8192, 8192, 8192, EPI_BF16),
    ("aligned q (89 x 256 rows)", 22784, 3072, 3072, EPI_BF16),
    ("aligned ffn.0 shape", 22784, 14336, 3072, EPI_BF16),
    ("aligned ffn.2 shape", 22784, 3072, 14336, EPI_BF16),
    ("ranker q/k/v/o (64 x 256 tokens)", 16384, 768, 768, EPI_BF16),
    ("ranker fc1", 16384, 3072, 768, EPI_GELU_BF16),
    ("ranker fc2", 16384, 768, 3072, EPI_BF16),
    ("q+k as ONE N=6144 launch", L2, 6144, 3072, EPI_BF16),       # measured 652-654 us against 2 x 327.7 us (shape 0): no gain
    ("q+k+v-sized N=9216 launch", L2, 9216, 3072, EPI_BF16),      # 960 us against 3 x 327.7 (-2.4 %; the V third would need the transposed epilogue)
    ("aligned ffn.0 shape, GELU", 22784, 14336, 3072, EPI_GELU_BF16),   # against shape 14 (plain bf16 epilogue): the GELU epilogue's surcharge
    ("ranker q+k+v as ONE N=2304 launch", 16384, 2304, 768, EPI_BF16),  # round 6, against 3 x shape 16: 576 tiles (2.25 rounds) against 3 x 192 (3 x 0.75)
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfgs", default="0,1,2,5,6")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--shapes", default="")
    ap.add_argument("--ws", action="store_true", help="pass a split-K workspace (tile_cfg 19 / 20; tile_cfg 0's ffn.2 strip)")
    a = ap.parse_args()
    cfgs = [int(c) for c in a.cfgs.split(",")]
    _lib.init()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(1)
    sel = [int(s) for s in a.shapes.split(",")] if a.shapes else range(len(SHAPES))
    for si in sel:
        name, M, N, K, epi = SHAPES[si]
        A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        W = ((torch.rand(N, K, device=dev, generator=g) * 2 - 1) * 0.05).to(torch.bfloat16)
        bias = (torch.rand(N, device=dev, generator=g) - 0.5).to(torch.bfloat16)
        gate = gate_tid = None
        if epi in (EPI_F32_FROM_BF16, EPI_RESID_F32):
            out = torch.rand(M, N, device=dev, generator=g)
        elif epi == EPI_GATE_RESID_F32:
            out = torch.rand(M, N, device=dev, generator=g)
            gate = torch.rand(2, N, device=dev, generator=g)
            gate_tid = (torch.arange(M, device=dev) * 2 // M).to(torch.int32)
        elif epi == EPI_BF16_T:
            out = torch.zeros(N, (M + 63) // 64 * 64, device=dev, dtype=torch.bfloat16)
        else:
            out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        times = {c: [] for c in cfgs}
        # split-K configurations (19 / 20) and the automatic choice's split-K strip take a workspace (uv_gemm_bf16_nt_ws)
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        ws = torch.empty(max(4096 + tiles * 4 * 262144 if tiles <= 1024 else 0, _lib.gemm_splitk_ws_bytes(M, N, K), 256), dtype=torch.uint8, device=dev) if a.ws else None
        for r in range(a.rounds + 1):
            for c in cfgs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    _lib.gemm_bf16(A, W, bias, out, epi, gate=gate, gate_tid=gate_tid, tile_cfg=c, ws=ws)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    times[c].append(e0.elapsed_time(e1) / a.iters)
        fl = 2.0 * M * N * K
        line = f"{name:24s} M={M:6d} N={N:6d} K={K:6d}"
        for c in cfgs:
            med, mn = statistics.median(times[c]), min(times[c])
            line += f" | cfg{c}: {fl / med / 1e9:7.1f} (best {fl / mn / 1e9:7.1f}) TF/s {med * 1e3:7.1f} us"
        print(line, flush=True)
        if a.check:
            rows = torch.randint(0, M, (64,), device=dev, generator=g)
            rows[0], rows[1] = 0, M - 1
            ref = A[rows].double() @ W.double().t() + bias.double()
            for c in cfgs:
                if epi == EPI_GATE_RESID_F32:
                    base = torch.rand(M, N, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
                    o = base.clone()
                    _lib.gemm_bf16(A, W, bias, o, epi, gate=gate, gate_tid=gate_tid, tile_cfg=c)
                    want = base[rows].double() + ref.to(torch.bfloat16).double() * gate[gate_tid[rows].long()].double()
                    err = (o[rows].double() - want).abs().max().item()
                elif epi == EPI_BF16_T:
                    o = torch.zeros_like(out)
                    _lib.gemm_bf16(A, W, bias, o, epi, tile_cfg=c)
                    err = (o[:, rows].t().double() - ref).abs().max().item()
                else:
                    o = torch.zeros_like(out)
                    _lib.gemm_bf16(A, W, bias, o, epi, tile_cfg=c)
                    want = torch.nn.functional.gelu(ref.to(torch.bfloat16).float(), approximate="tanh").double() if epi == EPI_GELU_BF16 else ref
                    err = (o[rows].double() - want).abs().max().item()
                print(f"    check cfg{c}: max abs err {err:.3e} (ref absmax {ref.abs().max().item():.2f})", flush=True)


if __name__ == "__main__":
    main()
