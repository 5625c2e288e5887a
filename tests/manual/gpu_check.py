"""Developer diagnostic (not a test): runs every kernel once on the GPU, prints error statistics against
CPU/torch computations and a few timings. Used through `gpurun` while bringing kernels up."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from univid_amd import _lib  # noqa: E402
from univid_amd._lib import *  # noqa: E402,F401,F403

dev = "cuda"
BF16 = torch.bfloat16
torch.manual_seed(0)


def stats(name, got, ref, tol=None):
    got, ref = got.float().cpu(), ref.float().cpu()
    d = (got - ref).abs()
    rel = d / (ref.abs() + 1e-6)
    bad = int((d > 1e-4 + 1e-3 * ref.abs()).sum())
    print(f"  {name:46s} max_abs={d.max().item():.3e} mean_abs={d.mean().item():.3e} ref_max={ref.abs().max().item():.3e} "
          f"viol(1e-3/1e-4)={bad}/{d.numel()} nan={int(torch.isnan(got).sum())}", flush=True)


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def check_gemm():
    print("== gemm_bf16")
    for (M, N, K) in [(300, 512, 256), (1000, 768, 192), (257, 3072, 3072)]:
        a = (torch.randn(M, K, device=dev) * 0.5).to(BF16)
        w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
        bias = (torch.randn(N, device=dev) * 0.1).to(BF16)
        acc = a.float() @ w.float().t() + bias.float()
        yb = acc.to(BF16)
        for cfg in (1, 2, 3):
            out = torch.zeros(M, N, dtype=BF16, device=dev)
            _lib.gemm_bf16(a, w, bias, out, EPI_BF16, tile_cfg=cfg)
            stats(f"M{M} N{N} K{K} cfg{cfg} EPI_BF16", out, yb)
        out = torch.zeros(M, N, dtype=BF16, device=dev)
        _lib.gemm_bf16(a, w, bias, out, EPI_GELU_BF16)
        stats("EPI_GELU_BF16", out, torch.nn.functional.gelu(yb, approximate="tanh"))
        out = torch.zeros(M, N, dtype=torch.float32, device=dev)
        _lib.gemm_bf16(a, w, bias, out, EPI_F32_FROM_BF16)
        stats("EPI_F32_FROM_BF16", out, yb.float())
        x0 = torch.randn(M, N, device=dev)
        x = x0.clone()
        _lib.gemm_bf16(a, w, bias, x, EPI_RESID_F32)
        stats("EPI_RESID_F32", x, x0 + yb.float())
        gate = torch.randn(3, N, device=dev)
        tid = torch.randint(0, 3, (M,), device=dev, dtype=torch.int32)
        x = x0.clone()
        _lib.gemm_bf16(a, w, bias, x, EPI_GATE_RESID_F32, gate=gate, gate_tid=tid)
        stats("EPI_GATE_RESID_F32", x, x0 + yb.float() * gate[tid.long()])
        Mp = (M + 63) // 64 * 64
        outT = torch.zeros(N, Mp, dtype=BF16, device=dev)
        _lib.gemm_bf16(a, w, bias, outT, EPI_BF16_T)
        stats("EPI_BF16_T", outT[:, :M], yb.t())
        print("   pad region zero:", bool((outT[:, M:] == 0).all()))


def check_gemm_f32():
    print("== gemm_f32")
    for (M, N, K) in [(300, 192, 256), (1000, 48, 48), (513, 192, 3072)]:
        a = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.05
        b = torch.randn(N, device=dev)
        r = torch.randn(M, N, device=dev)
        out = torch.zeros(M, N, device=dev)
        _lib.gemm_f32(a, w, b, out, resid=r)
        ref = (a.double() @ w.double().t() + b.double() + r.double()).float()
        stats(f"M{M} N{N} K{K}", out, ref)


def check_attn():
    print("== flash_attn")
    for (Lq, Lk, H, D) in [(300, 300, 2, 128), (256, 512, 3, 128), (1000, 77, 2, 128), (260, 260, 4, 64), (1200, 1200, 2, 128)]:
        C = H * D
        q = torch.randn(Lq, C, device=dev).to(BF16)
        k = torch.randn(Lk, C, device=dev).to(BF16)
        v = torch.randn(Lk, C, device=dev).to(BF16)
        Lkp = (Lk + 63) // 64 * 64
        vt = torch.zeros(C, Lkp, dtype=BF16, device=dev)
        vt[:, :Lk] = v.t()
        out = torch.zeros(Lq, C, dtype=BF16, device=dev)
        _lib.flash_attn(q, k, vt, out, Lq, Lk, H, D, 1.0 / math.sqrt(D))
        qf, kf, vf = (t.float().view(-1, H, D).transpose(0, 1) for t in (q, k, v))
        s = (qf @ kf.transpose(1, 2)) / math.sqrt(D)
        ref = (torch.softmax(s, -1) @ vf).transpose(0, 1).reshape(Lq, C)
        stats(f"Lq{Lq} Lk{Lk} H{H} D{D}", out, ref.to(BF16))
    # spike test: one key dominates late (forces the online-softmax rescale)
    Lq, Lk, H, D = 256, 640, 1, 128
    q = torch.randn(Lq, D, device=dev).to(BF16)
    k = torch.randn(Lk, D, device=dev).to(BF16)
    k[500] = q[3] * 4
    v = torch.randn(Lk, D, device=dev).to(BF16)
    vt = v.t().contiguous()
    out = torch.zeros(Lq, D, dtype=BF16, device=dev)
    _lib.flash_attn(q, k, vt, out, Lq, Lk, H, D, 1.0 / math.sqrt(D))
    s = (q.float() @ k.float().t()) / math.sqrt(D)
    stats("spike (rescale path)", out, (torch.softmax(s, -1) @ v.float()).to(BF16))


def check_glue():
    print("== glue")
    from oracle import wan_dit
    L, C, D = 70, 3072, 128
    x = torch.randn(L, C) * 2 + 0.3
    tab = torch.randn(2, 6 * C) * 0.3
    tid = torch.randint(0, 2, (L,), dtype=torch.int32)
    xd, tabd, tidd = x.to(dev), tab.to(dev), tid.to(dev)
    for rnd in (False, True):
        out = torch.zeros(L, C, dtype=BF16, device=dev)
        _lib.layernorm_mod(xd, out, L, C, 1e-6, mode=1, tab=tabd, shift_off=3 * C, scale_off=4 * C, tid=tidd, round_ln=rnd)
        xin = x.to(BF16) if rnd else x
        ln = wan_dit.layer_norm(xin, 1e-6).float()
        ref = (ln * (1 + tab[tid.long(), 4 * C:5 * C]) + tab[tid.long(), 3 * C:4 * C]).to(BF16)
        # NOTE round_ln rounds the LN output; the oracle rounds the INPUT too when x is bf16 (block 0) - feed same x
        if rnd:
            out2 = torch.zeros(L, C, dtype=BF16, device=dev)
            _lib.layernorm_mod(xin.float().to(dev), out2, L, C, 1e-6, mode=1, tab=tabd, shift_off=3 * C, scale_off=4 * C, tid=tidd, round_ln=True)
            stats("layernorm mode1 round_ln (bf16 x)", out2, ref)
        else:
            stats("layernorm mode1", out, ref)
    w, b = torch.randn(C) * 0.1 + 1, torch.randn(C) * 0.1
    out = torch.zeros(L, C, dtype=BF16, device=dev)
    _lib.layernorm_mod(xd, out, L, C, 1e-6, mode=2, w=w.to(dev), b=b.to(dev))
    stats("layernorm mode2 (affine)", out, wan_dit.layer_norm(x, 1e-6, w, b).to(BF16))
    out = torch.zeros(L, C, dtype=torch.float32, device=dev)
    _lib.layernorm_mod(xd, out, L, C, 1e-6, mode=0)
    stats("layernorm mode0 f32", out, wan_dit.layer_norm(x, 1e-6))
    # rmsnorm + rope
    for (dim, heads, grid) in [(3072, 24, (2, 5, 7)), (256, 4, (2, 5, 7))]:
        L = 70
        xq = (torch.randn(1, L, dim) * 1.5).to(BF16)
        wq = torch.randn(dim) * 0.1 + 1
        freqs = wan_dit.rope_table(dim // heads)
        ref = wan_dit.rope_apply(wan_dit.rms_norm(xq, wq, 1e-6).view(1, L, heads, dim // heads), torch.tensor([grid]), freqs)
        fr = torch.view_as_real(freqs).contiguous().to(dev)
        out = torch.zeros(L, dim, dtype=BF16, device=dev)
        _lib.rmsnorm_rope(xq[0].to(dev), out, wq.to(dev), L, dim, dim // heads, 1e-6, fr, grid)
        stats(f"rmsnorm_rope dim{dim}", out, ref.view(L, dim).to(BF16))
        eq = (out.cpu() == ref.view(L, dim).to(BF16)).float().mean().item()
        print(f"   exact-match fraction {eq:.6f}")
        out = torch.zeros(L, dim, dtype=BF16, device=dev)
        _lib.rmsnorm_rope(xq[0].to(dev), out, wq.to(dev), L, dim, dim // heads, 1e-6)
        stats(f"rmsnorm (no rope) dim{dim}", out, wan_dit.rms_norm(xq, wq, 1e-6)[0].to(BF16))


def check_dit():
    print("== tiny DiT forward vs oracle")
    from oracle import wan_dit
    from univid_amd.wan.model import WanModel
    for cfg, shape in [(wan_dit.TINY_CFG, (48, 4, 16, 16)),
                       (dict(wan_dit.TINY_CFG, num_heads=2, dim=256), (48, 3, 10, 12))]:
        sd = wan_dit.make_state_dict(cfg, 0)
        m = WanModel(model_type="ti2v", **{k: cfg[k] for k in ("patch_size", "text_len", "in_dim", "dim", "ffn_dim", "freq_dim", "text_dim", "out_dim", "num_heads", "num_layers", "eps")})
        m.load_state_dict(sd)
        m = m.to(dev).eval()
        g = torch.Generator().manual_seed(42)
        x = torch.randn(*shape, generator=g)
        ctx = [torch.randn(20, cfg["text_dim"], generator=g)]
        L = shape[1] * (shape[2] // 2) * (shape[3] // 2)
        for tmode in ("scalar", "two"):
            t = torch.full((1, L), 937.0)
            if tmode == "two":
                t[0, :(shape[2] // 2) * (shape[3] // 2)] = 0.0
            with torch.no_grad():
                ref, hid, _ = wan_dit.dit_forward(sd, cfg, [x], t, ctx, L, return_hidden=True)
                out = m([x.to(dev)], t.to(dev), [c.to(dev) for c in ctx], L)[0]
            stats(f"dit heads{cfg['num_heads']} t={tmode}", out, ref[0])


def check_vae():
    print("== VAE vs oracle")
    from oracle import wan_vae
    from univid_amd.wan.vae2_2 import WanVAE_
    cfg = wan_vae.SMALL_CFG
    sd = wan_vae.make_state_dict(cfg, 1)
    m = WanVAE_(dim=cfg["dim"], dec_dim=cfg["dec_dim"], z_dim=cfg["z_dim"], dim_mult=cfg["dim_mult"],
                temperal_downsample=cfg["temperal_downsample"])
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    ora = wan_vae.WanVAE(sd, cfg)
    scale = wan_vae.scale_tensors()
    scale_d = [s.to(dev) for s in scale]
    g = torch.Generator().manual_seed(3)
    for shape in [(3, 9, 32, 48), (3, 5, 64, 32), (3, 1, 32, 32)]:
        vid = torch.tanh(torch.randn(*shape, generator=g))
        with torch.no_grad():
            ref = ora.encode(vid.unsqueeze(0), scale)
            got = m.encode(vid.unsqueeze(0).to(dev), scale_d)
        stats(f"encode {shape}", got, ref)
    for shape in [(48, 3, 2, 3), (48, 2, 4, 2), (48, 1, 2, 2)]:
        z = torch.randn(1, *shape, generator=g)
        with torch.no_grad():
            ref = ora.decode(z, scale).clamp(-1, 1)
            got = m.decode(z.to(dev), scale_d)
        stats(f"decode {shape}", got, ref)
    # per-op: one conv against F.conv3d
    import torch.nn.functional as F
    x = torch.randn(1, 64, 5, 6, 7)
    w = torch.randn(96, 64, 3, 3, 3) * 0.05
    b = torch.randn(96)
    ref = wan_vae.causal_conv3d(x, w, b, pad=(1, 1, 1))
    from univid_amd.wan.vae2_2 import _Engine, _ConvOp
    import torch.nn as nn
    conv = nn.Conv3d(64, 96, 3)
    conv.weight.data.copy_(w); conv.bias.data.copy_(b)
    conv = conv.to(dev)
    op = _ConvOp(conv)
    ring = op.ring(6, 7, 5)
    ring[2:7].copy_(x[0].permute(1, 2, 3, 0).to(dev))
    out = torch.empty(5, 6, 7, 96, device=dev)
    _lib.call("uv_conv3d_f32", _lib.ptr(ring), 64, 7, 6, 7, _lib.ptr(op.w), _lib.ptr(op.b), _lib.ptr(out), 96, 5, 6, 7, 64, 96, 3, 3, 3,
              1, 1, 1, 0, 1, 1, 0, 0, None, 0, _lib.stream_ptr())
    stats("conv3d 3x3x3 64->96", out.permute(3, 0, 1, 2), ref[0])


def perf_vae():
    from univid_amd.wan.vae2_2 import Wan2_2_VAE
    from oracle import wan_vae
    g = torch.Generator().manual_seed(3)
    cfg = wan_vae.SMALL_CFG
    ora = wan_vae.WanVAE(wan_vae.make_state_dict(cfg, 1), cfg)
    v3 = Wan2_2_VAE(c_dim=32, dec_dim=32, device=dev, seed=1, precision="bf16x3")
    for shape in [(48, 3, 2, 3), (48, 2, 4, 2)]:
        z = torch.randn(*shape, generator=g)
        with torch.no_grad():
            ref = ora.decode(z.unsqueeze(0), wan_vae.scale_tensors()).clamp(-1, 1)[0]
            got = v3.decode([z.to(dev)])[0]
        stats(f"bf16x3 decode {shape}", got, ref)
    vid = torch.tanh(torch.randn(3, 9, 32, 48, generator=g))
    with torch.no_grad():
        stats("bf16x3 encode", v3.encode([vid.to(dev)])[0], ora.encode(vid.unsqueeze(0), wan_vae.scale_tensors())[0])
    prec = os.environ.get("VAE_PREC", "bf16x3")
    print("== VAE perf (full width)", prec)
    vae = Wan2_2_VAE(device=dev, precision=prec)
    for shape in [(48, 2, 16, 16), (48, 3, 45, 80)]:
        z = torch.randn(*shape, device=dev)
        torch.cuda.synchronize(); t0 = time.time()
        with torch.no_grad():
            v = vae.decode([z])[0]
        torch.cuda.synchronize(); dt = time.time() - t0
        print(f"  decode {shape} -> {tuple(v.shape)} in {dt:.3f} s, finite={bool(torch.isfinite(v).all())}", flush=True)
    _lib.PROFILE, _lib.PROFILE_ALL = {}, True
    z = torch.randn(48, 3, 45, 80, device=dev)
    torch.cuda.synchronize(); t0 = time.time()
    with torch.no_grad():
        v = vae.decode([z])[0]
    torch.cuda.synchronize(); dt = time.time() - t0
    prof = _lib.PROFILE; _lib.PROFILE, _lib.PROFILE_ALL = None, False
    print(f"  decode again: {dt:.3f} s for {v.shape[1]} frames 720x1280 -> {v.numel()*4/dt/1e9:.3f} GB/s")
    for name, evs in prof.items():
        ms = [s.elapsed_time(e) for s, e, _ in evs]
        fl = sum(f for _, _, f in evs)
        print(f"    {name:24s} launches={len(ms):5d} total={sum(ms):9.2f} ms  {fl/ (sum(ms)*1e-3)/1e12 if fl else 0:.1f} TFLOP/s")


def perf():
    print("== perf (full-size shapes)")
    L, C, F = int(os.environ.get('GEMM_M', 11440)), 3072, 14336
    a = (torch.randn(L, C, device=dev) * 0.5).to(BF16)
    for (N, K, name) in [(C, C, "qkvo"), (F, C, "ffn0"), (C, F, "ffn2")]:
        w = (torch.randn(N, K, device=dev) * 0.02).to(BF16)
        aa = a if K == C else (torch.randn(L, K, device=dev) * 0.5).to(BF16)
        out = torch.empty(L, N, dtype=BF16, device=dev)
        for cfg in (0, 4, 5, 6):
            ms = timeit(lambda: _lib.gemm_bf16(aa, w, None, out, EPI_BF16, tile_cfg=cfg))
            print(f"  gemm {name} M{L} N{N} K{K} cfg{cfg}: {ms:.3f} ms  {2 * L * N * K / ms / 1e9:.1f} TFLOP/s", flush=True)
    H, D = 24, 128
    q = torch.randn(L, C, device=dev).to(BF16)
    k = torch.randn(L, C, device=dev).to(BF16)
    vt = torch.randn(C, (L + 63) // 64 * 64, device=dev).to(BF16)
    out = torch.empty(L, C, dtype=BF16, device=dev)
    ms = timeit(lambda: _lib.flash_attn(q, k, vt, out, L, L, H, D, 1 / math.sqrt(D)), iters=5)
    print(f"  attention L{L} H{H}: {ms:.3f} ms  {4 * L * L * C / ms / 1e9:.1f} TFLOP/s")
    x = torch.randn(L, C, device=dev)
    tab = torch.randn(1, 6 * C, device=dev)
    h = torch.empty(L, C, dtype=BF16, device=dev)
    ms = timeit(lambda: _lib.layernorm_mod(x, h, L, C, 1e-6, mode=1, tab=tab, shift_off=0, scale_off=C))
    print(f"  layernorm_mod: {ms * 1e3:.1f} us  {(L * C * 6) / ms / 1e6:.1f} GB/s")


if __name__ == "__main__":
    _lib.init()
    which = sys.argv[1:] or ["gemm", "gemm_f32", "attn", "glue", "dit", "vae", "perf"]
    for w in which:
        try:
            {"gemm": check_gemm, "gemm_f32": check_gemm_f32, "attn": check_attn, "glue": check_glue, "dit": check_dit,
             "vae": check_vae, "perf_vae": perf_vae, "perf": perf}[w]()
        except Exception as ex:  # keep going: one round trip should tell as much as possible
            import traceback
            traceback.print_exc()
            print(f"!! {w} failed: {ex}", flush=True)
