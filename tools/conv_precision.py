"""Accuracy and speed of the three convolution arithmetics on one VAE-sized layer (developer tool): exact f32 MFMA, bf16x6 (exact
3-way operand split, 6 bf16 MFMA passes) and bf16x3 (2-way split, 3 passes), each against an fp64 convolution of the same operands."""
import os, sys, statistics, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univid_amd import _lib
_lib.init()
dev = "cuda"
g = torch.Generator().manual_seed(0)
_SPLIT = {}


def run(name, x_cl, wp, b, out, T, H, W, C, co):
    if name.endswith("x6"):
        key = wp.data_ptr()
        if key not in _SPLIT:
            _SPLIT[key] = torch.empty(wp.numel() * 3, dtype=torch.bfloat16, device=dev)
            _lib.call("uv_split_weights_bf16x6", _lib.ptr(wp), _lib.ptr(_SPLIT[key]), wp.numel(), _lib.stream_ptr())
        wp = _SPLIT[key]
    _lib.call(name, _lib.ptr(x_cl), C, T + 2, H, W, _lib.ptr(wp), _lib.ptr(b), _lib.ptr(out), co, T, H, W, C, co, 3, 3, 3, 1, 1, 1, 0, 1, 1,
              0, 0, None, 0, _lib.stream_ptr())


# accuracy: small spatial size, production channel counts, against fp64 on the CPU
for C, co in ((256, 256), (512, 512), (1024, 1024)):
    T, H, W = 2, 12, 16
    x = torch.randn(1, C, T + 2, H, W, generator=g)
    x = torch.nn.functional.silu(x)                                    # activations as the convolutions see them (post-SiLU)
    w = torch.randn(co, C, 3, 3, 3, generator=g) / (27 * C) ** 0.5
    b = torch.randn(co, generator=g) * 0.1
    ref = F.conv3d(F.pad(x.double(), (1, 1, 1, 1, 0, 0)), w.double(), b.double())[0]          # time: the 2 leading frames are the cache
    ref = ref.permute(1, 2, 3, 0)
    x_cl = x[0].permute(1, 2, 3, 0).contiguous().to(dev)
    wp = w.permute(0, 2, 3, 4, 1).reshape(co, -1).contiguous().to(dev)
    res = {}
    for name in ("uv_conv3d_f32", "uv_conv3d_bf16x6"):
        out = torch.empty(T, H, W, co, device=dev)
        run(name, x_cl, wp, b.to(dev), out, T, H, W, C, co)
        d = (out.cpu().double() - ref)
        res[name] = (float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()), float(d.abs().max() / ref.abs().max()))
    cpu32 = F.conv3d(F.pad(x, (1, 1, 1, 1, 0, 0)), w, b)[0].permute(1, 2, 3, 0)
    d = cpu32.double() - ref
    res["torch CPU fp32 conv3d"] = (float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()), float(d.abs().max() / ref.abs().max()))
    print(f"C = {C} -> {co}, K = {27 * C}: error vs fp64 (rel rms, max / range)")
    for k, (a, m) in res.items():
        print(f"   {k:24s} {a:.3e}  {m:.3e}")

# speed: the decoder's largest layer shape class (256 -> 256 channels, 16 frames of 360 x 640)
T, H, W, C, co = 16, 360, 640, 256, 256
x_cl = torch.randn(T + 2, H, W, C, device=dev)
wp = torch.randn(co, 27 * C, device=dev) * 0.01
b = torch.zeros(co, device=dev)
out = torch.empty(T, H, W, co, device=dev)
fl = 2.0 * T * H * W * co * 27 * C
ts = {"uv_conv3d_f32": [], "uv_conv3d_bf16x6": []}
for rnd in range(4):
    for name in ts:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); run(name, x_cl, wp, b, out, T, H, W, C, co); e.record(); torch.cuda.synchronize()
        if rnd:
            ts[name].append(s.elapsed_time(e))
for name, v in ts.items():
    t = statistics.median(v)
    print(f"{name:18s} {T}x{H}x{W}, {C}->{co}: {t:8.2f} ms  {fl / t / 1e9:7.1f} TFLOP/s (algorithmic)")
