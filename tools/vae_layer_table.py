"""Per-layer table of a full-size VAE decode / encode (developer tool): every convolution launch of the engine timed with HIP events (one
synchronisation per launch: the sum is a little above the untimed decode), grouped by geometry, with the algorithmic and - f16x3 / bf16x6 -
the EXECUTED MFMA rate (3 / 6 passes).   python3 tools/vae_layer_table.py [decode|encode] [precision] [UV_OPT_CONV_HALO value: -1 auto, 0 gather kernels only, 1]"""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univid_amd import _lib
from univid_amd.wan import vae2_2
from univid_amd.wan.vae2_2 import Wan2_2_VAE
what = sys.argv[1] if len(sys.argv) > 1 else "decode"
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
_lib.init()
if len(sys.argv) > 3:
    _lib.set_option(_lib.OPT_CONV_HALO, int(sys.argv[3]))
vae = Wan2_2_VAE(device="cuda", seed=0, precision=prec)
g = torch.Generator(device="cuda").manual_seed(7)
z = torch.randn(48, 13, 45, 80, device="cuda", generator=g)
vid = torch.tanh(torch.randn(3, 49, 720, 1280, device="cuda", generator=g))
run = (lambda: vae.decode([z])) if what == "decode" else (lambda: vae.encode([vid]))
with torch.no_grad():
    run()                                   # warm-up: weight preparation
    rows = collections.OrderedDict()
    orig = vae2_2._Engine._conv

    def timed(self, op, src, Tin, Hin, Win, Tout, Hout, Wout, *a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(self, op, src, Tin, Hin, Win, Tout, Hout, Wout, *a, **kw)
        e1.record(); e1.synchronize()
        key = (op.cin, op.cout, op.kt, op.kh, op.kw, Tout, Hout, Wout, kw.get("up", 0), kw.get("in_split", 0), kw.get("sh", 1))
        r = rows.setdefault(key, [0, 0.0, 0.0])
        r[0] += 1; r[1] += e0.elapsed_time(e1); r[2] += 2.0 * Tout * Hout * Wout * op.cout * op.kt * op.kh * op.kw * op.cin
        return out
    vae2_2._Engine._conv = timed
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    vae2_2._Engine._conv = orig
tot = sum(r[1] for r in rows.values())
print(f"{what} {prec} 49x720x1280: {e0.elapsed_time(e1):.1f} ms with per-launch synchronisation, {tot:.1f} ms inside the convolution launches")
print("| Cin | Cout | kt x kh x kw | Tout x Hout x Wout | up | split | launches | ms | % of conv time | TFLOP/s algorithmic | executed (passes) |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for (ci, co, kt, kh, kw_, T, H, W, up, sp, sh), (n, ms, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    passes = 1 if prec == "fp32" else 3 if (prec == "bf16x3" or (prec == "f16x3" and sp == 2)) else 6
    print(f"| {ci} | {co} | {kt}x{kh}x{kw_}{' /2' if sh == 2 else ''} | {T}x{H}x{W} | {up} | {sp} | {n} | {ms:.1f} | {100 * ms / tot:.1f} | {fl / ms / 1e9:.0f} | {passes * fl / ms / 1e9:.0f} ({passes}) |")
