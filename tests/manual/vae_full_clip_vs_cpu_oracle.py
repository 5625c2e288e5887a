"""BASELINE config 4 at its full size against the PINNED CPU oracle, once per round (many minutes of host time - a tool, not a test): HIP decode
(and encode) of the 49 x 720 x 1280 clip against oracle/wan_vae.WanVAE on the host cores, every element inside rtol 1e-3 / atol 1e-4;
every f32-grade mode (PRECS = fp32,bf16x6,f16x3). F = frames (1 + 4k). The round's output is profiles/rNN_vae_full_clip_vs_cpu_oracle.log.
    F=49 python3 tests/manual/vae_full_clip_vs_cpu_oracle.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import wan_vae
from univid_amd.wan.vae2_2 import Wan2_2_VAE
dev = "cuda"
torch.set_num_threads(min(128, os.cpu_count() or 8))
F = int(os.environ.get("F", 49))
vae = Wan2_2_VAE(device=dev, seed=2, precision="fp32")
sd = {k: v.detach().cpu() for k, v in vae.model.state_dict().items()}
ora = wan_vae.WanVAE(sd, wan_vae.FULL_CFG)
scale = wan_vae.scale_tensors()
g = torch.Generator().manual_seed(8)
vid = torch.tanh(torch.randn(3, F, 720, 1280, generator=g))


def cmp(name, got, ref):
    d = (got.float().cpu() - ref).abs()
    bad = int((d > 1e-4 + 1e-3 * ref.abs()).sum())
    print(f"{name}: max abs err {float(d.max()):.3e} (|ref| max {float(ref.abs().max()):.3f}), outside rtol 1e-3 / atol 1e-4: {bad} of {d.numel()}", flush=True)


with torch.no_grad():
    t0 = time.time(); zr = ora.encode(vid.unsqueeze(0), scale).float().squeeze(0); print(f"cpu oracle encode {time.time() - t0:.1f} s on {torch.get_num_threads()} threads", flush=True)
    do_dec = os.environ.get("DECODE", "1") != "0"        # DECODE=0: the encoder only (the oracle's decode is the long part)
    if do_dec:
        t0 = time.time(); vr = ora.decode(zr.unsqueeze(0), scale).float().clamp_(-1, 1).squeeze(0); print(f"cpu oracle decode {time.time() - t0:.1f} s", flush=True)
    for prec in os.environ.get("PRECS", "fp32,bf16x6,f16x3").split(","):
        v = vae if prec == "fp32" else Wan2_2_VAE(device=dev, seed=2, precision=prec)
        torch.cuda.synchronize(); t0 = time.time(); z = v.encode([vid.to(dev)])[0]; torch.cuda.synchronize(); te = time.time() - t0
        print(f"hip {prec}: encode {te:.2f} s (first call: includes weight preparation)", flush=True)
        cmp(f"  {prec} encode {tuple(z.shape)}", z, zr)
        if do_dec:
            t0 = time.time(); out = v.decode([zr.to(dev)])[0]; torch.cuda.synchronize(); td = time.time() - t0
            print(f"hip {prec}: decode {td:.2f} s (first call: includes weight preparation)", flush=True)
            cmp(f"  {prec} decode {tuple(out.shape)}", out, vr)
