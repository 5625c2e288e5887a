"""Full-RESOLUTION VAE parity (BASELINE config 4, 720 x 1280): HIP encode / decode of an F-frame clip against the oracle's text executed by
torch-ROCm eager on the same GPU (MIOpen fp32 convolutions). Minutes of MIOpen kernel search on a fresh box - a tool, not a test; the
round's output is profiles/r02_vae_fullres_vs_eager.log.   F=5 python3 tests/manual/vae_fullres_check.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import wan_vae
from univid_amd.wan.vae2_2 import Wan2_2_VAE
dev = "cuda"
vae = Wan2_2_VAE(device=dev, seed=2, precision="fp32")
sd = {k: v.detach() for k, v in vae.model.state_dict().items()}
ora = wan_vae.WanVAE(sd, wan_vae.FULL_CFG)
scale = [t.to(dev) for t in wan_vae.scale_tensors()]
g = torch.Generator(device=dev).manual_seed(8)
F = int(os.environ.get("F", 5))
vid = torch.tanh(torch.randn(3, F, 720, 1280, device=dev, generator=g))
with torch.no_grad():
    t0 = time.time(); z = vae.encode([vid])[0]; torch.cuda.synchronize(); print("hip encode", time.time() - t0, flush=True)
    t0 = time.time(); zr = ora.encode(vid.unsqueeze(0), scale).float().squeeze(0); torch.cuda.synchronize(); print("eager encode", time.time() - t0, flush=True)
    d = (z - zr).abs(); print("encode max abs", float(d.max()), "ref absmax", float(zr.abs().max()), "outside 1e-3/1e-4:", int((d > 1e-4 + 1e-3 * zr.abs()).sum()), flush=True)
    t0 = time.time(); v = vae.decode([zr])[0]; torch.cuda.synchronize(); print("hip decode", time.time() - t0, flush=True)
    t0 = time.time(); vr = ora.decode(zr.unsqueeze(0), scale).float().clamp_(-1, 1).squeeze(0); torch.cuda.synchronize(); print("eager decode", time.time() - t0, flush=True)
    d = (v - vr).abs(); print("decode max abs", float(d.max()), "outside 1e-3/1e-4:", int((d > 1e-4 + 1e-3 * vr.abs()).sum()), "of", d.numel(), flush=True)
    # the f32-grade bf16x6 arithmetic at the same resolution: against the same eager-oracle outputs and against the exact-f32 HIP mode
    vae6 = Wan2_2_VAE(device=dev, seed=2, precision="bf16x6")
    z6 = vae6.encode([vid])[0]
    d = (z6 - zr).abs(); print("bf16x6 encode max abs", float(d.max()), "outside 1e-3/1e-4:", int((d > 1e-4 + 1e-3 * zr.abs()).sum()),
                               "| vs exact-f32 HIP: max abs", float((z6 - z).abs().max()), flush=True)
    v6 = vae6.decode([zr])[0]
    d = (v6 - vr).abs(); print("bf16x6 decode max abs", float(d.max()), "outside 1e-3/1e-4:", int((d > 1e-4 + 1e-3 * vr.abs()).sum()), "of", d.numel(),
                               "| vs exact-f32 HIP: max abs", float((v6 - v).abs().max()), flush=True)
