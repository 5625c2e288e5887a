"""Same-process A/B of the convolution kernels behind a full-size VAE decode (developer tool): UV_CONV_HALO=0 (gather kernel) against 1
(LDS-halo kernel wherever the geometry fits), interleaved, after a warm-up decode of each.   python3 tools/vae_halo_ab.py [fp32|bf16x6] [rounds]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univid_amd import _lib
from univid_amd.wan.vae2_2 import Wan2_2_VAE
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
_lib.init()
vae = Wan2_2_VAE(device="cuda", seed=0, precision=prec)
g = torch.Generator(device="cuda").manual_seed(7)
z = torch.randn(48, 13, 45, 80, device="cuda", generator=g)
res = {"0": [], "1": []}
outs = {}
with torch.no_grad():
    for r in range(rounds + 1):
        for h in ("0", "1"):
            _lib.set_option(_lib.OPT_CONV_HALO, int(h))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            v = vae.decode([z])[0]
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            if r:
                res[h].append(dt)
            else:
                outs[h] = v
    d = (outs["0"] - outs["1"]).abs().max().item()
print(f"decode {prec} 49x720x1280: gather (UV_CONV_HALO=0) " + " ".join(f"{t:.3f}" for t in res["0"]) + " s | halo (=1) " + " ".join(f"{t:.3f}" for t in res["1"]) +
      f" s | max |gather - halo| = {d:.2e}")
