"""Full-size Wan2.2 VAE run for the rocprofv3 passes behind profiles/r02_vae_*.md (BASELINE config 4):
encode of a [3,49,720,1280] clip -> latent [48,13,45,80] -> decode back to [3,49,720,1280], fp32 (the reference dtype).

    python3 tools/vae_trace.py [encode|decode|both] [fp32|bf16x6|f16x3|bf16x3]
A two-latent-frame decode / five-frame encode runs first (untimed): the one-time weight preparation of the mode (splits, phase sums) and
the allocator's first touches stay out of the timed call, as in bench.py."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univid_amd import _lib                      # noqa: E402
from univid_amd.wan.vae2_2 import Wan2_2_VAE     # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "both"
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
_lib.init()
vae = Wan2_2_VAE(device="cuda", seed=0, precision=prec)
g = torch.Generator(device="cuda").manual_seed(7)
with torch.no_grad():
    if what in ("decode", "both"):
        vae.decode([torch.randn(48, 2, 45, 80, device="cuda", generator=g)])
    if what in ("encode", "both"):
        video = torch.randn(3, 49, 720, 1280, device="cuda", generator=g).tanh_()
        vae.encode([video[:, :5].contiguous()])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        z = vae.encode([video])[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"encode {prec}: {dt:.3f} s  latent {tuple(z.shape)} finite={bool(torch.isfinite(z).all())} "
              f"{159.1 / dt:.1f} TFLOP/s  {video.numel() * 4 / dt / 1e9:.3f} GB/s in", flush=True)
        del video
    else:
        z = torch.randn(48, 13, 45, 80, device="cuda", generator=g)
    if what in ("decode", "both"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        v = vae.decode([z])[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"decode {prec}: {dt:.3f} s  video {tuple(v.shape)} finite={bool(torch.isfinite(v).all())} "
              f"{835.4 / dt:.1f} TFLOP/s  {v.numel() * 4 / dt / 1e9:.3f} GB/s out", flush=True)
