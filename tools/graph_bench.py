"""BASELINE config 1 (tiny DiT, 16-frame 256x256 clip -> latent [48,4,16,16], 10 UniPC steps) and a small TI2V-5B-width case:
denoise steps/s with the CFG pair's forward launched eagerly vs replayed from a captured HIP graph (WanTI2V.denoise(graph=...)).
Small latents are launch-bound (~60 launches of a few microseconds per forward); the graph removes the host gaps. Outputs are
compared bit for bit."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univid_amd import _lib                                              # noqa: E402
from univid_amd.wan.model import WanModel                                # noqa: E402
from univid_amd.wan.textimage2video import TI2VConfig, WanTI2V           # noqa: E402

_lib.init()
dev = "cuda"
TINY = dict(model_type="ti2v", dim=256, ffn_dim=512, num_heads=4, num_layers=2, freq_dim=256, text_len=32, text_dim=64, in_dim=48,
            out_dim=48, patch_size=(1, 2, 2), eps=1e-6)
WIDE = dict(TI2VConfig.dit, num_layers=4)
for name, cfg, shape, steps in (("config 1: tiny DiT (dim 256, 2 layers), latent [48,4,16,16], L=256", TINY, (48, 4, 16, 16), 10),
                                ("TI2V-5B width, 4 layers, latent [48,4,16,16], L=256", WIDE, (48, 4, 16, 16), 10),
                                ("TI2V-5B width, 4 layers, latent [48,5,24,32], L=960", WIDE, (48, 5, 24, 32), 10)):
    with torch.device(dev):
        m = WanModel.from_config(cfg)
    m = m.eval().requires_grad_(False)
    m.init_weights(0)
    pipe = WanTI2V(TI2VConfig, model=m, device=dev)
    g = torch.Generator(device=dev).manual_seed(1)
    noise = torch.randn(*shape, device=dev, generator=g)
    ctx = [torch.randn(20, cfg["text_dim"], device=dev, generator=g)]
    ctxn = [torch.randn(7, cfg["text_dim"], device=dev, generator=g)]
    res = {}
    for mode, flag in (("eager", False), ("hipGraph", True)):
        with torch.no_grad():
            out = pipe.denoise(noise, ctx, ctxn, steps, 5.0, 5.0, graph=flag)      # warm-up (and the capture)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                out = pipe.denoise(noise, ctx, ctxn, steps, 5.0, 5.0, graph=flag)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
        res[mode] = (sorted(ts)[2], out)
        print(f"{name:72s} {mode:9s} {steps / res[mode][0]:8.1f} steps/s  ({res[mode][0] * 1e3 / steps:6.2f} ms/step, incl. the capture for hipGraph)", flush=True)
    print(f"{'':72s} bit-identical: {bool(torch.equal(res['eager'][1], res['hipGraph'][1]))}  speed-up x{res['eager'][0] / res['hipGraph'][0]:.2f}", flush=True)
    del m, pipe
    torch.cuda.empty_cache()
