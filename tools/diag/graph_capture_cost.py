"""How long does building the HIP-graph runner of the CFG pair's forward take at production size (round 4)? Breakdown: eager warm-up forward,
capture + instantiate, first replay, second replay - what a new prompt cost when the graph was keyed on the context tensors - and
_GraphedPair.refresh (the runner-owned context buffers recomputed in place: what a new prompt costs now)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from univid_amd import _lib
from univid_amd.wan.model import WanModel
from univid_amd.wan.textimage2video import TI2VConfig, _GraphedPair
_lib.init()
dev = "cuda"
with torch.device(dev):
    m = WanModel.from_config(TI2VConfig.dit)
m = m.eval().requires_grad_(False); m.init_weights(0); m.prepare()
g = torch.Generator(device=dev).manual_seed(1)
F = int(os.environ.get("FRAMES", 13))
lat = torch.randn(48, F, 44, 80, device=dev, generator=g)
L = F * 22 * 40


def sync():
    torch.cuda.synchronize(); return time.perf_counter()


with torch.no_grad(), m.context_cached():
    for rep in range(3):
        pe = [torch.randn(40, 4096, device=dev, generator=g) * 0.1]
        ne = [torch.randn(9, 4096, device=dev, generator=g) * 0.1]
        t0 = sync()
        r = _GraphedPair(m, lat, pe, ne, L, None)
        t1 = sync()
        r(lat, 500.0); t2 = sync()
        r(lat, 400.0); t3 = sync()
        pe2 = [torch.randn(33, 4096, device=dev, generator=g) * 0.1]
        t4 = sync(); r.refresh(pe2, ne); t5 = sync()
        print(f"runner {rep} (new prompt tensors): build {t1 - t0:.3f} s, first replay {t2 - t1:.3f} s, second replay {t3 - t2:.3f} s; "
              f"refresh for another prompt (no recapture) {1e3 * (t5 - t4):.2f} ms", flush=True)
        del r
