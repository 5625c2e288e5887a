"""Experiment: the CFG pair as one stacked forward vs two single-sample forwards on two HIP streams."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from univid_amd import _lib
from univid_amd.wan.textimage2video import TI2VConfig

cfg = {k: v for k, v in TI2VConfig.dit.items() if k not in ("model_type", "window_size", "qk_norm", "cross_attn_norm")}
cfg["num_layers"] = int(os.environ.get("LAYERS", 30))
_lib.init()
dev = torch.device("cuda", 0)
m = bench.build_model(cfg, dev)
g = torch.Generator(device=dev).manual_seed(1)
lat = torch.randn(*bench.LATENT, device=dev, generator=g)
ca = torch.randn(77, cfg["text_dim"], device=dev, generator=g) * 0.1
cb = torch.randn(12, cfg["text_dim"], device=dev, generator=g) * 0.1
L = bench.L_TOKENS
tv = torch.full((1, L), 500.0, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def stacked():
    return m([lat, lat], t=torch.cat([tv, tv]), context=[ca, cb], seq_len=L)

def two_streams():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        a = m([lat], t=tv, context=[ca], seq_len=L)[0]
    with torch.cuda.stream(s2):
        b = m([lat], t=tv, context=[cb], seq_len=L)[0]
    cur.wait_stream(s1); cur.wait_stream(s2)
    return a, b

def sequential():
    return m([lat], t=tv, context=[ca], seq_len=L)[0], m([lat], t=tv, context=[cb], seq_len=L)[0]

with torch.no_grad():
    ref = stacked(); two = two_streams()
    print("two-stream == stacked:", torch.equal(ref[0], two[0]), torch.equal(ref[1], two[1]))
    for name, fn in (("stacked", stacked), ("two_streams", two_streams), ("sequential", sequential), ("stacked", stacked), ("two_streams", two_streams)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        print(f"{name:12s} {(time.perf_counter() - t0) / 3 * 1e3:8.2f} ms per CFG pair", flush=True)
