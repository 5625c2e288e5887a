"""The headline configuration against the PINNED oracle, once per round (minutes of host time - a tool, not a test): the whole 30-block
TI2V-5B DiT forward at the bench's length (L = 13 x 22 x 40 = 11 440 tokens, two timesteps, 77-row prompt) through WanModel.forward
on the GPU against oracle/wan_dit.dit_forward on the host cores and against its no-rounding truth run; residual stream compared after
blocks 1, 2, 4, 8, 16, 30 and at the output. The round's output is profiles/rNN_full_forward_vs_cpu_oracle.log.
GRID = the token grid (frames, rows, columns after patching): "31,22,40" is UniVid's own default workload (121 frames of 704 x 1280,
L = 27 280, inference.py:48-50; the oracle's attention grows with L^2 - use LAYERS=16 there).
    LAYERS=30 python3 tests/manual/full_forward_vs_cpu_oracle.py
    GRID=31,22,40 LAYERS=16 python3 tests/manual/full_forward_vs_cpu_oracle.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import lora as ora_lora, wan_dit
from univid_amd.wan.model import WanModel
dev = "cuda"
torch.set_num_threads(min(128, os.cpu_count() or 8))
n = int(os.environ.get("LAYERS", 30))
cfg = dict(wan_dit.TI2V_5B_CFG, num_layers=n)
with torch.device(dev):
    m = WanModel.from_config(dict(cfg, model_type="ti2v"))
m = m.eval().requires_grad_(False)
m.init_weights(0)
sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
grid = tuple(int(v) for v in os.environ.get("GRID", "13,22,40").split(","))
DEPTHS = [d for d in (1, 2, 4, 8, 16, 30) if d <= n]
Lt = grid[0] * grid[1] * grid[2]
g = torch.Generator().manual_seed(41)
x = torch.randn(48, grid[0], 2 * grid[1], 2 * grid[2], generator=g)
ctx = [torch.randn(77, cfg["text_dim"], generator=g) * 0.1]
t = torch.full((1, Lt), 812.0)
t[0, :grid[1] * grid[2]] = 0.0
hidden = {}
for i, blk in enumerate(m.blocks):
    def run(xs, *a, _orig=blk._run, _d=i + 1, **kw):
        _orig(xs, *a, **kw)
        if _d in DEPTHS:
            hidden[_d] = xs.float().cpu()
    blk._run = run
with torch.no_grad():
    t0 = time.time(); out = m([x.to(dev)], t.to(dev), [c.to(dev) for c in ctx], Lt)[0].cpu(); print(f"hip forward (incl. weight preparation and copies) {time.time() - t0:.1f} s", flush=True)
    t0 = time.time(); ref, ref_h, _ = wan_dit.dit_forward(sd, cfg, [x], t, ctx, Lt, return_hidden=True); print(f"cpu oracle {time.time() - t0:.1f} s on {torch.get_num_threads()} threads", flush=True)
    ref_h = {d: ref_h[d - 1][0] for d in DEPTHS}
    old = wan_dit.BF16
    wan_dit.BF16 = ora_lora.BF16 = torch.float32
    try:
        t0 = time.time(); tru, tru_h, _ = wan_dit.dit_forward(sd, cfg, [x], t, ctx, Lt, return_hidden=True); print(f"cpu truth (no rounding) {time.time() - t0:.1f} s", flush=True)
        tru_h = {d: tru_h[d - 1][0] for d in DEPTHS}
    finally:
        wan_dit.BF16 = ora_lora.BF16 = old


def report(name, got, r, tr):
    got, r, tr = got.float(), r.float(), tr.float()
    d = (got - r).abs()
    inside = (d <= 1e-4 + 1e-3 * r.abs()).float().mean().item()
    e_hip, e_ora = (got - tr).pow(2).mean().sqrt().item(), (r - tr).pow(2).mean().sqrt().item()
    print(f"{name:34s} inside rtol 1e-3/atol 1e-4: {100 * inside:5.1f} %   max err / range {float(d.max() / r.abs().max()):.2e}   "
          f"rms vs truth: hip {e_hip:.3e}  oracle {e_ora:.3e}  ratio {e_hip / e_ora:.4f}", flush=True)


for depth in DEPTHS:
    report(f"residual stream after block {depth}", hidden[depth], ref_h[depth], tru_h[depth])
report(f"{n}-block forward output", out, ref[0], tru[0])
