import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conftest import load_golden, bf16_ulp
from univid_amd import _lib
from univid_amd._lib import *
_lib.init()
DEV="cuda"; BF16=torch.bfloat16
print("== unipc")
from univid_amd.wan.fm_solvers_unipc import FlowUniPCMultistepScheduler
from oracle import unipc as ou
g = load_golden("unipc")
s = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1); s.set_timesteps(10, device="cpu", shift=5.0)
o = ou.FlowUniPC(1000, shift=1); o.set_timesteps(10, shift=5.0)
lat = g["x"].to(DEV); lo = g["x"]
for i, t in enumerate(s.timesteps):
    lat = s.step(g["model_outputs"][i].to(DEV), t, lat, return_dict=False)[0]
    lo = o.step(g["model_outputs"][i], t, lo)
    d = (lat.cpu() - g["trajectory"][i]).abs()
    nz = (d > 0).sum().item()
    print(i, "mismatch", nz, "max", d.max().item(), "this_order", s.this_order, "oracle eq golden", torch.equal(lo, g["trajectory"][i]))
    if nz:
        idx = (d > 0).nonzero()[0]
        print("   at", idx.tolist(), lat.cpu()[tuple(idx)].item(), g["trajectory"][i][tuple(idx)].item())
        lat = g["trajectory"][i].to(DEV)  # resync
print("== gemm")
for (M,N,K) in [(1000,768,192),(257,3072,3072)]:
    gg = torch.Generator().manual_seed(M+N+K)
    a=(torch.randn(M,K,generator=gg)*0.5).to(BF16); w=(torch.randn(N,K,generator=gg)*0.05).to(BF16); bias=(torch.randn(N,generator=gg)*0.1).to(BF16)
    yb=(a.double()@w.double().t()+bias.double()).to(BF16)
    out=torch.zeros(M,N,dtype=BF16,device=DEV); _lib.gemm_bf16(a.to(DEV),w.to(DEV),bias.to(DEV),out,EPI_BF16)
    d=(out.float().cpu()-yb.float()).abs(); u=bf16_ulp(torch.maximum(yb.float().abs(), out.float().cpu().abs()))
    print(M,N,K,"max ulp",(d/u).max().item(),"exact frac",(d==0).float().mean().item())
    out=torch.zeros(M,N,dtype=BF16,device=DEV); _lib.gemm_bf16(a.to(DEV),w.to(DEV),bias.to(DEV),out,EPI_GELU_BF16)
    ref=torch.nn.functional.gelu(yb,approximate="tanh")
    d=(out.float().cpu()-ref.float()).abs(); u=bf16_ulp(torch.maximum(ref.float().abs(), out.float().cpu().abs()))
    print("  gelu max ulp",(d/u).max().item(),"exact frac",(d==0).float().mean().item(), "max abs", d.max().item())
print("== block 3072")
from univid_amd import detinit
from univid_amd.wan.model import WanAttentionBlock, rope_params
g = load_golden("dit_block_3072")
dim, ffn, heads, Lt = 3072, 14336, 24, 48
with torch.device(DEV):
    blk = WanAttentionBlock(dim, ffn, heads, (-1, -1), True, True, 1e-6)
sd = {"blocks.0." + k: v for k, v in blk.state_dict(keep_vars=True).items()}
detinit.init_state_dict_(sd, g["seed"]); blk.eval()
d_ = dim // heads
freqs = torch.cat([rope_params(1024, d_ - 4 * (d_ // 6)), rope_params(1024, 2 * (d_ // 6)), rope_params(1024, 2 * (d_ // 6))], dim=1)
e0 = g["e_rows"][g["tid"]].unsqueeze(0)
with torch.no_grad():
    out = blk(g["x"].to(DEV), e0.to(DEV), torch.tensor([Lt]), g["grid"], freqs, g["ctx"].to(DEV), None)
from oracle import wan_dit
sdc = {k: v.detach().cpu() for k, v in sd.items()}
old = wan_dit.BF16; wan_dit.BF16 = torch.float32
with torch.no_grad():
    truth = wan_dit.block_forward(sdc, "blocks.0.", g["x"], e0, torch.tensor([Lt]), g["grid"], wan_dit.rope_table(d_), g["ctx"].float(), heads, 1e-6)
wan_dit.BF16 = old
ref = g["out_f32"]; got = out.cpu()
dd = (got-ref).abs()
print("hip-vs-oracle: max", dd.max().item(), "mean", dd.mean().item(), "ref absmax", ref.abs().max().item(), "ref rms", ref.pow(2).mean().sqrt().item(),
      "inside", (dd <= 1e-4+1e-3*ref.abs()).float().mean().item())
print("delta rms (out - x): ", (ref-g["x"]).pow(2).mean().sqrt().item())
print("rms err hip vs truth", (got-truth).pow(2).mean().sqrt().item(), "oracle vs truth", (ref-truth).pow(2).mean().sqrt().item(), "hip vs oracle", (got-ref).pow(2).mean().sqrt().item())
