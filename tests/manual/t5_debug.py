import os, sys, torch, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univid_amd import _lib
from conftest import load_golden
from oracle import t5 as ot5
_lib.init()
DEV="cuda"; BF=torch.bfloat16
g=load_golden("t5_tiny"); cfg=ot5.TINY_CFG
sd=ot5.make_state_dict(cfg,int(g["seed"]))
from univid_amd.wan.t5 import T5Encoder
m=T5Encoder(vocab=cfg["vocab_size"],dim=256,dim_attn=256,dim_ffn=512,num_heads=4,num_layers=2,num_buckets=32)
m.load_state_dict(sd); m=m.to(device=DEV,dtype=BF).eval()
n=48; ids=g["ids_48"]
x=sd["token_embedding.weight"][ids]
# stage 1: norm
y_ref=ot5._norm(x, sd["blocks.0.norm1.weight"])
wf=m._norm_weights()
y=m._rms(x.to(DEV).contiguous(), wf[0][0], 1e-6)
print("norm1 diff", (y.float().cpu()-y_ref.float()).abs().max().item(), (y.cpu()==y_ref).float().mean().item())
# stage 2: attention alone with reference q,k,v
import torch.nn.functional as F
H=4;c=64
q=F.linear(y_ref, sd["blocks.0.attn.q.weight"]); k=F.linear(y_ref, sd["blocks.0.attn.k.weight"]); v=F.linear(y_ref, sd["blocks.0.attn.v.weight"])
rel=torch.arange(n).unsqueeze(0)-torch.arange(n).unsqueeze(1)
e=sd["blocks.0.pos_embedding.embedding.weight"][ot5.relative_position_bucket(rel)].permute(2,0,1)
attn=torch.einsum("inc,jnc->nij", q.view(n,H,c), k.view(n,H,c))+e
attn=F.softmax(attn.float(),-1).type_as(attn)
a_ref=torch.einsum("nij,jnc->inc", attn, v.view(n,H,c)).reshape(n,H*c)
tab=m.blocks[0].pos_embedding.table(n)
qd,kd=q.to(DEV).contiguous(),k.to(DEV).contiguous()
vd=v.to(DEV).contiguous()
att=torch.empty(n,256,dtype=BF,device=DEV)
_lib.call("uv_t5_attention_bf16", _lib.ptr(qd), qd.stride(0), _lib.ptr(kd), kd.stride(0), _lib.ptr(vd), vd.stride(0), _lib.ptr(att), att.stride(0), n, H, _lib.ptr(tab), n, _lib.stream_ptr())
d=(att.float().cpu()-a_ref.float()).abs()
print("attention diff max", d.max().item(), "rel rms", (d.pow(2).mean().sqrt()/a_ref.float().pow(2).mean().sqrt()).item(), "exact", (d==0).float().mean().item())
# gated gelu
gg=torch.randn(n,512).to(BF); ff=torch.randn(n,512).to(BF)
ref=(ff*ot5._gelu(gg))
gd,fd=gg.to(DEV),ff.to(DEV); od=torch.empty_like(gd)
_lib.call("uv_t5_gated_gelu_bf16", _lib.ptr(gd), _lib.ptr(fd), _lib.ptr(od), gd.numel(), _lib.stream_ptr())
print("gated gelu exact frac", (od.cpu()==ref).float().mean().item(), (od.float().cpu()-ref.float()).abs().max().item())

for nn in (48,33,5):
    got=m.encode(g[f"ids_{nn}"]).float().cpu(); ref=g[f"out_{nn}"].float(); d=(got-ref).abs()
    print(nn, "rel rms", (d.pow(2).mean().sqrt()/ref.pow(2).mean().sqrt()).item(), "max", d.max().item(), "exact", (d==0).float().mean().item())
