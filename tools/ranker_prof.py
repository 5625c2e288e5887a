import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from univid_amd import _lib
from univid_amd.understanding import Siglip2Model
_lib.init()
dev="cuda"
V = dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12, num_channels=3, patch_size=16, num_patches=256, layer_norm_eps=1e-6)
T = dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12, vocab_size=32000, max_position_embeddings=64, projection_size=768, layer_norm_eps=1e-6)
with torch.device(dev):
    m = Siglip2Model(dict(vision=V, text=T), dtype=torch.float16)
m.init_weights(0).eval()
B,N=64,256
pv=torch.randn(B,N,768,device=dev); mask=torch.ones(B,N,dtype=torch.int64,device=dev); shapes=torch.tensor([[16,16]]*B,device=dev)
for _ in range(3): m.get_image_features(pv,mask,shapes)
torch.cuda.synchronize()
_lib.PROFILE={}; _lib.PROFILE_ALL=True
t0=time.perf_counter(); m.get_image_features(pv,mask,shapes); torch.cuda.synchronize(); dt=time.perf_counter()-t0
prof=_lib.PROFILE; _lib.PROFILE=None; _lib.PROFILE_ALL=False
tot=0; n=0
for k,evs in prof.items():
    ms=sum(s.elapsed_time(e) for s,e,_ in evs); tot+=ms; n+=len(evs)
    print(f"{k:24s} {len(evs):4d} {ms:7.3f} ms")
print("launches",n,"kernel sum %.3f ms"%tot, "wall (with event overhead) %.3f ms"%(dt*1e3))
t0=time.perf_counter()
for _ in range(10): m.get_image_features(pv,mask,shapes)
torch.cuda.synchronize(); print("wall per call %.3f ms"%((time.perf_counter()-t0)/10*1e3))
