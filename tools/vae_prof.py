import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from univid_amd import _lib
from univid_amd.wan.vae2_2 import Wan2_2_VAE
_lib.init()
prec = os.environ.get("PREC", "bf16x3")
vae = Wan2_2_VAE(device="cuda", seed=0, precision=prec)
g = torch.Generator(device="cuda").manual_seed(7)
z = torch.randn(48, 13, 45, 80, device="cuda", generator=g)
with torch.no_grad():
    vae.decode([z[:, :2].contiguous()])
    torch.cuda.synchronize()
    _lib.PROFILE = {}
    _lib.PROFILE_ALL = True
    t0 = time.perf_counter()
    v = vae.decode([z])[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
prof = _lib.PROFILE
_lib.PROFILE, _lib.PROFILE_ALL = None, False
print(f"decode {prec}: {dt:.3f} s")
tot = 0
for name, evs in sorted(prof.items(), key=lambda kv: -sum(s.elapsed_time(e) for s, e, _ in kv[1])):
    ms = sum(s.elapsed_time(e) for s, e, _ in evs)
    tot += ms
    print(f"  {name:28s} {len(evs):6d} launches {ms:9.1f} ms")
print("  sum of kernels", round(tot, 1), "ms")
