"""Attention-only micro benchmark (developer tool): full-size self-attention launches for profiling."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univid_amd import _lib
_lib.init()
dev = "cuda"; BF16 = torch.bfloat16
L, H, D = int(os.environ.get("L", 11440)), 24, 128
C = H * D
torch.manual_seed(0)
q = torch.randn(L, C, device=dev).to(BF16); k = torch.randn(L, C, device=dev).to(BF16)
vt = torch.randn(C, (L + 63) // 64 * 64, device=dev).to(BF16)
out = torch.empty(L, C, dtype=BF16, device=dev)
n = int(os.environ.get("N", 5))
for _ in range(n):
    _lib.flash_attn(q, k, vt, out, L, L, H, D, 1 / math.sqrt(D))
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(n):
    _lib.flash_attn(q, k, vt, out, L, L, H, D, 1 / math.sqrt(D))
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / n
print(f"attention L{L}: {ms:.3f} ms {4*L*L*C/ms/1e9:.1f} TFLOP/s")

if os.environ.get("STAMPS"):
    import ctypes
    lib = _lib.load()
    for nw, extra in ((4, 0), (44, 0), (42, 0)):
        nblk = ((L + 127) // 128) * H if nw == 44 else ((L + 255) // 256) * H if nw >= 42 else ((L + nw * 32 - 1) // (nw * 32)) * H
        st = torch.zeros(nblk * (4 if nw >= 42 else nw) * 6, dtype=torch.int64, device=dev)
        fn = lib.uvdbg_flash_attn_stamps
        fn.argtypes = [ctypes.c_void_p, ctypes.c_long] * 3 + [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        for _ in range(2):
            torch.cuda.synchronize(); import time; t0 = time.time()
            rc = fn(q.data_ptr(), C, k.data_ptr(), C, vt.data_ptr(), vt.stride(0), out.data_ptr(), C, L, L, H, 1 / math.sqrt(D), nw, st.data_ptr(), extra, None)
            torch.cuda.synchronize(); dt = time.time() - t0
        print(f"nw={nw} extra_lds={extra}: wall {dt*1e3:.2f} ms")
        v = st.view(-1, 6).double()
        tiles = (L + 63) // 64
        per = v.median(0).values / tiles
        if nw in (42, 43):
            print(f"QB=2 sgb={nw == 43}: median cycles per 64-query tile per wave  step1 QK_A|expB={per[0]:.0f} step2 PV_B1|maxA+decide={per[1]:.0f} step3 PV_B2,QK_B|expA={per[2]:.0f} step4 PV_A1|maxB+decide={per[3]:.0f} step5 PV_A2+wait+barrier={per[4]:.0f} total={per.sum():.0f}")
        else:
            print(f"nw={nw}: median cycles per tile per wave  dma_issue={per[5]:.0f} qk={per[0]:.0f} softmax={per[1]:.0f} pv={per[2]:.0f} commit={per[3]:.0f} barrier={per[4]:.0f} total={per.sum():.0f}")
