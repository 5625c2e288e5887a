"""In-kernel cycle anatomy of the ping-pong GEMM (diagnostic build, s_memtime stamps)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univid_amd import _lib
_lib.init()
lib = _lib._LIB if hasattr(_lib, "_LIB") else ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libunivid_hip.so"))
fn = lib.uvdbg_gemm_stamps
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
               ctypes.c_long, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = "cuda"
M, N, K = int(os.environ.get("M", 22880)), int(os.environ.get("N", 3072)), int(os.environ.get("K", 3072))
g = torch.Generator(device=dev).manual_seed(0)
A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
W = ((torch.rand(N, K, device=dev, generator=g) * 2 - 1) * 0.05).to(torch.bfloat16)
out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
tiles = ((M + 255) // 256) * ((N + 255) // 256)
dbg = torch.zeros(tiles * 8 * 16, dtype=torch.int64, device=dev)
for _ in range(20):
    _lib.gemm_bf16(A, W, None, out, 0, tile_cfg=7)     # heat the chip like the real loop
for _ in range(3):
    rc = fn(A.data_ptr(), A.stride(0), W.data_ptr(), W.stride(0), M, N, K, out.data_ptr(), out.stride(0), dbg.data_ptr(), int(os.environ.get('VARIANT', 2)), None)
    assert rc == 0
torch.cuda.synchronize()
d = dbg.view(tiles, 8, 16).double().cpu()
nk = K // 64
variant = int(os.environ.get('VARIANT', 2))
names6 = ["A load+waits", "A barrier1", "A mfma(32)", "A barrier2", "B load+waits", "B barrier1", "B mfma(32)", "B barrier2", "-", "-", "-", "-"]
names = names6 if variant == 6 else ["P1 load+b1", "P1 lds+mfma", "P1 b2", "P2 load+b1", "P2 lds+mfma", "P2 b2", "P3 load+b1", "P3 lds+mfma", "P3 b2",
         "P4 load+vm+b1", "P4 mfma", "P4 b2"]
for grp, sl in (("group 0 (waves 0-3)", slice(0, 4)), ("group 1 (waves 4-7)", slice(4, 8))):
    x = d[:, sl, :].reshape(-1, 16)
    per = x[:, :12].median(0).values / nk
    print(grp, " loop cycles/K-tile %.0f" % (x[:, 12].median().item() / nk), " in-kernel clock %.2f GHz" % (x[:, 12].median().item() / max(x[:, 14].median().item(), 1) * 0.1))
    for n, v in zip(names, per):
        print(f"   {n:16s} {v.item():7.1f}")
    print("   sum %.0f" % per.sum().item())

x = d.reshape(-1, 16)
print("per workgroup (100 MHz ticks -> us): prologue %.2f us, loop %.2f us, epilogue+drain %.2f us" % (
    x[:, 8].median().item() / 100, x[:, 14].median().item() / 100, x[:, 9].median().item() / 100))
ent, end = d[:, 0, 11], d[:, 0, 10]
print("kernel span %.1f us; workgroup lifetime median %.1f us; %d tiles on %d CUs" % ((end.max() - ent.min()).item() / 100,
      (end - ent).median().item() / 100, tiles, torch.cuda.get_device_properties(0).multi_processor_count))
order = torch.argsort(ent)
starts = (ent[order] - ent.min()) / 100
print("entry time of the 1st/256th/257th/512th/last workgroup: %.1f %.1f %.1f %.1f %.1f us" % (
    starts[0].item(), starts[min(255, tiles - 1)].item(), starts[min(256, tiles - 1)].item(), starts[min(511, tiles - 1)].item(), starts[-1].item()))
