"""Same-process A/B of uv_gemm_bf16_nt from TWO builds of the library (pattern of tools/attn_so_ab.py): the tree's against another .so."""
import ctypes, os, statistics, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.cuda.init()
_mode = os.RTLD_NOW | os.RTLD_LOCAL | os.RTLD_DEEPBIND
libs = {"tree": ctypes.CDLL(os.path.join(ROOT, "univid_amd", "libunivid_hip.so"), mode=_mode), "other": ctypes.CDLL(os.path.abspath(sys.argv[1]), mode=_mode)}
P, L_, I_ = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
for lib in libs.values():
    lib.uv_init()
    lib.uv_gemm_bf16_nt.argtypes = [P, L_, P, L_, P, I_, I_, I_, I_, P, L_, P, P, L_, I_, P]
dev, BF16 = "cuda", torch.bfloat16
for (M, N, K, epi, n) in ((23040, 14336, 3072, 1, 5), (23040, 14336, 3072, 0, 5), (16384, 3072, 768, 1, 20)):
    g = torch.Generator(device=dev).manual_seed(M + N)
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(BF16)
    w = ((torch.rand(N, K, device=dev, generator=g) * 2 - 1) * 0.05).to(BF16)
    b = (torch.rand(N, device=dev, generator=g) - 0.5).to(BF16)
    outs = {n_: torch.zeros(M, N, dtype=BF16, device=dev) for n_ in libs}
    st = torch.cuda.current_stream().cuda_stream
    def run(name):
        rc = libs[name].uv_gemm_bf16_nt(a.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), M, N, K, epi, outs[name].data_ptr(), N, None, None, 0, 0, st)
        assert rc == 0
    res = {n_: [] for n_ in libs}
    for r in range(8):
        for name in (list(libs) if r % 2 == 0 else list(libs)[::-1]):
            run(name); run(name)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(n): run(name)
            e.record(); torch.cuda.synchronize()
            res[name].append(s.elapsed_time(e) / n * 1e3)
    same = torch.equal(outs["tree"].view(torch.int16), outs["other"].view(torch.int16))
    print(f"M={M} N={N} K={K} epilogue {epi}: " + "  ".join(f"{n_}: {statistics.median(v):.1f} us" for n_, v in res.items()) + f"  bit-identical: {same}", flush=True)
