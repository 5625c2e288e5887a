"""Same-process A/B of uv_gemm_bf16_nt from TWO builds of the library (developer tool, the GEMM twin of tools/attn_so_ab.py): the tree's
libunivid_hip.so against another .so given on the command line (e.g. a copy made before a kernel change); interleaved rounds, random data,
bit-identity of the outputs.
    python3 tools/gemm_so_ab.py tools/diag/libunivid_hip_prev.so [shape indices of tools/gemm_bench.py, default 2,4,11,0,3]"""
import ctypes, os, statistics, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
torch.cuda.init()
from univid_amd._lib import EPI_GATE_RESID_F32, EPI_RESID_F32, EPI_F32_FROM_BF16, EPI_BF16_T   # noqa: E402
import importlib.util                                                                                # noqa: E402
_spec = importlib.util.spec_from_file_location("gemm_bench", os.path.join(ROOT, "tools", "gemm_bench.py"))
_gb = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_gb)
TREE_LIB = os.path.join(ROOT, "univid_amd", "libunivid_hip.so")
_mode = os.RTLD_NOW | os.RTLD_LOCAL | os.RTLD_DEEPBIND      # (see tools/attn_so_ab.py)
libs = {"tree": ctypes.CDLL(TREE_LIB, mode=_mode), "other": ctypes.CDLL(os.path.abspath(sys.argv[1]), mode=_mode)}
P, L_, I_ = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
for lib in libs.values():
    lib.uv_init()
    lib.uv_gemm_bf16_nt.argtypes = [P, L_, P, L_, P, I_, I_, I_, I_, P, L_, P, P, L_, I_, P]
dev = "cuda"
sel = [int(s) for s in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 4, 11, 0, 3]
for si in sel:
    name, M, N, K, epi = _gb.SHAPES[si]
    g = torch.Generator(device=dev).manual_seed(si)
    A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    W = ((torch.rand(N, K, device=dev, generator=g) * 2 - 1) * 0.05).to(torch.bfloat16)
    bias = (torch.rand(N, device=dev, generator=g) - 0.5).to(torch.bfloat16)
    gate = gate_tid = None
    f32out = epi in (EPI_F32_FROM_BF16, EPI_RESID_F32, EPI_GATE_RESID_F32)
    if epi == EPI_GATE_RESID_F32:
        gate = torch.rand(2, N, device=dev, generator=g)
        gate_tid = (torch.arange(M, device=dev) * 2 // M).to(torch.int32)
    x0 = torch.rand(M, N, device=dev, generator=g) if f32out else None
    def fresh():
        if f32out:
            return x0.clone()
        if epi == EPI_BF16_T:
            return torch.zeros(N, (M + 63) // 64 * 64, device=dev, dtype=torch.bfloat16)
        return torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    outs = {n_: fresh() for n_ in libs}
    st = torch.cuda.current_stream().cuda_stream

    def run(nm, o):
        rc = libs[nm].uv_gemm_bf16_nt(A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), M, N, K, epi, o.data_ptr(), o.stride(0),
                                      None if gate is None else gate.data_ptr(), None if gate_tid is None else gate_tid.data_ptr(),
                                      0 if gate is None else gate.stride(0), 0, st)
        assert rc == 0
    for nm in libs:          # bit-identity on ONE application (the read-modify-write epilogues accumulate)
        run(nm, outs[nm])
    torch.cuda.synchronize()
    same = torch.equal(outs["tree"], outs["other"])
    scratch = fresh()
    res = {n_: [] for n_ in libs}
    n = 8
    for r in range(8):
        for nm in (list(libs) if r % 2 == 0 else list(libs)[::-1]):
            run(nm, scratch); run(nm, scratch)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(n):
                run(nm, scratch)
            e.record(); torch.cuda.synchronize()
            res[nm].append(s.elapsed_time(e) / n * 1e3)
    fl = 2.0 * M * N * K
    print(f"{name:32s} {M}x{N}x{K}: " + "  ".join(f"{n_}: {statistics.median(v):8.1f} us (min {min(v):8.1f}, {fl / statistics.median(v) / 1e6:.0f} TF/s)" for n_, v in res.items()) +
          f"  bit-identical: {same}", flush=True)
