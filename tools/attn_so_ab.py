"""Same-process A/B of uv_flash_attn_bf16 from TWO builds of the library (developer tool): the tree's libunivid_hip.so against another
.so given on the command line (e.g. a copy made before a kernel change), interleaved rounds, random data, bit-identity of the outputs.
    python3 tools/attn_so_ab.py tools/diag/libunivid_hip_prev.so"""
import ctypes, math, os, statistics, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
torch.cuda.init()
TREE_LIB = os.path.join(ROOT, "univid_amd", "libunivid_hip.so")
# RTLD_DEEPBIND: the two libraries export the SAME symbols (kernel host stubs included); without it the second library's internal references
# resolve to the first one's globally visible definitions and both "builds" launch the same kernels (found the hard way in round 4)
_mode = os.RTLD_NOW | os.RTLD_LOCAL | os.RTLD_DEEPBIND
libs = {"tree": ctypes.CDLL(TREE_LIB, mode=_mode), "other": ctypes.CDLL(os.path.abspath(sys.argv[1]), mode=_mode)}
P, L_, I_ = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
for lib in libs.values():
    lib.uv_init()
    lib.uv_flash_attn_bf16.argtypes = [P, L_, P, L_, P, L_, P, L_, I_, I_, I_, I_, I_, ctypes.c_float, P]
dev, BF16, H, D = "cuda", torch.bfloat16, 24, 128
C = H * D
for (Lq, Lk, B, n) in ((11440, 11440, 2, 5), (11440, 512, 2, 20), (27280, 27280, 2, 2), (11440, 11440, 1, 5)):
    g = torch.Generator(device=dev).manual_seed(Lq + Lk)
    q = torch.randn(B * Lq, C, device=dev, generator=g).to(BF16)
    k = torch.randn(B * Lk, C, device=dev, generator=g).to(BF16)
    vt = torch.randn(C, (B - 1) * Lk + (Lk + 63) // 64 * 64, device=dev, generator=g).to(BF16)
    outs = {n_: torch.zeros(B * Lq, C, dtype=BF16, device=dev) for n_ in libs}
    st = torch.cuda.current_stream().cuda_stream

    def run(name):
        o = outs[name]
        rc = libs[name].uv_flash_attn_bf16(q.data_ptr(), C, k.data_ptr(), C, vt.data_ptr(), vt.stride(0), o.data_ptr(), C, B, Lq, Lk, H, D, 1 / math.sqrt(D), st)
        assert rc == 0
    res = {n_: [] for n_ in libs}
    for r in range(6):
        for name in (list(libs) if r % 2 == 0 else list(libs)[::-1]):      # alternate the order: the second of a pair measured 0.3-2.5 % faster
            run(name); run(name)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(n):
                run(name)
            e.record(); torch.cuda.synchronize()
            res[name].append(s.elapsed_time(e) / n)
    same = torch.equal(outs["tree"].view(torch.int16), outs["other"].view(torch.int16))
    fl = 4.0 * B * Lq * Lk * C
    print(f"Lq={Lq} Lk={Lk} B={B}: " + "  ".join(f"{n_}: {statistics.median(v):.4f} ms ({fl / statistics.median(v) / 1e9:.0f} TF/s)" for n_, v in res.items()) +
          f"  bit-identical: {same}", flush=True)
