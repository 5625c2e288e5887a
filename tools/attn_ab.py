"""A/B of attention kernel variants in ONE process (developer tool): runs tools/attn_bench-style timing for each UV_ATTN_* setting
in child processes and checks that the outputs are bit-identical to the default kernel's."""
import math, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child():
    from univid_amd import _lib
    _lib.init()
    dev = "cuda"; BF16 = torch.bfloat16
    L, H, D = int(os.environ.get("L", 11440)), 24, 128
    C = H * D
    B = int(os.environ.get("B", 2))
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn(B * L, C, device=dev, generator=g).to(BF16); k = torch.randn(B * L, C, device=dev, generator=g).to(BF16)
    vt = torch.randn(C, (B * L + 63) // 64 * 64, device=dev, generator=g).to(BF16)
    out = torch.empty(B * L, C, dtype=BF16, device=dev)
    n = int(os.environ.get("N", 10))
    for _ in range(3):
        _lib.flash_attn(q, k, vt, out, L, L, H, D, 1 / math.sqrt(D), batch=B)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        _lib.flash_attn(q, k, vt, out, L, L, H, D, 1 / math.sqrt(D), batch=B)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    ref_path = "/tmp/attn_ab_default.pt"
    if os.environ.get("TAG", "default") == "default":
        torch.save(out.cpu(), ref_path)
        cmp = ""
    elif os.path.exists(ref_path):
        ref = torch.load(ref_path).float()
        d = (out.cpu().float() - ref).abs()
        cmp = f"  vs default: max {float(d.max()):.3e} rel rms {float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.2e} exact {float((d == 0).float().mean()):.4f}"
    import hashlib
    hsh = hashlib.sha1(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12]
    print(f"{os.environ.get('TAG', 'default'):>12}: L={L} B={B}: {ms:.3f} ms {B*4*L*L*C/ms/1e9:.1f} TFLOP/s  sha1 {hsh}  finite {bool(torch.isfinite(out.float()).all())}{cmp}")

if __name__ == "__main__":
    if os.environ.get("ATTN_AB_CHILD"):
        child()
    else:
        variants = [("default", {"UV_ATTN_W3": "0"})] + [(f"W3={v}", {"UV_ATTN_W3": v}) for v in (sys.argv[1:] or ["1"])]
        for tag, env in variants + variants:
            subprocess.run([sys.executable, __file__], env=dict(os.environ, ATTN_AB_CHILD="1", TAG=tag, **env))
