"""Builds tools/diag/libuv_diag.so: diagnostic kernels that are NOT part of the product library (developer tools and cross-check tests).
Today: the 4-wave x 64-query self-attention kernel attn_pw4.hip (C entry uv_diag_flash_attn_pw4, bit-identical to the product's
flash_attn_fwd12_kernel). In-tree like the product's .so, so it travels to the GPU box with the snapshot.   python tools/diag/build_diag.py"""
import hashlib, os, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
LIB = os.path.join(HERE, "libuv_diag.so")


def build(verbose=True):
    from univid_amd import build as b
    src = os.path.join(HERE, "attn_pw4.hip")
    h = hashlib.sha256()
    for p in [src, *b.headers()]:
        h.update(open(p, "rb").read())
    h.update(" ".join(b.FLAGS + b.DIAG_PW4_FLAGS).encode())
    dig, stamp = h.hexdigest(), LIB + ".sha"
    if not (os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig):
        cmd = [b._hipcc(), *b.FLAGS, *b.DIAG_PW4_FLAGS, "-shared", "-I", b.CSRC, src, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-4000:]}")
        open(stamp, "w").write(dig)
    if verbose:
        print(f"[tools/diag] {LIB} ({os.path.getsize(LIB) >> 10} KiB)")
    return LIB


if __name__ == "__main__":
    build()
