"""Builds libunivid_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

One translation unit per .hip file, compiled in parallel, linked into a single shared library next to this
package so that it travels with the source tree (gpurun snapshot) and shows up as an in-tree native module.
"""
import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libunivid_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-value", "-DNDEBUG",
         # no implicit FMA contraction: a*b+c keeps the two roundings PyTorch eager has (explicit fmaf/MFMA are unaffected)
         "-ffp-contract=off"]


# per-file additions. The diagnostic kernel tools/diag/attn_pw4.hip is built by
# tools/diag/build_diag.py with DIAG_PW4_FLAGS: its slots are hand-placed scalar f32 operations beside MFMAs; SLP-packing them into
# v_pk_*_f32 costs issue cycles there (MI355X_MICROARCH.md, "price of one filler beside MFMAs").
# attention.hip: -fno-honor-nans. fmaxf on values hipcc cannot prove canonical (MFMA outputs, v_permlane results, loop-carried maxima)
# is lowered as v_max_f32 x, x, x (quieting a possible signalling NaN) in front of the real maximum: 6 of the 16 maximum instructions per
# 32-key half of the softmax were such canonicalisations (24 of 64 VALU maxima per two key tiles; MI355X_MICROARCH.md, "inserts
# canonicalising v_max before fmaxf on MFMA outputs"). Without NaN honouring they are gone: bit-identical outputs on NaN-free data
# (identity on every non-NaN value; +-inf is still honoured - the masked tile's -inf scores rely on it), -1.0 ... -1.8 % on the self-
# attention launch in a two-library same-process A/B (tools/attn_so_ab.py). A NaN in q / k still poisons its rows through exp2 and the row sum.
FILE_FLAGS = {"attention.hip": ["-fno-honor-nans"]}
DIAG_PW4_FLAGS = ["-fno-slp-vectorize", "-Wno-inline-asm"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found; univid_amd needs the ROCm toolchain to build its HIP kernels")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))


_TOOL = None


def _toolchain_id():
    """hipcc's version banner: an object built by another compiler is stale even if the source did not change."""
    global _TOOL
    if _TOOL is None:
        try:
            _TOOL = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True).stdout.strip()
        except Exception:
            _TOOL = "unknown"
    return _TOOL


def _digest(path, extra=""):
    h = hashlib.sha256()
    for p in [path, *headers()]:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    h.update(_toolchain_id().encode())
    h.update(extra.encode())
    return h.hexdigest()


def source_id():
    """Digest of every kernel source + header + flags (NOT the toolchain: the GPU box may not have the same hipcc banner and
    only checks that the .so it received belongs to the sources it received). Compiled into the library as uv_build_id();
    univid_amd._lib.load() compares the two, so a stale .so next to newer sources fails loudly instead of running old kernels
    behind new ctypes signatures."""
    h = hashlib.sha256()
    for p in sources() + headers():
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(FILE_FLAGS.items())).encode())
    return h.hexdigest()[:24]


def _compile(src):
    obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
    stamp = obj + ".sha"        # lives next to the object it describes; both are git-ignored
    extra = list(FILE_FLAGS.get(os.path.basename(src), []))
    if os.path.basename(src) == "capi.hip":
        extra = [f'-DUV_BUILD_ID="{source_id()}"']
    dig = _digest(src, " ".join(extra))
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj
    cmd = [_hipcc(), *FLAGS, *extra, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-4000:]}")
    with open(stamp, "w") as f:
        f.write(dig)
    return obj


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sources()
    if force:
        for f in os.listdir(OBJDIR):
            os.remove(os.path.join(OBJDIR, f))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    newest = max(os.path.getmtime(o) for o in objs)
    linked = LIB + ".objs"      # digest list of the objects the library was linked from
    want = "\n".join(open(o + ".sha").read() for o in objs)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < newest or not os.path.exists(linked) or open(linked).read() != want:
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        with open(linked, "w") as f:
            f.write(want)
    # a kernel whose host stub was silently dropped shows up as an undefined symbol only at dlopen time
    r = subprocess.run([sys.executable, "-c", f"import ctypes; ctypes.CDLL({LIB!r})"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{LIB} does not load:\n{r.stderr[-2000:]}")
    if verbose:
        print(f"[univid_amd.build] {LIB} ({os.path.getsize(LIB) >> 10} KiB, {len(objs)} objects)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
