"""ctypes binding of libunivid_hip.so (the C ABI declared in include/univid_hip.h).

PyTorch-ROCm tensors in, raw device pointers out: this module is the only place that turns a tensor into
`data_ptr()` + sizes + the current HIP stream. There is NO fallback: if the shared library is missing or a
call is rejected, a RuntimeError is raised.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libunivid_hip.so")

_c = ctypes
_P, _L, _I, _F = _c.c_void_p, _c.c_long, _c.c_int, _c.c_float

# name -> argtypes (all return int unless listed in _RESTYPE)
SIGNATURES = {
    "uv_version": [],
    "uv_init": [],
    "uv_last_error": [],
    "uv_build_id": [],
    "uv_device_arch": [_c.c_char_p, _I],
    "uv_host_blocking_sync": [_I],
    "uv_set_option": [_I, _I],
    "uv_get_option": [_I, _c.POINTER(_I)],
    "uv_reset_options": [],
    "uv_gemm_bf16_nt": [_P, _L, _P, _L, _P, _I, _I, _I, _I, _P, _L, _P, _P, _L, _I, _P],
    "uv_gemm_bf16_nt_ws": [_P, _L, _P, _L, _P, _I, _I, _I, _I, _P, _L, _P, _P, _L, _I, _P, _L, _P],
    "uv_gemm_splitk_ws_bytes": [_I, _I, _I],
    "uv_gemm_f16_nt": [_P, _L, _P, _L, _P, _I, _I, _I, _I, _P, _L, _P, _P, _L, _I, _P],
    "uv_gemm_f32_nt": [_P, _L, _P, _L, _P, _I, _I, _I, _P, _L, _P, _L, _P],
    "uv_gemm_bf16_nt_ssq": [_P, _L, _P, _L, _P, _I, _I, _I, _P, _L, _P, _L, _I, _P],
    "uv_rms_scale_from_ssq": [_P, _L, _I, _I, _I, _F, _P, _P],
    "uv_flash_attn_bf16_qnorm": [_P, _L, _P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _I, _F, _P, _P, _P],
    "uv_flash_attn_bf16": [_P, _L, _P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _I, _F, _P],
    "uv_flash_attn_f16": [_P, _L, _P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _I, _F, _P],
    "uv_flash_attn_kernel_name": [_I, _I, _L, _L, _I, _c.c_char_p, _I],
    "uv_transpose_16": [_P, _L, _P, _L, _I, _I, _I, _P],
    "uv_cast_f32_to16": [_P, _P, _L, _I, _P],
    "uv_cast_16_to_f32": [_P, _P, _L, _I, _P],
    "uv_layernorm_mod": [_P, _L, _P, _L, _I, _I, _F, _I, _P, _L, _I, _I, _P, _P, _P, _I, _I, _P],
    "uv_t5_attention_bf16": [_P, _L, _P, _L, _P, _L, _P, _L, _I, _I, _P, _I, _P],
    "uv_add_bf16": [_P, _P, _P, _L, _P],
    "uv_t5_gated_gelu_bf16": [_P, _P, _P, _L, _P],
    "uv_gelu_erf_bf16": [_P, _P, _L, _P],
    "uv_interp_linear_rows_bf16": [_P, _L, _P, _L, _I, _I, _I, _P],
    "uv_l2_normalize_rows_f32": [_P, _L, _P, _L, _I, _I, _F, _P],
    "uv_rmsnorm_rope": [_P, _L, _P, _L, _P, _I, _I, _I, _F, _P, _I, _I, _I, _I, _P],
    "uv_rmsnorm_rope_qk": [_P, _P, _P, _P, _P, _P, _L, _L, _I, _I, _I, _I, _F, _P, _I, _I, _I, _I, _P],
    "uv_patchify_bf16": [_P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "uv_unpatchify_f32": [_P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "uv_sinusoid_f32": [_P, _P, _I, _I, _P],
    "uv_linear_rows_f32": [_P, _L, _P, _P, _P, _L, _I, _I, _I, _I, _P],
    "uv_add_rows_f32": [_P, _P, _P, _I, _L, _P],
    "uv_cast_f32_bf16": [_P, _P, _L, _P],
    "uv_add_bf16_resid": [_P, _L, _P, _L, _I, _I, _P],
    "uv_text_weight_rows_bf16": [_P, _L, _P, _L, _I, _I, _I, _F, _P],
    "uv_cfg_convert": [_P, _P, _P, _F, _F, _P, _P, _L, _P],
    "uv_unipc_corrector": [_P, _P, _P, _P, _P, _F, _F, _F, _F, _F, _F, _I, _L, _P],
    "uv_unipc_predictor": [_P, _P, _P, _P, _F, _F, _F, _F, _I, _L, _P],
    "uv_dpmpp_update": [_P, _P, _P, _P, _F, _F, _F, _I, _L, _P],
    "uv_conv3d_f32": [_P, _L, _I, _I, _I, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I,
                      _P, _L, _P],
    "uv_conv3d_bf16x6": [_P, _L, _I, _I, _I, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I,
                         _P, _L, _P],
    "uv_conv3d_bf16x3": [_P, _L, _I, _I, _I, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I,
                         _P, _L, _I, _P],
    "uv_conv3d_f16x3": [_P, _L, _I, _I, _I, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I,
                        _P, _L, _F, _P, _P],
    "uv_vae_split_f16": [_P, _L, _P, _L, _L, _I, _P, _P],
    "uv_split_weights_f16x3": [_P, _P, _L, _F, _P],
    "uv_split_weights_bf16x3": [_P, _P, _L, _P],
    "uv_split_weights_bf16x6": [_P, _P, _L, _P],
    "uv_vae_rms_silu": [_P, _L, _P, _P, _L, _L, _I, _I, _I, _P],
    "uv_softmax_rows_f32": [_P, _L, _I, _I, _F, _P],
    "uv_vae_dupup_add": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "uv_vae_avgdown_add": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "uv_vae_latent_in": [_P, _P, _P, _P, _L, _I, _L, _P],
    "uv_vae_latent_out": [_P, _L, _P, _P, _P, _I, _L, _P],
    "uv_vae_video_in": [_P, _P, _L, _I, _I, _I, _I, _I, _P],
    "uv_vae_video_out": [_P, _L, _P, _I, _I, _I, _I, _I, _P],
}
_RESTYPE = {"uv_last_error": _c.c_char_p, "uv_build_id": _c.c_char_p, "uv_gemm_splitk_ws_bytes": _L}

OPT_CONV_HALO, OPT_GEMM_GM, OPT_ATTN_CUT = range(3)      # include/univid_hip.h: UV_OPT_*
EPI_BF16, EPI_GELU_BF16, EPI_F32_FROM_BF16, EPI_RESID_F32, EPI_GATE_RESID_F32, EPI_BF16_T = range(6)

_lib = None
_inited = set()        # device indices whose arch check + per-device library state (zero page) are done


class UnividHipError(RuntimeError):
    pass


def load(path=None):
    """Loads the shared library (torch first, so the HIP runtime torch ships is the one both sides use)."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise UnividHipError(
            f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `python -m univid_amd.build`). univid_amd has no CPU/eager fallback.")
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.argtypes = args
        fn.restype = _RESTYPE.get(name, _I)
    _check_build_id(lib, path)
    _lib = lib
    return lib


def _check_build_id(lib, path):
    """A stale .so next to newer kernel sources (git pull onto old local objects) would run old kernels behind new ctypes
    signatures: compare the digest compiled into the library with the one of the sources in the tree."""
    if os.path.abspath(path) != LIB_PATH or not os.path.isdir(os.path.join(_HERE, "csrc")):
        return
    from . import build as _build
    have, want = lib.uv_build_id().decode(), _build.source_id()
    if have != want:
        raise UnividHipError(
            f"{path} was built from other kernel sources (library {have}, tree {want}): rebuild with "
            "`python -m univid_amd.build`. univid_amd never runs a stale extension.")


def init(device=None):
    """One-time checks for `device` (index; default = the current one): a HIP device is visible, it is a gfx950, and the
    library's per-device state exists. Called by every entry point with the device of its tensors."""
    lib = load()
    if device is None:
        if not torch.cuda.is_available():
            raise UnividHipError("no HIP device visible: univid_amd's hot path only runs on an MI355X (gfx950)")
        device = torch.cuda.current_device()
    if device not in _inited:
        if not torch.cuda.is_available():
            raise UnividHipError("no HIP device visible: univid_amd's hot path only runs on an MI355X (gfx950)")
        torch.cuda.init()
        with torch.cuda.device(device):
            rc = lib.uv_init()
            if rc != 0:
                raise UnividHipError(f"uv_init failed: {lib.uv_last_error().decode()}")
            buf = ctypes.create_string_buffer(64)
            lib.uv_device_arch(buf, 64)
        arch = buf.value.decode()
        if not arch.startswith("gfx950"):
            raise UnividHipError(f"device {device}: arch {arch!r} is not gfx950: the kernels are built for MI355X only")
        _inited.add(device)
    return lib


def set_option(key, value):
    """uv_set_option: explicit developer switch (A/B tools, tests); the library never reads the environment."""
    lib = load()
    if lib.uv_set_option(int(key), int(value)) != 0:
        raise UnividHipError(lib.uv_last_error().decode())


def get_option(key):
    lib = load()
    v = _I(0)
    if lib.uv_get_option(int(key), ctypes.byref(v)) != 0:
        raise UnividHipError(lib.uv_last_error().decode())
    return v.value


def reset_options():
    load().uv_reset_options()


class _DevPtr(_c.c_void_p):
    """A device pointer that remembers which GPU it lives on, so `call` can make that GPU current for the launch."""
    dev = None


class _Stream:
    """Placeholder for 'the current HIP stream of the device the tensors of this call live on'; resolved in `call`."""


_STREAM = _Stream()


def stream_ptr():
    return _STREAM


def ptr(t):
    if t is None:
        return None
    p = _DevPtr(t.data_ptr())
    p.dev = t.device.index if t.device.type == "cuda" else -1
    return p


# Optional live kernel timing (bench.py): PROFILE = {entry_point: []} times those entry points with HIP events
# recorded on the launch stream; PROFILE_ALL times every entry point. Off (None) in normal use.
PROFILE = None
PROFILE_ALL = False
CALL_COUNT = 0          # entry-point launches since import (bench.py reports launches per step)


def call(name, *args, flops=0):
    """Launches one entry point on the device its pointer arguments live on (all on ONE device, else an error) and on that
    device's current stream: `WanTI2V(device_id=1)` / the reference's manual model placement work without the process ever
    calling torch.cuda.set_device."""
    dev = None
    for a in args:
        if type(a) is _DevPtr:
            if a.dev != dev:
                if dev is not None or a.dev < 0:
                    raise UnividHipError(f"{name}: tensors must live on ONE GPU (got devices {dev} and {a.dev}; -1 = host memory)")
                dev = a.dev
    if dev is None:
        dev = torch.cuda.current_device() if torch.cuda.is_available() else None
    global CALL_COUNT
    lib = init(dev)
    if dev != torch.cuda.current_device():
        with torch.cuda.device(dev):
            return call(name, *args, flops=flops)
    stream = torch.cuda.current_stream(dev)
    args = tuple(_c.c_void_p(stream.cuda_stream) if a is _STREAM else a for a in args)
    prof = PROFILE
    timed = prof is not None and (PROFILE_ALL or name in prof)
    if timed:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(stream)
    CALL_COUNT += 1
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise UnividHipError(f"{name} failed ({rc}): {lib.uv_last_error().decode()}")
    if timed:
        e.record(stream)
        prof.setdefault(name, []).append((s, e, flops))


# ---- thin typed wrappers (tensor checks live here so the C side only sees valid pointers) -----------------------

def _chk(t, dtype, name):
    if t.device.type != "cuda":
        raise UnividHipError(f"{name}: tensor must live on the GPU (got {t.device})")
    if t.dtype != dtype:
        raise UnividHipError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.stride(-1) != 1:
        raise UnividHipError(f"{name}: innermost dimension must be contiguous")


def host_blocking_sync(on=True, device=None):
    """uv_host_blocking_sync for `device` (default: the current one): a host thread waiting in synchronize sleeps instead of spinning.
    Process-wide policy: the APPLICATION decides (bench.py's ranks and univid_amd.parallel's workers call it; a library import never does)."""
    dev = torch.device("cuda" if device is None else device)
    idx = torch.cuda.current_device() if dev.index is None else dev.index
    lib = load()
    with torch.cuda.device(idx):
        if lib.uv_host_blocking_sync(1 if on else 0) != 0:
            raise UnividHipError(lib.uv_last_error().decode())


def gemm_splitk_ws_bytes(M, N, K, device=None):
    """Bytes of workspace with which gemm_bf16(..., ws=) runs the leftover-row strip of an [M, K] x [N, K]^T projection as one round of
    split-K workgroups (0: this shape has no such strip). The answer depends on the device's CU count."""
    dev = torch.device("cuda" if device is None else device)
    idx = torch.cuda.current_device() if dev.index is None else dev.index
    lib = init(idx)
    with torch.cuda.device(idx):
        return int(lib.uv_gemm_splitk_ws_bytes(int(M), int(N), int(K)))


def gemm_bf16(a, w, bias, out, epi, M=None, gate=None, gate_tid=None, tile_cfg=0, ws=None):
    """a [M,K] bf16, w [N,K] bf16, bias bf16 [N] | None; out per epilogue (see include/univid_hip.h).
    ws: uint8 scratch tensor of >= gemm_splitk_ws_bytes(M, N, K) bytes (uv_gemm_bf16_nt_ws), or None."""
    f16 = a.dtype == torch.float16      # IEEE fp16 operands (SigLIP2 ranker): same kernels, fp16 MFMA / conversions
    _chk(a, torch.float16 if f16 else torch.bfloat16, "gemm_bf16.a")
    _chk(w, a.dtype, "gemm_bf16.w")
    M = a.shape[0] if M is None else M
    N, K = w.shape
    if ws is not None and not f16:
        _chk(ws, torch.uint8, "gemm_bf16.ws")
        call("uv_gemm_bf16_nt_ws", ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), M, N, K, epi, ptr(out), out.stride(0),
             ptr(gate), ptr(gate_tid), 0 if gate is None else gate.stride(0), tile_cfg, ptr(ws), ws.numel(), stream_ptr(), flops=2 * M * N * K)
        return out
    call("uv_gemm_f16_nt" if f16 else "uv_gemm_bf16_nt", ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), M, N, K, epi, ptr(out), out.stride(0),
         ptr(gate), ptr(gate_tid), 0 if gate is None else gate.stride(0), tile_cfg, stream_ptr(), flops=2 * M * N * K)
    return out


def gemm_bf16_ssq(a, w, bias, out, ssq, M=None, tile_cfg=0):
    """out = bf16(a w^T + bias) and ssq[m][g] = the output row's sum of squares over columns [32 g, 32 g + 32) (f32 [M, N / 32])."""
    _chk(a, torch.bfloat16, "gemm_bf16_ssq.a")
    _chk(w, torch.bfloat16, "gemm_bf16_ssq.w")
    _chk(ssq, torch.float32, "gemm_bf16_ssq.ssq")
    M = a.shape[0] if M is None else M
    N, K = w.shape
    call("uv_gemm_bf16_nt_ssq", ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), M, N, K, ptr(out), out.stride(0), ptr(ssq), ssq.stride(0),
         tile_cfg, stream_ptr(), flops=2 * M * N * K)
    return out


def rms_scale_from_ssq(ssq, rs, M, C, eps):
    """rs[m] = 1 / sqrt(sum(ssq[m]) / C + eps): the RMSNorm scale of row m from its per-group sums of squares."""
    _chk(ssq, torch.float32, "rms_scale_from_ssq.ssq")
    _chk(rs, torch.float32, "rms_scale_from_ssq.rs")
    call("uv_rms_scale_from_ssq", ptr(ssq), ssq.stride(0), M, ssq.shape[1], C, float(eps), ptr(rs), stream_ptr())
    return rs


def gemm_f32(a, w, bias, out, resid=None, M=None):
    _chk(a, torch.float32, "gemm_f32.a")
    _chk(w, torch.float32, "gemm_f32.w")
    M = a.shape[0] if M is None else M
    N, K = w.shape
    call("uv_gemm_f32_nt", ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), M, N, K, ptr(out), out.stride(0),
         ptr(resid), 0 if resid is None else resid.stride(0), stream_ptr())
    return out


def flash_attn(q, k, vt, out, Lq, Lk, H, D, scale, batch=1, q_rs=None, q_weight=None):
    """q [batch*Lq, C], k [batch*Lk, C], vt [C, >= (batch-1)*Lk + roundup(Lk, 64)] (sample b = columns b*Lk..), out [batch*Lq, C].
    q_rs / q_weight: q is the RAW projection and the kernel's Q prologue applies norm_q (per-row scale f32 [batch*Lq], weight f32 [C])."""
    f16 = q.dtype == torch.float16
    for t, n in ((q, "q"), (k, "k"), (vt, "vt"), (out, "out")):
        _chk(t, q.dtype if f16 else torch.bfloat16, "flash_attn." + n)
    if q_rs is not None:
        _chk(q_rs, torch.float32, "flash_attn.q_rs")
        _chk(q_weight, torch.float32, "flash_attn.q_weight")
        call("uv_flash_attn_bf16_qnorm", ptr(q), q.stride(0), ptr(k), k.stride(0), ptr(vt), vt.stride(0), ptr(out), out.stride(0),
             batch, Lq, Lk, H, D, float(scale), ptr(q_rs), ptr(q_weight), stream_ptr(), flops=4 * batch * Lq * Lk * H * D)
        return out
    call("uv_flash_attn_f16" if f16 else "uv_flash_attn_bf16", ptr(q), q.stride(0), ptr(k), k.stride(0), ptr(vt), vt.stride(0), ptr(out), out.stride(0),
         batch, Lq, Lk, H, D, float(scale), stream_ptr(), flops=4 * batch * Lq * Lk * H * D)
    return out


def attn_kernel_name(Lq, Lk, D, batch=1, H=None, f16=False):
    """Name of the kernel `flash_attn` dispatches for this geometry (dense [tokens, H*D] rows as the DiT uses them)."""
    H = H or 1
    buf = ctypes.create_string_buffer(96)
    ldvt = (batch - 1) * Lk + (Lk + 63) // 64 * 64
    rc = load().uv_flash_attn_kernel_name(Lk, D, H * D, ldvt, int(f16), buf, 96)
    if rc != 0:
        raise UnividHipError(load().uv_last_error().decode())
    return buf.value.decode()


def layernorm_mod(x, out, L, C, eps, mode=0, tab=None, shift_off=0, scale_off=0, tid=None, w=None, b=None,
                  round_ln=False):
    _chk(x, torch.float32, "layernorm_mod.x")
    call("uv_layernorm_mod", ptr(x), x.stride(0), ptr(out), out.stride(0), L, C, float(eps), mode, ptr(tab),
         0 if tab is None else tab.stride(0), shift_off, scale_off, ptr(tid), ptr(w), ptr(b), int(round_ln),
         2 if out.dtype == torch.float16 else int(out.dtype == torch.bfloat16), stream_ptr())
    return out


def rmsnorm_rope(x, out, weight, L, C, D, eps, freqs=None, grid=(0, 0, 0), row0=0):
    _chk(x, torch.bfloat16, "rmsnorm_rope.x")
    _chk(out, torch.bfloat16, "rmsnorm_rope.out")
    call("uv_rmsnorm_rope", ptr(x), x.stride(0), ptr(out), out.stride(0), ptr(weight), L, C, D, float(eps), ptr(freqs),
         int(grid[0]), int(grid[1]), int(grid[2]), int(row0), stream_ptr())
    return out


def text_weight_rows(x, out, n_scaled, w):
    """out[r] = bf16(x[r] * bf16(w)) for the first n_scaled rows of one sample's embedded context [R, C] bf16, the other rows copied:
    UniVid's dynamic text weight (model_pipeline.py:1787-1797)."""
    _chk(x, torch.bfloat16, "text_weight_rows.x")
    _chk(out, torch.bfloat16, "text_weight_rows.out")
    R, C = x.shape
    call("uv_text_weight_rows_bf16", ptr(x), x.stride(0), ptr(out), out.stride(0), R, int(n_scaled), C, float(w), stream_ptr())
    return out


def rmsnorm_rope_qk(q, k, q_weight, k_weight, L, Ls, C, D, eps, freqs, grid, row0=0):
    """In-place QK RMSNorm + RoPE of q and k (same strides) in one launch; rows = L / Ls stacked samples of Ls tokens."""
    _chk(q, torch.bfloat16, "rmsnorm_rope_qk.q")
    _chk(k, torch.bfloat16, "rmsnorm_rope_qk.k")
    if q.stride(0) != k.stride(0):
        raise UnividHipError("rmsnorm_rope_qk: q and k must have the same row stride")
    call("uv_rmsnorm_rope_qk", ptr(q), ptr(q), ptr(q_weight), ptr(k), ptr(k), ptr(k_weight), q.stride(0), q.stride(0), L, Ls, C, D,
         float(eps), ptr(freqs), int(grid[0]), int(grid[1]), int(grid[2]), int(row0), stream_ptr())
