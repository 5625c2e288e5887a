"""`flash_attention` / `attention`: the reference's attention operator seam on the MI355X kernels.

Mirrors /root/reference/models/wan/utils/modules/attention.py (flash_attention :24-130, attention :133-179): the same
signature, argument meaning and result dtype, so its callers (model.py:145-150 self-attention with `k_lens=seq_lens`,
model.py:175 cross-attention, distributed/ulysses.py:37) can import this module instead of the flash-attn wheel:

    q [B, Lq, N, C], k [B, Lk, N, C], v [B, Lk, N, C] in any float dtype -> [B, Lq, N, C] in q's dtype;
    operands that are not fp16/bf16 are cast to `dtype` first (:59-83); sample b attends keys [0, k_lens[b]).

The core is `uv_flash_attn_bf16` / `uv_flash_attn_f16` (csrc/attention.hip); this file only arranges memory for it: casts
(`uv_cast_f32_to16`), the V^T operand (`uv_transpose_16`, zero-filled to the 64-key tile), the widening of the result
(`uv_cast_16_to_f32`). The fused DiT path (`univid_amd/wan/model.py`) does not come through here - there V^T is written by
the V projection's GEMM epilogue - so this is the path for code that calls the operator directly.

What the reference function accepts but UniVid never uses is rejected loudly instead of being computed wrongly:
causal / sliding-window masks, dropout, `q_scale`, grouped K/V heads, head sizes other than 64 / 128, and `q_lens` that
differ from Lq (the reference itself cannot return those: its `.unflatten(0, (b, lq))` needs every q_len == Lq).
There is no eager fallback: without the HIP extension or a gfx950 device the call raises.
"""
import torch

from .. import _lib

__all__ = ["flash_attention", "attention"]

_HALF = (torch.float16, torch.bfloat16)


def _half(x, dtype):
    """attention.py:59-60: keep fp16 / bf16 tensors, cast everything else to `dtype` (round to nearest even)."""
    if x.dtype in _HALF:
        return x.contiguous()
    if x.dtype != torch.float32:
        x = x.float()
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    if x.numel():
        _lib.call("uv_cast_f32_to16", _lib.ptr(x), _lib.ptr(out), x.numel(), int(dtype == torch.float16), _lib.stream_ptr())
    return out


def _lens(lens, b, full, name):
    if lens is None:
        return [full] * b
    vals = [int(v) for v in (lens.tolist() if torch.is_tensor(lens) else lens)]
    if len(vals) != b or min(vals) < 1 or max(vals) > full:
        raise ValueError(f"flash_attention: {name} must hold {b} lengths in [1, {full}], got {vals}")
    return vals


def flash_attention(q, k, v, q_lens=None, k_lens=None, dropout_p=0., softmax_scale=None, q_scale=None, causal=False,
                    window_size=(-1, -1), deterministic=False, dtype=torch.bfloat16, version=None):
    """Reference signature (attention.py:24-38). `deterministic` and `version` select nothing here: the kernel is always
    deterministic (no atomics, fixed reduction order) and there is one implementation."""
    assert dtype in _HALF
    if q.device.type != "cuda":
        raise _lib.UnividHipError(f"flash_attention: tensors must live on the GPU (got {q.device}); univid_amd has no CPU path")
    if causal or tuple(window_size) != (-1, -1):
        raise NotImplementedError("flash_attention: causal / sliding-window attention is not built (UniVid's DiT uses global "
                                  "attention, window_size=(-1, -1): configs/wan_ti2v_5B.py)")
    if dropout_p:
        raise NotImplementedError("flash_attention: dropout_p > 0 is a training option; the inference kernels have no dropout")
    if q_scale is not None:
        raise NotImplementedError("flash_attention: q_scale is never passed on UniVid's path; fold it into softmax_scale")
    if q.dim() != 4 or k.dim() != 4 or v.dim() != 4:
        raise ValueError("flash_attention: q, k, v must be [B, L, N, C]")
    b, lq, n, c = q.shape
    lk = k.size(1)
    if k.shape != (b, lk, n, c) or v.shape != (b, lk, n, c):
        raise NotImplementedError(f"flash_attention: k / v must be [B, Lk, {n}, {c}] like q's heads (grouped K/V heads and "
                                  f"C2 != C1 are not built); got k {tuple(k.shape)}, v {tuple(v.shape)}")
    if c not in (64, 128):
        raise NotImplementedError(f"flash_attention: head size {c} (the kernels are built for 64 and 128)")
    out_dtype = q.dtype
    if any(ql != lq for ql in _lens(q_lens, b, lq, "q_lens")):
        raise ValueError("flash_attention: q_lens other than Lq cannot be returned as [B, Lq, N, C] (attention.py:107,127)")
    kls = _lens(k_lens, b, lk, "k_lens")

    qh, kh, vh = _half(q, dtype), _half(k, dtype), _half(v, dtype)
    if not (qh.dtype == kh.dtype == vh.dtype):          # attention.py:82-83 casts q and k to v's dtype
        raise NotImplementedError("flash_attention: q, k, v of different half dtypes; pass one dtype")
    f16 = int(vh.dtype == torch.float16)
    C = n * c
    dev = q.device
    scale = float(softmax_scale) if softmax_scale is not None else c ** -0.5
    out = torch.empty(b, lq, C, dtype=vh.dtype, device=dev)
    q2, k2, v2 = qh.view(b * lq, C), kh.view(b * lk, C), vh.view(b * lk, C)

    def vt_of(rows, L, vt, col0):
        """vt[:, col0 : col0 + roundup(L, 64)] = rows[:L]^T, zero beyond L."""
        pad = (L + 63) // 64 * 64
        dst = vt[:, col0:]
        _lib.call("uv_transpose_16", _lib.ptr(rows), rows.stride(0), _lib.ptr(dst), vt.stride(0), L, C, pad, _lib.stream_ptr())

    if len(set(kls)) == 1 and kls[0] == lk and (b == 1 or lk % 8 == 0):
        # every sample attends all Lk keys: ONE launch over the stacked samples (sample s = V^T columns [s*Lk, (s+1)*Lk))
        cols = (b - 1) * lk + (lk + 63) // 64 * 64
        vt = torch.empty(C, cols, dtype=vh.dtype, device=dev)
        for s in range(b):      # later samples overwrite the previous sample's zero padding, the last one keeps it
            vt_of(v2[s * lk:(s + 1) * lk], lk, vt, s * lk)
        _lib.flash_attn(q2, k2, vt, out.view(b * lq, C), lq, lk, n, c, scale, batch=b)
    else:
        for s in range(b):      # ragged key lengths: one launch per sample on its first k_lens[s] keys
            L = kls[s]
            vt = torch.empty(C, (L + 63) // 64 * 64, dtype=vh.dtype, device=dev)
            vt_of(v2[s * lk:s * lk + L], L, vt, 0)
            _lib.flash_attn(q2[s * lq:(s + 1) * lq], k2[s * lk:s * lk + L], vt, out[s], lq, L, n, c, scale)
    out = out.view(b, lq, n, c)
    if out_dtype == out.dtype:
        return out
    if out_dtype == torch.float32:                                            # attention.py:130
        wide = torch.empty(b, lq, n, c, dtype=torch.float32, device=dev)
        _lib.call("uv_cast_16_to_f32", _lib.ptr(out), _lib.ptr(wide), out.numel(), f16, _lib.stream_ptr())
        return wide
    return out.type(out_dtype)   # fp64 / other-half callers: a dtype view change of the finished result, no arithmetic


def attention(q, k, v, q_lens=None, k_lens=None, dropout_p=0., softmax_scale=None, q_scale=None, causal=False,
              window_size=(-1, -1), deterministic=False, dtype=torch.bfloat16, fa_version=None):
    """attention.py:133-179. The reference falls back to `scaled_dot_product_attention` WITHOUT the padding mask when no
    flash-attn wheel is installed; here the kernel is always present, so this is `flash_attention` (mask included)."""
    return flash_attention(q=q, k=k, v=v, q_lens=q_lens, k_lens=k_lens, dropout_p=dropout_p, softmax_scale=softmax_scale,
                           q_scale=q_scale, causal=causal, window_size=window_size, deterministic=deterministic, dtype=dtype,
                           version=fa_version)
