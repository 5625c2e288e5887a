"""Wan DiT (`WanModel`) on MI355X: the reference's module tree / parameter names / call signatures over the
hand-written HIP kernels of libunivid_hip.so.

Mirrors /root/reference/models/wan/utils/modules/model.py (WanModel :294-546, WanAttentionBlock :183-259,
WanSelfAttention :101-155, WanCrossAttention :158-180, Head :262-291, WanRMSNorm :69-85, WanLayerNorm :88-98)
so that UniVid's own code keeps working against it: `.blocks` is an nn.ModuleList, cross-attention modules are
instances of a class NAMED `WanCrossAttention` whose `.forward(x, context, context_lens)` can be re-assigned
per instance (models/model_pipeline.py:1745-1807), the nn.Linear children are called
self_attn.{q,k,v,o} / cross_attn.{q,k,v,o} / ffn.0 / ffn.2 (LoRA targets, model_pipeline.py:474-493), and the
state-dict keys equal the reference's, so its checkpoints load.

Numerics contract = UniVid's runtime setting (fp32 parameters, ambient autocast(bf16), fp32 islands; SURVEY.md
Appendix A): parameters stay fp32 nn.Parameters; `prepare()` makes the bf16 copies autocast would make on
every call. Every op runs in a HIP kernel; torch is used for memory, streams and views only. There is no
eager fallback: without the extension or a gfx950 device, forward raises.

Per-token timesteps (`t` of shape [B, seq_len], textimage2video.py:373-378) are handled without materialising
the reference's [L, 6, dim] fp32 modulation tensor (843 MB at L = 11 440): the distinct timestep values
(1 for t2v, 2 for i2v) go through the time MLPs once, and kernels index the resulting rows with a
token -> row map.
"""
import contextlib
import math
from typing import List, Optional

import torch
import torch.nn as nn

from .. import _lib
from .._lib import (EPI_BF16, EPI_BF16_T, EPI_F32_FROM_BF16, EPI_GATE_RESID_F32, EPI_GELU_BF16, EPI_RESID_F32)

__all__ = ["WanModel"]

BF16 = torch.bfloat16


def _round_up(x, m):
    return (x + m - 1) // m * m


def rope_params(max_seq_len, dim, theta=10000):
    """complex128 phasor table (model.py:27-35); built once on the host in fp64."""
    ang = torch.outer(torch.arange(max_seq_len),
                      1.0 / torch.pow(theta, torch.arange(0, dim, 2).to(torch.float64).div(dim)))
    return torch.polar(torch.ones_like(ang), ang)


class _Prepared:
    """bf16 weight/bias copies of one nn.Linear (what autocast's weight cache holds in the reference)."""
    __slots__ = ("w", "b")

    def __init__(self, lin: nn.Linear, pad_k: int = 0):
        if hasattr(lin, "base_layer") or hasattr(lin, "lora_A"):
            # PEFT's lora.Linear (what the reference's LoRAManager installs on q/k/v/o/ffn.0/ffn.2, model_pipeline.py:340-380)
            # exposes `.weight` = base_layer.weight: reading it would silently DROP the adapter. The fused kernels consume one
            # dense weight per projection, so the deltas have to be folded in first.
            raise NotImplementedError(
                "a LoRA-wrapped nn.Linear (PEFT lora.Linear) was found in the DiT: the HIP path reads dense weights and would "
                "ignore the adapter. Load the adapter directory with `univid_amd.lora.LoRAManager().load_lora_weights(dir, model)` "
                "(folds it into the dense weights), or fold it in with `peft_model.merge_and_unload()` and call "
                "`WanModel.invalidate()` before the next forward.")
        w = lin.weight.detach()
        if pad_k and w.shape[1] % pad_k:
            w = torch.nn.functional.pad(w, (0, pad_k - w.shape[1] % pad_k))
        self.w = w.to(BF16).contiguous()
        self.b = None if lin.bias is None else lin.bias.detach().to(BF16).contiguous()


class WanRMSNorm(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Parameter(torch.ones(dim))


class WanLayerNorm(nn.LayerNorm):
    def __init__(self, dim, eps=1e-6, elementwise_affine=False):
        super().__init__(dim, elementwise_affine=elementwise_affine, eps=eps)


class WanSelfAttention(nn.Module):
    def __init__(self, dim, num_heads, window_size=(-1, -1), qk_norm=True, eps=1e-6):
        assert dim % num_heads == 0
        super().__init__()
        self.dim, self.num_heads, self.head_dim = dim, num_heads, dim // num_heads
        self.window_size, self.qk_norm, self.eps = window_size, qk_norm, eps
        if tuple(window_size) != (-1, -1) or not qk_norm:
            raise NotImplementedError("only the TI2V-5B setting (global attention, qk_norm) is built")
        self.q = nn.Linear(dim, dim)
        self.k = nn.Linear(dim, dim)
        self.v = nn.Linear(dim, dim)
        self.o = nn.Linear(dim, dim)
        self.norm_q = WanRMSNorm(dim, eps=eps)
        self.norm_k = WanRMSNorm(dim, eps=eps)
        self._prep = None
        self._kv_cache = {}

    def prepare(self):
        self._prep = {n: _Prepared(getattr(self, n)) for n in ("q", "k", "v", "o")}
        self._kv_cache = {}

    def _self_attn(self, h, L, grid, freqs, x_resid, gate, gate_tid, batch=1, sp=None):
        """h: bf16 [batch*L, C] modulated input (samples stacked along the token axis). Adds o(attn) * gate into x_resid
        (fp32) in the GEMM epilogue. sp = (SeqParallel, L_total, r0): h holds this rank's token range of every sample."""
        if sp is not None:
            return self._self_attn_sp(h, L, grid, freqs, x_resid, gate, gate_tid, batch, sp)
        C, H, D = self.dim, self.num_heads, self.head_dim
        p = self._prep
        dev = h.device
        M = batch * L
        ql = torch.empty(M, C, dtype=BF16, device=dev)
        kl = torch.empty(M, C, dtype=BF16, device=dev)
        vt = _vt_scratch("vt", C, batch, L, dev)
        _lib.gemm_bf16(h, p["q"].w, p["q"].b, ql, EPI_BF16, M=M)
        _lib.gemm_bf16(h, p["k"].w, p["k"].b, kl, EPI_BF16, M=M)
        _lib.gemm_bf16(h, p["v"].w, p["v"].b, vt, EPI_BF16_T, M=M)   # V^T [C, tokens]: sample b = columns b*L..
        # q and k of every stacked sample in one launch (RoPE positions restart with every sample)
        _lib.rmsnorm_rope_qk(ql, kl, self.norm_q.weight, self.norm_k.weight, M, L, C, D, self.eps, freqs, grid)
        att = torch.empty(M, C, dtype=BF16, device=dev)
        _lib.flash_attn(ql, kl, vt, att, L, L, H, D, 1.0 / math.sqrt(D), batch=batch)
        _lib.gemm_bf16(att, p["o"].w, p["o"].b, x_resid, EPI_GATE_RESID_F32, M=M, gate=gate, gate_tid=gate_tid)

    def _self_attn_sp(self, h, n, grid, freqs, x_resid, gate, gate_tid, batch, sp):
        """Ulysses form (distributed/sequence_parallel.py:147-176, ulysses.py:9-47): projections, RMSNorm and RoPE on the
        rank's n tokens of each sample; q/k/V^T exchanged to [all tokens, heads/p]; attention over the whole sequence for
        the rank's heads; output exchanged back; o-projection + gated residual on the rank's tokens."""
        par, Lt, r0 = sp
        C, H, D = self.dim, self.num_heads, self.head_dim
        P = par.size
        if H % P or Lt % 8:
            raise NotImplementedError(f"sequence parallel needs heads % ranks == 0 and tokens % 8 == 0 (H={H}, p={P}, L={Lt})")
        p = self._prep
        dev = h.device
        M = batch * n
        Cp, Hp = C // P, H // P
        ql = torch.empty(M, C, dtype=BF16, device=dev)
        kl = torch.empty(M, C, dtype=BF16, device=dev)
        vt = torch.zeros(C, _round_up(max(M, 1), 64), dtype=BF16, device=dev)
        qf = torch.empty(batch * Lt, Cp, dtype=BF16, device=dev)
        kf = torch.empty(batch * Lt, Cp, dtype=BF16, device=dev)
        vtf = _vt_scratch("vt_sp", Cp, batch, Lt, dev)
        if M:
            _lib.gemm_bf16(h, p["q"].w, p["q"].b, ql, EPI_BF16, M=M)
            _lib.gemm_bf16(h, p["k"].w, p["k"].b, kl, EPI_BF16, M=M)
            _lib.gemm_bf16(h, p["v"].w, p["v"].b, vt, EPI_BF16_T, M=M)
        for b in range(batch):
            rows = slice(b * n, (b + 1) * n)
            if n:
                _lib.rmsnorm_rope(ql[rows], ql[rows], self.norm_q.weight, n, C, D, self.eps, freqs, grid, row0=r0)
                _lib.rmsnorm_rope(kl[rows], kl[rows], self.norm_k.weight, n, C, D, self.eps, freqs, grid, row0=r0)
            par.heads_to_tokens(ql[rows], Lt, qf[b * Lt:(b + 1) * Lt])
            par.heads_to_tokens(kl[rows], Lt, kf[b * Lt:(b + 1) * Lt])
            par.heads_to_tokens_T(vt[:, b * n:(b + 1) * n], Lt, vtf, col0=b * Lt)
        attf = torch.empty(batch * Lt, Cp, dtype=BF16, device=dev)
        _lib.flash_attn(qf, kf, vtf, attf, Lt, Lt, Hp, D, 1.0 / math.sqrt(D), batch=batch)
        att = torch.empty(M, C, dtype=BF16, device=dev)
        for b in range(batch):
            par.tokens_to_heads(attf[b * Lt:(b + 1) * Lt], Lt, att[b * n:(b + 1) * n])
        if M:
            _lib.gemm_bf16(att, p["o"].w, p["o"].b, x_resid, EPI_GATE_RESID_F32, M=M, gate=gate, gate_tid=gate_tid)

    def forward(self, x, seq_lens, grid_sizes, freqs):
        """Reference signature (model.py:126-155): x [B, L, C] -> [B, L, C] bf16. Keys >= seq_lens[b] are masked,
        which here means: only the first seq_lens[b] tokens are attended (the rest of L is padding)."""
        _ensure_prepared(self)
        outs = []
        fr = _freqs_device(freqs, x.device)
        for b in range(x.size(0)):
            L = x.size(1)
            Lk = int(seq_lens[b])
            h = x[b].to(BF16).contiguous()
            C, H, D = self.dim, self.num_heads, self.head_dim
            p = self._prep
            ql = torch.empty(L, C, dtype=BF16, device=x.device)
            kl = torch.empty(L, C, dtype=BF16, device=x.device)
            vt = torch.zeros(C, _round_up(L, 64), dtype=BF16, device=x.device)
            _lib.gemm_bf16(h, p["q"].w, p["q"].b, ql, EPI_BF16)
            _lib.gemm_bf16(h, p["k"].w, p["k"].b, kl, EPI_BF16)
            _lib.gemm_bf16(h, p["v"].w, p["v"].b, vt, EPI_BF16_T)
            grid = tuple(int(v) for v in grid_sizes[b])
            _lib.rmsnorm_rope(ql, ql, self.norm_q.weight, L, C, D, self.eps, fr, grid)
            _lib.rmsnorm_rope(kl, kl, self.norm_k.weight, L, C, D, self.eps, fr, grid)
            att = torch.empty(L, C, dtype=BF16, device=x.device)
            _lib.flash_attn(ql, kl, vt, att, L, Lk, H, D, 1.0 / math.sqrt(D))
            y = torch.empty(L, C, dtype=BF16, device=x.device)
            _lib.gemm_bf16(att, p["o"].w, p["o"].b, y, EPI_BF16)
            outs.append(y)
        return torch.stack(outs)


class WanCrossAttention(WanSelfAttention):
    """Text cross-attention (model.py:158-180). The class name and the (x, context, context_lens) signature are
    part of UniVid's contract: Wan22ContextWrapper finds modules by `__class__.__name__ == 'WanCrossAttention'`
    and replaces `module.forward` with a closure that rescales `context` (model_pipeline.py:1745-1807)."""
    fuse_q_norm = True      # norm_q inside the attention kernel's Q prologue instead of a pass over q (A/B switch; same arithmetic but for
                            # the summation order of the row's mean square)

    def _context_kv(self, ctx, Lc, batch, kv_key, out=None):
        """k = norm_k(Wk ctx) [batch*Lc, C] and V^T = (Wv ctx)^T of the embedded context (model.py:170-172). They depend on the
        context and this block's weights only, not on the latent or the timestep, so across the steps of a sampling loop they are
        computed ONCE: `kv_key` identifies the (context generation, sample group) WanModel.forward is running; None = no caching
        (the reference-signature forward, a context under UniVid's dynamic text weight - it changes from forward to forward -, or
        out=(k, V^T): the caller's own buffers, written in place - the HIP-graph runner's, which its captured attention launches read)."""
        hit = self._kv_cache.get(kv_key) if kv_key is not None else None
        if hit is not None:
            return hit
        C, D = self.dim, self.head_dim
        p = self._prep
        dev = ctx.device
        if out is not None:
            kl, vt = out
            _lib.gemm_bf16(ctx, p["k"].w, p["k"].b, kl, EPI_BF16, M=batch * Lc)
            _lib.gemm_bf16(ctx, p["v"].w, p["v"].b, vt, EPI_BF16_T, M=batch * Lc)
            _lib.rmsnorm_rope(kl, kl, self.norm_k.weight, batch * Lc, C, D, self.eps)
            return kl, vt
        kl = torch.empty(batch * Lc, C, dtype=BF16, device=dev)
        if kv_key is None:
            vt = _vt_scratch("cvt", C, batch, Lc, dev)
        else:
            if batch > 1 and Lc % 8:
                raise NotImplementedError(f"stacked samples need a context length divisible by 8 (got {Lc})")
            vt = torch.zeros(C, (batch - 1) * Lc + _round_up(Lc, 64), dtype=BF16, device=dev)
        _lib.gemm_bf16(ctx, p["k"].w, p["k"].b, kl, EPI_BF16, M=batch * Lc)
        _lib.gemm_bf16(ctx, p["v"].w, p["v"].b, vt, EPI_BF16_T, M=batch * Lc)
        _lib.rmsnorm_rope(kl, kl, self.norm_k.weight, batch * Lc, C, D, self.eps)
        if kv_key is not None:
            if self._kv_cache and next(iter(self._kv_cache))[0] != kv_key[0]:
                self._kv_cache.clear()          # a new context generation: the old entries can never hit again
            self._kv_cache[kv_key] = (kl, vt)
        return kl, vt

    def _attend(self, hq, ctx, L, Lc, batch=1, kv_key=None):
        """hq bf16 [batch*L, C] (normed queries' input), ctx bf16 [batch*Lc, C] -> attention output bf16 [batch*L, C]."""
        C, H, D = self.dim, self.num_heads, self.head_dim
        p = self._prep
        dev = hq.device
        M = batch * L
        ql = torch.empty(M, C, dtype=BF16, device=dev)
        kl, vt = self._context_kv(ctx, Lc, batch, kv_key)
        att = torch.empty(M, C, dtype=BF16, device=dev)
        if self.fuse_q_norm and C % 32 == 0:
            # norm_q without a pass of its own over q (round 5): the q GEMM's epilogue leaves each output row's sums of squares per 32-column
            # group, a tiny kernel turns them into the row scale 1 / sqrt(mean + eps), and the attention kernel's Q prologue applies scale and
            # weight with WanRMSNorm's rounding points (model.py:82-85) while it loads the raw projection
            ssq = torch.empty(M, C // 32, dtype=torch.float32, device=dev)
            rs = torch.empty(M, dtype=torch.float32, device=dev)
            _lib.gemm_bf16_ssq(hq, p["q"].w, p["q"].b, ql, ssq, M=M)
            _lib.rms_scale_from_ssq(ssq, rs, M, C, self.eps)
            _lib.flash_attn(ql, kl, vt, att, L, Lc, H, D, 1.0 / math.sqrt(D), batch=batch, q_rs=rs, q_weight=self.norm_q.weight)
            return att
        _lib.gemm_bf16(hq, p["q"].w, p["q"].b, ql, EPI_BF16, M=M)
        _lib.rmsnorm_rope(ql, ql, self.norm_q.weight, M, C, D, self.eps)
        _lib.flash_attn(ql, kl, vt, att, L, Lc, H, D, 1.0 / math.sqrt(D), batch=batch)
        return att

    def _cross_fused(self, hq, ctx, L, Lc, x_resid, batch=1, kv_key=None):
        att = self._attend(hq, ctx, L, Lc, batch, kv_key)
        p = self._prep["o"]
        _lib.gemm_bf16(att, p.w, p.b, x_resid, EPI_RESID_F32, M=batch * L)

    def forward(self, x, context, context_lens=None):
        """x [B, L1, C], context [B, L2, C] -> [B, L1, C] bf16 (the caller adds it to the residual stream)."""
        if context_lens is not None:
            raise NotImplementedError("context_lens is always None on UniVid's path (model.py:472)")
        _ensure_prepared(self)
        outs = []
        for b in range(x.size(0)):
            L, Lc = x.size(1), context.size(1)
            att = self._attend(x[b].to(BF16).contiguous(), context[b].to(BF16).contiguous(), L, Lc)
            y = torch.empty(L, self.dim, dtype=BF16, device=x.device)
            p = self._prep["o"]
            _lib.gemm_bf16(att, p.w, p.b, y, EPI_BF16)
            outs.append(y)
        return torch.stack(outs)


class WanAttentionBlock(nn.Module):
    def __init__(self, dim, ffn_dim, num_heads, window_size=(-1, -1), qk_norm=True, cross_attn_norm=False, eps=1e-6):
        super().__init__()
        self.dim, self.ffn_dim, self.num_heads, self.eps = dim, ffn_dim, num_heads, eps
        self.window_size, self.qk_norm, self.cross_attn_norm = window_size, qk_norm, cross_attn_norm
        self.norm1 = WanLayerNorm(dim, eps)
        self.self_attn = WanSelfAttention(dim, num_heads, window_size, qk_norm, eps)
        self.norm3 = WanLayerNorm(dim, eps, elementwise_affine=True) if cross_attn_norm else nn.Identity()
        self.cross_attn = WanCrossAttention(dim, num_heads, (-1, -1), qk_norm, eps)
        self.norm2 = WanLayerNorm(dim, eps)
        self.ffn = nn.Sequential(nn.Linear(dim, ffn_dim), nn.GELU(approximate="tanh"), nn.Linear(ffn_dim, dim))
        self.modulation = nn.Parameter(torch.randn(1, 6, dim) / dim ** 0.5)
        self._prep = None

    def prepare(self):
        self.self_attn.prepare()
        self.cross_attn.prepare()
        self._prep = {"ffn0": _Prepared(self.ffn[0]), "ffn2": _Prepared(self.ffn[2])}

    def _run(self, x, L, e0_rows, tid, grid, freqs, ctx, first_block, batch=1, sp=None, kv_key=None, twin_rows=False):
        """x: fp32 [batch*L, C] residual stream (independent samples stacked along the token axis), updated IN PLACE.
        e0_rows: fp32 [n_t, 6C]; tid int32 [batch*L] | None; ctx: bf16 [batch*Lc, C] embedded context(s); kv_key: cache key of
        the context's cross-attention K / V^T (WanCrossAttention._context_kv), None = recompute.
        twin_rows: the stacked samples enter this block with IDENTICAL rows and timesteps (the CFG pair on one latent, before its
        first cross-attention): the self-attention half runs on sample 0 only and its result is copied to the others."""
        C = self.dim
        dev = x.device
        n_t = e0_rows.shape[0]
        Ls, L = L, batch * L          # Ls = tokens per sample; L = rows of every row-wise kernel below
        tab = torch.empty(n_t, 6 * C, dtype=torch.float32, device=dev)
        _lib.call("uv_add_rows_f32", _lib.ptr(self.modulation), _lib.ptr(e0_rows), _lib.ptr(tab), n_t, 6 * C,
                  _lib.stream_ptr())                                                               # model.py:239
        # (rows of the block's bf16 activations: L, or L rounded up to whole 256-row tiles for ffn.0 - see below; the pad rows are never
        # written and their products never read, so they need no zeroing and no scratch that outlives the forward)
        Lf = _ffn0_rows(L, self.ffn_dim, dev)
        h_full = torch.empty(Lf, C, dtype=BF16, device=dev)
        h = h_full[:L]
        # NOTE for consumers of `mid` (ffn.0's output, below): its rows >= L are the GELU of whatever the allocator left in h_full[L:]
        # - UNDEFINED, possibly NaN / Inf. Rows are independent in every kernel of the block and ffn.2 reads L rows only.
        # self-attention (model.py:243-247)
        if twin_rows and batch > 1 and sp is None:
            t1 = None if tid is None else tid[:Ls]
            _lib.layernorm_mod(x[:Ls], h[:Ls], Ls, C, self.eps, mode=1, tab=tab, shift_off=0, scale_off=C, tid=t1, round_ln=first_block)
            self.self_attn._self_attn(h[:Ls], Ls, grid, freqs, x[:Ls], tab[:, 2 * C:], t1, 1, None)
            for b in range(1, batch):
                x[b * Ls:(b + 1) * Ls].copy_(x[:Ls])
        else:
            _lib.layernorm_mod(x, h, L, C, self.eps, mode=1, tab=tab, shift_off=0, scale_off=C, tid=tid,
                               round_ln=first_block)
            self.self_attn._self_attn(h, Ls, grid, freqs, x, tab[:, 2 * C:], tid, batch, sp)
        # cross-attention (model.py:251)
        if self.cross_attn_norm:
            _lib.layernorm_mod(x, h, L, C, self.eps, mode=2, w=self.norm3.weight, b=self.norm3.bias)
        else:
            _lib.call("uv_cast_f32_bf16", _lib.ptr(x), _lib.ptr(h), L * C, _lib.stream_ptr())
        Lc = ctx.shape[0] // batch
        if "forward" in self.cross_attn.__dict__:
            # forward was re-assigned on the instance (UniVid hook): honour it, then add the residual un-fused
            y = self.cross_attn.forward(h.view(batch, Ls, C), ctx.view(batch, Lc, C), None)
            y = y.reshape(L, C).contiguous()
            _lib.call("uv_add_bf16_resid", _lib.ptr(x), x.stride(0), _lib.ptr(y), y.stride(0), L, C, _lib.stream_ptr())
        else:
            self.cross_attn._cross_fused(h, ctx, Ls, Lc, x, batch, kv_key)
        # FFN (model.py:252-255)
        # ffn.0 on rows rounded up to whole 256-row tiles where the extra tiles ride in the last, partial round of the persistent GEMM
        # (22 880 rows x 14 336 columns: 19.47 -> 19.69 rounds, both 20) instead of a leftover-row launch behind it: the pad rows of the
        # input hold whatever the buffer held (rows are independent: nothing of them reaches a row that is read), their outputs are never
        # read (ffn.2 runs on L rows); results unchanged (a row's arithmetic is the same in both kernels)
        _lib.layernorm_mod(x, h, L, C, self.eps, mode=1, tab=tab, shift_off=3 * C, scale_off=4 * C, tid=tid)
        mid = torch.empty(Lf, self.ffn_dim, dtype=BF16, device=dev)
        _lib.gemm_bf16(h_full, self._prep["ffn0"].w, self._prep["ffn0"].b, mid, EPI_GELU_BF16, M=Lf)
        # ffn.2's leftover rows (1 120 of 22 880) as one round of 256 x 256 tiles x split-K 4 where the library has that strip for the
        # shape (K >= 8192): it needs scratch for the partial tiles (63 MB at this shape; the allocator hands every block the same one)
        nws = _splitk_ws_bytes(L, C, self.ffn_dim, dev)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev) if nws else None
        _lib.gemm_bf16(mid, self._prep["ffn2"].w, self._prep["ffn2"].b, x, EPI_GATE_RESID_F32, M=L,
                       gate=tab[:, 5 * C:], gate_tid=tid, ws=ws)

    def forward(self, x, e, seq_lens, grid_sizes, freqs, context, context_lens=None):
        """Reference signature (model.py:219-259): x [B, L, C], e [B, L, 6, C] fp32 -> [B, L, C] fp32.
        Every token carries its own modulation row here (the general case the reference materialises)."""
        assert e.dtype == torch.float32
        _ensure_prepared(self)
        fr = _freqs_device(freqs, x.device)
        outs = []
        for b in range(x.size(0)):
            L = x.size(1)
            if int(seq_lens[b]) != L:
                raise NotImplementedError("padded sequences: call WanModel.forward, which strips the padding")
            xb = x[b].float().contiguous().clone()
            tid = torch.arange(L, dtype=torch.int32, device=x.device)
            self._run(xb, L, e[b].reshape(L, -1).contiguous(), tid, tuple(int(v) for v in grid_sizes[b]), fr,
                      context[b].to(BF16).contiguous(), first_block=(x.dtype == BF16))
            outs.append(xb)
        return torch.stack(outs)


class Head(nn.Module):
    def __init__(self, dim, out_dim, patch_size, eps=1e-6):
        super().__init__()
        self.dim, self.out_dim, self.patch_size, self.eps = dim, out_dim, patch_size, eps
        self.norm = WanLayerNorm(dim, eps)
        self.head = nn.Linear(dim, math.prod(patch_size) * out_dim)
        self.modulation = nn.Parameter(torch.randn(1, 2, dim) / dim ** 0.5)

    def _run(self, x, L, e_rows, tid):
        """fp32 island (model.py:286-290): LN -> modulate -> Linear, all fp32. Returns [L, P*Cout] fp32."""
        C = self.dim
        dev = x.device
        n_t = e_rows.shape[0]
        tab = torch.empty(n_t, 2 * C, dtype=torch.float32, device=dev)
        e2 = e_rows.repeat(1, 2).contiguous()       # e.unsqueeze(2) broadcast over the 2 modulation rows
        _lib.call("uv_add_rows_f32", _lib.ptr(self.modulation), _lib.ptr(e2), _lib.ptr(tab), n_t, 2 * C, _lib.stream_ptr())
        hh = torch.empty(L, C, dtype=torch.float32, device=dev)
        _lib.layernorm_mod(x, hh, L, C, self.eps, mode=1, tab=tab, shift_off=0, scale_off=C, tid=tid)
        out = torch.empty(L, self.head.out_features, dtype=torch.float32, device=dev)
        _lib.gemm_f32(hh, self.head.weight, self.head.bias, out, M=L)
        return out


_zero_cache = {}


_ncu = {}


_ws_bytes = {}
SPLITK_STRIP = False     # opt-in (bench.py --splitk-strip): ffn.2's leftover rows as one round of 256 x 256 tiles x split-K 4 (uv_gemm_bf16_nt_ws).
# OFF by default: measured -0.35 ... -0.5 ms of a 251 ms step (round 6), and it ends the property every other schedule of the GEMM keeps - a row's
# bits do not depend on which rows it is launched with - so the stacked CFG pair would no longer be bit-identical to two batch-1 forwards


def _splitk_ws_bytes(M, N, K, dev):
    if not SPLITK_STRIP:
        return 0
    key = (M, N, K, str(dev))
    n = _ws_bytes.get(key)
    if n is None:
        n = _ws_bytes[key] = _lib.gemm_splitk_ws_bytes(M, N, K, dev)
    return n


def _ffn0_rows(L, n_cols, dev):
    """Rows to run the first FFN projection on: L, or L rounded up to 256 when the padded row tile does not add a round of 256 x 256 tiles."""
    Lp = _round_up(L, 256)
    if Lp == L or n_cols % 256:
        return L
    ncu = _ncu.get(dev)
    if ncu is None:
        ncu = _ncu[dev] = max(8, torch.cuda.get_device_properties(dev).multi_processor_count & ~7)
    tn = n_cols // 256
    full, padded = (L // 256) * tn, (Lp // 256) * tn
    return Lp if full >= 2 * ncu and -(-padded // ncu) == -(-full // ncu) else L


def _zeros_cached(key, shape, dtype, device):
    """Zero-initialised scratch whose padding region is never written (V^T column padding). One tensor per key: a new shape
    replaces the old one (the caching allocator frees it in stream order), so the cache does not grow with the resolutions served."""
    t = _zero_cache.get(key)
    # (a scratch created under torch.inference_mode() cannot be re-zeroed in place outside it - e.g. by a HIP-graph capture, which runs
    # outside inference mode -: such a tensor is replaced)
    if t is None or tuple(t.shape) != tuple(shape) or (t.is_inference() and not torch.is_inference_mode_enabled()):
        t = torch.zeros(shape, dtype=dtype, device=device)
        _zero_cache[key] = t
    return t


_vt_user = {}


def _vt_scratch(tag, C, batch, L, device):
    """V^T [C, columns]: sample b's keys are columns [b*L, (b+1)*L); the tail [batch*L, columns) up to the 64-key tile bound is
    zero. The attention kernels mask the scores of those columns to -inf (P = 0 exactly), so the contract they need is only
    that the tail is FINITE (0 * NaN would poison the row); it is kept zero here: different (batch, L) pairs can give the same
    column count, and then the previous user's V values would sit in the new user's tail - it is re-zeroed on such a change."""
    if batch > 1 and L % 8:
        raise NotImplementedError(f"stacked samples need a token count divisible by 8 (got {L}); run them one by one")
    cols = (batch - 1) * L + _round_up(L, 64)
    # one scratch per stream: forwards running concurrently on different streams must not share it
    key = (tag, device, torch.cuda.current_stream(device).cuda_stream)
    t = _zeros_cached(key, (C, cols), BF16, device)
    if _vt_user.get(key, (batch, L, t.data_ptr())) != (batch, L, t.data_ptr()) and cols > batch * L:
        t[:, batch * L:].zero_()
    _vt_user[key] = (batch, L, t.data_ptr())
    return t


def scratch_snapshot():
    """Identity of every per-stream scratch tensor (before a HIP-graph capture)."""
    return {k: id(t) for k, t in _zero_cache.items()}


def scratch_take_new(snapshot):
    """Removes - and returns, for the caller to keep alive - the scratch tensors created since `snapshot`: those of a HIP-graph capture
    live in that graph's private memory pool, and torch's capture stream (part of their cache key) is shared by all captures."""
    taken = []
    for k in list(_zero_cache):
        if snapshot.get(k) != id(_zero_cache[k]):
            taken.append(_zero_cache.pop(k))
            _vt_user.pop(k, None)
    return taken


def tensor_version(u):
    """`u._version`, or -1 for a tensor created under torch.inference_mode() (no version counter; such a tensor cannot be written
    in place outside inference mode, and inside a `context_cached()` scope the caller vouches for it)."""
    try:
        return u._version
    except RuntimeError:
        return -1


def _ensure_prepared(mod):
    if getattr(mod, "_prep", None) is None:
        mod.prepare()


def _freqs_device(freqs, device):
    """complex128 [1024, D/2] -> float64 [1024, D/2, 2] on the device (cached on the tensor object)."""
    cached = getattr(freqs, "_uv_dev", None)
    if cached is not None and cached.device == device:
        return cached
    fr = torch.view_as_real(freqs.to(torch.complex128)).contiguous().to(device)
    try:
        freqs._uv_dev = fr
    except Exception:
        pass
    return fr


class WanModel(nn.Module):
    """Wan diffusion backbone (reference WanModel, model.py:294-546), HIP-backed."""

    def __init__(self, model_type="t2v", patch_size=(1, 2, 2), text_len=512, in_dim=16, dim=2048, ffn_dim=8192,
                 freq_dim=256, text_dim=4096, out_dim=16, num_heads=16, num_layers=32, window_size=(-1, -1), qk_norm=True,
                 cross_attn_norm=True, eps=1e-6):
        super().__init__()
        assert model_type in ["t2v", "i2v", "ti2v", "s2v"]
        self.model_type = model_type
        self.patch_size = tuple(patch_size)
        self.text_len, self.in_dim, self.dim, self.ffn_dim = text_len, in_dim, dim, ffn_dim
        self.freq_dim, self.text_dim, self.out_dim = freq_dim, text_dim, out_dim
        self.num_heads, self.num_layers = num_heads, num_layers
        self.window_size, self.qk_norm, self.cross_attn_norm, self.eps = window_size, qk_norm, cross_attn_norm, eps
        self.config = dict(model_type=model_type, patch_size=self.patch_size, text_len=text_len, in_dim=in_dim, dim=dim,
                           ffn_dim=ffn_dim, freq_dim=freq_dim, text_dim=text_dim, out_dim=out_dim, num_heads=num_heads,
                           num_layers=num_layers, window_size=window_size, qk_norm=qk_norm,
                           cross_attn_norm=cross_attn_norm, eps=eps)

        self.patch_embedding = nn.Conv3d(in_dim, dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.text_embedding = nn.Sequential(nn.Linear(text_dim, dim), nn.GELU(approximate="tanh"), nn.Linear(dim, dim))
        self.time_embedding = nn.Sequential(nn.Linear(freq_dim, dim), nn.SiLU(), nn.Linear(dim, dim))
        self.time_projection = nn.Sequential(nn.SiLU(), nn.Linear(dim, dim * 6))
        self.blocks = nn.ModuleList([
            WanAttentionBlock(dim, ffn_dim, num_heads, window_size, qk_norm, cross_attn_norm, eps)
            for _ in range(num_layers)
        ])
        self.head = Head(dim, out_dim, self.patch_size, eps)

        assert (dim % num_heads) == 0 and (dim // num_heads) % 2 == 0
        d = dim // num_heads
        if d not in (64, 128):
            raise NotImplementedError(f"head_dim {d}: the attention kernel is built for 64 and 128")
        # plain attribute like the reference (model.py:397-405), so .to() does not change its dtype
        with torch.device("cpu"):  # fp64 host table, also when the module is built under torch.device('cuda')
            self.freqs = torch.cat([rope_params(1024, d - 4 * (d // 6)), rope_params(1024, 2 * (d // 6)),
                                    rope_params(1024, 2 * (d // 6))], dim=1)
        self._prep = None
        self._ctx_cache = None   # (key, input tensors kept alive, embedded contexts, generation): see _embedded_context
        self._ctx_gen = 0
        self._prep_gen = 0       # bumped by prepare() and invalidate(): the identity of "the prepared weights" for cache / graph keys
        # Step-constant context work (text_embedding, cross-attention K / V^T) computed once per context. OFF for the bare
        # reference-signature forward: the cache key is tensor identity + version counter, which cannot see writes that bypass the
        # counter (`.data` writes, raw-pointer kernels). A sampling loop that OWNS its context tensors for its duration switches it
        # on around the loop: `with model.context_cached(): ...` (WanTI2V.denoise and bench.py do).
        self.cache_context = False
        self.sp = None   # SeqParallel when Ulysses sequence parallelism is enabled (enable_sequence_parallel)
        self._text_weight = None  # (per-sample weights, rows, layers | None): set_text_weight
        self.dedup_twins = True   # block 0's self-attention half once for samples that enter with identical rows (the CFG pair); A/B switch
        self.register_load_state_dict_post_hook(lambda m, _k: m.invalidate())

    # ---- weight preparation -------------------------------------------------------------------------------
    def invalidate(self):
        """Forget the bf16 weight copies (call after changing parameters in place)."""
        self._prep = None
        self._prep_gen += 1
        self._ctx_cache = None
        for b in self.blocks:
            b._prep = None
            b.self_attn._prep = None
            b.cross_attn._prep = None
            b.cross_attn._kv_cache = {}

    def prepare(self):
        """Materialise the bf16 operand copies autocast would create on every call of the reference."""
        dev = self.patch_embedding.weight.device
        if dev.type != "cuda":
            raise _lib.UnividHipError("WanModel.prepare: parameters must be on the GPU (model.to('cuda')) - there is "
                                      "no CPU path in univid_amd")
        _lib.init()
        pe = nn.Linear(1, 1, bias=True)
        pe.weight = nn.Parameter(self.patch_embedding.weight.detach().flatten(1), requires_grad=False)
        pe.bias = nn.Parameter(self.patch_embedding.bias.detach(), requires_grad=False)
        self._prep_gen += 1
        self._prep = {
            "patch": _Prepared(pe, pad_k=64),
            "text0": _Prepared(self.text_embedding[0], pad_k=64),
            "text2": _Prepared(self.text_embedding[2]),
        }
        for b in self.blocks:
            b.prepare()
        return self

    # ---- pieces of forward --------------------------------------------------------------------------------
    def _time_rows(self, tvals):
        """tvals: fp32 [n_t] distinct timesteps -> (e [n_t, C], e0 [n_t, 6C]) fp32 (model.py:462-469)."""
        n_t = tvals.numel()
        dev = tvals.device
        C = self.dim
        sp = _lib.stream_ptr()
        emb = torch.empty(n_t, self.freq_dim, dtype=torch.float32, device=dev)
        _lib.call("uv_sinusoid_f32", _lib.ptr(tvals), _lib.ptr(emb), n_t, self.freq_dim, sp)
        te0, te2, tp = self.time_embedding[0], self.time_embedding[2], self.time_projection[1]
        h1 = torch.empty(n_t, C, dtype=torch.float32, device=dev)
        e = torch.empty(n_t, C, dtype=torch.float32, device=dev)
        e0 = torch.empty(n_t, 6 * C, dtype=torch.float32, device=dev)
        _lib.call("uv_linear_rows_f32", _lib.ptr(emb), emb.stride(0), _lib.ptr(te0.weight), _lib.ptr(te0.bias), _lib.ptr(h1),
                  h1.stride(0), n_t, C, self.freq_dim, 0, sp)
        _lib.call("uv_linear_rows_f32", _lib.ptr(h1), h1.stride(0), _lib.ptr(te2.weight), _lib.ptr(te2.bias), _lib.ptr(e),
                  e.stride(0), n_t, C, C, 1, sp)
        _lib.call("uv_linear_rows_f32", _lib.ptr(e), e.stride(0), _lib.ptr(tp.weight), _lib.ptr(tp.bias), _lib.ptr(e0),
                  e0.stride(0), n_t, 6 * C, C, 1, sp)
        return e, e0

    def embed_context(self, context: List[torch.Tensor]):
        """text_embedding on the zero-padded prompt embeddings (model.py:472-478) -> bf16 [B, text_len, C]."""
        _ensure_prepared(self)
        dev = self.patch_embedding.weight.device
        Kp = self._prep["text0"].w.shape[1]
        outs = []
        for u in context:
            if u.size(0) > self.text_len:
                raise ValueError(f"context has {u.size(0)} rows > text_len {self.text_len}")
            a = torch.zeros(self.text_len, Kp, dtype=BF16, device=dev)
            a[:u.size(0), :u.size(1)] = u.to(device=dev, dtype=BF16)
            h = torch.empty(self.text_len, self.dim, dtype=BF16, device=dev)
            _lib.gemm_bf16(a, self._prep["text0"].w, self._prep["text0"].b, h, EPI_GELU_BF16)
            c = torch.empty(self.text_len, self.dim, dtype=BF16, device=dev)
            _lib.gemm_bf16(h, self._prep["text2"].w, self._prep["text2"].b, c, EPI_BF16)
            outs.append(c)
        return torch.stack(outs)

    @contextlib.contextmanager
    def context_cached(self):
        """Scope in which the caller guarantees that the context tensors it passes are not written behind the version counter
        (a sampling loop over loop-constant prompt embeddings): text_embedding and the blocks' cross-attention K / V^T are then
        computed once per context. The cache is dropped when the outermost scope ends."""
        prev = self.cache_context
        self.cache_context = True
        try:
            yield self
        finally:
            self.cache_context = prev
            if not prev:
                self._ctx_cache = None
                for b in self.blocks:
                    b.cross_attn._kv_cache = {}

    # ---- UniVid's dynamic text weight, native (models/model_pipeline.py:1699-1810) -----------------------------
    def set_text_weight(self, weights=None, text_len=0, layers=None):
        """The per-layer cross-attention hook of UniVid's Wan22ContextWrapper (model_pipeline.py:1742-1810) as model state instead of
        30 re-assigned `forward` closures: in the blocks of `layers` (None = every block; an iterable of block indices = the wrapper's
        `injection_layers`) the first `text_len` rows of sample i's EMBEDDED context are multiplied by bf16(weights[i]) before that
        block's K / V projections - exactly `context * weight_mask` of :1787-1797 - for every forward until the next call.
        weights: one float per sample of the forward's x / context lists (the CFG pair stacked as [cond, uncond] carries the two
        consecutive values of the wrapper's per-forward counter, :1856-1864); None, or all 1.0, switches it off (the hook's own
        `text_weight_multiplier != 1.0` test): the forward is then the plain one, with its cached context K / V^T.
        Because the rest of the forward is untouched, the stacked CFG pair, block 0's shared self-attention half, the fused
        residual epilogue and (WanTI2V.denoise) the HIP-graph replay all stay in use."""
        if weights is not None:
            weights = tuple(float(w) for w in weights)
            if all(w == 1.0 for w in weights) or int(text_len) <= 0:
                weights = None
        self._text_weight = None if weights is None else (weights, int(text_len), None if layers is None else frozenset(int(i) for i in layers))

    def weighted_context(self, ctx_all, idx, tw):
        """Embedded contexts of the samples `idx` stacked [len(idx) * text_len, C] with sample i's first tw[1] rows scaled by bf16(tw[0][i])."""
        weights, n_scaled, _ = tw
        Lc = ctx_all.shape[1]
        out = torch.empty(len(idx) * Lc, self.dim, dtype=BF16, device=ctx_all.device)
        for j, i in enumerate(idx):
            _lib.text_weight_rows(ctx_all[i], out[j * Lc:(j + 1) * Lc], min(n_scaled, Lc) if weights[i] != 1.0 else 0, weights[i])
        return out

    def _embedded_context(self, context):
        """text_embedding of the prompt embeddings, cached across forwards: in a sampling loop the same context tensors come
        back every step (textimage2video.py:380-385), and neither text_embedding (model.py:472-478) nor the blocks'
        cross-attention K / V projections of it (model.py:170-172) depend on the latent or the timestep. The key is the
        identity AND version counter of every input tensor (an in-place edit bumps `_version`; the tensors are kept referenced
        here so their storage cannot be recycled under the key); parameter changes go through invalidate().
        Returns (embedded contexts [B, text_len, C] bf16, generation id or None when caching is off)."""
        if not self.cache_context:
            return self.embed_context(context), None
        key = tuple((u.data_ptr(), tensor_version(u), tuple(u.shape), u.dtype, u.device) for u in context)
        c = self._ctx_cache
        if c is None or c[0] != key:
            self._ctx_gen += 1
            c = self._ctx_cache = (key, list(context), self.embed_context(context), self._ctx_gen)
        return c[2], c[3]

    def forward(self, x, t, context, seq_len, y=None, *, t_rows=None):
        r"""Same contract as the reference (model.py:410-497).

        x: List[Tensor[C_in, F, H, W]] fp32; t: Tensor[B] or Tensor[B, seq_len]; context: List[Tensor[L<=text_len,
        text_dim]]; seq_len: int. Returns List[Tensor[C_out, F, H, W]] float32.

        t_rows (extension, keyword-only): the caller's own table of the DISTINCT timesteps instead of `t`:
        `(tvals fp32 [n_t] on the device, sorted ascending, tid int32 [B*L] token -> row | None when n_t == 1[, same: bool - every
        sample carries the same token -> row map, which lets identical samples share block 0's self-attention half])`. With it the
        forward contains no device -> host round trip (finding the distinct values of a [B, seq_len] tensor needs one), which
        is what makes it capturable in a HIP graph (WanTI2V.denoise builds the table from the sampler's scalar timestep and
        the i2v mask). Only for equal-shape samples (one stacked group).
        """
        if self.model_type == "i2v":
            assert y is not None
        _ensure_prepared(self)
        dev = self.patch_embedding.weight.device
        if y is not None:
            x = [torch.cat([u, v], dim=0) for u, v in zip(x, y)]
        if t_rows is None and t.dim() == 1:  # one timestep per sample (model.py:460-461)
            t = t.view(-1, 1).expand(-1, seq_len)
        ctx_all, ctx_gen = self._embedded_context(context)
        fr = _freqs_device(self.freqs, dev)
        pt, ph, pw = self.patch_size
        C = self.dim
        sp = _lib.stream_ptr
        # Samples of identical shape run as ONE stacked pass ([B*L, C] rows: every GEMM / norm kernel sees B*L rows, the
        # attention kernel gets a batch dimension). Row-wise kernels and per-(sample, head) attention do not mix samples,
        # so each sample's result is bit-identical to running it alone; the benefit is occupancy (e.g. CFG's cond + uncond
        # pair: 4320 attention workgroups instead of 2 x 2160 on 512 slots) and weight reuse. Mixed shapes run one by one.
        xs_in = [u.to(device=dev, dtype=torch.float32).contiguous() for u in x]
        same = len({tuple(u.shape) for u in xs_in}) == 1
        if same:   # stacking needs 16-byte aligned per-sample columns in V^T: token count and context length % 8 == 0
            _, F0, H0, W0 = xs_in[0].shape
            same = ((F0 // pt) * (H0 // ph) * (W0 // pw)) % 8 == 0 and self.text_len % 8 == 0
        groups = [list(range(len(xs_in)))] if same else [[i] for i in range(len(xs_in))]
        outs = [None] * len(xs_in)
        for idx in groups:
            B = len(idx)
            cin, F, H, W = xs_in[idx[0]].shape
            Fp, Hp, Wp = F // pt, H // ph, W // pw
            L = Fp * Hp * Wp
            assert L <= seq_len, "sequence longer than seq_len (model.py:453)"
            # patch embedding: im2col -> GEMM, bf16 result stored as the fp32 residual stream (model.py:448-451)
            Kp = self._prep["patch"].w.shape[1]
            a = torch.empty(B * L, Kp, dtype=BF16, device=dev)
            for j, i in enumerate(idx):
                _lib.call("uv_patchify_bf16", _lib.ptr(xs_in[i]), _lib.ptr(a[j * L:]), a.stride(0), cin, F, H, W, pt, ph, pw, Kp, sp())
            # timesteps: distinct values -> rows; token -> row map (padding tokens beyond L are never computed)
            if t_rows is not None:
                if len(groups) != 1:
                    raise ValueError("t_rows= needs samples of one shape (a single stacked group)")
                tvals, tid = t_rows[0], t_rows[1]
                twin_t = tid is None or (len(t_rows) > 2 and bool(t_rows[2]))     # third entry: every sample has the same token -> row map
                if tid is not None and tid.numel() != B * L:
                    raise ValueError(f"t_rows: tid must hold {B * L} token -> row indices, got {tid.numel()}")
            else:
                tb = torch.cat([t[i].to(device=dev, dtype=torch.float32).flatten()[:L] for i in idx])
                tvals, inv = torch.unique(tb, return_inverse=True)          # device -> host sync (sizes the table)
                tid = None if tvals.numel() == 1 else inv.to(torch.int32).contiguous()
                # (i2v: two timesteps per sample - the samples of a twin pair must also agree token by token)
                twin_t = tid is None or (B > 1 and bool(torch.equal(inv.view(B, L), inv[:L].expand(B, L))))
            e_rows, e0_rows = self._time_rows(tvals.contiguous())
            ctx = ctx_all[idx[0]] if B == 1 else torch.cat([ctx_all[i] for i in idx], 0)
            kv_key = None if ctx_gen is None else (ctx_gen, tuple(idx))
            par = self.sp if (self.sp is not None and self.sp.size > 1) else None
            if par is None:
                n, sp_arg = L, None
            else:
                # sequence parallel (distributed/sequence_parallel.py:116-119): this rank keeps tokens [r0, r1) of every sample
                r0, r1 = par.token_range(L)
                n, sp_arg = r1 - r0, (par, L, r0)
                a = torch.cat([a[j * L + r0:j * L + r1] for j in range(B)], 0)
                if tid is not None:
                    tid = torch.cat([tid[j * L + r0:j * L + r1] for j in range(B)], 0).contiguous()
            xs = torch.empty(B * n, C, dtype=torch.float32, device=dev)
            if B * n:
                _lib.gemm_bf16(a, self._prep["patch"].w, self._prep["patch"].b, xs, EPI_F32_FROM_BF16)
            # The CFG pair of a sampling step is the SAME latent under two prompts (textimage2video.py:380-385): until the first
            # cross-attention the two samples' rows are identical, so block 0's self-attention half (LayerNorm, q/k/v, RoPE,
            # attention, o-projection: 1/60 of a step's attention and projection work) is computed once and copied. Per-sample results do
            # not depend on the stacking (tested), so the output is bit-identical. Detected, not assumed: the same tensor object for
            # every sample of the group and the same token -> timestep map in every sample.
            twin = (self.dedup_twins and B > 1 and par is None and all(xs_in[i] is xs_in[idx[0]] for i in idx) and
                    (tid is None or twin_t))
            # UniVid's dynamic text weight (set_text_weight): the hooked blocks read the row-scaled context, and compute their K / V^T
            # of it in this forward (it changes from forward to forward while the schedule runs; the cache holds the plain one)
            tw = self._text_weight
            if tw is not None and len(tw[0]) != len(xs_in):
                raise ValueError(f"set_text_weight was given {len(tw[0])} weight(s); this forward has {len(xs_in)} sample(s)")
            ctx_w = self.weighted_context(ctx_all, idx, tw) if (tw is not None and any(tw[0][i] != 1.0 for i in idx)) else None
            for li, blk in enumerate(self.blocks):
                hooked = ctx_w is not None and (tw[2] is None or li in tw[2])
                if par is None or n:
                    blk._run(xs, n, e0_rows, tid, (Fp, Hp, Wp), fr, ctx_w if hooked else ctx, first_block=(li == 0), batch=B, sp=sp_arg,
                             kv_key=None if hooked else kv_key, twin_rows=(twin and li == 0))
                else:   # a rank without tokens still takes part in the self-attention exchanges
                    blk.self_attn._self_attn_sp(xs.new_empty(0, C).to(BF16), 0, (Fp, Hp, Wp), fr, xs, None, None, B, sp_arg)
            yh = self.head._run(xs, B * n, e_rows, tid) if B * n else xs.new_empty(0, self.head.head.out_features)
            for j, i in enumerate(idx):
                out = torch.empty(self.out_dim, Fp * pt, Hp * ph, Wp * pw, dtype=torch.float32, device=dev)
                yj = yh[j * n:(j + 1) * n]
                if par is not None:
                    yj = par.gather_rows(yj, L).contiguous()     # gather_forward (sequence_parallel.py:139)
                _lib.call("uv_unpatchify_f32", _lib.ptr(yj), yj.stride(0), _lib.ptr(out), self.out_dim, Fp, Hp, Wp, pt, ph, pw,
                          sp())
                outs[i] = out
        return outs

    def enable_sequence_parallel(self, group=None):
        """Ulysses sequence parallelism over the ranks of `group` (what the reference's `use_sp=True` installs,
        textimage2video.py:106-118): every rank calls forward with the SAME inputs and gets the full outputs; tokens are
        sharded inside (univid_amd/parallel.py: SeqParallel). Pass group=False to switch it off."""
        from ..parallel import SeqParallel
        self.sp = None if group is False else SeqParallel(group)
        return self

    def unpatchify(self, x, grid_sizes):
        """model.py:499-522, for callers that use it directly."""
        outs = []
        pt, ph, pw = self.patch_size
        for u, v in zip(x, grid_sizes.tolist()):
            u = u[:math.prod(v)].float().contiguous()
            out = torch.empty(self.out_dim, v[0] * pt, v[1] * ph, v[2] * pw, dtype=torch.float32, device=u.device)
            _lib.call("uv_unpatchify_f32", _lib.ptr(u), u.stride(0), _lib.ptr(out), self.out_dim, v[0], v[1], v[2], pt, ph,
                      pw, _lib.stream_ptr())
            outs.append(out)
        return outs

    def init_weights(self, seed=0):
        """Deterministic synthetic init (there are no checkpoints offline); see univid_amd.detinit."""
        from .. import detinit
        detinit.init_module_(self, seed)
        self.invalidate()
        return self

    @classmethod
    def from_pretrained(cls, checkpoint_dir, subfolder=None, device="cpu"):
        """diffusers-layout directory (config.json + [sharded] safetensors), as the reference's ModelMixin.from_pretrained
        (models/wan/textimage2video.py:103)."""
        from .checkpoint import load_wan_model
        return load_wan_model(checkpoint_dir, device=device, subfolder=subfolder)

    @classmethod
    def from_config(cls, cfg: dict):
        keys = ("model_type", "patch_size", "text_len", "in_dim", "dim", "ffn_dim", "freq_dim", "text_dim", "out_dim",
                "num_heads", "num_layers", "window_size", "qk_norm", "cross_attn_norm", "eps")
        return cls(**{k: cfg[k] for k in keys if k in cfg})
