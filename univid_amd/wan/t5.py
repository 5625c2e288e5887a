"""umT5 text encoder on the MI355X kernels (reference models/wan/utils/modules/t5.py: T5Encoder :267-312, T5SelfAttention :144-177,
T5Attention :62-120, T5FeedForward :123-141, T5LayerNorm :53-66, T5RelativeEmbedding :205-264, T5EncoderModel :473-513).

The reference runs the encoder as a bf16 module (`dtype=torch.bfloat16`): bf16 parameters, every op rounded to bf16, softmax in
fp32. The same roundings are kept here: RMS norm = `uv_rmsnorm_rope` without RoPE (fp32 statistics, bf16 result, bf16 weight
product), projections = `uv_gemm_bf16_nt` on the bf16 parameters themselves (no operand copies), attention =
`uv_t5_attention_bf16` (scores rounded to bf16, + bf16 relative-position bias, no scaling, probabilities normalised before their bf16 rounding), residual adds = `uv_add_bf16`,
gated FFN = `uv_t5_gated_gelu_bf16` (the reference's op-by-op bf16 GELU). Only the valid token prefix of a prompt is computed:
padded keys are masked in the reference and the caller slices the output to the prompt length (t5.py:507-513).
Parameter names equal the reference's (`models_t5_umt5-xxl-enc-bf16.pth` loads key for key)."""
import math

import torch
import torch.nn as nn

from .. import _lib
from .._lib import EPI_BF16

BF16 = torch.bfloat16


def _round_up(a, b):
    return (a + b - 1) // b * b


class T5LayerNorm(nn.Module):
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Parameter(torch.ones(dim))


class T5Attention(nn.Module):
    def __init__(self, dim, dim_attn, num_heads):
        super().__init__()
        self.dim, self.dim_attn, self.num_heads, self.head_dim = dim, dim_attn, num_heads, dim_attn // num_heads
        self.q = nn.Linear(dim, dim_attn, bias=False)
        self.k = nn.Linear(dim, dim_attn, bias=False)
        self.v = nn.Linear(dim, dim_attn, bias=False)
        self.o = nn.Linear(dim_attn, dim, bias=False)


class T5FeedForward(nn.Module):
    def __init__(self, dim, dim_ffn):
        super().__init__()
        self.gate = nn.Sequential(nn.Linear(dim, dim_ffn, bias=False))     # + GELU (parameter-free; applied in the fused kernel)
        self.fc1 = nn.Linear(dim, dim_ffn, bias=False)
        self.fc2 = nn.Linear(dim_ffn, dim, bias=False)


class T5RelativeEmbedding(nn.Module):
    def __init__(self, num_buckets, num_heads, max_dist=128):
        super().__init__()
        self.num_buckets, self.num_heads, self.max_dist = num_buckets, num_heads, max_dist
        self.embedding = nn.Embedding(num_buckets, num_heads)

    def bucket(self, rel_pos):
        """_relative_position_bucket, bidirectional (t5.py:241-264)."""
        nb = self.num_buckets // 2
        rel_buckets = (rel_pos > 0).long() * nb
        rel_pos = torch.abs(rel_pos)
        max_exact = nb // 2
        large = max_exact + (torch.log(rel_pos.float() / max_exact) / math.log(self.max_dist / max_exact) * (nb - max_exact)).long()
        large = torch.min(large, torch.full_like(large, nb - 1))
        return rel_buckets + torch.where(rel_pos < max_exact, rel_pos, large)

    def table(self, span):
        """fp32 [H, 2*span - 1]: the (bf16) bias of relative position r = key - query at column r + span - 1."""
        w = self.embedding.weight
        rel = torch.arange(-(span - 1), span, device=w.device)
        return w.detach()[self.bucket(rel)].float().t().contiguous()


class T5SelfAttention(nn.Module):
    def __init__(self, dim, dim_attn, dim_ffn, num_heads, num_buckets):
        super().__init__()
        self.norm1 = T5LayerNorm(dim)
        self.attn = T5Attention(dim, dim_attn, num_heads)
        self.norm2 = T5LayerNorm(dim)
        self.ffn = T5FeedForward(dim, dim_ffn)
        self.pos_embedding = T5RelativeEmbedding(num_buckets, num_heads)     # shared_pos=False (umt5_xxl, t5.py:465)


class T5Encoder(nn.Module):
    def __init__(self, vocab, dim, dim_attn, dim_ffn, num_heads, num_layers, num_buckets, shared_pos=False, dropout=0.1):
        super().__init__()
        if shared_pos:
            raise NotImplementedError("shared_pos=True is not umT5's setting (t5.py:465)")
        if dim_attn // num_heads != 64 or dim % 8:
            raise NotImplementedError("the T5 attention kernel is built for head_dim 64")
        self.dim, self.dim_attn, self.dim_ffn, self.num_heads, self.num_layers = dim, dim_attn, dim_ffn, num_heads, num_layers
        self.token_embedding = vocab if isinstance(vocab, nn.Embedding) else nn.Embedding(vocab, dim)
        self.blocks = nn.ModuleList([T5SelfAttention(dim, dim_attn, dim_ffn, num_heads, num_buckets) for _ in range(num_layers)])
        self.norm = T5LayerNorm(dim)
        self._wf, self._tabs = None, {}

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._wf, self._tabs = None, {}
        return r

    def _norm_weights(self):
        if self._wf is None:      # the norm kernel takes its weight as fp32 (exact copies of the bf16 parameters)
            self._wf = [(b.norm1.weight.detach().float().contiguous(), b.norm2.weight.detach().float().contiguous()) for b in self.blocks]
            self._wf.append(self.norm.weight.detach().float().contiguous())
        return self._wf

    def _rms(self, x, w, eps):
        y = torch.empty_like(x)
        _lib.rmsnorm_rope(x, y, w, x.shape[0], x.shape[1], 64, eps)
        return y

    @torch.no_grad()
    def encode(self, ids):
        """ids [n] valid tokens of one prompt -> [n, dim] bf16 (= T5Encoder(ids_padded, mask)[0, :n] of the reference)."""
        if self.token_embedding.weight.dtype != BF16:
            raise NotImplementedError("the HIP umT5 encoder runs the module in bfloat16 (T5EncoderModel's dtype): call .to(torch.bfloat16)")
        dev = self.token_embedding.weight.device
        ids = ids.to(dev)
        n, d, H = ids.shape[0], self.dim, self.num_heads
        if n == 0:                    # a tokenizer that emits nothing (not even an end token) for "": the reference slices u[:0] (t5.py:512)
            return torch.empty(0, d, dtype=BF16, device=dev)
        wf = self._norm_weights()
        x = self.token_embedding.weight.detach()[ids].contiguous()
        span = _round_up(n, 512)      # one bias table per layer serves every prompt length up to text_len
        for li, blk in enumerate(self.blocks):
            key = (li, span)
            if key not in self._tabs:
                self._tabs[key] = blk.pos_embedding.table(span)
            y = self._rms(x, wf[li][0], blk.norm1.eps)
            q = torch.empty(n, self.dim_attn, dtype=BF16, device=dev)
            k = torch.empty(n, self.dim_attn, dtype=BF16, device=dev)
            v = torch.empty(n, self.dim_attn, dtype=BF16, device=dev)
            _lib.gemm_bf16(y, blk.attn.q.weight.detach(), None, q, EPI_BF16)
            _lib.gemm_bf16(y, blk.attn.k.weight.detach(), None, k, EPI_BF16)
            _lib.gemm_bf16(y, blk.attn.v.weight.detach(), None, v, EPI_BF16)
            att = torch.empty(n, self.dim_attn, dtype=BF16, device=dev)
            _lib.call("uv_t5_attention_bf16", _lib.ptr(q), q.stride(0), _lib.ptr(k), k.stride(0), _lib.ptr(v), v.stride(0), _lib.ptr(att),
                      att.stride(0), n, H, _lib.ptr(self._tabs[key]), span, _lib.stream_ptr())
            o = torch.empty(n, d, dtype=BF16, device=dev)
            _lib.gemm_bf16(att, blk.attn.o.weight.detach(), None, o, EPI_BF16)
            _lib.call("uv_add_bf16", _lib.ptr(x), _lib.ptr(o), _lib.ptr(x), x.numel(), _lib.stream_ptr())
            y = self._rms(x, wf[li][1], blk.norm2.eps)
            g = torch.empty(n, self.dim_ffn, dtype=BF16, device=dev)
            f = torch.empty(n, self.dim_ffn, dtype=BF16, device=dev)
            _lib.gemm_bf16(y, blk.ffn.gate[0].weight.detach(), None, g, EPI_BF16)
            _lib.gemm_bf16(y, blk.ffn.fc1.weight.detach(), None, f, EPI_BF16)
            _lib.call("uv_t5_gated_gelu_bf16", _lib.ptr(g), _lib.ptr(f), _lib.ptr(f), f.numel(), _lib.stream_ptr())
            _lib.gemm_bf16(f, blk.ffn.fc2.weight.detach(), None, o, EPI_BF16)
            _lib.call("uv_add_bf16", _lib.ptr(x), _lib.ptr(o), _lib.ptr(x), x.numel(), _lib.stream_ptr())
        return self._rms(x, wf[-1], self.norm.eps)

    def forward(self, ids, mask=None):
        """Reference signature (t5.py:301-311): ids [B, L], mask [B, L] -> [B, L, dim]; rows of padded positions are zero here
        (the reference computes values there that every caller discards)."""
        B, L = ids.shape
        out = torch.zeros(B, L, self.dim, dtype=BF16, device=self.token_embedding.weight.device)
        for b in range(B):
            n = L if mask is None else int(mask[b].gt(0).sum())
            if n:
                out[b, :n] = self.encode(ids[b, :n])
        return out


def umt5_xxl_encoder(**kw):
    """umt5_xxl(encoder_only=True) (t5.py:455-470)."""
    cfg = dict(vocab=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=24, num_buckets=32, shared_pos=False)
    cfg.update(kw)
    return T5Encoder(**cfg)


class T5EncoderModel:
    """t5.py:473-513: tokenizer_path= builds the HuggingfaceTokenizer (whitespace cleaning, padded / truncated to text_len) as the
    reference does, or inject tokenizer=; pass model= to reuse an encoder, or checkpoint_path= to load
    `models_t5_umt5-xxl-enc-bf16.pth`."""

    def __init__(self, text_len, dtype=torch.bfloat16, device="cuda", checkpoint_path=None, tokenizer_path=None, shard_fn=None, *,
                 tokenizer=None, model: T5Encoder = None, encoder_kwargs=None):
        if dtype != torch.bfloat16:
            raise NotImplementedError("the HIP umT5 encoder is the bf16 module of the reference configuration (config.t5_dtype)")
        if shard_fn is not None:
            raise NotImplementedError("FSDP sharding of the text encoder (t5_fsdp) is not part of this build")
        self.text_len, self.dtype, self.device = text_len, dtype, torch.device(device)
        if model is None:
            with torch.device(self.device):      # encoder_kwargs: T5Encoder arguments of a non-XXL checkpoint (tests)
                model = T5Encoder(**encoder_kwargs) if encoder_kwargs else umt5_xxl_encoder()
            if checkpoint_path is not None:
                model.load_state_dict(torch.load(checkpoint_path, map_location="cpu"))
        self.model = model.to(device=self.device, dtype=dtype).eval().requires_grad_(False)
        if tokenizer is None:
            if tokenizer_path is None:
                raise ValueError("pass tokenizer_path= (a Hugging Face tokenizer directory, e.g. google/umt5-xxl) or tokenizer= "
                                 "(texts -> (ids [B, text_len], mask [B, text_len]))")
            from .tokenizers import HuggingfaceTokenizer
            tokenizer = HuggingfaceTokenizer(name=tokenizer_path, seq_len=text_len, clean="whitespace")   # t5.py:509-510
        self.tokenizer = tokenizer

    def __call__(self, texts, device=None):
        ids, mask = self.tokenizer(texts, return_mask=True, add_special_tokens=True)
        seq_lens = mask.gt(0).sum(dim=1).long()
        return [self.model.encode(ids[i, :int(n)]) for i, n in enumerate(seq_lens)]
