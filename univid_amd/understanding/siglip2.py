"""SigLIP2 (NaFlex) vision + text towers on the MI355X kernels of this package.

What UniVid's ranker calls (models/BAGEL/eval_understanding.py:175-192): `AutoModel.from_pretrained(ckpt)` ->
`get_text_features(**inputs)` / `get_image_features(**inputs)`. The towers' arithmetic is HF transformers'
`models/siglip2/modeling_siglip2.py` (pinned 4.56.1 by the reference): patch-embedding Linear + resized position table, pre-LN
encoder layers (MHA with bias, gelu_pytorch_tanh MLP), post LayerNorm, multi-head attention pooling with a learned probe;
text: token + position embeddings, the same encoder without a causal mask, final LayerNorm, LAST token, `head` Linear.

Parameter names and shapes are HF's, so a `Siglip2Model` state dict loads key for key. Compute: the DiT's kernels -
`uv_gemm_{f16,bf16}_nt` (bias / GELU-tanh / fp32-residual / transposed-V epilogues), `uv_flash_attn_{f16,bf16}` (head_dim 64),
`uv_layernorm_mod` (affine). Operands are fp16 (the reference runs the HF model under fp16 autocast,
`Siglip2Scorer(dtype=torch.float16)`) or bf16, with fp32 accumulation and an fp32 residual stream. Padded patches / tokens are never computed: only the valid prefix of every sequence is
embedded, attended and pooled, which is what the reference's attention masks amount to.
"""
import json
import math
import os

import torch
import torch.nn as nn

from .. import _lib
from .._lib import EPI_BF16, EPI_BF16_T, EPI_F32_FROM_BF16, EPI_GELU_BF16, EPI_RESID_F32

BF16 = torch.bfloat16


def _round_up(a, b):
    return (a + b - 1) // b * b


class _W:
    """16-bit operand copies of one Linear (weight [N, K] with K padded to the GEMM's 64 granularity, bias)."""
    __slots__ = ("w", "b")

    def __init__(self, weight, bias, op):
        w = weight.detach()
        if w.shape[1] % 64:
            w = torch.nn.functional.pad(w, (0, 64 - w.shape[1] % 64))
        self.w = w.to(op).contiguous()
        self.b = None if bias is None else bias.detach().to(op).contiguous()


class Siglip2MLP(nn.Module):
    def __init__(self, hidden, inter):
        super().__init__()
        self.fc1 = nn.Linear(hidden, inter)
        self.fc2 = nn.Linear(inter, hidden)


class Siglip2Attention(nn.Module):
    def __init__(self, hidden, heads):
        super().__init__()
        self.num_heads, self.head_dim = heads, hidden // heads
        self.k_proj = nn.Linear(hidden, hidden)
        self.v_proj = nn.Linear(hidden, hidden)
        self.q_proj = nn.Linear(hidden, hidden)
        self.out_proj = nn.Linear(hidden, hidden)


class Siglip2EncoderLayer(nn.Module):
    def __init__(self, hidden, inter, heads, eps):
        super().__init__()
        self.layer_norm1 = nn.LayerNorm(hidden, eps=eps)
        self.self_attn = Siglip2Attention(hidden, heads)
        self.layer_norm2 = nn.LayerNorm(hidden, eps=eps)
        self.mlp = Siglip2MLP(hidden, inter)
        self._p = None
        self._op = BF16

    def prepare(self, op=BF16):
        a = self.self_attn
        self._op = op
        self._p = {n: _W(getattr(a, n).weight, getattr(a, n).bias, op) for n in ("q_proj", "k_proj", "v_proj", "out_proj")}
        self._p["fc1"] = _W(self.mlp.fc1.weight, self.mlp.fc1.bias, op)
        self._p["fc2"] = _W(self.mlp.fc2.weight, self.mlp.fc2.bias, op)

    def run(self, x, batch, Lq, Lk):
        """x fp32 [batch*Lq, h] residual stream, updated in place; the keys / values of sample b are its first Lk rows."""
        p, a = self._p, self.self_attn
        h, H, D = x.shape[1], a.num_heads, a.head_dim
        dev = x.device
        BF16 = self._op          # operand dtype of this pass (bf16 or fp16); the name is kept for brevity below
        M = batch * Lq
        y = torch.empty(M, h, dtype=BF16, device=dev)
        _lib.layernorm_mod(x, y, M, h, self.layer_norm1.eps, mode=2, w=self.layer_norm1.weight, b=self.layer_norm1.bias)
        q = torch.empty(M, h, dtype=BF16, device=dev)
        _lib.gemm_bf16(y, p["q_proj"].w, p["q_proj"].b, q, EPI_BF16)
        att = torch.empty(M, h, dtype=BF16, device=dev)
        stacked = batch > 1 and Lq == Lk and Lk % 8 == 0
        if stacked or batch == 1:
            yk = y if Lq == Lk else y[:Lk]
            k = torch.empty(batch * Lk, h, dtype=BF16, device=dev)
            vt = torch.zeros(h, (batch - 1) * Lk + _round_up(Lk, 64), dtype=BF16, device=dev)
            _lib.gemm_bf16(yk, p["k_proj"].w, p["k_proj"].b, k, EPI_BF16, M=batch * Lk)
            _lib.gemm_bf16(yk, p["v_proj"].w, p["v_proj"].b, vt, EPI_BF16_T, M=batch * Lk)
            _lib.flash_attn(q, k, vt, att, Lq, Lk, H, D, D ** -0.5, batch=batch)
        else:   # ragged key counts or unaligned lengths: one attention problem per sample
            for b_ in range(batch):
                yb = y[b_ * Lq:b_ * Lq + Lk]
                k = torch.empty(Lk, h, dtype=BF16, device=dev)
                vt = torch.zeros(h, _round_up(Lk, 64), dtype=BF16, device=dev)
                _lib.gemm_bf16(yb, p["k_proj"].w, p["k_proj"].b, k, EPI_BF16)
                _lib.gemm_bf16(yb, p["v_proj"].w, p["v_proj"].b, vt, EPI_BF16_T)
                _lib.flash_attn(q[b_ * Lq:(b_ + 1) * Lq], k, vt, att[b_ * Lq:(b_ + 1) * Lq], Lq, Lk, H, D, D ** -0.5)
        _lib.gemm_bf16(att, p["out_proj"].w, p["out_proj"].b, x, EPI_RESID_F32)
        _lib.layernorm_mod(x, y, M, h, self.layer_norm2.eps, mode=2, w=self.layer_norm2.weight, b=self.layer_norm2.bias)
        mid = torch.empty(M, p["fc1"].w.shape[0], dtype=BF16, device=dev)
        _lib.gemm_bf16(y, p["fc1"].w, p["fc1"].b, mid, EPI_GELU_BF16)
        _lib.gemm_bf16(mid, p["fc2"].w, p["fc2"].b, x, EPI_RESID_F32)


class Siglip2Encoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layers = nn.ModuleList([Siglip2EncoderLayer(c["hidden_size"], c["intermediate_size"], c["num_attention_heads"],
                                                         c["layer_norm_eps"]) for _ in range(c["num_hidden_layers"])])


class Siglip2VisionEmbeddings(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.patch_embedding = nn.Linear(c["num_channels"] * c["patch_size"] ** 2, c["hidden_size"])
        self.position_embedding = nn.Embedding(c["num_patches"], c["hidden_size"])
        self._pe_cache = {}

    def positions(self, hw):
        """Position table resized to an (h, w) patch grid (resize_positional_embeddings: bilinear, antialias, fp32). The native
        grid is returned as is; other grids are resampled once per shape and cached (a weight-table preprocessing step)."""
        t = self.position_embedding.weight
        side = int(math.isqrt(t.shape[0]))
        if tuple(hw) == (side, side):
            return t.detach().float()
        key = (tuple(hw), t.data_ptr(), t._version)
        if key not in self._pe_cache:
            pe = t.detach().float().view(side, side, -1).permute(2, 0, 1).unsqueeze(0)
            pe = torch.nn.functional.interpolate(pe, size=tuple(hw), mode="bilinear", align_corners=False, antialias=True)
            self._pe_cache = {key: pe.reshape(t.shape[1], hw[0] * hw[1]).t().contiguous()}
        return self._pe_cache[key]


class Siglip2PoolingHead(nn.Module):
    def __init__(self, c):
        super().__init__()
        h = c["hidden_size"]
        self.probe = nn.Parameter(torch.randn(1, 1, h))
        self.attention = nn.MultiheadAttention(h, c["num_attention_heads"], batch_first=True)
        self.layernorm = nn.LayerNorm(h, eps=c["layer_norm_eps"])
        self.mlp = Siglip2MLP(h, c["intermediate_size"])
        self.num_heads = c["num_attention_heads"]


class Siglip2VisionModel(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.cfg = dict(c)
        self.embeddings = Siglip2VisionEmbeddings(c)
        self.encoder = Siglip2Encoder(c)
        self.post_layernorm = nn.LayerNorm(c["hidden_size"], eps=c["layer_norm_eps"])
        self.head = Siglip2PoolingHead(c)
        self._p = None
        self.op_dtype = BF16
        self.use_graph = False     # replay each (image count, grid) group from a captured HIP graph (see _graphed_group)
        self._graphs = {}

    def prepare(self, op=BF16):
        for l in self.encoder.layers:
            l.prepare(op)
        h = self.cfg["hidden_size"]
        self._op = op
        self._graphs = {}          # graphs hold pointers into the previous operand copies
        W, B = self.head.attention.in_proj_weight, self.head.attention.in_proj_bias
        self._p = {"patch": _W(self.embeddings.patch_embedding.weight, self.embeddings.patch_embedding.bias, op),
                   "hq": _W(W[:h], B[:h], op), "hk": _W(W[h:2 * h], B[h:2 * h], op), "hv": _W(W[2 * h:], B[2 * h:], op),
                   "ho": _W(self.head.attention.out_proj.weight, self.head.attention.out_proj.bias, op),
                   "hfc1": _W(self.head.mlp.fc1.weight, self.head.mlp.fc1.bias, op),
                   "hfc2": _W(self.head.mlp.fc2.weight, self.head.mlp.fc2.bias, op),
                   "probe": self.head.probe.detach().reshape(1, h).to(op).contiguous()}

    def pooled(self, pixel_values, pixel_attention_mask, spatial_shapes):
        """[B, N, 3*p*p], [B, N], [B, 2] -> pooler_output [B, h] fp32. Images with the same grid run as one stacked pass."""
        if self._p is None:
            self.prepare(self.op_dtype)
        dev = self.embeddings.patch_embedding.weight.device
        c, p = self.cfg, self._p
        BF16 = self._op
        h, H = c["hidden_size"], c["num_attention_heads"]
        D = h // H
        B = pixel_values.shape[0]
        shapes = [tuple(int(v) for v in s) for s in spatial_shapes.tolist()]
        out = torch.empty(B, h, dtype=torch.float32, device=dev)
        groups = {}
        for i, s in enumerate(shapes):
            groups.setdefault(s, []).append(i)
        for hw, idx in groups.items():
            n, G = hw[0] * hw[1], len(idx)
            if n > pixel_values.shape[1] or int(pixel_attention_mask[idx[0]].sum()) != n:
                raise ValueError("pixel_attention_mask does not mark exactly the first h*w patches of an image")
            ii = torch.tensor(idx, device=dev)
            px = pixel_values.to(dev)[ii, :n].reshape(G * n, -1).float().contiguous()
            if self.use_graph:
                out[ii] = self._graphed_group(px, hw, G)
            else:
                out[ii] = self._pooled_group(px, hw, G)
        return out

    def _graphed_group(self, px, hw, G):
        """One captured HIP graph per (image count, grid): the ~150 launches of a 12-layer tower take ~3-4 us of host time each,
        about as long as the kernels themselves at 64 frames x 256 patches; replaying them as one graph removes the gaps.
        Static input / output buffers; the kernels and their order are those of the eager path (bit-identical)."""
        key = (G, hw, px.shape[1], self._op, px.device)
        with torch.cuda.device(px.device):                      # torch's capture stream belongs to the current device
            ent = self._graphs.get(key)
            if ent is None:
                buf = torch.empty_like(px)
                buf.copy_(px)
                self._pooled_group(buf, hw, G)                  # eager warm-up: scratch, function attributes
                torch.cuda.synchronize(px.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    z = self._pooled_group(buf, hw, G)
                ent = self._graphs[key] = (g, buf, z)
            g, buf, z = ent
            buf.copy_(px)
            g.replay()
            return z.clone()

    def _pooled_group(self, px, hw, G):
        """px fp32 [G*n, 3*p*p]: the patches of G images with the same (h, w) grid -> pooler_output [G, h] fp32."""
        dev = px.device
        c, p = self.cfg, self._p
        BF16 = self._op
        h, H = c["hidden_size"], c["num_attention_heads"]
        D = h // H
        n = hw[0] * hw[1]
        Kp = p["patch"].w.shape[1]
        a = torch.zeros(G * n, Kp, dtype=BF16, device=dev)
        a[:, :px.shape[1]] = px.to(BF16)
        x = self.embeddings.positions(hw).to(dev).repeat(G, 1).contiguous()       # residual stream starts as the positions
        _lib.gemm_bf16(a, p["patch"].w, p["patch"].b, x, EPI_RESID_F32)            # + patch embedding
        for l in self.encoder.layers:
            l.run(x, G, n, n)
        y = torch.empty(G * n, h, dtype=BF16, device=dev)
        _lib.layernorm_mod(x, y, G * n, h, self.post_layernorm.eps, mode=2, w=self.post_layernorm.weight, b=self.post_layernorm.bias)
        # attention pooling: one probe query per image over its n tokens (nn.MultiheadAttention with packed in_proj)
        q1 = torch.empty(1, h, dtype=BF16, device=dev)
        _lib.gemm_bf16(p["probe"], p["hq"].w, p["hq"].b, q1, EPI_BF16)
        q = q1.expand(G, h).contiguous()
        att = torch.empty(G, h, dtype=BF16, device=dev)
        if G == 1 or n % 8 == 0:
            k = torch.empty(G * n, h, dtype=BF16, device=dev)
            vt = torch.zeros(h, (G - 1) * n + _round_up(n, 64), dtype=BF16, device=dev)
            _lib.gemm_bf16(y, p["hk"].w, p["hk"].b, k, EPI_BF16)
            _lib.gemm_bf16(y, p["hv"].w, p["hv"].b, vt, EPI_BF16_T)
            _lib.flash_attn(q, k, vt, att, 1, n, H, D, D ** -0.5, batch=G)
        else:
            for g in range(G):
                yb = y[g * n:(g + 1) * n]
                k = torch.empty(n, h, dtype=BF16, device=dev)
                vt = torch.zeros(h, _round_up(n, 64), dtype=BF16, device=dev)
                _lib.gemm_bf16(yb, p["hk"].w, p["hk"].b, k, EPI_BF16)
                _lib.gemm_bf16(yb, p["hv"].w, p["hv"].b, vt, EPI_BF16_T)
                _lib.flash_attn(q[g:g + 1], k, vt, att[g:g + 1], 1, n, H, D, D ** -0.5)
        z = torch.empty(G, h, dtype=torch.float32, device=dev)
        _lib.gemm_bf16(att, p["ho"].w, p["ho"].b, z, EPI_F32_FROM_BF16)
        zn = torch.empty(G, h, dtype=BF16, device=dev)
        _lib.layernorm_mod(z, zn, G, h, self.head.layernorm.eps, mode=2, w=self.head.layernorm.weight, b=self.head.layernorm.bias)
        mid = torch.empty(G, p["hfc1"].w.shape[0], dtype=BF16, device=dev)
        _lib.gemm_bf16(zn, p["hfc1"].w, p["hfc1"].b, mid, EPI_GELU_BF16)
        _lib.gemm_bf16(mid, p["hfc2"].w, p["hfc2"].b, z, EPI_RESID_F32)
        return z


class Siglip2TextEmbeddings(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.token_embedding = nn.Embedding(c["vocab_size"], c["hidden_size"])
        self.position_embedding = nn.Embedding(c["max_position_embeddings"], c["hidden_size"])


class Siglip2TextModel(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.cfg = dict(c)
        self.embeddings = Siglip2TextEmbeddings(c)
        self.encoder = Siglip2Encoder(c)
        self.final_layer_norm = nn.LayerNorm(c["hidden_size"], eps=c["layer_norm_eps"])
        self.head = nn.Linear(c["hidden_size"], c["projection_size"])
        self._p = None
        self.op_dtype = BF16

    def prepare(self, op=BF16):
        for l in self.encoder.layers:
            l.prepare(op)
        self._op = op
        self._p = {"head": _W(self.head.weight, self.head.bias, op)}

    def pooled(self, input_ids, attention_mask=None):
        """[B, T] token ids (right-padded to max_length by the tokenizer) -> [B, projection] fp32: last token's state -> head."""
        if self._p is None:
            self.prepare(self.op_dtype)
        BF16 = self._op
        dev = self.head.weight.device
        ids = input_ids.to(dev)
        B, T = ids.shape
        if T > self.embeddings.position_embedding.weight.shape[0]:
            raise ValueError("Sequence length must be less than max_position_embeddings")
        h = self.cfg["hidden_size"]
        out = torch.empty(B, self.cfg["projection_size"], dtype=torch.float32, device=dev)
        lens = [T] * B if attention_mask is None else [int(v) for v in attention_mask.sum(1).tolist()]
        for n in sorted(set(lens)):
            idx = [i for i, v in enumerate(lens) if v == n]
            G = len(idx)
            ii = torch.tensor(idx, device=dev)
            x = (self.embeddings.token_embedding.weight.detach()[ids[ii]].float() +
                 self.embeddings.position_embedding.weight.detach()[:T].float()).reshape(G * T, h).contiguous()
            for l in self.encoder.layers:
                l.run(x, G, T, n)
            y = torch.empty(G * T, h, dtype=BF16, device=dev)
            _lib.layernorm_mod(x, y, G * T, h, self.final_layer_norm.eps, mode=2, w=self.final_layer_norm.weight, b=self.final_layer_norm.bias)
            last = y.view(G, T, h)[:, -1].contiguous()
            z = torch.empty(G, out.shape[1], dtype=torch.float32, device=dev)
            _lib.gemm_bf16(last, self._p["head"].w, self._p["head"].b, z, EPI_F32_FROM_BF16)
            out[ii] = z
        return out


class Siglip2Model(nn.Module):
    """`transformers.Siglip2Model` surface used by the ranker: get_image_features / get_text_features, .to(device), .eval()."""

    def __init__(self, config: dict, dtype: torch.dtype = BF16):
        super().__init__()
        self.config = {"vision": dict(config["vision"]), "text": dict(config["text"])}
        for c in (self.config["vision"], self.config["text"]):
            if c["hidden_size"] % 256 or (c["hidden_size"] // c["num_attention_heads"]) not in (64, 128):
                raise NotImplementedError("hidden_size must be a multiple of 256 with head_dim 64 or 128 (SigLIP2 base: 768 = 12 x 64)")
        self.vision_model = Siglip2VisionModel(self.config["vision"])
        self.text_model = Siglip2TextModel(self.config["text"])
        self.logit_scale = nn.Parameter(torch.zeros(1))
        self.logit_bias = nn.Parameter(torch.zeros(1))
        self.set_operand_dtype(dtype)
        self.register_load_state_dict_post_hook(lambda m, _k: m.invalidate())

    def set_operand_dtype(self, dtype):
        """torch.bfloat16 or torch.float16: the 16-bit operand type of every GEMM / attention (fp32 accumulation and residual
        stream either way). The reference runs the HF model under fp16 autocast (eval_understanding.py:172)."""
        if dtype not in (torch.bfloat16, torch.float16):
            raise NotImplementedError(f"operand dtype {dtype}: bfloat16 or float16")
        self.vision_model.op_dtype = self.text_model.op_dtype = dtype
        self.invalidate()
        return self

    def invalidate(self):
        self.vision_model._p = None
        self.text_model._p = None

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.invalidate()
        return r

    def init_weights(self, seed=0):
        from .. import detinit
        detinit.init_state_dict_({k: v for k, v in self.state_dict(keep_vars=True).items() if not k.startswith("logit_")}, seed)
        self.invalidate()
        return self

    @torch.no_grad()
    def get_image_features(self, pixel_values, pixel_attention_mask, spatial_shapes, **_):
        return self.vision_model.pooled(pixel_values, pixel_attention_mask, spatial_shapes)

    @torch.no_grad()
    def get_text_features(self, input_ids, attention_mask=None, **_):
        return self.text_model.pooled(input_ids, attention_mask)

    @classmethod
    def from_pretrained(cls, ckpt_dir, device="cpu"):
        """HF checkpoint directory: config.json (vision_config / text_config) + model.safetensors (possibly sharded)."""
        from safetensors import safe_open
        with open(os.path.join(ckpt_dir, "config.json")) as f:
            raw = json.load(f)
        keys_v = ("hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads", "num_channels", "patch_size",
                  "num_patches", "layer_norm_eps")
        keys_t = ("hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads", "vocab_size",
                  "max_position_embeddings", "projection_size", "layer_norm_eps")
        dv = dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12, num_channels=3, patch_size=16,
                  num_patches=256, layer_norm_eps=1e-6)
        dt = dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12, vocab_size=32000,
                  max_position_embeddings=64, projection_size=None, layer_norm_eps=1e-6)
        v = {k: raw.get("vision_config", {}).get(k, dv[k]) for k in keys_v}
        t = {k: raw.get("text_config", {}).get(k, dt[k]) for k in keys_t}
        if t["projection_size"] is None:
            t["projection_size"] = t["hidden_size"]
        with torch.device(device):
            m = cls({"vision": v, "text": t})
        index = os.path.join(ckpt_dir, "model.safetensors.index.json")
        shards = sorted(set(json.load(open(index))["weight_map"].values())) if os.path.exists(index) else ["model.safetensors"]
        sd = {}
        for sh in shards:
            with safe_open(os.path.join(ckpt_dir, sh), framework="pt", device=str(device)) as f:
                for k in f.keys():
                    sd[k] = f.get_tensor(k)
        missing, unexpected = m.load_state_dict(sd, strict=False)
        if unexpected or [k for k in missing if not k.startswith("logit_")]:
            raise RuntimeError(f"checkpoint {ckpt_dir} does not match Siglip2Model: missing {missing[:5]} unexpected {unexpected[:5]}")
        return m.float().eval()
