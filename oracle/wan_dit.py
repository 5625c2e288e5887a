"""TEST INFRASTRUCTURE ONLY (the checker, never the product path; `univid_amd` must not import this).

CPU restatement of the Wan DiT forward as UniVid runs it: fp32 parameters, ambient autocast(bf16), fp32
islands. Written functionally over a state dict with the reference's key names,
with every autocast cast made explicit, so that the same text documents the dtype contract the HIP path
implements. Citations are to /root/reference/models/wan/utils/modules/model.py unless noted.

Pinned against the reference's own modules by `oracle/gen_golden.py` (bit-exact on CPU) and the vectors
under `tests/golden/`.
"""
import math

import torch
import torch.nn.functional as F

BF16 = torch.bfloat16


def linear_ac(x, w, b):
    """nn.Linear under autocast(bf16): input, weight and bias are cast to bf16, output is bf16."""
    return F.linear(x.to(BF16), w.to(BF16), None if b is None else b.to(BF16))


def lin(sd, key, x):
    """The nn.Linear `key` of the state dict under autocast(bf16). If the dict also holds `key.lora_A.weight` / `key.lora_B.weight`
    / `key.lora_scaling` (test-only extension: an UN-MERGED PEFT adapter on that layer, oracle/lora.py) the adapter branch is added
    the way PEFT's lora.Linear does."""
    if key + ".lora_A.weight" in sd:
        from . import lora
        return lora.lora_linear_ac(x, sd[key + ".weight"], sd[key + ".bias"], sd[key + ".lora_A.weight"], sd[key + ".lora_B.weight"],
                                   float(sd[key + ".lora_scaling"]))
    return linear_ac(x, sd[key + ".weight"], sd[key + ".bias"])


def sinusoidal_embedding_1d(dim, position):
    """model.py:14-24 - fp64 outer product, cos || sin."""
    half = dim // 2
    position = position.to(torch.float64)
    freqs = torch.pow(10000, -torch.arange(half).to(position).div(half))
    ang = torch.outer(position, freqs)
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=1)


def rope_params(max_seq_len, dim, theta=10000):
    """model.py:27-35 - complex128 unit phasors [max_seq_len, dim/2]."""
    ang = torch.outer(torch.arange(max_seq_len),
                      1.0 / torch.pow(theta, torch.arange(0, dim, 2).to(torch.float64).div(dim)))
    return torch.polar(torch.ones_like(ang), ang)


def rope_table(head_dim):
    """model.py:398-405 - the three axis tables concatenated along the complex column axis."""
    d = head_dim
    return torch.cat([rope_params(1024, d - 4 * (d // 6)), rope_params(1024, 2 * (d // 6)),
                      rope_params(1024, 2 * (d // 6))], dim=1)


def rope_apply(x, grid_sizes, freqs):
    """model.py:38-66 - x [B, L, N, D]; rotation in complex128, fp32 result; rows past f*h*w pass through."""
    n, c = x.size(2), x.size(3) // 2
    fa, fb, fc = freqs.to(x.device).split([c - 2 * (c // 3), c // 3, c // 3], dim=1)   # (device move: a no-op on the CPU)
    out = []
    for i, (f, h, w) in enumerate(grid_sizes.tolist()):
        s = f * h * w
        xi = torch.view_as_complex(x[i, :s].to(torch.float64).reshape(s, n, -1, 2))
        fi = torch.cat([fa[:f].view(f, 1, 1, -1).expand(f, h, w, -1), fb[:h].view(1, h, 1, -1).expand(f, h, w, -1),
                        fc[:w].view(1, 1, w, -1).expand(f, h, w, -1)], dim=-1).reshape(s, 1, -1)
        xi = torch.view_as_real(xi * fi).flatten(2)
        out.append(torch.cat([xi, x[i, s:]]))
    return torch.stack(out).float()


def rms_norm(x, weight, eps):
    """WanRMSNorm model.py:82-85: fp32 normalise, cast back to x's dtype, times the fp32 weight."""
    xf = x.float()
    y = xf * torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + eps)
    return y.type_as(x) * weight


def layer_norm(x, eps, weight=None, bias=None):
    """WanLayerNorm model.py:93-98: fp32 LayerNorm, cast back to x's dtype."""
    return F.layer_norm(x.float(), (x.shape[-1],), weight, bias, eps).type_as(x)


def attention_core(q, k, v, k_lens=None):
    """flash_attention attention.py:24-130 semantics: bf16 operands, fp32 softmax, result in q's dtype."""
    out_dtype = q.dtype
    b, lk = q.size(0), k.size(1)
    q, k, v = q.to(BF16), k.to(BF16), v.to(BF16)
    mask = None
    if k_lens is not None and int(k_lens.min()) < lk:
        mask = (torch.arange(lk)[None, :] < k_lens[:, None]).view(b, 1, 1, lk)
    o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=mask)
    return o.transpose(1, 2).contiguous().type(out_dtype)


def self_attention(sd, pre, x, seq_lens, grid_sizes, freqs, num_heads, eps):
    """WanSelfAttention.forward model.py:126-155."""
    b, s = x.shape[:2]
    d = x.shape[2] // num_heads
    q = rms_norm(lin(sd, pre + "q", x), sd[pre + "norm_q.weight"], eps).view(b, s, num_heads, d)
    k = rms_norm(lin(sd, pre + "k", x), sd[pre + "norm_k.weight"], eps).view(b, s, num_heads, d)
    v = lin(sd, pre + "v", x).view(b, s, num_heads, d)
    o = attention_core(rope_apply(q, grid_sizes, freqs), rope_apply(k, grid_sizes, freqs), v, k_lens=seq_lens)
    return lin(sd, pre + "o", o.flatten(2))


def cross_attention(sd, pre, x, context, num_heads, eps):
    """WanCrossAttention.forward model.py:160-180 (context_lens=None: all text_len rows attended)."""
    b = x.size(0)
    d = x.shape[2] // num_heads
    q = rms_norm(lin(sd, pre + "q", x), sd[pre + "norm_q.weight"], eps).view(b, -1, num_heads, d)
    k = rms_norm(lin(sd, pre + "k", context), sd[pre + "norm_k.weight"], eps).view(b, -1, num_heads, d)
    v = lin(sd, pre + "v", context).view(b, -1, num_heads, d)
    o = attention_core(q, k, v)
    return lin(sd, pre + "o", o.flatten(2))


def block_forward(sd, pre, x, e0, seq_lens, grid_sizes, freqs, context, num_heads, eps, context_scale=None):
    """WanAttentionBlock.forward model.py:219-259.

    context_scale: optional [B, text_len, 1] multiplier applied to the embedded context right before
    this block's cross-attention (UniVid's per-layer text-weight hook, models/model_pipeline.py:1787-1797).
    """
    e = (sd[pre + "modulation"].unsqueeze(0) + e0).chunk(6, dim=2)                           # :239 fp32
    h = layer_norm(x, eps).float() * (1 + e[1].squeeze(2)) + e[0].squeeze(2)                  # :244
    y = self_attention(sd, pre + "self_attn.", h, seq_lens, grid_sizes, freqs, num_heads, eps)
    x = x + y * e[2].squeeze(2)                                                              # :247 fp32
    ctx = context if context_scale is None else context * context_scale
    hn = layer_norm(x, eps, sd[pre + "norm3.weight"], sd[pre + "norm3.bias"])               # :251 cross_attn_norm
    x = x + cross_attention(sd, pre + "cross_attn.", hn, ctx, num_heads, eps)
    h = layer_norm(x, eps).float() * (1 + e[4].squeeze(2)) + e[3].squeeze(2)                  # :253
    y = lin(sd, pre + "ffn.0", h)
    y = F.gelu(y, approximate="tanh")
    y = lin(sd, pre + "ffn.2", y)
    return x + y * e[5].squeeze(2)                                                           # :255 fp32


def unpatchify(x, grid_sizes, patch_size, out_dim):
    """model.py:499-522."""
    out = []
    for u, v in zip(x, grid_sizes.tolist()):
        u = u[:math.prod(v)].view(*v, *patch_size, out_dim)
        u = torch.einsum("fhwpqrc->cfphqwr", u)
        out.append(u.reshape(out_dim, *[i * j for i, j in zip(v, patch_size)]))
    return out


def dit_forward(sd, cfg, x, t, context, seq_len, context_scale_fn=None, return_hidden=False):
    """WanModel.forward model.py:410-497 (t2v/ti2v: y=None).

    sd: state dict (fp32) with the reference key names; cfg: dict(dim, ffn_dim, num_heads, num_layers,
    freq_dim, text_len, text_dim, in_dim, out_dim, patch_size, eps).
    context_scale_fn(layer_idx) -> None | [B, text_len, 1] tensor (UniVid hook).
    """
    dim, nh, eps, ps = cfg["dim"], cfg["num_heads"], cfg["eps"], tuple(cfg["patch_size"])
    freqs = rope_table(dim // nh)
    pw, pb = sd["patch_embedding.weight"], sd["patch_embedding.bias"]
    # :448 Conv3d under autocast -> bf16
    xs = [F.conv3d(u.unsqueeze(0).to(BF16), pw.to(BF16), pb.to(BF16), stride=ps) for u in x]
    grid_sizes = torch.stack([torch.tensor(u.shape[2:], dtype=torch.long) for u in xs])
    xs = [u.flatten(2).transpose(1, 2) for u in xs]
    seq_lens = torch.tensor([u.size(1) for u in xs], dtype=torch.long)
    assert int(seq_lens.max()) <= seq_len
    xt = torch.cat([torch.cat([u, u.new_zeros(1, seq_len - u.size(1), u.size(2))], dim=1) for u in xs])

    if t.dim() == 1:
        t = t.expand(t.size(0), seq_len)
    bt = t.size(0)
    # :462-469 fp32 island
    emb = sinusoidal_embedding_1d(cfg["freq_dim"], t.flatten()).unflatten(0, (bt, seq_len)).float()
    e = F.linear(F.silu(F.linear(emb, sd["time_embedding.0.weight"], sd["time_embedding.0.bias"])),
                 sd["time_embedding.2.weight"], sd["time_embedding.2.bias"])
    e0 = F.linear(F.silu(e), sd["time_projection.1.weight"], sd["time_projection.1.bias"]).unflatten(2, (6, dim))

    # :472-478 text embedding on the zero-padded context (autocast -> bf16)
    ctx = torch.stack([torch.cat([u, u.new_zeros(cfg["text_len"] - u.size(0), u.size(1))]) for u in context])
    ctx = linear_ac(ctx, sd["text_embedding.0.weight"], sd["text_embedding.0.bias"])
    ctx = linear_ac(F.gelu(ctx, approximate="tanh"), sd["text_embedding.2.weight"], sd["text_embedding.2.bias"])

    hidden = []
    for i in range(cfg["num_layers"]):
        cs = None if context_scale_fn is None else context_scale_fn(i)
        xt = block_forward(sd, f"blocks.{i}.", xt, e0, seq_lens, grid_sizes, freqs, ctx, nh, eps, cs)
        if return_hidden:
            hidden.append(xt)

    # Head.forward :279-291 (fp32 island)
    eh = (sd["head.modulation"].unsqueeze(0) + e.unsqueeze(2)).chunk(2, dim=2)
    xh = F.linear(layer_norm(xt, eps) * (1 + eh[1].squeeze(2)) + eh[0].squeeze(2), sd["head.head.weight"],
                  sd["head.head.bias"])
    out = [u.float() for u in unpatchify(xh, grid_sizes, ps, cfg["out_dim"])]
    if return_hidden:
        return out, hidden, dict(e=e, e0=e0, ctx=ctx, x0=None)
    return out


def state_dict_shapes(cfg):
    """Key -> shape of a WanModel state dict (model.py:378-395), used to build synthetic weights."""
    d, f, td, fd = cfg["dim"], cfg["ffn_dim"], cfg["text_dim"], cfg["freq_dim"]
    ps = tuple(cfg["patch_size"])
    s = {
        "patch_embedding.weight": (d, cfg["in_dim"], *ps), "patch_embedding.bias": (d,),
        "text_embedding.0.weight": (d, td), "text_embedding.0.bias": (d,),
        "text_embedding.2.weight": (d, d), "text_embedding.2.bias": (d,),
        "time_embedding.0.weight": (d, fd), "time_embedding.0.bias": (d,),
        "time_embedding.2.weight": (d, d), "time_embedding.2.bias": (d,),
        "time_projection.1.weight": (6 * d, d), "time_projection.1.bias": (6 * d,),
    }
    for i in range(cfg["num_layers"]):
        p = f"blocks.{i}."
        s[p + "modulation"] = (1, 6, d)
        for a in ("self_attn.", "cross_attn."):
            for n in ("q", "k", "v", "o"):
                s[p + a + n + ".weight"] = (d, d)
                s[p + a + n + ".bias"] = (d,)
            s[p + a + "norm_q.weight"] = (d,)
            s[p + a + "norm_k.weight"] = (d,)
        s[p + "norm3.weight"] = (d,)
        s[p + "norm3.bias"] = (d,)
        s[p + "ffn.0.weight"] = (f, d)
        s[p + "ffn.0.bias"] = (f,)
        s[p + "ffn.2.weight"] = (d, f)
        s[p + "ffn.2.bias"] = (d,)
    s["head.head.weight"] = (math.prod(ps) * cfg["out_dim"], d)
    s["head.head.bias"] = (math.prod(ps) * cfg["out_dim"],)
    s["head.modulation"] = (1, 2, d)
    return s


def make_state_dict(cfg, seed=0):
    from univid_amd import detinit
    sd = {k: torch.empty(v, dtype=torch.float32) for k, v in state_dict_shapes(cfg).items()}
    return detinit.init_state_dict_(sd, seed)


TINY_CFG = dict(dim=256, ffn_dim=512, num_heads=4, num_layers=2, freq_dim=256, text_len=32, text_dim=64, in_dim=48,
                out_dim=48, patch_size=(1, 2, 2), eps=1e-6)
TI2V_5B_CFG = dict(dim=3072, ffn_dim=14336, num_heads=24, num_layers=30, freq_dim=256, text_len=512, text_dim=4096,
                   in_dim=48, out_dim=48, patch_size=(1, 2, 2), eps=1e-6)  # configs/wan_ti2v_5B.py:17-29
