"""CPU restatement of the umT5 text encoder forward (models/wan/utils/modules/t5.py:267-312 with :46-120, :123-141, :144-177,
:205-264) - TEST INFRASTRUCTURE ONLY. The reference runs the module in bf16 (`T5EncoderModel(dtype=torch.bfloat16)`, t5.py:476-
491): every op below is the same torch op on bf16 tensors, so the restatement is bit-identical to the reference module on the same
host (`oracle/gen_golden.py t5` asserts it). Only the valid token prefix of a prompt matters: padded keys are masked and the
caller slices the output to the prompt length (t5.py:507-513)."""
import math

import torch
import torch.nn.functional as F

TINY_CFG = dict(vocab_size=100, dim=256, dim_attn=256, dim_ffn=512, num_heads=4, num_layers=2, num_buckets=32)
UMT5_XXL_CFG = dict(vocab_size=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=24, num_buckets=32)


def state_dict_shapes(cfg):
    d, da, df, H = cfg["dim"], cfg["dim_attn"], cfg["dim_ffn"], cfg["num_heads"]
    s = {"token_embedding.weight": (cfg["vocab_size"], d), "norm.weight": (d,)}
    for i in range(cfg["num_layers"]):
        p = f"blocks.{i}."
        s[p + "norm1.weight"] = (d,)
        for n, shp in (("q", (da, d)), ("k", (da, d)), ("v", (da, d)), ("o", (d, da))):
            s[p + f"attn.{n}.weight"] = shp
        s[p + "norm2.weight"] = (d,)
        s[p + "ffn.gate.0.weight"], s[p + "ffn.fc1.weight"], s[p + "ffn.fc2.weight"] = (df, d), (df, d), (d, df)
        s[p + "pos_embedding.embedding.weight"] = (cfg["num_buckets"], H)
    return s


def make_state_dict(cfg, seed=0):
    from univid_amd import detinit
    sd = {k: torch.empty(v, dtype=torch.float32) for k, v in state_dict_shapes(cfg).items()}
    detinit.init_state_dict_(sd, seed)
    for k in sd:
        if k.endswith("pos_embedding.embedding.weight"):
            sd[k] *= 4.0          # visible relative-position biases
    return {k: v.to(torch.bfloat16) for k, v in sd.items()}


def relative_position_bucket(rel_pos, num_buckets=32, max_dist=128):
    """T5RelativeEmbedding._relative_position_bucket, bidirectional (t5.py:241-264)."""
    nb = num_buckets // 2
    rel_buckets = (rel_pos > 0).long() * nb
    rel_pos = torch.abs(rel_pos)
    max_exact = nb // 2
    large = max_exact + (torch.log(rel_pos.float() / max_exact) / math.log(max_dist / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return rel_buckets + torch.where(rel_pos < max_exact, rel_pos, large)


def _norm(x, w, eps=1e-6):
    x = x * torch.rsqrt(x.float().pow(2).mean(dim=-1, keepdim=True) + eps)
    return w * x.type_as(w)


def _gelu(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def encode(sd, cfg, ids):
    """ids [n] (the valid tokens of ONE prompt) -> [n, dim] bf16 = T5Encoder(ids_padded, mask)[0, :n]."""
    H = cfg["num_heads"]
    x = sd["token_embedding.weight"][ids]
    n = x.shape[0]
    rel = torch.arange(n).unsqueeze(0) - torch.arange(n).unsqueeze(1)
    buckets = relative_position_bucket(rel, cfg["num_buckets"])
    for i in range(cfg["num_layers"]):
        p = f"blocks.{i}."
        e = sd[p + "pos_embedding.embedding.weight"][buckets].permute(2, 0, 1)            # [H, n, n]
        y = _norm(x, sd[p + "norm1.weight"])
        c = cfg["dim_attn"] // H
        q = F.linear(y, sd[p + "attn.q.weight"]).view(n, H, c)
        k = F.linear(y, sd[p + "attn.k.weight"]).view(n, H, c)
        v = F.linear(y, sd[p + "attn.v.weight"]).view(n, H, c)
        attn = torch.einsum("inc,jnc->nij", q, k) + e
        attn = F.softmax(attn.float(), dim=-1).type_as(attn)
        a = torch.einsum("nij,jnc->inc", attn, v).reshape(n, H * c)
        x = x + F.linear(a, sd[p + "attn.o.weight"])
        y = _norm(x, sd[p + "norm2.weight"])
        x = x + F.linear(F.linear(y, sd[p + "ffn.fc1.weight"]) * _gelu(F.linear(y, sd[p + "ffn.gate.0.weight"])), sd[p + "ffn.fc2.weight"])
    return _norm(x, sd["norm.weight"])
