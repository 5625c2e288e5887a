"""CPU restatement of the SigLIP2 ranker of UniVid's understanding path (BASELINE.json config 5) - TEST INFRASTRUCTURE ONLY.

Reference call sites: models/BAGEL/eval_understanding.py:171-206 (`Siglip2Scorer.emb_text / emb_imgs / rank_frames`) and
:225-240 (`mmr_select`). The arithmetic of the towers is not in the reference repo: it lives in HF `transformers`
(pinned 4.56.1 in the reference's environment.yaml; 5.15.0 in this image) - `models/siglip2/modeling_siglip2.py`:
`Siglip2VisionEmbeddings`, `Siglip2EncoderLayer` (pre-LN, MHA, gelu_pytorch_tanh MLP), `Siglip2MultiheadAttentionPoolingHead`,
`Siglip2TextModel` (no causal mask, pooled = LAST token, then `head`). This file restates that algorithm in plain fp32 torch
ops on a state dict with HF's parameter names; `oracle/gen_golden.py siglip2` pins it against `transformers.Siglip2Model` itself
(constructed from a config with the same deterministic weights) in the build container and stores the vectors in tests/golden/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import math

import torch
import torch.nn.functional as F

TINY_CFG = dict(
    vision=dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_channels=3, patch_size=4,
                num_patches=64, layer_norm_eps=1e-6),
    text=dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, vocab_size=97,
              max_position_embeddings=16, projection_size=256, layer_norm_eps=1e-6))   # head_dim 64, widths the kernels accept
# google/siglip2-base-patch16-naflex geometry (768-d, 12 layers, 12 heads x 64, patch 16, 256 patches)
BASE_CFG = dict(
    vision=dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12, num_channels=3, patch_size=16,
                num_patches=256, layer_norm_eps=1e-6),
    text=dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12, vocab_size=256000,
              max_position_embeddings=64, projection_size=768, layer_norm_eps=1e-6))


def _encoder_shapes(prefix, c, n_layers):
    h, f = c["hidden_size"], c["intermediate_size"]
    s = {}
    for i in range(n_layers):
        p = f"{prefix}.encoder.layers.{i}."
        for ln in ("layer_norm1", "layer_norm2"):
            s[p + ln + ".weight"] = (h,)
            s[p + ln + ".bias"] = (h,)
        for proj in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"self_attn.{proj}.weight"] = (h, h)
            s[p + f"self_attn.{proj}.bias"] = (h,)
        s[p + "mlp.fc1.weight"], s[p + "mlp.fc1.bias"] = (f, h), (f,)
        s[p + "mlp.fc2.weight"], s[p + "mlp.fc2.bias"] = (h, f), (h,)
    return s


def state_dict_shapes(cfg, towers=("vision", "text")):
    s = {}
    if "vision" in towers:
        c = cfg["vision"]
        h, f = c["hidden_size"], c["intermediate_size"]
        pd = c["num_channels"] * c["patch_size"] ** 2
        s["vision_model.embeddings.patch_embedding.weight"] = (h, pd)
        s["vision_model.embeddings.patch_embedding.bias"] = (h,)
        s["vision_model.embeddings.position_embedding.weight"] = (c["num_patches"], h)
        s.update(_encoder_shapes("vision_model", c, c["num_hidden_layers"]))
        s["vision_model.post_layernorm.weight"], s["vision_model.post_layernorm.bias"] = (h,), (h,)
        hp = "vision_model.head."
        s[hp + "probe"] = (1, 1, h)
        s[hp + "attention.in_proj_weight"], s[hp + "attention.in_proj_bias"] = (3 * h, h), (3 * h,)
        s[hp + "attention.out_proj.weight"], s[hp + "attention.out_proj.bias"] = (h, h), (h,)
        s[hp + "layernorm.weight"], s[hp + "layernorm.bias"] = (h,), (h,)
        s[hp + "mlp.fc1.weight"], s[hp + "mlp.fc1.bias"] = (f, h), (f,)
        s[hp + "mlp.fc2.weight"], s[hp + "mlp.fc2.bias"] = (h, f), (h,)
    if "text" in towers:
        c = cfg["text"]
        h = c["hidden_size"]
        s["text_model.embeddings.token_embedding.weight"] = (c["vocab_size"], h)
        s["text_model.embeddings.position_embedding.weight"] = (c["max_position_embeddings"], h)
        s.update(_encoder_shapes("text_model", c, c["num_hidden_layers"]))
        s["text_model.final_layer_norm.weight"], s["text_model.final_layer_norm.bias"] = (h,), (h,)
        s["text_model.head.weight"], s["text_model.head.bias"] = (c["projection_size"], h), (c["projection_size"],)
    return s


def make_state_dict(cfg, seed=0, towers=("vision", "text")):
    from univid_amd import detinit
    sd = {k: torch.empty(v, dtype=torch.float32) for k, v in state_dict_shapes(cfg, towers).items()}
    return detinit.init_state_dict_(sd, seed)


def _ln(x, sd, name, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def _mha(xq, xkv, wq, bq, wk, bk, wv, bv, wo, bo, heads):
    """xq [Lq, h], xkv [Lk, h] (only the attended keys are passed) -> [Lq, h]."""
    d = xq.shape[-1] // heads
    q = (xq @ wq.t() + bq).view(-1, heads, d).transpose(0, 1)
    k = (xkv @ wk.t() + bk).view(-1, heads, d).transpose(0, 1)
    v = (xkv @ wv.t() + bv).view(-1, heads, d).transpose(0, 1)
    p = torch.softmax(q @ k.transpose(1, 2) * d ** -0.5, -1)
    return (p @ v).transpose(0, 1).reshape(xq.shape[0], -1) @ wo.t() + bo


def _mlp(x, sd, p):
    y = F.gelu(x @ sd[p + "fc1.weight"].t() + sd[p + "fc1.bias"], approximate="tanh")   # hidden_act = gelu_pytorch_tanh
    return y @ sd[p + "fc2.weight"].t() + sd[p + "fc2.bias"]


def _encoder(x, n_keys, sd, prefix, c):
    """Siglip2EncoderLayer stack on ONE sequence x [L, h]; keys/values restricted to the first n_keys rows (the padding mask of
    create_bidirectional_mask: padded positions are never attended; they still produce (unused) query rows)."""
    for i in range(c["num_hidden_layers"]):
        p = f"{prefix}.encoder.layers.{i}."
        y = _ln(x, sd, p + "layer_norm1", c["layer_norm_eps"])
        a = p + "self_attn."
        x = x + _mha(y, y[:n_keys], sd[a + "q_proj.weight"], sd[a + "q_proj.bias"], sd[a + "k_proj.weight"], sd[a + "k_proj.bias"],
                     sd[a + "v_proj.weight"], sd[a + "v_proj.bias"], sd[a + "out_proj.weight"], sd[a + "out_proj.bias"],
                     c["num_attention_heads"])
        x = x + _mlp(_ln(x, sd, p + "layer_norm2", c["layer_norm_eps"]), sd, p + "mlp.")
    return x


def resized_position_embedding(table, hw):
    """Siglip2VisionEmbeddings.resize_positional_embeddings for one image: bilinear, align_corners=False, antialias=True, fp32."""
    side = int(math.isqrt(table.shape[0]))
    h, w = hw
    pe = table.view(side, side, -1).permute(2, 0, 1).unsqueeze(0).float()
    pe = F.interpolate(pe, size=(h, w), mode="bilinear", align_corners=False, antialias=True)
    return pe.reshape(table.shape[1], h * w).t()


def image_features(sd, cfg, pixel_values, pixel_attention_mask, spatial_shapes):
    """get_image_features of transformers 4.56 (the pooled output): pixel_values [B, N, 3*p*p], mask [B, N], shapes [B, 2] -> [B, h]."""
    c = cfg["vision"]
    out = []
    for b in range(pixel_values.shape[0]):
        n = int(pixel_attention_mask[b].sum())
        hh, ww = (int(v) for v in spatial_shapes[b])
        x = pixel_values[b].float() @ sd["vision_model.embeddings.patch_embedding.weight"].t() + sd["vision_model.embeddings.patch_embedding.bias"]
        pe = resized_position_embedding(sd["vision_model.embeddings.position_embedding.weight"], (hh, ww))
        pos = torch.cat([pe, pe[:1].expand(x.shape[0] - hh * ww, -1)], 0)     # padding rows get the first embedding
        x = _encoder(x + pos, n, sd, "vision_model", c)
        x = _ln(x, sd, "vision_model.post_layernorm", c["layer_norm_eps"])
        hp = "vision_model.head."
        h = c["hidden_size"]
        W, Bv = sd[hp + "attention.in_proj_weight"], sd[hp + "attention.in_proj_bias"]
        probe = sd[hp + "probe"].view(1, h)
        y = _mha(probe, x[:n], W[:h], Bv[:h], W[h:2 * h], Bv[h:2 * h], W[2 * h:], Bv[2 * h:], sd[hp + "attention.out_proj.weight"],
                 sd[hp + "attention.out_proj.bias"], c["num_attention_heads"])
        y = y + _mlp(_ln(y, sd, hp + "layernorm", c["layer_norm_eps"]), sd, hp + "mlp.")
        out.append(y[0])
    return torch.stack(out)


def text_features(sd, cfg, input_ids, attention_mask=None):
    """get_text_features: input_ids [B, T] (padded to max_length by the tokenizer) -> [B, projection]."""
    c = cfg["text"]
    out = []
    for b in range(input_ids.shape[0]):
        T = input_ids.shape[1]
        x = sd["text_model.embeddings.token_embedding.weight"][input_ids[b]] + sd["text_model.embeddings.position_embedding.weight"][:T]
        n = T if attention_mask is None else int(attention_mask[b].sum())
        x = _encoder(x, n, sd, "text_model", c)
        x = _ln(x, sd, "text_model.final_layer_norm", c["layer_norm_eps"])
        out.append(x[-1] @ sd["text_model.head.weight"].t() + sd["text_model.head.bias"])     # last token, may be padding
    return torch.stack(out)


def rank_frames(img_feats, txt_feat, topk):
    """eval_understanding.py:196-206 after the towers: cosine similarity, top-k."""
    v = F.normalize(img_feats, dim=-1)
    t = F.normalize(txt_feat, dim=-1)
    sims = (v @ t.T).squeeze(-1).float()
    vals, idx = torch.topk(sims, k=min(topk, sims.shape[0]))
    return idx.tolist(), [float(x) for x in vals.tolist()]


def mmr_select(embs, query_emb, K, lam=0.5):
    """eval_understanding.py:225-240: greedy maximal-marginal-relevance selection (first maximum wins ties, ascending index)."""
    sims_q = (embs @ query_emb.T).squeeze(-1)
    sims_ii = embs @ embs.T
    N = embs.shape[0]
    selected, candidate = [], list(range(N))
    while len(selected) < min(K, N) and candidate:
        best_i, best_score = None, -1e9
        for i in candidate:
            div = 0.0 if not selected else torch.max(sims_ii[i, selected]).item()
            score = lam * sims_q[i].item() - (1.0 - lam) * div
            if score > best_score:
                best_score, best_i = score, i
        selected.append(best_i)
        candidate.remove(best_i)
    return selected
