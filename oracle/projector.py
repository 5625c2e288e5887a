"""CPU restatement of UniVid's ContextProjector forward (models/model_pipeline.py:1506-1574) - TEST INFRASTRUCTURE ONLY.

The reference module is `nn.Sequential(Linear, LayerNorm, GELU, Dropout, Linear, LayerNorm).to(bfloat16)` run on bf16 input
outside autocast, followed by `F.interpolate(x^T, size=wan_text_length, mode='linear', align_corners=False)^T`. Every torch op on
a bf16 tensor computes in fp32 and rounds once, which is what this restatement spells out (and what `oracle/gen_golden.py
projector` checks bit for bit against the reference class, executed from its source)."""
import torch
import torch.nn.functional as F


def forward(sd, tokens, target_len):
    """sd: bf16 state dict with keys bagel_to_t5_projector.{0,1,4,5}.{weight,bias}; tokens [B, L, D] -> list of [target_len, Dout] bf16."""
    bf = torch.bfloat16
    p = "bagel_to_t5_projector."
    x = tokens.to(bf)
    x = F.linear(x, sd[p + "0.weight"], sd[p + "0.bias"])
    x = F.layer_norm(x, (x.shape[-1],), sd[p + "1.weight"], sd[p + "1.bias"], 1e-5)
    x = F.gelu(x)
    x = F.linear(x, sd[p + "4.weight"], sd[p + "4.bias"])
    x = F.layer_norm(x, (x.shape[-1],), sd[p + "5.weight"], sd[p + "5.bias"], 1e-5)
    if x.shape[1] != target_len:
        x = F.interpolate(x.transpose(1, 2), size=target_len, mode="linear", align_corners=False).transpose(1, 2)
    return [x[b] for b in range(x.shape[0])]


def make_state_dict(bagel_dim, text_dim, seed=0):
    from univid_amd import detinit
    names = {"0.weight": ("proj.lin1.weight", (2 * text_dim, bagel_dim)), "0.bias": ("proj.lin1.bias", (2 * text_dim,)),
             "1.weight": ("proj.norm1.weight", (2 * text_dim,)), "1.bias": ("proj.norm1.bias", (2 * text_dim,)),
             "4.weight": ("proj.lin2.weight", (text_dim, 2 * text_dim)), "4.bias": ("proj.lin2.bias", (text_dim,)),
             "5.weight": ("proj.norm2.weight", (text_dim,)), "5.bias": ("proj.norm2.bias", (text_dim,))}
    sd = {}
    for k, (alias, shape) in names.items():
        t = torch.empty(shape, dtype=torch.float32)
        detinit.fill_(alias, t, seed)
        sd["bagel_to_t5_projector." + k] = t.to(torch.bfloat16)
    return sd
