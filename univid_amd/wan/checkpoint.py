"""Checkpoint (wire) formats of the hot path: diffusers-style DiT directories and the VAE .pth.

The reference loads the DiT with `WanModel.from_pretrained(checkpoint_dir)` (models/wan/textimage2video.py:103), i.e. the
diffusers `ModelMixin` layout: `config.json` (the `register_to_config` arguments of WanModel.__init__, model.py:304-320) next
to `diffusion_pytorch_model.safetensors`, or a sharded set `diffusion_pytorch_model-0000X-of-0000N.safetensors` with
`diffusion_pytorch_model.safetensors.index.json` ({"weight_map": {param: shard}}). The VAE is a plain `torch.save`d state
dict (`Wan2.2_VAE.pth`, vae2_2.py:877-883). Parameter names are identical here, so loading is a key-for-key copy.
"""
import json
import os

import torch

from .model import WanModel

WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"
INDEX_NAME = WEIGHTS_NAME + ".index.json"
_CONFIG_KEYS = ("model_type", "patch_size", "text_len", "in_dim", "dim", "ffn_dim", "freq_dim", "text_dim", "out_dim",
                "num_heads", "num_layers", "window_size", "qk_norm", "cross_attn_norm", "eps")


def load_wan_model(checkpoint_dir, device="cpu", subfolder=None) -> WanModel:
    """WanModel.from_pretrained equivalent. Parameters stay fp32 (UniVid's `convert_model_dtype=False`)."""
    from safetensors import safe_open
    root = os.path.join(checkpoint_dir, subfolder) if subfolder else checkpoint_dir
    with open(os.path.join(root, "config.json")) as f:
        raw = json.load(f)
    cfg = {k: raw[k] for k in _CONFIG_KEYS if k in raw}
    for k in ("patch_size", "window_size"):
        if k in cfg:
            cfg[k] = tuple(cfg[k])
    with torch.device(device):
        model = WanModel(**cfg)
    index = os.path.join(root, INDEX_NAME)
    if os.path.exists(index):
        with open(index) as f:
            shards = sorted(set(json.load(f)["weight_map"].values()))
    else:
        shards = [WEIGHTS_NAME]
    sd = {}
    for shard in shards:
        with safe_open(os.path.join(root, shard), framework="pt", device=str(device)) as f:
            for k in f.keys():
                sd[k] = f.get_tensor(k)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    if missing or unexpected:
        raise RuntimeError(f"checkpoint {root} does not match WanModel: missing {missing[:5]} unexpected {unexpected[:5]}")
    return model.float().eval()


def save_wan_model(model: WanModel, checkpoint_dir, max_shard_bytes=5 << 30):
    """Writes the same layout (used by the tests and to re-export converted weights)."""
    from safetensors.torch import save_file
    os.makedirs(checkpoint_dir, exist_ok=True)
    cfg = {k: (list(v) if isinstance(v, tuple) else v) for k, v in model.config.items()}
    cfg["_class_name"] = "WanModel"
    with open(os.path.join(checkpoint_dir, "config.json"), "w") as f:
        json.dump(cfg, f, indent=2)
    sd = {k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}
    shards, cur, size = [], {}, 0
    for k, v in sd.items():
        n = v.numel() * v.element_size()
        if cur and size + n > max_shard_bytes:
            shards.append(cur)
            cur, size = {}, 0
        cur[k] = v
        size += n
    shards.append(cur)
    if len(shards) == 1:
        save_file(shards[0], os.path.join(checkpoint_dir, WEIGHTS_NAME))
        return
    weight_map = {}
    for i, sh in enumerate(shards):
        name = f"diffusion_pytorch_model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
        save_file(sh, os.path.join(checkpoint_dir, name))
        weight_map.update({k: name for k in sh})
    with open(os.path.join(checkpoint_dir, INDEX_NAME), "w") as f:
        json.dump({"metadata": {"total_size": sum(v.numel() * v.element_size() for v in sd.values())}, "weight_map": weight_map}, f)


def load_vae_state_dict(vae_pth, device="cpu"):
    """`Wan2.2_VAE.pth` (vae2_2.py:877-883): a plain state dict whose keys equal univid_amd.wan.vae2_2.WanVAE_'s."""
    return torch.load(vae_pth, map_location=device)
