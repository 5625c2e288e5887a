"""Inference-side LoRA for the HIP DiT: PEFT adapter directories are read and FOLDED INTO the dense projection weights.

The reference wraps the DiT's `nn.Linear`s (`self_attn.{q,k,v,o}`, `cross_attn.{q,k,v,o}`, `ffn.0`, `ffn.2`) with PEFT
(`LoRAManager`, /root/reference/models/model_pipeline.py:325-835; `inference.py --use_lora`, :198-264) and keeps the adapters
un-merged: y = base(x) + lora_B(lora_A(x)) * scaling (peft==0.17.1, environment.yaml:417; `peft/tuners/lora/layer.py`
`Linear.forward`). The HIP path consumes ONE dense bf16 weight per projection (its GEMM epilogues - GELU, gated residual,
transposed V - are fused behind it), so an adapter is applied the way PEFT's own `merge_and_unload()` applies it
(`Linear.merge` / `get_delta_weight`): W <- W + scaling * (B @ A) on the fp32 master weights, then the bf16 operand copies are
rebuilt (`WanModel.invalidate()`). The merged projection differs from the un-merged one only by bf16 rounding placement (one
rounding of W + dW instead of separate roundings of the two branches); `tests/test_gpu_parity.py::test_lora_adapter_*` gates
it against the un-merged arithmetic restated in `oracle/lora.py`.

On-disk format read here (what `lora_model.save_pretrained(dir)` writes, model_pipeline.py:608-616, and the manual fallback of
:627-640): `adapter_config.json` (r, lora_alpha, use_rslora, use_dora, bias, fan_in_fan_out, target_modules) and
`adapter_model.safetensors` | `adapter_model.bin` | `lora_weights.pt` with keys
`base_model.model.<module path>.lora_A[.default].weight` [r, in] and `...lora_B[.default].weight` [out, r].
Training-side features (applying fresh adapters, gradients, saving) are out of scope.
"""
import json
import math
import os
import re
from typing import Dict, Tuple

import torch
from torch import nn


def read_adapter(load_path) -> Tuple[dict, Dict[str, torch.Tensor]]:
    """(adapter_config dict, raw tensor dict) of a PEFT adapter directory."""
    load_path = str(load_path)
    if not os.path.isdir(load_path):
        raise FileNotFoundError(f"LoRA weights path does not exist: {load_path}")
    cfg = {}
    for name in ("adapter_config.json", "lora_config.json"):       # PEFT's own file first, the reference's side file second
        p = os.path.join(load_path, name)
        if os.path.exists(p):
            with open(p) as f:
                cfg = json.load(f)
            break
    st = os.path.join(load_path, "adapter_model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        return cfg, load_file(st)
    for name in ("adapter_model.bin", "lora_weights.pt"):
        p = os.path.join(load_path, name)
        if os.path.exists(p):
            return cfg, torch.load(p, map_location="cpu", weights_only=True)
    raise FileNotFoundError(f"no adapter_model.safetensors / adapter_model.bin / lora_weights.pt under {load_path}")


_KEY = re.compile(r"^(?:base_model\.model\.)?(?P<mod>.+?)\.lora_(?P<ab>[AB])(?:\.[^.]+)?\.weight$")


def adapter_factors(tensors: Dict[str, torch.Tensor]) -> Dict[str, Tuple[torch.Tensor, torch.Tensor]]:
    """{module path: (A [r, in], B [out, r])} from PEFT state-dict keys (with or without the adapter name)."""
    ab: Dict[str, dict] = {}
    for k, v in tensors.items():
        m = _KEY.match(k)
        if m is None:
            if "lora_" in k:
                raise ValueError(f"unsupported LoRA tensor {k!r} (only lora_A / lora_B weights of Linear layers are handled; "
                                 f"DoRA magnitudes and embedding adapters are not)")
            continue
        ab.setdefault(m.group("mod"), {})[m.group("ab")] = v
    out = {}
    for mod, d in ab.items():
        if "A" not in d or "B" not in d:
            raise ValueError(f"adapter for {mod!r} is missing its lora_{'B' if 'A' in d else 'A'} weight")
        a, b = d["A"], d["B"]
        if a.dim() != 2 or b.dim() != 2 or a.shape[0] != b.shape[1]:
            raise ValueError(f"adapter for {mod!r}: lora_A {tuple(a.shape)} and lora_B {tuple(b.shape)} do not form a rank-r pair")
        out[mod] = (a, b)
    return out


def adapter_scaling(cfg: dict, r: int) -> float:
    """peft LoraLayer.update_layer: lora_alpha / r, or lora_alpha / sqrt(r) with use_rslora. The config MUST name lora_alpha: PEFT
    refuses an adapter directory without its config, and guessing alpha (= r, scaling 1) folds the adapter in at the wrong strength."""
    if "lora_alpha" not in cfg and "alpha" not in cfg:
        raise ValueError("the adapter's config (adapter_config.json / lora_config.json) is missing or does not name lora_alpha: "
                         "the merge scaling lora_alpha / r cannot be guessed")
    for pat in ("rank_pattern", "alpha_pattern"):
        if cfg.get(pat):
            raise NotImplementedError(f"adapter config has a non-empty {pat} (per-module rank / alpha): one global scaling would merge "
                                      f"those modules at the wrong strength")
    alpha = cfg.get("lora_alpha", cfg.get("alpha"))
    return alpha / math.sqrt(r) if cfg.get("use_rslora", False) else alpha / r


def merge_adapter_(model: nn.Module, factors, cfg: dict):
    """W += scaling * (B @ A) on the fp32 master weight of every targeted nn.Linear (peft Linear.get_delta_weight).
    Returns {module path: original weight clone} so that the merge can be undone bit-exactly."""
    if cfg.get("use_dora", False):
        raise NotImplementedError("DoRA adapters (use_dora=True) are not supported: only plain LoRA deltas can be folded in")
    if cfg.get("bias", "none") not in ("none", None):
        raise NotImplementedError(f"LoRA bias mode {cfg.get('bias')!r} is not supported (the reference uses 'none')")
    mods = dict(model.named_modules())
    saved = {}
    # validate EVERY (module, shape, scaling) before the first weight is touched: an error must not leave a half-merged model
    for name, (a, b) in factors.items():
        lin = mods.get(name)
        if not isinstance(lin, nn.Linear):
            raise KeyError(f"adapter targets {name!r}, which is not an nn.Linear of this model")
        if tuple(lin.weight.shape) != (b.shape[0], a.shape[1]):
            raise ValueError(f"adapter for {name!r} is {b.shape[0]} x {a.shape[1]}, the layer is {tuple(lin.weight.shape)}")
        adapter_scaling(cfg, a.shape[0])
    with torch.no_grad():
        for name, (a, b) in factors.items():
            lin = mods[name]
            w = lin.weight
            delta = (b.to(w.device, torch.float32) @ a.to(w.device, torch.float32)) * adapter_scaling(cfg, a.shape[0])
            if cfg.get("fan_in_fan_out", False):
                delta = delta.t()
            saved[name] = w.detach().clone()
            w.add_(delta.to(w.dtype))
    if hasattr(model, "invalidate"):
        model.invalidate()          # bf16 operand copies and the cached context K/V are rebuilt from the merged weights
    return saved


class LoRAManager:
    """Inference half of the reference's LoRAManager (model_pipeline.py:325-835): `load_lora_weights(path, model)` and
    `merge_and_unload()` with the same names; the adapter is merged at load (see the module docstring), `unload()` restores the
    base weights bit for bit. `apply_lora_to_dit` (fresh trainable adapters) belongs to training and raises."""

    def __init__(self, config=None, logger=None):
        self.config = config
        self.logger = logger
        self.lora_model = None
        self.original_model = None
        self.lora_config = None
        self.applied_modules = []
        self._saved = {}

    def apply_lora_to_dit(self, dit_model):
        raise NotImplementedError("applying fresh (trainable) LoRA adapters is training-side (SURVEY.md section 2, row 15); "
                                  "for inference call load_lora_weights(adapter_dir, model)")

    def load_lora_weights(self, load_path, model):
        """model_pipeline.py:724-750. Returns the model with the adapter folded in. Errors raise (the reference logs them and
        returns the un-adapted model, which silently generates with the wrong weights)."""
        if self._saved:
            raise RuntimeError("an adapter is already merged into this model: call unload() first")
        cfg, tensors = read_adapter(load_path)
        factors = adapter_factors(tensors)
        if not factors:
            raise ValueError(f"no lora_A / lora_B tensors found under {load_path}")
        self._saved = merge_adapter_(model, factors, cfg)
        self.lora_config = cfg
        self.applied_modules = sorted(factors)
        self.original_model = self.lora_model = model
        if self.logger is not None:
            r = next(iter(factors.values()))[0].shape[0]
            self.logger.info(f"LoRA adapter merged from {load_path}: {len(factors)} layers, rank {r}, "
                             f"scaling {adapter_scaling(cfg, r):g}")
        return model

    def merge_and_unload(self):
        """model_pipeline.py:752-764: the adapter is already merged; the dense model is returned and the saved base weights dropped."""
        if self.lora_model is None:
            raise RuntimeError("no LoRA model to merge")
        self._saved = {}
        return self.lora_model

    def unload(self):
        """Restores the base weights saved at load (bit for bit) and rebuilds the bf16 operands."""
        if self.lora_model is None:
            return None
        mods = dict(self.lora_model.named_modules())
        with torch.no_grad():
            for name, w in self._saved.items():
                mods[name].weight.copy_(w)
        self._saved = {}
        if hasattr(self.lora_model, "invalidate"):
            self.lora_model.invalidate()
        model, self.lora_model, self.applied_modules = self.lora_model, None, []
        return model

    def get_statistics(self):
        """model_pipeline.py:766-800 (the fields that exist without trainable parameters)."""
        if self.lora_model is None:
            return {}
        m = self.applied_modules
        return {
            "lora_modules": len(m),
            "module_breakdown": {"cross_attention": sum("cross_attn" in x for x in m), "self_attention": sum("self_attn" in x for x in m),
                                 "ffn": sum("ffn" in x for x in m), "total": len(m)},
            "lora_config": {"rank": self.lora_config.get("r"), "alpha": self.lora_config.get("lora_alpha"),
                            "use_rslora": self.lora_config.get("use_rslora", False), "use_dora": self.lora_config.get("use_dora", False)},
        }
