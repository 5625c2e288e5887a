"""TEST INFRASTRUCTURE ONLY - never imported by the product package `univid_amd`.

Loads individual files of the UniVid reference (read-only mount at /root/reference) so that
`oracle/gen_golden.py` can (a) prove the CPU restatement in `oracle/` equal to the reference's own
modules and (b) emit the golden vectors committed under `tests/golden/`.

The reference only exists in the BUILD container; nothing here runs on the GPU box. The three shims
follow SURVEY.md section 8(c) / Appendix D:

  1. `torch.amp.autocast('cuda', ...)` is remapped to CPU autocast (bf16) / disabled (fp32 islands), so
     the dtype flow of the CUDA path is reproduced on CPU (decorators at model.py:27,38 run at import).
  2. `diffusers` (not installed) is replaced by inert mixin stubs (model.py:6-7,
     fm_solvers_unipc.py:10-16).
  3. `flash_attention` (attention.py:24-130, asserts CUDA + needs the flash-attn wheel) is replaced by
     a function that keeps its pre/post-processing (cast q/k/v to bf16 :59-83, cast result back to q's
     dtype :130, k_lens masking) and uses `F.scaled_dot_product_attention` as the core.
"""
import dataclasses
import enum
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = os.environ.get("UNIVID_REFERENCE", "/root/reference")
_MODDIR = os.path.join(REF_ROOT, "models/wan/utils/modules")
_UTILDIR = os.path.join(REF_ROOT, "models/wan/utils")


def available():
    return os.path.isfile(os.path.join(_MODDIR, "model.py"))


_installed = False
_real_autocast = torch.amp.autocast


class _AutocastRemap(_real_autocast):
    """autocast('cuda', dtype=bf16) -> autocast('cpu', bf16); fp32 / disabled -> disabled."""

    def __init__(self, device_type, dtype=None, enabled=True, cache_enabled=None):
        if device_type == "cuda":
            device_type = "cpu"
            if dtype == torch.float32 or dtype is None:
                # autocast(dtype=float32) on CUDA means "run autocast-eligible ops in fp32": with fp32
                # weights and fp32 inputs that is exactly "autocast off".
                enabled = False
                dtype = torch.bfloat16
        super().__init__(device_type, dtype=dtype, enabled=enabled, cache_enabled=cache_enabled)


def _install_shims():
    global _installed
    if _installed:
        return
    torch.amp.autocast = _AutocastRemap
    torch.amp.autocast_mode.autocast = _AutocastRemap

    d = types.ModuleType("diffusers")
    cu = types.ModuleType("diffusers.configuration_utils")
    mu = types.ModuleType("diffusers.models.modeling_utils")
    mm = types.ModuleType("diffusers.models")
    su = types.ModuleType("diffusers.schedulers.scheduling_utils")
    ss = types.ModuleType("diffusers.schedulers")
    uu = types.ModuleType("diffusers.utils")
    tu = types.ModuleType("diffusers.utils.torch_utils")

    class ConfigMixin:
        def register_to_config(self, **kw):
            for k, v in kw.items():
                setattr(self.config, k, v)

    def register_to_config(init):
        import functools
        import inspect

        sig = inspect.signature(init)

        @functools.wraps(init)
        def wrapper(self, *a, **kw):
            bound = sig.bind(self, *a, **kw)
            bound.apply_defaults()
            cfg = types.SimpleNamespace(**{k: v for k, v in bound.arguments.items() if k != "self"})
            object.__setattr__(self, "config", cfg) if not isinstance(self, nn.Module) else self.__dict__.__setitem__("config", cfg)
            return init(self, *a, **kw)

        return wrapper

    class ModelMixin(nn.Module):
        pass

    class SchedulerMixin:
        pass

    @dataclasses.dataclass
    class SchedulerOutput:
        prev_sample: torch.Tensor

    class KarrasDiffusionSchedulers(enum.Enum):
        UniPCMultistepScheduler = 1

    cu.ConfigMixin = ConfigMixin
    cu.register_to_config = register_to_config
    mu.ModelMixin = ModelMixin
    su.SchedulerMixin = SchedulerMixin
    su.SchedulerOutput = SchedulerOutput
    su.KarrasDiffusionSchedulers = KarrasDiffusionSchedulers
    uu.deprecate = lambda *a, **k: None
    uu.is_scipy_available = lambda: True
    tu.randn_tensor = lambda shape, generator=None, device=None, dtype=None: torch.randn(
        shape, generator=generator, device=device, dtype=dtype)
    for name, m in [("diffusers", d), ("diffusers.configuration_utils", cu), ("diffusers.models", mm),
                    ("diffusers.models.modeling_utils", mu), ("diffusers.schedulers", ss),
                    ("diffusers.schedulers.scheduling_utils", su), ("diffusers.utils", uu),
                    ("diffusers.utils.torch_utils", tu)]:
        sys.modules.setdefault(name, m)
    _installed = True


def _load(name, path, package=None):
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=None)
    mod = importlib.util.module_from_spec(spec)
    if package:
        mod.__package__ = package
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def fa_cpu(q, k, v, q_lens=None, k_lens=None, dropout_p=0.0, softmax_scale=None, q_scale=None, causal=False,
           window_size=(-1, -1), deterministic=False, dtype=torch.bfloat16, version=None):
    """CPU core for attention.py:24-130 keeping its casts; k_lens masks keys >= k_lens[b]."""
    half = (torch.float16, torch.bfloat16)
    out_dtype = q.dtype
    cast = lambda x: x if x.dtype in half else x.to(dtype)
    q, k, v = cast(q), cast(k), cast(v)
    q = q.to(v.dtype)
    k = k.to(v.dtype)
    if q_scale is not None:
        q = q * q_scale
    b, lq, lk = q.size(0), q.size(1), k.size(1)
    mask = None
    if k_lens is not None and int(k_lens.min()) < lk:
        mask = (torch.arange(lk)[None, :] < k_lens[:, None]).view(b, 1, 1, lk)
    out = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=mask,
                                         is_causal=causal, scale=softmax_scale)
    return out.transpose(1, 2).contiguous().type(out_dtype)


_cache = {}


def load_reference():
    """Returns a namespace with the reference's model / vae / scheduler modules."""
    if "ns" in _cache:
        return _cache["ns"]
    assert available(), f"reference not mounted at {REF_ROOT}"
    _install_shims()
    pkg = types.ModuleType("uvref_modules")
    pkg.__path__ = [_MODDIR]
    sys.modules["uvref_modules"] = pkg
    attention = _load("uvref_modules.attention", os.path.join(_MODDIR, "attention.py"), "uvref_modules")
    model = _load("uvref_modules.model", os.path.join(_MODDIR, "model.py"), "uvref_modules")
    model.flash_attention = fa_cpu
    vae = _load("uvref_modules.vae2_2", os.path.join(_MODDIR, "vae2_2.py"), "uvref_modules")
    unipc = _load("uvref_unipc", os.path.join(_UTILDIR, "fm_solvers_unipc.py"))
    dpm = _load("uvref_dpm", os.path.join(_UTILDIR, "fm_solvers.py"))
    ns = types.SimpleNamespace(attention=attention, model=model, vae=vae, unipc=unipc, dpm=dpm)
    _cache["ns"] = ns
    return ns


def ref_masks_like(tensor, zero=False):
    """Executes masks_like from models/wan/utils/utils.py:172-199 without importing the file's
    unrelated top-level dependencies (imageio / torchvision): only that function's source is compiled."""
    import ast
    src = open(os.path.join(_UTILDIR, "utils.py")).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "masks_like"][0]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "utils.py:masks_like", "exec"), ns)
    return ns["masks_like"](tensor, zero=zero)


def ref_mmr_select(embs, query_emb, K, lam=0.5):
    """Executes mmr_select from models/BAGEL/eval_understanding.py:225-240 without importing the file's top-level dependencies
    (decord, openai, PIL, ...): only that function's source is compiled."""
    import ast
    from typing import List
    path = os.path.join(REF_ROOT, "models", "BAGEL", "eval_understanding.py")
    tree = ast.parse(open(path).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "mmr_select"][0]
    ns = {"torch": torch, "List": List}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "eval_understanding.py:mmr_select", "exec"), ns)
    return ns["mmr_select"](embs, query_emb, K, lam)


def ref_context_projector(config):
    """Builds the reference's ContextProjector (models/model_pipeline.py:1506-1574) from its source alone (the module's file
    imports cv2 / torchvision / easydict at top level and cannot be imported here)."""
    import ast
    import contextlib
    import io
    from typing import Dict, List
    import torch.nn as nn
    import torch.nn.functional as F
    path = os.path.join(REF_ROOT, "models", "model_pipeline.py")
    tree = ast.parse(open(path).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "ContextProjector"][0]
    ns = {"torch": torch, "nn": nn, "F": F, "List": List, "Dict": Dict, "GLOBAL_TARGET_DTYPE": torch.bfloat16,
          "CrossAttentionConfig": object}
    exec(compile(ast.Module(body=[cls], type_ignores=[]), "model_pipeline.py:ContextProjector", "exec"), ns)
    with contextlib.redirect_stdout(io.StringIO()):
        return ns["ContextProjector"](config)


def ref_t5_encoder(cfg):
    """The reference's T5Encoder (models/wan/utils/modules/t5.py:267-312) built from its file with a stub for the sibling
    `tokenizers` module (which needs ftfy / regex data not in this image)."""
    import importlib.util
    import sys
    import types
    pkg = "uvref_t5pkg"
    if pkg not in sys.modules:
        mod_pkg = types.ModuleType(pkg)
        mod_pkg.__path__ = [_MODDIR]
        sys.modules[pkg] = mod_pkg
        tok = types.ModuleType(pkg + ".tokenizers")
        tok.HuggingfaceTokenizer = object
        sys.modules[pkg + ".tokenizers"] = tok
        spec = importlib.util.spec_from_file_location(pkg + ".t5", os.path.join(_MODDIR, "t5.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[pkg + ".t5"] = m
        real = torch.cuda.current_device          # t5.py:479 evaluates torch.cuda.current_device() in a default argument
        torch.cuda.current_device = lambda: 0
        try:
            spec.loader.exec_module(m)
        finally:
            torch.cuda.current_device = real
    t5 = sys.modules[pkg + ".t5"]
    return t5.T5Encoder(vocab=cfg["vocab_size"], dim=cfg["dim"], dim_attn=cfg["dim_attn"], dim_ffn=cfg["dim_ffn"],
                        num_heads=cfg["num_heads"], num_layers=cfg["num_layers"], num_buckets=cfg["num_buckets"], shared_pos=False,
                        dropout=0.1)
