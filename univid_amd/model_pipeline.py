"""UniVid's fusion-pipeline API surface over the MI355X hot path.

Mirrors the parts of /root/reference/models/model_pipeline.py that sit directly on the denoise path:
  * CrossAttentionConfig (:154-296)        - same field names and defaults (the fields inference.py:145-194 passes)
  * Wan22ContextWrapper (:1624-1900)       - per-layer cross-attention hook = dynamic text-weight schedule
                                             (:1699-1810) and the DiT-forward step counter (:1844-1886)
  * CrossAttentionFusionPipeline (:2110-)  - generate_video_with_bagel_context(text, image, steps=, guidance_scale=,
                                             frames=, size=, shift=, seed=) -> (video | None, path | None) (:2577-2655)
so that inference.py can drive this package unchanged. Everything off the hot path is pluggable instead of
re-built (SURVEY.md section 2 marks it out of scope): the BAGEL-7B semantic extractor, the ContextProjector adapter,
the umT5 text encoder, training and video file I/O are callables/objects the caller supplies. LoRA adapters are loaded
for inference by `univid_amd.lora.LoRAManager` (merged into the dense weights).

Behaviour kept on purpose: the text-encoder override of the reference is dead code (Python resolves
`obj(...)` through the type), so the DiT receives the text encoder's embeddings; the per-layer forward hook IS live
and rescales the first min(bagel_sequence_length, text_len // 2) rows of the embedded context by w(step), where
`step` counts DiT FORWARD calls (2 per sampler step).
"""
import contextlib
import logging
import math
import os
import time
from dataclasses import dataclass, field
from pathlib import Path
from typing import Callable, List, Optional, Tuple

import torch

from .lora import LoRAManager
from .wan.textimage2video import TI2VConfig, WanTI2V


@dataclass
class CrossAttentionConfig:
    bagel_model_path: str = os.getenv("BAGEL_MODEL_PATH", "your_bagel_model_path_here")
    wan_model_path: str = os.getenv("WAN_MODEL_PATH", "your_wan_model_path_here")
    bagel_gpu: int = 0
    wan_gpu: int = 1
    cross_attn_gpu: int = 2
    backup_gpu: int = 3
    fusion_mode: str = "context_replacement"
    enable_bagel_extraction: bool = True
    enable_wan_injection: bool = True
    bagel_sequence_length: int = 128
    wan_text_length: int = 512
    bagel_hidden_dim: int = 3584
    wan_text_dim: int = 4096
    use_lora: bool = True
    lora_rank: int = 8
    lora_alpha: int = 16
    lora_dropout: float = 0.1
    lora_target_strategy: str = "your_method_here"
    guidance_strength: float = 1.0
    bagel_cross_attn_layers: List[int] = None
    freeze_bagel: bool = True
    freeze_wan_vae: bool = True
    freeze_t5: bool = True
    skip_t5_loading: bool = True
    train_wan_dit: bool = True
    train_cross_attn: bool = True
    use_dynamic_text_weight: bool = True
    text_weight_max: float = 1.3
    text_weight_min: float = 1.0
    text_weight_schedule: str = "cosine"
    text_weight_transition_ratio: float = 0.4
    total_sampling_steps: int = 25
    use_bfloat16: bool = True
    enable_autocast: bool = True
    video_length: int = 121
    video_size: Tuple[int, int] = (1280, 704)
    video_fps: int = 8
    output_dir: str = "./cross_attention_outputs"
    save_video_mp4: bool = True
    save_tensor_backup: bool = True

    def __post_init__(self):
        if self.bagel_cross_attn_layers is None:
            self.bagel_cross_attn_layers = [8, 15, 22, 28]


class _TextWeightSchedule:
    """The state of one generation under UniVid's dynamic text weight, as WanTI2V.denoise reads it: the wrapper's forward counter
    (model_pipeline.py:1844-1886: reset per generate(), +1 per DiT forward, the weight of a forward = _calculate_text_weight(counter)),
    the hook's conditions (:1767-1773: BAGEL context set, dynamic weight on) evaluated when the forward runs, the number of scaled
    rows (:1790 min(bagel_sequence_length, context rows // 2)) and the hooked layers (`injection_layers`, None = all)."""

    def __init__(self, wrapper):
        self.wrapper = wrapper

    @property
    def layers(self):
        return self.wrapper.injection_layers

    def rows(self, context_rows):
        return min(self.wrapper.config.bagel_sequence_length, context_rows // 2)

    def next_weight(self):
        wr = self.wrapper
        wr.set_timestep(wr.sampling_step_counter)
        wr.sampling_step_counter += 1
        on = wr.use_bagel_context and wr.bagel_context is not None and wr.config.use_dynamic_text_weight
        return float(wr.text_weight_multiplier) if on else 1.0

    def next_pair(self):
        """Weights of the cond and the uncond forward of the next sampler step (the reference runs them in this order)."""
        wc = self.next_weight()
        return wc, self.next_weight()


class Wan22ContextWrapper:
    """model_pipeline.py:1624-1900. Wraps a WanTI2V and applies UniVid's dynamic text weight to every WanCrossAttention of its DiT.

    native=True (default): the schedule is handed to the loop as DATA - WanTI2V.denoise asks for the two weights of each step's CFG pair
    and the DiT scales the context rows in front of the hooked blocks' K / V projections (WanModel.set_text_weight; under the HIP graph
    the runner's K / V^T buffers are refreshed in place) - so the generation runs on the fast path (stacked CFG pair, cached context
    work, fused residual epilogue, graph replay) and is bit-identical to the closures' result. Nothing is re-assigned on the model.
    native=False: the reference's mechanism literally - a closure re-assigned as `forward` on every WanCrossAttention instance plus a
    counting closure as the DiT's `forward` during generate() (:1742-1810, 1856-1868) - which the model honours on its generic path.
    """

    def __init__(self, original_wan_pipeline, context_projector, logger, config: CrossAttentionConfig, native: bool = True):
        self.original_pipeline = original_wan_pipeline
        self.context_projector = context_projector
        self.logger = logger
        self.config = config
        self.native = bool(native)
        self.dit_model = original_wan_pipeline.model
        self.original_forward_methods = {}
        self.fusion_alpha = 1.0
        self.injection_layers = None
        self.bagel_context = None
        self.use_bagel_context = False
        self.current_timestep = None
        self.text_weight_multiplier = 1.0
        self._hook_cross_attention_layers()

    def _calculate_text_weight(self, timestep: int) -> float:
        """:1699-1735."""
        c = self.config
        if not c.use_dynamic_text_weight:
            return 1.0
        transition_steps = int(c.total_sampling_steps * c.text_weight_transition_ratio)
        if timestep >= transition_steps:
            return c.text_weight_min
        progress = timestep / max(transition_steps, 1)
        if c.text_weight_schedule == "linear":
            return c.text_weight_max - (c.text_weight_max - c.text_weight_min) * progress
        if c.text_weight_schedule == "cosine":
            return c.text_weight_min + (c.text_weight_max - c.text_weight_min) * (1 + math.cos(math.pi * progress)) / 2
        if c.text_weight_schedule == "exponential":
            return c.text_weight_min + (c.text_weight_max - c.text_weight_min) * math.exp(-5 * progress)
        return 1.0

    def set_timestep(self, timestep: int):
        self.current_timestep = timestep
        self.text_weight_multiplier = self._calculate_text_weight(timestep)

    def _hook_cross_attention_layers(self):
        """:1742-1810: finds every WanCrossAttention (by class name, in module order = layer index). native: records them - the
        scaling itself happens inside the DiT; otherwise re-assigns `forward` on every instance, as the reference does."""
        layer_idx = 0
        for name, module in self.dit_model.named_modules():
            if module.__class__.__name__ != "WanCrossAttention":
                continue
            original_forward = module.forward
            self.original_forward_methods[name] = original_forward
            if not self.native:
                module.forward = self._make_hook(layer_idx, original_forward)
            layer_idx += 1
        self.logger.info(f"Hooked {layer_idx} cross-attention layers ({'native schedule' if self.native else 'forward closures'})")

    def _make_hook(self, layer_index, original_fn):
        def hooked_forward(x, context, context_lens, *args, **kwargs):
            if (self.use_bagel_context and self.bagel_context is not None
                    and (self.injection_layers is None or layer_index in self.injection_layers)
                    and self.config.use_dynamic_text_weight and self.text_weight_multiplier != 1.0
                    and context is not None):
                seq_len = context.shape[1] if context.dim() > 1 else context.shape[0]
                text_len = min(self.config.bagel_sequence_length, seq_len // 2)
                weight_mask = torch.ones_like(context)
                if context.dim() == 3:
                    weight_mask[:, :text_len, :] *= self.text_weight_multiplier
                elif context.dim() == 2:
                    weight_mask[:text_len, :] *= self.text_weight_multiplier
                context = context * weight_mask
            return original_fn(x, context, context_lens, *args, **kwargs)
        return hooked_forward

    def set_bagel_context(self, bagel_tokens, fusion_alpha=None, injection_layers=None):
        self.bagel_context = self.context_projector(bagel_tokens) if self.context_projector is not None else bagel_tokens
        self.use_bagel_context = True
        if fusion_alpha is not None:
            self.fusion_alpha = fusion_alpha
        if injection_layers is not None:
            self.injection_layers = injection_layers

    def clear_bagel_context(self):
        self.bagel_context = None
        self.use_bagel_context = False

    def restore_original_methods(self):
        for name, module in self.dit_model.named_modules():
            if name in self.original_forward_methods:
                module.__dict__.pop("forward", None)
        self.original_forward_methods = {}

    @contextlib.contextmanager
    def scheduled(self):
        """The lifetime of the forward counter = one generation (:1851-1876): reset to 0, advanced by every DiT forward of the loop
        that runs inside, removed afterwards. native: the loop reads it through WanTI2V.text_weight_schedule; otherwise a counting
        closure is re-assigned as the DiT's `forward`, as in the reference."""
        if self.native:
            # the native schedule exists only inside WanTI2V.denoise, which reads `text_weight_schedule`: a wrapped pipeline without that
            # attribute (duck-typed / foreign) would silently generate with NO text weight, and a foreign re-assigned `forward` on the
            # DiT or a cross-attention makes the loop take its generic path, which does not read the schedule either
            if not hasattr(self.original_pipeline, "text_weight_schedule"):
                raise TypeError(f"{type(self.original_pipeline).__name__} does not read a native text-weight schedule (no `text_weight_schedule` "
                                f"attribute): wrap it with Wan22ContextWrapper(..., native=False) to install the reference's forward closures")
            if "forward" in self.dit_model.__dict__ or any("forward" in m.__dict__ for m in self.dit_model.modules()
                                                           if m.__class__.__name__ == "WanCrossAttention"):
                raise RuntimeError("a `forward` re-assigned by someone else is installed on the DiT / a WanCrossAttention: the loop would run "
                                   "its generic path and drop the native text-weight schedule - remove it or use native=False")
        self.sampling_step_counter = 0
        if self.native:
            self.original_pipeline.text_weight_schedule = _TextWeightSchedule(self)
        else:
            original_dit_forward = self.dit_model.forward
            wrapper_self = self

            def hooked_dit_forward(hidden_states, t, *args, **kw):
                wrapper_self.set_timestep(wrapper_self.sampling_step_counter)
                wrapper_self.sampling_step_counter += 1
                return original_dit_forward(hidden_states, t, *args, **kw)

            self.dit_model.forward = hooked_dit_forward
        try:
            yield self
        finally:
            if self.native:
                self.original_pipeline.text_weight_schedule = None
            else:
                self.dit_model.__dict__.pop("forward", None)
            del self.sampling_step_counter

    def generate(self, **kwargs):
        """:1844-1886: one generation under the forward counter that drives the text weight."""
        if not self.config.use_dynamic_text_weight:
            return self.original_pipeline.generate(**kwargs)
        with self.scheduled():
            return self.original_pipeline.generate(**kwargs)


class ContextProjector(torch.nn.Module):
    """BAGEL semantic tokens [B, L, 3584] -> list of Wan text-context tensors [512, 4096] (models/model_pipeline.py:1506-1574):
    Linear(3584 -> 8192) -> LayerNorm -> GELU (exact) -> Dropout (identity in eval) -> Linear(8192 -> 4096) -> LayerNorm, all in
    bf16 (`.to(dtype=GLOBAL_TARGET_DTYPE)`), then linear interpolation of the token axis to `wan_text_length`.

    Same module tree (`bagel_to_t5_projector.{0,1,4,5}`), so a trained projector's state dict (`training_state.pt
    ['context_projector']`, inference.py:227-236) loads unchanged. Compute: `uv_gemm_bf16_nt` (bias, bf16-rounded fp32 out),
    `uv_layernorm_mod` (affine, bf16 out), `uv_gelu_erf_bf16`, `uv_interp_linear_rows_bf16`."""

    def __init__(self, config):
        super().__init__()
        nn = torch.nn
        self.config = config
        self.bagel_dim, self.wan_text_dim = config.bagel_hidden_dim, config.wan_text_dim
        self.bagel_to_t5_projector = nn.Sequential(
            nn.Linear(self.bagel_dim, self.wan_text_dim * 2), nn.LayerNorm(self.wan_text_dim * 2), nn.GELU(), nn.Dropout(0.1),
            nn.Linear(self.wan_text_dim * 2, self.wan_text_dim), nn.LayerNorm(self.wan_text_dim)).to(dtype=torch.bfloat16)

    @torch.no_grad()
    def forward(self, bagel_tokens: torch.Tensor):
        from . import _lib
        from ._lib import EPI_F32_FROM_BF16
        if self.training:
            raise NotImplementedError("the HIP ContextProjector is inference-only (Dropout is the identity)")
        seq = self.bagel_to_t5_projector
        l1, n1, l2, n2 = seq[0], seq[1], seq[4], seq[5]
        dev = l1.weight.device
        B, L, _ = bagel_tokens.shape
        bf = torch.bfloat16
        out = []
        for b in range(B):
            x = bagel_tokens[b].to(device=dev, dtype=bf).contiguous()
            K1 = (x.shape[1] + 63) // 64 * 64
            w1 = l1.weight.detach()
            if K1 != x.shape[1]:      # the GEMM's K granularity is 64: zero-pad the operand columns
                x = torch.nn.functional.pad(x, (0, K1 - x.shape[1]))
                w1 = torch.nn.functional.pad(w1, (0, K1 - w1.shape[1]))
            h1 = torch.empty(L, w1.shape[0], dtype=torch.float32, device=dev)
            _lib.gemm_bf16(x, w1.contiguous(), l1.bias.detach(), h1, EPI_F32_FROM_BF16)        # bf16 Linear output, held as fp32
            y1 = torch.empty(L, w1.shape[0], dtype=bf, device=dev)
            _lib.layernorm_mod(h1, y1, L, w1.shape[0], n1.eps, mode=2, w=n1.weight.detach().float(), b=n1.bias.detach().float())
            _lib.call("uv_gelu_erf_bf16", _lib.ptr(y1), _lib.ptr(y1), y1.numel(), _lib.stream_ptr())
            h2 = torch.empty(L, l2.weight.shape[0], dtype=torch.float32, device=dev)
            _lib.gemm_bf16(y1, l2.weight.detach(), l2.bias.detach(), h2, EPI_F32_FROM_BF16)
            y2 = torch.empty(L, l2.weight.shape[0], dtype=bf, device=dev)
            _lib.layernorm_mod(h2, y2, L, l2.weight.shape[0], n2.eps, mode=2, w=n2.weight.detach().float(), b=n2.bias.detach().float())
            T = self.config.wan_text_length
            if L != T:
                z = torch.empty(T, y2.shape[1], dtype=bf, device=dev)
                _lib.call("uv_interp_linear_rows_bf16", _lib.ptr(y2), y2.stride(0), _lib.ptr(z), z.stride(0), L, T, y2.shape[1],
                          _lib.stream_ptr())
                y2 = z
            out.append(y2)
        return out


_BAGEL_EXTRACTOR_FACTORY: Optional[Callable] = None


def register_bagel_extractor(factory: Optional[Callable]):
    """The ONE hook-up point of the BAGEL-7B semantic extractor (reference `BagelSemanticExtractor`, model_pipeline.py:837-1503: a
    7 B multimodal LLM, SURVEY.md section 2 row 14 - outside the denoise hot path and not rebuilt here). `factory` is called by
    `CrossAttentionFusionPipeline(config)` exactly as the reference constructs its extractor (:2153-2158):

        factory(model_path=config.bagel_model_path, device_id=config.bagel_gpu, use_bfloat16=config.use_bfloat16, config=config)

    and must return an object with `extract_semantic_tokens(text, image_or_None) -> Tensor[1, L, bagel_hidden_dim]`
    (:1240). Passing the reference's own class registers the reference's extractor unchanged. `None` un-registers. Returns the
    previous factory."""
    global _BAGEL_EXTRACTOR_FACTORY
    prev, _BAGEL_EXTRACTOR_FACTORY = _BAGEL_EXTRACTOR_FACTORY, factory
    return prev


class CrossAttentionFusionPipeline:
    """model_pipeline.py:2110-3230, inference side: `CrossAttentionFusionPipeline(config)` as inference.py:196 constructs it.

    From `config` alone the constructor composes what the reference's `_initialize_components` (:2151-2174) composes:

        .wan_pipeline       WanTI2V(wan_config, checkpoint_dir=config.wan_model_path, device_id=config.wan_gpu)    (:2193-2204) - the
                            diffusers-layout DiT, Wan2.2_VAE.pth and the umT5 .pth + tokenizer of that directory, all on the HIP path
        .context_projector  the native ContextProjector(config) on cuda:{config.cross_attn_gpu}                     (:2160-2161) - an
                            nn.Module with the reference's tree, so inference.py:227-236's load_state_dict works
        .bagel_extractor    the registered factory's object (register_bagel_extractor) when config.enable_bagel_extraction  (:2153-2158)
        .lora_manager       LoRAManager(config) when config.use_lora                                                (:2117)
        .dit_model / .vae_model / .wan_wrapper / .text_encoder                                                      (:2206-2207,2166)

    Every component can instead be INJECTED (tests, services that already hold a WanTI2V): wan_pipeline=, bagel_extractor=,
    context_projector=. `wan_config` is the WAN_CONFIGS['ti2v-5B'] counterpart (TI2VConfig); a subclass names other checkpoint
    files / a width-reduced VAE or T5. Differences from the reference, on purpose: a missing extractor is an error naming the
    registration point, never a stub; the reference's fresh trainable adapters (`_apply_lora_to_dit`, :2254-2297) are training-side -
    an inference adapter arrives through `lora_manager.load_lora_weights` (inference.py:218-224); generation errors propagate.
    """

    def __init__(self, config: CrossAttentionConfig, wan_pipeline: Optional[WanTI2V] = None, bagel_extractor=None,
                 context_projector: Optional[Callable] = None, save_fn: Optional[Callable] = None, native_text_weight: bool = True,
                 wan_config=TI2VConfig):
        self.config = config
        self.logger = logging.getLogger("univid_amd.pipeline")
        # model_pipeline.py:2117: inference.py:218-224 calls `pipeline.lora_manager.load_lora_weights(path, pipeline.dit_model)`
        self.lora_manager = LoRAManager(config, self.logger) if getattr(config, "use_lora", False) else None
        if wan_pipeline is None and not os.path.exists(config.wan_model_path):     # :2178-2180, checked before anything is built
            raise FileNotFoundError(f"Wan2.2 model path not found: {config.wan_model_path}")
        if bagel_extractor is None and getattr(config, "enable_bagel_extraction", True) and wan_pipeline is None:
            # the injected form keeps its meaning (no extractor given = plain text-encoder context); the config-only form follows the
            # reference, which always builds one (:2153)
            if _BAGEL_EXTRACTOR_FACTORY is None:
                raise RuntimeError(
                    "config.enable_bagel_extraction is set and no BAGEL extractor is available: call "
                    "univid_amd.model_pipeline.register_bagel_extractor(factory) first (factory(model_path=, device_id=, use_bfloat16=, "
                    "config=) -> object with extract_semantic_tokens(text, image)), pass bagel_extractor=, or set "
                    "enable_bagel_extraction=False to generate from the text encoder's context alone")
            bagel_extractor = _BAGEL_EXTRACTOR_FACTORY(model_path=config.bagel_model_path, device_id=config.bagel_gpu,
                                                       use_bfloat16=config.use_bfloat16, config=config)
        self.bagel_extractor = bagel_extractor
        if context_projector is None and wan_pipeline is None:
            self._check_gpu("cross_attn_gpu")
            context_projector = ContextProjector(config).to(f"cuda:{config.cross_attn_gpu}").eval()
        self.context_projector = context_projector
        if wan_pipeline is None:
            wan_pipeline = self._initialize_wan22(wan_config)
        self.wan_pipeline = wan_pipeline
        self.dit_model = wan_pipeline.model
        self.vae_model = wan_pipeline.vae
        self.text_encoder = wan_pipeline.text_encoder
        # native_text_weight=False: the reference's closures on the model's generic path (kept as the comparison the tests and bench use)
        self.wan_wrapper = Wan22ContextWrapper(wan_pipeline, self.context_projector, self.logger, config, native=native_text_weight)
        self.save_fn = save_fn

    def _check_gpu(self, field_name):
        idx, n = getattr(self.config, field_name), torch.cuda.device_count()
        if not 0 <= idx < n:
            raise RuntimeError(f"config.{field_name} = {idx}, but this process sees {n} GPU(s): the reference's defaults place BAGEL / Wan / "
                               f"the projector on cuda:0 / 1 / 2 (inference.py:43-45); on one MI355X (288 GB) set all three to 0")

    def _initialize_wan22(self, wan_config):
        """:2176-2243. WanTI2V from the checkpoint directory. `skip_t5_loading` only moves the reference's T5 to the CPU (t5_cpu=True,
        :2186-2191, the encoder is still what produces the context); here the encoder runs on the GPU either way (288 GB)."""
        path = self.config.wan_model_path
        self._check_gpu("wan_gpu")
        try:
            return WanTI2V(config=wan_config, checkpoint_dir=path, device_id=self.config.wan_gpu, rank=0, t5_fsdp=False, dit_fsdp=False,
                           use_sp=False)
        except Exception as e:
            raise RuntimeError(f"Wan2.2 initialization failed: {e}") from e

    def generate_video_with_bagel_context(self, text: str, image=None, **kwargs):
        """:2577-2655. Returns (video [3, T, H, W] in [-1, 1], path | None). Extra keyword `prompt_embeds` /
        `negative_prompt_embeds` bypass the (pluggable) text encoder. Unlike the reference, errors propagate."""
        start = time.time()
        try:
            if self.bagel_extractor is not None:
                tokens = self.bagel_extractor.extract_semantic_tokens(text, image)
                self.wan_wrapper.set_bagel_context(tokens.to(self.wan_pipeline.device), fusion_alpha=self.config.guidance_strength)
            params = dict(input_prompt=text, img=image, size=kwargs.get("size", (1280, 704)),
                          frame_num=kwargs.get("frames", self.config.video_length), shift=kwargs.get("shift", 5.0),
                          sample_solver="unipc", sampling_steps=kwargs.get("steps", self.config.total_sampling_steps),
                          guide_scale=kwargs.get("guidance_scale", 5.0), seed=kwargs.get("seed", -1), offload_model=False)
            for k in ("prompt_embeds", "negative_prompt_embeds", "noise", "decode"):
                if k in kwargs:
                    params[k] = kwargs[k]
            video = self.wan_wrapper.generate(**params)
            self.logger.info(f"generation time: {time.time() - start:.2f}s")
            path = self.save_fn(video, text) if (self.save_fn is not None and video is not None) else None
            return video, path
        finally:
            self.wan_wrapper.clear_bagel_context()

    def get_fusion_info(self):
        return dict(fusion_mode=self.config.fusion_mode, dynamic_text_weight=self.config.use_dynamic_text_weight,
                    hooked_layers=len(self.wan_wrapper.original_forward_methods))

    def cleanup_resources(self):
        self.wan_wrapper.restore_original_methods()
