"""GPU (-m gpu): `CrossAttentionFusionPipeline(config)` exactly as the reference's inference.py constructs and drives it.

The call sequence below is inference.py's own, RESTATED here (the reference file is not read or executed):
  inference.py:145-194  CrossAttentionConfig(**34 keyword arguments)
  inference.py:196      CrossAttentionFusionPipeline(ca_config)                        - config only, nothing injected
  inference.py:218-224  pipeline.lora_manager.load_lora_weights(path, pipeline.dit_model)
  inference.py:227-236  state = torch.load(path / "training_state.pt"); pipeline.context_projector.load_state_dict(state['context_projector'])
  inference.py:296-320  pipeline.generate_video_with_bagel_context(text=, steps=, guidance_scale=, frames=, size=, shift=, seed=)
  inference.py:365-385  pipeline.generate_video_with_bagel_context(text=, image=<PIL image>, ...same keywords)
on a checkpoint DIRECTORY written here with plain json / safetensors / torch.save in the layouts the reference reads
(textimage2video.py:88-103: diffusers-layout DiT, Wan2.2_VAE.pth, the T5 .pth + a tokenizer directory). The result must be
bit-identical to the same generation through the INJECTED construction (components built one by one), and the DiT it loaded must
reproduce the reference-generated golden.
"""
import json
import logging
import math
import os

import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
T5_DIM, TEXT_LEN = 256, 48
WORDS = ["a", "cat", "on", "the", "mat", "walks", "slowly", "through", "tall", "grass", "at", "dawn", "from", "image"]


def _write_checkpoint_dir(d, seed):
    """DiT (3 safetensors shards + index + config.json), Wan2.2_VAE.pth, the T5 .pth and its tokenizer directory under `d`."""
    from safetensors.torch import save_file
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast
    from oracle import t5 as ot5
    from oracle import wan_dit, wan_vae
    d.mkdir()
    cfg = dict(wan_dit.TINY_CFG, text_dim=T5_DIM, text_len=TEXT_LEN)
    sd = wan_dit.make_state_dict(cfg, seed)
    json.dump({"_class_name": "WanModel", "_diffusers_version": "0.35.1", "model_type": "ti2v", "patch_size": [1, 2, 2],
               "text_len": cfg["text_len"], "in_dim": 48, "dim": cfg["dim"], "ffn_dim": cfg["ffn_dim"], "freq_dim": 256,
               "text_dim": cfg["text_dim"], "out_dim": 48, "num_heads": cfg["num_heads"], "num_layers": cfg["num_layers"],
               "window_size": [-1, -1], "qk_norm": True, "cross_attn_norm": True, "eps": 1e-6}, open(d / "config.json", "w"))
    names = sorted(sd)
    cuts = [0, len(names) // 3, 2 * len(names) // 3, len(names)]
    weight_map = {}
    for i in range(3):
        fn = f"diffusion_pytorch_model-{i + 1:05d}-of-00003.safetensors"
        save_file({k: sd[k].contiguous() for k in names[cuts[i]:cuts[i + 1]]}, str(d / fn), metadata={"format": "pt"})
        weight_map.update({k: fn for k in names[cuts[i]:cuts[i + 1]]})
    json.dump({"metadata": {"total_size": sum(v.numel() * 4 for v in sd.values())}, "weight_map": weight_map},
              open(d / "diffusion_pytorch_model.safetensors.index.json", "w"))
    torch.save(wan_vae.make_state_dict(wan_vae.SMALL_CFG, load_golden("vae_small")["seed"]), str(d / "Wan2.2_VAE.pth"))
    tcfg = ot5.TINY_CFG
    assert tcfg["dim"] == T5_DIM
    torch.save(ot5.make_state_dict(tcfg, int(load_golden("t5_tiny")["seed"])), str(d / "models_t5_tiny.pth"))
    vocab = {"<pad>": 0, "</s>": 1, "<unk>": 2, **{w: 3 + i for i, w in enumerate(WORDS)}}
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    tok.post_processor = processors.TemplateProcessing(single="$A </s>", special_tokens=[("</s>", 1)])   # as umT5's: "" -> [</s>]
    (d / "google" / "umt5-xxl").mkdir(parents=True)
    PreTrainedTokenizerFast(tokenizer_object=tok, pad_token="<pad>", eos_token="</s>", unk_token="<unk>").save_pretrained(d / "google" / "umt5-xxl")
    return cfg, sd


def _wan_config():
    from oracle import t5 as ot5
    from univid_amd.wan.textimage2video import TI2VConfig
    tcfg = ot5.TINY_CFG

    class TinyTI2V(TI2VConfig):          # the WAN_CONFIGS['ti2v-5B'] counterpart of the tiny checkpoint directory
        text_len = TEXT_LEN
        vae_kwargs = dict(c_dim=32, dec_dim=32)
        t5_checkpoint = "models_t5_tiny.pth"
        t5_tokenizer = "google/umt5-xxl"
        t5_kwargs = dict(vocab=tcfg["vocab_size"], dim=tcfg["dim"], dim_attn=tcfg["dim_attn"], dim_ffn=tcfg["dim_ffn"],
                         num_heads=tcfg["num_heads"], num_layers=tcfg["num_layers"], num_buckets=tcfg["num_buckets"])
    return TinyTI2V


def _write_lora_checkpoint(path, cfg, seed):
    """What the reference's training leaves under --lora_checkpoint_path: a PEFT adapter directory + training_state.pt whose
    'context_projector' entry is the projector's state dict (inference.py:227-233)."""
    from safetensors.torch import save_file
    os.makedirs(path)
    g = torch.Generator().manual_seed(seed)
    names = [f"blocks.{i}.{a}.{p}" for i in range(2) for a in ("cross_attn", "self_attn") for p in "qkvo"] + ["blocks.1.ffn.0", "blocks.0.ffn.2"]
    shapes = {"ffn.0": (cfg["ffn_dim"], cfg["dim"]), "ffn.2": (cfg["dim"], cfg["ffn_dim"])}
    r, alpha = 8, 16
    t = {}
    for n in names:
        o, i = next((v for k, v in shapes.items() if n.endswith(k)), (cfg["dim"], cfg["dim"]))
        t[f"base_model.model.{n}.lora_A.default.weight"] = (torch.randn(r, i, generator=g) / math.sqrt(i)).contiguous()
        t[f"base_model.model.{n}.lora_B.default.weight"] = (torch.randn(o, r, generator=g) * 0.2).contiguous()
    json.dump(dict(peft_type="LORA", r=r, lora_alpha=alpha, lora_dropout=0.0, bias="none", use_rslora=False, use_dora=False,
                   fan_in_fan_out=False, target_modules=sorted(names), task_type="FEATURE_EXTRACTION", inference_mode=True),
              open(os.path.join(path, "adapter_config.json"), "w"))
    save_file(t, os.path.join(path, "adapter_model.safetensors"))
    # the projector at the width inference.py's 34 keyword arguments leave it (bagel_hidden_dim 3584 -> 8192 -> wan_text_dim 4096)
    psd = {}
    for k, shape in {"0.weight": (8192, 3584), "0.bias": (8192,), "1.weight": (8192,), "1.bias": (8192,), "4.weight": (4096, 8192),
                     "4.bias": (4096,), "5.weight": (4096,), "5.bias": (4096,)}.items():
        v = torch.randn(shape, generator=g) * (0.02 if len(shape) == 2 else 0.1)
        psd["bagel_to_t5_projector." + k] = (v + (1.0 if k in ("1.weight", "5.weight") else 0.0)).to(torch.bfloat16)
    torch.save({"context_projector": psd, "epoch": 3}, os.path.join(path, "training_state.pt"))
    return psd


class _Extractor:
    """Stands where the reference's BagelSemanticExtractor stands (model_pipeline.py:837-1503, out of scope): constructed with the
    reference's four keyword arguments (:2153-2158), extract_semantic_tokens(text, image) -> [1, L, 3584] (:1240)."""
    built = []

    def __init__(self, model_path, device_id=0, use_bfloat16=True, config=None):
        self.args = (model_path, device_id, use_bfloat16, config)
        self.calls = []
        _Extractor.built.append(self)

    def extract_semantic_tokens(self, text, images=None):
        self.calls.append((text, images is not None))
        g = torch.Generator().manual_seed(len(text) + (7 if images is not None else 0))
        return torch.randn(1, 20, 3584, generator=g).to(torch.bfloat16)


def _ca_config(wan_dir, out_dir, steps):
    """inference.py:145-194, keyword for keyword (values = what its InferenceConfig would carry on ONE GPU)."""
    from univid_amd.model_pipeline import CrossAttentionConfig
    return CrossAttentionConfig(
        bagel_model_path="/models/BAGEL-7B-MoT", wan_model_path=str(wan_dir),
        bagel_gpu=0, wan_gpu=0, cross_attn_gpu=0,
        video_length=5, video_fps=8, video_size=(256, 256),
        fusion_mode="context_replacement", guidance_strength=1.0, wan_text_length=TEXT_LEN,
        output_dir=str(out_dir), save_video_mp4=False,
        use_lora=True, lora_rank=8, lora_alpha=16, lora_dropout=0.0, lora_target_strategy="your_lora_target_strategy",
        use_bfloat16=True, enable_autocast=True, skip_t5_loading=True,
        use_dynamic_text_weight=True, text_weight_max=1.3, text_weight_min=1.0, text_weight_schedule="cosine",
        text_weight_transition_ratio=0.5, total_sampling_steps=steps,
        enable_bagel_extraction=True, enable_wan_injection=True, freeze_bagel=True, freeze_wan_vae=True, freeze_t5=True,
        train_wan_dit=False, train_cross_attn=False)


def test_config_only_construction_replays_inference_py(tmp_path):
    from PIL import Image
    from univid_amd import _lib
    from univid_amd import model_pipeline as mp
    from univid_amd.model_pipeline import ContextProjector, CrossAttentionFusionPipeline, register_bagel_extractor
    from univid_amd.wan.textimage2video import WanTI2V
    _lib.init()
    seed = load_golden("dit_tiny")["seed"]
    d = tmp_path / "Wan2.2-TI2V-tiny"
    cfg, sd = _write_checkpoint_dir(d, seed)
    lora_dir = tmp_path / "ckpt" / "best"
    psd = _write_lora_checkpoint(str(lora_dir), cfg, 3)
    steps = 4
    ca = _ca_config(d, tmp_path / "out", steps)
    Tiny = _wan_config()

    # no extractor registered: the config-only form refuses, naming the registration point (never a silent stub)
    prev = register_bagel_extractor(None)
    try:
        with pytest.raises(RuntimeError, match="register_bagel_extractor"):
            CrossAttentionFusionPipeline(ca, wan_config=Tiny)
        register_bagel_extractor(_Extractor)

        # ---- inference.py:196
        pipeline = CrossAttentionFusionPipeline(ca, wan_config=Tiny)
        ex = _Extractor.built[-1]
        assert ex.args == (ca.bagel_model_path, ca.bagel_gpu, ca.use_bfloat16, ca) and pipeline.bagel_extractor is ex
        assert isinstance(pipeline.wan_pipeline, WanTI2V) and pipeline.dit_model is pipeline.wan_pipeline.model
        assert pipeline.vae_model is pipeline.wan_pipeline.vae and pipeline.vae_model is not None
        assert pipeline.wan_pipeline.text_encoder is not None and pipeline.text_encoder is pipeline.wan_pipeline.text_encoder
        assert isinstance(pipeline.context_projector, ContextProjector) and pipeline.wan_wrapper.context_projector is pipeline.context_projector
        assert pipeline.context_projector.bagel_to_t5_projector[0].weight.device == torch.device(DEV)
        assert pipeline.lora_manager is not None and pipeline.dit_model.patch_embedding.weight.device == torch.device(DEV)
        # the DiT it loaded from the directory is the reference-initialised one
        g = load_golden("dit_tiny")
        base_args = ([g["x"].to(DEV)], g["t_one"].to(DEV), [torch.randn(20, T5_DIM, generator=torch.Generator().manual_seed(2)).to(DEV)], 256)
        with torch.no_grad():
            base = pipeline.dit_model(*base_args)[0].clone()

        # ---- inference.py:218-224
        pipeline.lora_manager.load_lora_weights(str(lora_dir), pipeline.dit_model)
        with torch.no_grad():
            adapted = pipeline.dit_model(*base_args)[0].clone()
        assert not torch.equal(base, adapted), "the adapter must change the DiT"
        # ---- inference.py:227-236
        state = torch.load(str(lora_dir / "training_state.pt"), map_location=f"cuda:{ca.cross_attn_gpu}")
        assert "context_projector" in state and hasattr(pipeline, "context_projector")
        pipeline.context_projector.load_state_dict(state["context_projector"])
        assert torch.equal(pipeline.context_projector.bagel_to_t5_projector[4].weight.cpu(), psd["bagel_to_t5_projector.4.weight"])
        # inference.py:238-247 counts `lora` parameter names on the dit model: a merged adapter has none (LoRAManager's statistics carry it)
        assert pipeline.lora_manager.get_statistics()["lora_modules"] == 18

        # ---- inference.py:296-320 (t2v) and :365-385 (i2v): the exact keyword set
        kw = {"steps": steps, "guidance_scale": 5.0, "frames": ca.video_length, "size": ca.video_size, "shift": 5.0, "seed": 42}
        prompt = "a cat walks slowly through tall grass at dawn"
        img = Image.fromarray((torch.rand(48, 80, 3, generator=torch.Generator().manual_seed(9)) * 255).to(torch.uint8).numpy(), "RGB")
        video, path = pipeline.generate_video_with_bagel_context(text=prompt, **kw)
        assert path is None and video.shape == (3, 5, 256, 256) and video.dtype == torch.float32
        assert torch.isfinite(video).all() and float(video.abs().max()) <= 1.0
        assert ex.calls[-1] == (prompt, False) and not pipeline.wan_wrapper.use_bagel_context
        assert pipeline.wan_wrapper.current_timestep == 2 * steps - 1, "the forward counter saw 2 DiT forwards per sampler step"
        video_i, _ = pipeline.generate_video_with_bagel_context(text=prompt, image=img, **kw)
        assert ex.calls[-1] == (prompt, True)
        # i2v sizes the clip from the image's aspect ratio at max_area 704x1280 (textimage2video.py:462-470; `size` is ignored)
        assert video_i.shape[0] == 3 and video_i.shape[1] == 5 and video_i.shape[2] * video_i.shape[3] <= 704 * 1280
        assert video_i.shape[2] % 32 == 0 and video_i.shape[3] % 32 == 0 and torch.isfinite(video_i).all()

        # ---- the same through the INJECTED construction, components built one by one: bit-identical
        wan = WanTI2V(Tiny, checkpoint_dir=str(d), device=DEV)
        proj = ContextProjector(ca).to(DEV).eval()
        proj.load_state_dict(psd)
        injected = CrossAttentionFusionPipeline(ca, wan_pipeline=wan, bagel_extractor=_Extractor("x"), context_projector=proj)
        injected.lora_manager.load_lora_weights(str(lora_dir), injected.dit_model)
        video2, _ = injected.generate_video_with_bagel_context(text=prompt, **kw)
        video2_i, _ = injected.generate_video_with_bagel_context(text=prompt, image=img, **kw)
        assert torch.equal(video, video2) and torch.equal(video_i, video2_i)
        # and the dynamic text weight + the adapter both acted: without either the clip differs
        ca_plain = _ca_config(d, tmp_path / "out", steps)
        ca_plain.use_dynamic_text_weight = False
        plain = CrossAttentionFusionPipeline(ca_plain, wan_pipeline=wan, bagel_extractor=_Extractor("x"), context_projector=proj)
        video3, _ = plain.generate_video_with_bagel_context(text=prompt, **kw)
        assert not torch.equal(video, video3)
        injected.lora_manager.unload()
        video4, _ = injected.generate_video_with_bagel_context(text=prompt, **kw)
        video6, _ = plain.generate_video_with_bagel_context(text=prompt, **kw)
        assert not torch.equal(video, video4) and not torch.equal(video4, video6)
        # enable_bagel_extraction=False: config-only construction without any extractor, context from the text encoder alone
        ca_off = _ca_config(d, tmp_path / "out", steps)
        ca_off.enable_bagel_extraction, ca_off.use_lora = False, False
        register_bagel_extractor(None)
        bare = CrossAttentionFusionPipeline(ca_off, wan_config=Tiny)
        assert bare.bagel_extractor is None and bare.lora_manager is None
        video5, _ = bare.generate_video_with_bagel_context(text=prompt, **kw)
        assert torch.equal(video5, video6), "no BAGEL context = no text weight = the plain base-model loop"
    finally:
        register_bagel_extractor(prev)


def test_config_only_construction_errors(tmp_path):
    from univid_amd.model_pipeline import CrossAttentionFusionPipeline, register_bagel_extractor
    ca = _ca_config(tmp_path / "no-such-dir", tmp_path / "out", 4)
    prev = register_bagel_extractor(_Extractor)
    try:
        with pytest.raises(FileNotFoundError, match="Wan2.2 model path not found"):
            CrossAttentionFusionPipeline(ca)
        ca.wan_gpu = torch.cuda.device_count()           # the reference's default layout (cuda:1 / cuda:2) on a box without them
        (tmp_path / "no-such-dir").mkdir()
        with pytest.raises(RuntimeError, match="wan_gpu"):
            CrossAttentionFusionPipeline(ca)
        ca.wan_gpu = 0                                   # an empty directory: the loader's own error, wrapped as the reference wraps it
        with pytest.raises(RuntimeError, match="Wan2.2 initialization failed"):
            CrossAttentionFusionPipeline(ca)
    finally:
        register_bagel_extractor(prev)
