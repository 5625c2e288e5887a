"""End-to-end generation time (developer tool): WanTI2V.t2v / i2v with the production model sizes (30-block TI2V-5B DiT, full-width VAE,
random-init weights, synthetic prompt embeddings): 50 UniPC steps with CFG + VAE decode, per stage, for the exact-f32 VAE and the
f32-grade bf16x6 / f16x3 modes - at the bench size (49 frames of 704 x 1280) and at UniVid's own default workload (121 frames of
704 x 1280, inference.py:48-50).   env: STEPS (50), FRAMES ("49,121"), PRECS ("fp32,bf16x6,f16x3")"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univid_amd import _lib
from univid_amd.wan.model import WanModel
from univid_amd.wan.textimage2video import TI2VConfig, WanTI2V
from univid_amd.wan.vae2_2 import Wan2_2_VAE
_lib.init()
dev = "cuda"
steps = int(os.environ.get("STEPS", 50))
with torch.device(dev):
    m = WanModel.from_config(TI2VConfig.dit)
m = m.eval().requires_grad_(False)
m.init_weights(0)
m.prepare()
g = torch.Generator(device=dev).manual_seed(1)
pe = [torch.randn(40, 4096, device=dev, generator=g) * 0.1]
ne = [torch.randn(9, 4096, device=dev, generator=g) * 0.1]


def clock(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return r, time.perf_counter() - t0


frames = [int(f) for f in os.environ.get("FRAMES", "49,121").split(",")]
precs = os.environ.get("PRECS", "fp32,bf16x6,f16x3").split(",")
for F in frames:
    for prec in precs:
        vae = Wan2_2_VAE(device=dev, seed=0, precision=prec)
        pipe = WanTI2V(model=m, vae=vae, device=dev)
        with torch.no_grad():
            # warm-up of both loops (2 steps each): one-time costs - the VAE's weight preparation, the HIP-graph captures of the t2v and i2v
            # forward pairs (one per latent shape and mode; a NEW PROMPT does not recapture) - stay out of the timed generations
            w = pipe.t2v("", size=(1280, 704), frame_num=F, sampling_steps=2, seed=7, prompt_embeds=pe, negative_prompt_embeds=ne)
            pipe.i2v("", w[:, 0].clamp(-1, 1).contiguous(), max_area=704 * 1280, frame_num=F, sampling_steps=2, seed=3, prompt_embeds=pe,
                     negative_prompt_embeds=ne)
            del w
            lat, t_den = clock(lambda: pipe.t2v("", size=(1280, 704), frame_num=F, sampling_steps=steps, seed=7, prompt_embeds=pe,
                                                negative_prompt_embeds=ne, decode=False))
            vid, t_dec = clock(lambda: vae.decode([lat])[0])
            img = vid[:, 0].clamp(-1, 1).contiguous()
            _, t_i2v = clock(lambda: pipe.i2v("", img, max_area=704 * 1280, frame_num=F, sampling_steps=steps, seed=3, prompt_embeds=pe,
                                              negative_prompt_embeds=ne))
        print(f"{F:3d} frames, VAE {prec:7s}: t2v {steps} steps {t_den:6.2f} s ({t_den / steps * 1e3:.1f} ms/step) + decode {t_dec:5.2f} s = "
              f"{t_den + t_dec:6.2f} s per {F}-frame 704x1280 clip (latent {list(lat.shape)}); i2v (encode 1 frame + {steps} steps + decode) "
              f"{t_i2v:6.2f} s; finite {bool(torch.isfinite(vid).all())}", flush=True)
        if prec == "f16x3":
            # the same clip through UniVid's OWN entry point (inference.py:311,377 -> CrossAttentionFusionPipeline.generate_video_with_bagel_context,
            # models/model_pipeline.py:2577-2655) with inference.py's dynamic text weight (cosine 1.3 -> 1.0 over the first 40 % of the steps):
            # the native schedule keeps it on the graph-replay path, so it must cost what the plain t2v + decode above cost
            import types
            from univid_amd.model_pipeline import CrossAttentionConfig, CrossAttentionFusionPipeline
            bagel = types.SimpleNamespace(extract_semantic_tokens=lambda text, image: torch.zeros(1, 128, 3584, device=dev, dtype=torch.bfloat16))
            ccfg = CrossAttentionConfig(use_lora=False, use_dynamic_text_weight=True, total_sampling_steps=steps)
            fusion = CrossAttentionFusionPipeline(ccfg, wan_pipeline=pipe, bagel_extractor=bagel, context_projector=None)
            kw = dict(steps=steps, guidance_scale=5.0, frames=F, size=(1280, 704), shift=5.0, seed=7, prompt_embeds=pe, negative_prompt_embeds=ne)
            with torch.no_grad():
                (v2, _), t_pipe = clock(lambda: fusion.generate_video_with_bagel_context("", **kw))
                (_, _), t_pipe_i2v = clock(lambda: fusion.generate_video_with_bagel_context("", image=img, **dict(kw, seed=3)))
            print(f"{F:3d} frames, VAE {prec:7s}: through CrossAttentionFusionPipeline (dynamic text weight, native schedule): t2v + decode {t_pipe:6.2f} s, "
                  f"i2v {t_pipe_i2v:6.2f} s; finite {bool(torch.isfinite(v2).all())}", flush=True)
            fusion.cleanup_resources()
            del fusion, v2
        del vae, pipe, vid, lat
        torch.cuda.empty_cache()
