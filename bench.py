#!/usr/bin/env python
"""Headline benchmark: denoise-steps/sec of the Wan2.2-TI2V-5B DiT hot path on MI355X.

One "step" = exactly what WanTI2V.t2v does per timestep (reference models/wan/textimage2video.py:367-394):
DiT forward with the prompt context + DiT forward with the negative context (CFG) + CFG combine + UniPC
update, on a 49-frame 704x1280 latent [48,13,44,80] (L = 11 440 tokens; SURVEY.md 8(d) config 3 shape A, the
"49-frame 720p latent" BASELINE.json's metric is quoted on), TI2V-5B dimensions (dim 3072, ffn 14336, 24 heads,
30 layers), random-init weights (no checkpoints offline), synthetic noise / prompt-embeds resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1: one process per GPU over RCCL (torch.distributed backend "nccl"): every rank denoises its OWN sample (independent
diffusion samples shard one per GPU, no data-path collective) and the final latents are all-gathered once over xGMI at
the end of the timed region; value = N*K / max-over-ranks time ("weak"). Rank 0 prints ONE JSON line.
The driver starts the ranks through `python -m torch.distributed.run ... bench.py --gpus N`; a BARE `python bench.py --gpus N`
works too: the parent, before it touches the GPU in any way, starts that same launcher as a CHILD process and relays its
output (never an exec of a process that initialised HIP).
"""
import argparse
import json
import math
import os
import sys
import time

# Host scheduling is decided BEFORE the HIP runtime initialises (it reads its environment then). With the runtime's default ("direct dispatch") a
# HIP-graph replay is submitted by a runtime thread that SPINS while launches are pending - 122-235 ms of CPU per 251 ms step, one busy core per
# rank (round 5 / 6 measurements), whatever hipDeviceScheduleBlockingSync says; with AMD_DIRECT_DISPATCH=0 the runtime's command thread blocks
# instead, and together with uv_host_blocking_sync a whole generation costs 3.4-5.4 ms of CPU per step at the same step time
# (profiles/r06_host_policy.md). `--host-sync auto` (the default): hipDeviceScheduleBlockingSync for every rank, the runtime's own dispatch mode -
# RCCL has never run under AMD_DIRECT_DISPATCH=0 here (no multi-GPU box in six rounds), and the first N > 1 run must not also be the first run of
# a dispatch mode; on this pool's 256-thread hosts eight spinning runtime threads cost nothing. `--host-sync blocking` adds AMD_DIRECT_DISPATCH=0
# (for hosts where the cores matter, once validated there); `default` leaves the runtime alone. Every N = 1 line carries the measurement of the
# `blocking` policy from a child process (`pipeline_path.host_sync.rank_policy_child`); the parent keeps direct dispatch also because HIP events
# around an EAGER launch read 5 % long under the command thread (2.857 -> 3.005 ms on the self-attention launch; 2.84 in the rocprofv3 trace
# either way) and `roofline` is measured exactly so.
def _host_sync_mode():
    for i, a in enumerate(sys.argv):
        if a.startswith("--host-sync="):
            return a.split("=", 1)[1]
        if a == "--host-sync" and i + 1 < len(sys.argv):
            return sys.argv[i + 1]
    return "auto"


if _host_sync_mode() == "blocking" or "--host-probe" in sys.argv:
    os.environ.setdefault("AMD_DIRECT_DISPATCH", "0")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LATENT = (48, 13, 44, 80)          # 49 frames, 704 x 1280 (vae stride 4,16,16)
L_TOKENS = 13 * 22 * 40            # 11 440
GUIDE, SHIFT, SAMPLING_STEPS = 5.0, 5.0, 50
PEAK_BF16_TFLOPS = 2500.0          # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def dit_forward_flops(L, cfg, executed=False):
    """SURVEY.md 8(d) 'Algorithmic work per unit'. executed=True: what a step of the sampling loop actually runs - the context's
    text_embedding and the blocks' cross-attention K / V projections (4 Lc d^2 per block) are step-constant and computed once per
    context (WanModel.context_cached), so they are NOT part of a timed step."""
    d, f, Lc, td, n = cfg["dim"], cfg["ffn_dim"], cfg["text_len"], cfg["text_dim"], cfg["num_layers"]
    ctx_block = 0 if executed else 4 * Lc * d * d
    ctx_embed = 0 if executed else 2 * Lc * (td * d + d * d)
    per_block = L * (12 * d * d + 4 * d * f) + ctx_block + 4 * L * L * d + 4 * L * Lc * d
    return n * per_block + 2 * L * 192 * d * 2 + ctx_embed


def hipblaslt_ref(device):
    """The vendor library on the ffn.0 shape (22880 x 14336 x 3072, the one DiT GEMM shape where it is ahead), next to
    uv_gemm_bf16_nt in the SAME run and OUTSIDE the timed region: interleaved rounds, random operands, plain bf16 epilogue.
    The uv side runs the launch the PRODUCT runs for this GEMM (WanAttentionBlock._run / _ffn0_rows: the input rows rounded up to whole
    256-row tiles, 23 040, inside one buffer - 19.69 rounds of the persistent kernel, no leftover-row launch); FLOPs are counted for the
    22 880 real rows on both sides. `uv_unpadded_*`: the same call on the bare 22 880 rows (main launch + a 96-row strip launch), what
    rounds 3-4 reported here. Calibration only - the product path never calls a vendor GEMM."""
    from univid_amd import _lib
    from univid_amd._lib import EPI_BF16
    from univid_amd.wan.model import _ffn0_rows
    M, N, K = 22880, 14336, 3072
    Mp = _ffn0_rows(M, N, torch.device(device))
    g = torch.Generator(device=device).manual_seed(0)
    Ap = torch.zeros(Mp, K, device=device, dtype=torch.bfloat16)
    Ap[:M] = (torch.rand(M, K, device=device, generator=g) * 2 - 1).to(torch.bfloat16)
    A = Ap[:M]
    W = ((torch.rand(N, K, device=device, generator=g) * 2 - 1) * 0.05).to(torch.bfloat16)
    out = torch.empty(Mp, N, device=device, dtype=torch.bfloat16)
    from univid_amd._lib import EPI_GELU_BF16
    gelu = torch.nn.functional.gelu
    fns = {"hipblaslt": lambda: torch.nn.functional.linear(A, W), "uv_gemm_bf16_nt": lambda: _lib.gemm_bf16(Ap, W, None, out, EPI_BF16, M=Mp),
           "uv_unpadded": lambda: _lib.gemm_bf16(A, W, None, out, EPI_BF16, M=M),
           # the operator the DiT block actually runs there (model.py:212-213: Linear -> GELU(tanh)): the reference's eager pair of kernels on the
           # vendor library against this library's fused epilogue
           "hipblaslt_then_gelu": lambda: gelu(torch.nn.functional.linear(A, W), approximate="tanh"),
           "uv_fused_gelu": lambda: _lib.gemm_bf16(Ap, W, None, out, EPI_GELU_BF16, M=Mp)}
    res = {k: [] for k in fns}
    for _ in range(3):
        for name, fn in fns.items():
            for _ in range(2):
                fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                fn()
            e.record()
            torch.cuda.synchronize()
            res[name].append(s.elapsed_time(e) / 5)
    fl = 2.0 * M * N * K
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    _lib.gemm_bf16(Ap, W, None, out, EPI_BF16, M=Mp)
    same = bool(torch.equal(out[:M], _lib.gemm_bf16(A, W, None, torch.empty(M, N, device=device, dtype=torch.bfloat16), EPI_BF16, M=M)))
    return {"shape": f"{M}x{N}x{K} bf16 (ffn.0)", "uv_rows_launched": Mp, "hipblaslt_tflops": round(fl / med["hipblaslt"] / 1e9, 1),
            "uv_gemm_bf16_nt_tflops": round(fl / med["uv_gemm_bf16_nt"] / 1e9, 1),
            "ratio": round(med["hipblaslt"] / med["uv_gemm_bf16_nt"], 3),
            "uv_unpadded_tflops": round(fl / med["uv_unpadded"] / 1e9, 1), "ratio_unpadded": round(med["hipblaslt"] / med["uv_unpadded"], 3),
            "padded_equals_unpadded_bitwise": same,
            "ffn0_with_gelu": {"hipblaslt_linear_then_torch_gelu_ms": round(med["hipblaslt_then_gelu"], 4), "uv_fused_epilogue_ms": round(med["uv_fused_gelu"], 4),
                               "ratio": round(med["hipblaslt_then_gelu"] / med["uv_fused_gelu"], 3),
                               "note": "Linear + GELU(tanh) as the reference's eager path runs it on this GPU (vendor GEMM, then an elementwise pass over 656 MB) against the fused epilogue"},
            "note": "same run, outside the timed region, interleaved; ratio = vendor time / this kernel's time in the product's launch form (rows padded to whole tiles, FLOPs of the real rows)"}


def self_attn_flops(L, d):
    return 4 * L * L * d


def twin_saved_flops(L, cfg):
    """Work the CFG pair shares: both samples enter block 0 with identical rows (same latent, same timesteps), so its q / k / v / o
    projections and self-attention are computed once (WanModel.dedup_twins; bit-identical)."""
    d = cfg["dim"]
    return 8 * L * d * d + 4 * L * L * d


def sdpa_ref(device, L=11440, H=24, D=128, B=2):
    """The vendor attention on the metric's self-attention shape next to uv_flash_attn_bf16 in the SAME run and OUTSIDE the timed region:
    torch-ROCm F.scaled_dot_product_attention (flash backend; whatever kernel torch 2.10 / ROCm 7 dispatches for bf16 head_dim 128 on
    gfx950) on q, k, v [B, H, L, D] bf16 - B = 2 stacked samples (cond + uncond), exactly one DiT self-attention launch of the step -
    interleaved rounds, the same random (gaussian) data through both. Calibration only - the product path never calls SDPA."""
    import torch.nn.functional as F
    from univid_amd import _lib
    C = H * D
    g = torch.Generator(device=device).manual_seed(3)
    q = torch.randn(B * L, C, device=device, generator=g).to(torch.bfloat16)
    k = torch.randn(B * L, C, device=device, generator=g).to(torch.bfloat16)
    v = torch.randn(B * L, C, device=device, generator=g).to(torch.bfloat16)
    cols = (B - 1) * L + (L + 63) // 64 * 64
    vt = torch.zeros(C, cols, device=device, dtype=torch.bfloat16)
    vt[:, :B * L] = v.t()
    out = torch.empty(B * L, C, device=device, dtype=torch.bfloat16)
    q4, k4, v4 = (t.view(B, L, H, D).transpose(1, 2) for t in (q, k, v))      # [B, H, L, D] views of the same rows
    backend = "flash"
    try:
        from torch.nn.attention import SDPBackend, sdpa_kernel

        def vendor():
            with sdpa_kernel([SDPBackend.FLASH_ATTENTION]):
                return F.scaled_dot_product_attention(q4, k4, v4)
        vendor()
    except Exception:
        backend = "default dispatch (flash backend refused)"

        def vendor():
            return F.scaled_dot_product_attention(q4, k4, v4)
        vendor()
    fns = {"sdpa": vendor, "uv_flash_attn_bf16": lambda: _lib.flash_attn(q, k, vt, out, L, L, H, D, D ** -0.5, batch=B)}
    res = {n: [] for n in fns}
    for _ in range(3):
        for name, fn in fns.items():
            for _ in range(2):
                fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                fn()
            e.record()
            torch.cuda.synchronize()
            res[name].append(s.elapsed_time(e) / 5)
    # same problem, same data: the two outputs agree to bf16 rounding
    o_sdpa = vendor().transpose(1, 2).reshape(B * L, C)
    _lib.flash_attn(q, k, vt, out, L, L, H, D, D ** -0.5, batch=B)
    err = float((o_sdpa.float() - out.float()).abs().max())
    fl = B * self_attn_flops(L, C)
    med = {n: sorted(t)[len(t) // 2] for n, t in res.items()}
    return {"shape": f"q,k,v [{B},{H},{L},{D}] bf16, non-causal (one DiT self-attention launch of the metric's step)", "backend": backend,
            "sdpa_ms": round(med["sdpa"], 4), "uv_flash_attn_bf16_ms": round(med["uv_flash_attn_bf16"], 4),
            "sdpa_tflops": round(fl / med["sdpa"] / 1e9, 1), "uv_flash_attn_bf16_tflops": round(fl / med["uv_flash_attn_bf16"] / 1e9, 1),
            "ratio": round(med["sdpa"] / med["uv_flash_attn_bf16"], 3), "max_abs_diff": round(err, 5),
            "note": "ratio = vendor time / this kernel's time (> 1: this kernel is faster); same run, outside the timed region, interleaved rounds, median of 3"}


def default_shape_probe(model, device, cfg, steps=3):
    """UniVid's OWN default workload (inference.py:48-50: video_length 121, video_size 1280 x 704 -> latent [48,31,44,80], L = 27 280
    tokens; 1 034 TFLOP per CFG step, self-attention 53 % of it): the full 30-block model, one warm-up step + `steps` timed steps of
    exactly what WanTI2V.denoise runs per timestep, then ONE more step with HIP events around the attention launches (outside the timed
    steps) for the self-attention kernel's own duration. OUTSIDE the metric's timed region; NOT the metric (BASELINE.json quotes the
    49-frame latent); reported beside it because it is what inference.py runs."""
    from univid_amd import _lib
    from univid_amd.wan.fm_solvers_unipc import FlowUniPCMultistepScheduler
    latent_shape, L = (48, 31, 44, 80), 31 * 22 * 40
    g = torch.Generator(device=device).manual_seed(11)
    lat = torch.randn(*latent_shape, device=device, generator=g)
    ctx = [torch.randn(77, cfg["text_dim"], device=device, generator=g) * 0.1, torch.randn(12, cfg["text_dim"], device=device, generator=g) * 0.1]
    sched = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1)
    sched.set_timesteps(SAMPLING_STEPS, device="cpu", shift=SHIFT)
    ts = sched.timesteps

    def one_step(i, x):
        tvec = torch.full((2, L), float(ts[i]), device=device)
        cond, uncond = model([x, x], t=tvec, context=ctx, seq_len=L)
        return sched.step_cfg(cond.unsqueeze(0), uncond.unsqueeze(0), GUIDE, ts[i], x.unsqueeze(0)).squeeze(0)

    with torch.no_grad(), model.context_cached():
        lat = one_step(0, lat)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(1, 1 + steps):
            lat = one_step(i, lat)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        _lib.PROFILE = {"uv_flash_attn_bf16": []}
        lat = one_step(1 + steps, lat)
        torch.cuda.synchronize()
        prof, _lib.PROFILE = _lib.PROFILE, None
    fl = 2 * dit_forward_flops(L, cfg, executed=True) - twin_saved_flops(L, cfg)
    sa = 2 * self_attn_flops(L, cfg["dim"])
    ev = [(s_, e_) for s_, e_, f in prof["uv_flash_attn_bf16"] if f >= sa * 0.99]
    att_ms = sum(s_.elapsed_time(e_) for s_, e_ in ev) / max(len(ev), 1)
    return {"workload": f"UniVid's default (inference.py:48-50), not the metric: 121-frame 704x1280 latent [48,31,44,80], L={L} tokens; full "
                        f"TI2V-5B DiT ({cfg['num_layers']} layers); 1 step = cond+uncond forward + CFG + UniPC; {steps} timed steps after 1 warm-up",
            "steps_per_sec": round(1.0 / dt, 4), "ms_per_step": round(dt * 1e3, 2), "step_tflop": round(fl / 1e12, 1),
            "tflops": round(fl / dt / 1e12, 1), "mfma_frac": round(fl / dt / 1e12 / PEAK_BF16_TFLOPS, 4),
            "self_attention": {"kernel": _lib.attn_kernel_name(L, L, cfg["dim"] // cfg["num_heads"], 2), "avg_launch_ms": round(att_ms, 3),
                               "launches_timed": len(ev), "timed": "one extra step after the timed ones, HIP events on the launch stream",
                               "tflops": round(sa / (att_ms * 1e-3) / 1e12, 1) if att_ms else None,
                               "frac": round(sa / (att_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4) if att_ms else None},
            "finite": bool(torch.isfinite(lat).all().item())}


def _pipeline_probe_parts(model, device, cfg):
    """What pipeline_path_probe and the host-policy child build: the WanTI2V around `model`, inference.py's config, a stub BAGEL extractor
    returning [1, 128, 3584] tokens, the HIP ContextProjector, and the keyword arguments of the generation call."""
    import types
    from univid_amd.model_pipeline import ContextProjector, CrossAttentionConfig
    from univid_amd.wan.textimage2video import TI2VConfig, WanTI2V
    g = torch.Generator(device=device).manual_seed(21)
    tokens = torch.randn(1, 128, 3584, device=device, generator=g).to(torch.bfloat16)
    bagel = types.SimpleNamespace(extract_semantic_tokens=lambda text, image: tokens)
    ccfg = CrossAttentionConfig(use_lora=False, use_dynamic_text_weight=True, text_weight_max=1.3, text_weight_min=1.0, text_weight_schedule="cosine",
                                text_weight_transition_ratio=0.4, total_sampling_steps=SAMPLING_STEPS)
    with torch.device(device):
        proj = ContextProjector(ccfg)
    from univid_amd import detinit
    detinit.init_module_(proj, 5)
    proj = proj.eval()
    pipe = WanTI2V(TI2VConfig, model=model, device=device)
    noise = torch.randn(*LATENT, device=device, generator=g)
    emb = [torch.randn(77, cfg["text_dim"], device=device, generator=g) * 0.1]
    emb_n = [torch.randn(12, cfg["text_dim"], device=device, generator=g) * 0.1]
    kw = dict(guidance_scale=GUIDE, frames=49, size=(1280, 704), shift=SHIFT, decode=False, prompt_embeds=emb, negative_prompt_embeds=emb_n, noise=noise)
    return pipe, ccfg, bagel, proj, kw, g


def host_probe_main():
    """`bench.py --host-probe` (a child of the N = 1 run, started before that run touches the GPU, with AMD_DIRECT_DISPATCH=0 in its environment): the
    host policy the ranks of an N > 1 job run under - command-thread dispatch + hipDeviceScheduleBlockingSync - on the metric's model and shape through
    UniVid's entry point: process CPU per step of one whole 12-step generation (after a 2-step one that captures the graph). Prints one JSON line."""
    from univid_amd import _lib
    from univid_amd.model_pipeline import CrossAttentionFusionPipeline
    from univid_amd.wan.textimage2video import TI2VConfig
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    _lib.host_blocking_sync(True, device)
    _lib.init()
    cfg = {k: v for k, v in TI2VConfig.dit.items() if k not in ("model_type", "window_size", "qk_norm", "cross_attn_norm")}
    model = build_model(cfg, device, seed=0)
    pipe, ccfg, bagel, proj, kw, _ = _pipeline_probe_parts(model, device, cfg)
    n = 12
    with torch.no_grad():
        fusion = CrossAttentionFusionPipeline(ccfg, wan_pipeline=pipe, bagel_extractor=bagel, context_projector=proj)
        fusion.generate_video_with_bagel_context("a prompt", steps=2, **kw)
        torch.cuda.synchronize()
        cpu0, t0 = time.process_time(), time.perf_counter()
        lat, _ = fusion.generate_video_with_bagel_context("a prompt", steps=n, **kw)
        cpu1 = time.process_time()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(json.dumps({"policy": "AMD_DIRECT_DISPATCH=%s + hipDeviceScheduleBlockingSync (bench.py --host-sync blocking; univid_amd.parallel.RANK_ENV / host_policy)"
                                % os.environ.get("AMD_DIRECT_DISPATCH"), "steps": n, "ms_per_step": round(dt / n * 1e3, 2),
                      "host_cpu_ms_per_step": round((cpu1 - cpu0) / n * 1e3, 3), "graph": pipe._runner is not None, "finite": bool(torch.isfinite(lat).all().item())}), flush=True)


def run_host_probe_child():
    """Starts host_probe_main in a child process. Called BEFORE this process initialises the GPU (a fork + exec from a process that has is what the
    pool forbids), exactly like the `--gpus N` launcher below."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--host-probe"], env=dict(os.environ, AMD_DIRECT_DISPATCH="0"),
                           capture_output=True, text=True, timeout=240)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        return json.loads(lines[-1]) if lines else {"error": f"child exited {r.returncode}: {r.stderr[-300:]}"}
    except Exception as ex:      # a side measurement: never fails the bench
        return {"error": repr(ex)[:300]}


def pipeline_path_probe(model, device, cfg, steps=SAMPLING_STEPS, closure_steps=10, blocking=True):
    """UniVid's OWN entry point on the metric's shape: CrossAttentionFusionPipeline.generate_video_with_bagel_context (reference
    models/model_pipeline.py:2577-2655, what inference.py:311,377 calls) -> Wan22ContextWrapper.generate -> WanTI2V.t2v, with inference.py's
    settings (:52-80: 50 steps, dynamic text weight cosine 1.3 -> 1.0 over int(50 * 0.4) = 20 forwards = the first 10 steps), a stub BAGEL
    extractor returning [1, 128, 3584] tokens, the HIP ContextProjector, prompt embeddings passed in, decode=False. Timed: ONE whole
    50-step generation (after a 2-step one that captures the graph), wall clock around the call. Beside it the same entry point with
    native_text_weight=False: the reference's closures on every WanCrossAttention.forward and the DiT forward, executed on the model's
    generic path (two batch-1 forwards per step, context K / V re-projected in every block, un-fused residual, no graph).
    OUTSIDE the metric's timed region; the metric's own loop is WanTI2V.denoise's plain step, which this path equals once w = 1."""
    from univid_amd import _lib
    from univid_amd.model_pipeline import CrossAttentionFusionPipeline
    pipe, ccfg, bagel, proj, kw, g = _pipeline_probe_parts(model, device, cfg)
    res = {"entry": "CrossAttentionFusionPipeline.generate_video_with_bagel_context(prompt_embeds=, decode=False) -> Wan22ContextWrapper -> WanTI2V.t2v",
           "workload": f"the metric's latent {list(LATENT)}, L={L_TOKENS}; inference.py's schedule: cosine text weight 1.3 -> 1.0 over the first "
                       f"{int(SAMPLING_STEPS * 0.4)} forwards of {SAMPLING_STEPS} steps, first 128 context rows, all {cfg['num_layers']} blocks"}

    def timed(fusion, n):
        torch.cuda.synchronize()
        c0, cpu0, t0 = _lib.CALL_COUNT, time.process_time(), time.perf_counter()
        lat, _ = fusion.generate_video_with_bagel_context("a prompt", steps=n, **kw)
        cpu1 = time.process_time()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return lat, {"steps": n, "seconds": round(dt, 3), "ms_per_step": round(dt / n * 1e3, 2), "steps_per_sec": round(n / dt, 4),
                     "host_cpu_ms_per_step": round((cpu1 - cpu0) / n * 1e3, 3), "launches_per_step": round((_lib.CALL_COUNT - c0) / n, 1),
                     "finite": bool(torch.isfinite(lat).all().item())}

    with torch.no_grad():
        fusion = CrossAttentionFusionPipeline(ccfg, wan_pipeline=pipe, bagel_extractor=bagel, context_projector=proj)
        fusion.generate_video_with_bagel_context("a prompt", steps=2, **kw)            # capture + kernel load
        lat_n, res["native"] = timed(fusion, steps)
        res["native"]["graph"] = pipe._runner is not None
        res["native"]["note"] = ("whole generation incl. ContextProjector, graph-buffer refreshes of the weighted steps, sampler set-up; "
                                 "text weight as data on the fast path (WanModel.set_text_weight / _GraphedPair.apply). host_cpu_ms_per_step here is process CPU "
                                 "time over the WHOLE call, i.e. including the time the host waits for the GPU (the loop keeps <= 4 steps queued and naps; what "
                                 "remains is a ROCm runtime thread that polls while work is pending) - the headline's figure is the enqueue time only")
        # the same call under the runtime's DEFAULT host scheduling (a waiting host thread spins): what host_cpu_ms_per_step was before round 6
        if blocking:
            _lib.host_blocking_sync(False, device)
            try:
                _, spin = timed(fusion, closure_steps)
            finally:
                _lib.host_blocking_sync(True, device)
            res["host_sync"] = {"policy": "hipDeviceScheduleBlockingSync, AMD_DIRECT_DISPATCH=%s (this process)" % os.environ.get("AMD_DIRECT_DISPATCH", "unset = direct dispatch"),
                                "host_cpu_ms_per_step": res["native"]["host_cpu_ms_per_step"],
                                "schedule_auto_host_cpu_ms_per_step": spin["host_cpu_ms_per_step"], "schedule_auto_ms_per_step": spin["ms_per_step"],
                                "schedule_auto_steps": closure_steps,
                                "note": "process CPU (all threads) per step of a whole generation. schedule_auto = the same call after uv_host_blocking_sync(0) "
                                        "(hipDeviceScheduleAuto: the waiting main thread spins too). rank_policy_child = the same entry point in a child process "
                                        "under AMD_DIRECT_DISPATCH=0 + blocking sync (--host-sync blocking: the policy for hosts whose cores matter)"}
        # i2v through the same entry point (inference.py:365-385: image=...): one 704x1280 frame encoded by the VAE (f16x3, random-init), its
        # latent frame held fixed through the loop (timestep 0 on its tokens, the {0, t} table of the graph runner); 50 steps, encode included
        try:
            from univid_amd.wan.vae2_2 import Wan2_2_VAE
            pipe.vae = Wan2_2_VAE(device=device, seed=0)
            img = torch.rand(3, 704, 1280, device=device, generator=g) * 2 - 1
            kw_i = {k: v for k, v in kw.items() if k != "noise"}
            fusion.generate_video_with_bagel_context("a prompt", image=img, steps=2, seed=7, **kw_i)      # capture of the i2v graph
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lat_i, _ = fusion.generate_video_with_bagel_context("a prompt", image=img, steps=steps, seed=7, **kw_i)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res["i2v"] = {"steps": steps, "seconds": round(dt, 3), "ms_per_step": round(dt / steps * 1e3, 2),
                          "vs_t2v_ms_per_step": round(dt / steps * 1e3 / res["native"]["ms_per_step"], 4), "finite": bool(torch.isfinite(lat_i).all().item()),
                          "note": "generate_video_with_bagel_context(text, image=[3,704,1280] tensor): VAE encode of the frame + 50 steps, decode=False; wall clock around the call"}
            del lat_i
        except Exception as ex:
            res["i2v"] = {"error": repr(ex)[:300]}
        finally:
            pipe.vae = None
        # UniVid's DEFAULT clip (inference.py:48-50: 121 frames 704x1280, L = 27 280) through the same entry point: 4 steps after a 2-step capture run
        try:
            kw_d = dict(kw, frames=121)
            kw_d.pop("noise")
            fusion.generate_video_with_bagel_context("a prompt", steps=2, seed=3, **kw_d)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lat_d, _ = fusion.generate_video_with_bagel_context("a prompt", steps=4, seed=3, **kw_d)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res["default_shape"] = {"workload": "121-frame 704x1280 latent [48,31,44,80], L=27280; 4 steps (all inside the weighted part of the 50-step schedule: every "
                                                "step refreshes the graph's context K / V buffers)", "steps": 4, "ms_per_step": round(dt / 4 * 1e3, 2),
                                    "finite": bool(torch.isfinite(lat_d).all().item())}
            del lat_d
        except Exception as ex:
            res["default_shape"] = {"error": repr(ex)[:300]}
        pipe._runner = None
        torch.cuda.empty_cache()
        # same first steps through both mechanisms: bit-identical latents (also tests/test_pipeline_path.py)
        lat_a, _ = fusion.generate_video_with_bagel_context("a prompt", steps=closure_steps, **kw)
        lat_a = lat_a.clone()
        fusion.cleanup_resources()
        pipe._runner = None
        torch.cuda.empty_cache()
        closures = CrossAttentionFusionPipeline(ccfg, wan_pipeline=pipe, bagel_extractor=bagel, context_projector=proj, native_text_weight=False)
        closures.generate_video_with_bagel_context("a prompt", steps=1, **kw)
        lat_c, res["closures"] = timed(closures, closure_steps)
        res["closures"]["note"] = "the reference's mechanism literally (forward closures), on the model's generic path; what rounds 1-4 ran under this entry point"
        closures.cleanup_resources()
        res["native_equals_closures_bitwise"] = bool(torch.equal(lat_a, lat_c))
    res["ms_per_step"], res["host_cpu_ms_per_step"], res["launches_per_step"] = (res["native"][k] for k in ("ms_per_step", "host_cpu_ms_per_step", "launches_per_step"))
    return res


def stress_shape_probe(device, base_cfg, blocks=30, model=None):
    """north_star's utilisation target is quoted at the literal 49 x 90 x 160 latent (L = 176 400 tokens; 13 PFLOP per forward, 26 per CFG step).
    Round 5 (verdict item 5): the WHOLE 30-block TI2V-5B stack - the bench's own model - one warm-up forward pair (context work, kernel
    load), then ONE timed forward pair (cond + uncond stacked, exactly the step's DiT work), outside the timed region of the metric.
    blocks < 30 (developer runs, --stress-blocks): a partial stack built for the probe; labelled. NOT the metric."""
    from univid_amd import _lib
    cfg = dict(base_cfg, num_layers=blocks)
    latent_shape, L = (48, 49, 90, 160), 49 * 45 * 80
    own = model is None
    if own:
        model = build_model(cfg, device, seed=0)
    g = torch.Generator(device=device).manual_seed(7)
    lat = torch.randn(*latent_shape, device=device, generator=g)
    ctx = [torch.randn(77, cfg["text_dim"], device=device, generator=g) * 0.1, torch.randn(12, cfg["text_dim"], device=device, generator=g) * 0.1]
    tvec = torch.full((2, L), 500.0, device=device)
    torch.cuda.reset_peak_memory_stats(device)
    with torch.no_grad(), model.context_cached():
        model([lat, lat], t=tvec, context=ctx, seq_len=L)                     # warm-up (and the context work)
        torch.cuda.synchronize()
        _lib.PROFILE = {"uv_flash_attn_bf16": []}
        t0 = time.perf_counter()
        out = model([lat, lat], t=tvec, context=ctx, seq_len=L)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        prof, _lib.PROFILE = _lib.PROFILE, None
    fl = 2 * dit_forward_flops(L, cfg, executed=True) - twin_saved_flops(L, cfg)
    self_ev = [(s_, e_, f) for s_, e_, f in prof["uv_flash_attn_bf16"] if f >= 2 * self_attn_flops(L, cfg["dim"]) * 0.99]
    att_ms = sum(s_.elapsed_time(e_) for s_, e_, _ in self_ev) / max(len(self_ev), 1)
    res = {"workload": f"STRESS SHAPE (not the metric): literal 49x90x160 latent [48,49,90,160], L={L} tokens, cond+uncond stacked, TI2V-5B width, "
                       + (f"ALL {blocks} blocks" if blocks == base_cfg["num_layers"] else f"partial stack: the first {blocks} of {base_cfg['num_layers']} blocks")
                       + " + embeddings + head; one forward pair timed after one warm-up pair",
           "blocks": blocks, "seconds": round(dt, 3), "tflop": round(fl / 1e12, 1), "tflops": round(fl / dt / 1e12, 1),
           "mfma_frac": round(fl / dt / 1e12 / PEAK_BF16_TFLOPS, 4), "target": "north_star: >= 0.40 at this latent",
           "memory_high_water_gb": round(torch.cuda.max_memory_allocated(device) / 1e9, 2),
           "self_attention": {"avg_launch_ms": round(att_ms, 2), "launches_timed": len(self_ev),
                              "tflops": round(2 * self_attn_flops(L, cfg["dim"]) / (att_ms * 1e-3) / 1e12, 1) if att_ms else None,
                              "frac": round(2 * self_attn_flops(L, cfg["dim"]) / (att_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4) if att_ms else None},
           "finite": bool(torch.isfinite(out[0]).all().item())}
    del out
    if own:
        del model
    torch.cuda.empty_cache()
    return res


def ranker_probe(device, frames=64, reps=9):
    """BASELINE.json config 5 (reference models/BAGEL/eval_understanding.py:171-206): SigLIP2-base patch16 encode of 64 keyframes per
    video for the Pyramid-Reflection ranker, fp16 as the reference loads it (:172), random-init weights of the siglip2-base geometry
    (768-d, 12 layers, 12 heads x 64, 256 patches of 16x16x3; NaFlex inputs pixel_values [64, 256, 768], full masks, 16x16 grids), one
    64-token text query: Siglip2Scorer.rank_frames = text tower + vision tower on the 64 frames + cosine top-8 (eager launches: a HIP-graph replay of
    the tower measures the same at this size, profiles/r02_ranker_bench.md). frames/s = 64 / median wall time of `reps` calls. FLOPs per frame: 12 layers x ((8 h^2 + 4 h f)
    per token x 256 + attention 4 L^2 h) + patch embedding + pooling head = 46.5 GFLOP. OUTSIDE the metric's timed region."""
    from univid_amd.understanding import Siglip2Model, Siglip2Scorer
    V = dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12, num_channels=3, patch_size=16,
             num_patches=256, layer_norm_eps=1e-6)
    T = dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12, vocab_size=32000,
             max_position_embeddings=64, projection_size=768, layer_norm_eps=1e-6)
    with torch.device(device):
        m = Siglip2Model(dict(vision=V, text=T))
    m.init_weights(0).eval()
    B, N = frames, 256
    g = torch.Generator(device=device).manual_seed(0)
    pv = torch.randn(B, N, 768, device=device, generator=g)
    mask = torch.ones(B, N, dtype=torch.int64, device=device)
    shapes = torch.tensor([[16, 16]] * B, device=device)
    ids = torch.randint(0, 32000, (1, 64), device=device, generator=g)

    class Proc:
        def __call__(self, images=None, text=None, return_tensors="pt"):
            if text is not None:
                return {"input_ids": ids}
            i = torch.tensor(images)
            return {"pixel_values": pv[i], "pixel_attention_mask": mask[i], "spatial_shapes": shapes[i]}

    sc = Siglip2Scorer(device=device, model=m, processor=Proc())
    h, f, L, nl = 768, 3072, 256, 12
    flops_frame = nl * (L * (8 * h * h + 4 * h * f) + 4 * L * L * h) + 2 * L * 768 * h + (2 * L * 2 * h * h + 4 * L * h + 2 * h * h + 4 * h * f)
    res = {}
    def serial_rank():
        sc.overlap_towers = False
        try:
            return sc.rank_frames(list(range(B)), "q", 8)
        finally:
            sc.overlap_towers = True

    for name, fn in (("rank_frames", lambda: sc.rank_frames(list(range(B)), "q", 8)), ("rank_frames_towers_in_series", serial_rank),
                     ("image_features", lambda: m.get_image_features(pv, mask, shapes))):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        t = sorted(ts)[len(ts) // 2]
        res[name] = {"ms": round(t * 1e3, 3), "frames_per_sec": round(B / t, 1), "tflops": round(B * flops_frame / t / 1e12, 1),
                     "frac_fp16_peak": round(B * flops_frame / t / 1e12 / PEAK_BF16_TFLOPS, 4)}
    return {"workload": f"BASELINE config 5: SigLIP2-base patch16, {B} keyframes x 256 patches, fp16, random-init; rank_frames = text query + {B} frames + cosine top-8 "
                        "(eval_understanding.py:171-206), the text tower on a side stream beside the vision tower (rank_frames_towers_in_series: one after the other, rounds 1-5); "
                        "image_features = the vision tower alone; median of %d calls" % reps,
            "metric": "ranker_frames_per_sec", "value": res["rank_frames"]["frames_per_sec"], "unit": "frames/s", "dtype": "fp16",
            "gflop_per_frame": round(flops_frame / 1e9, 1), **res,
            "note": "launch/latency-bound at this size (2.9 TFLOP per call): the fraction of the fp16 MFMA peak is reported, not a target"}


def build_model(cfg, device, seed=0):
    from univid_amd.wan.model import WanModel
    with torch.device(device):  # parameters are born on the GPU (5 B fp32 = 20 GB; no host staging)
        m = WanModel.from_config(dict(cfg, model_type="ti2v"))
    m = m.eval().requires_grad_(False)
    m.init_weights(seed)       # deterministic counter-based init, generated on the device
    m.prepare()
    return m


VAE_DECODE_TFLOP, VAE_ENCODE_TFLOP = 835.4, 159.1      # 49x720x1280, SURVEY 8(d) (FlopCounter on meta tensors)
PEAK_F32_MFMA_TFLOPS = 157.3


def _vae_bound(precision, tflops):
    """fp32 = exact f32 MFMA: bound is the f32 matrix peak. bf16x3 executes 3 bf16 MFMA passes per product, so its bound in
    ALGORITHMIC flops is the bf16 peak / 3 (an f32-peak fraction would read > 1 and means nothing)."""
    if precision == "fp32":
        return {"bound_tflops": PEAK_F32_MFMA_TFLOPS, "vs_bound": round(tflops / PEAK_F32_MFMA_TFLOPS, 4), "bound": "f32 MFMA peak"}
    n = 6 if precision == "bf16x6" else 3
    pipe = "fp16" if precision == "f16x3" else "bf16"       # same dense peak
    return {"bound_tflops": round(PEAK_BF16_TFLOPS / n, 1), "vs_bound": round(tflops / (PEAK_BF16_TFLOPS / n), 4),
            "bound": f"{pipe} MFMA peak / {n} passes"}


_VAE_DTYPE = {"fp32": "f32", "bf16x3": "bf16x3 (f32 accumulate)",
              "f16x3": "f32-grade in 3 passes: the convolutions behind an RMS_norm take both operands as two IEEE fp16 pieces (22 significant bits; "
                       "error vs fp64 = the f32 MFMA kernel's, accumulation-bound), products on the fp16 MFMA, f32 accumulate; the other convolutions as bf16x6",
              "bf16x6": "f32 operands, products on the bf16 MFMA by exact 3-way splitting (6 passes, error < 2^-26 per product), f32 accumulate"}


def vae_metrics(device, precision, encode=True, reps=3):
    """BASELINE.json's second metric (config 4): Wan2.2 VAE on a 49-frame 720x1280 clip, random-init weights: encode
    [3,49,720,1280] -> [48,13,45,80] (vae2_2.py:783-810) and decode back (:812-839). precision 'fp32' = exact f32 MFMA, the
    reference's dtype (vae2_2.py:897); 'f16x3' = Wan2_2_VAE's default (f32-grade in 3 fp16 passes); 'bf16x6' = the f32 operands split
    exactly into three bf16 planes, six passes; 'bf16x3' = two-way bf16 split (drops lo*lo: ~1e-5) - each under its own key.
    Timing (round-4 verdict: the decode's wall clock held allocator stalls): ONE full-size warm-up call fills the VAE's workspace arena
    (WanVAE_._arena), then `reps` timed calls, each bracketed by a device synchronise; `seconds` = their MEDIAN, min / max beside it,
    and the number of device allocations (hipMalloc calls of torch's allocator) the timed calls caused - 0 means a call's wall time is
    its kernel time. GB/s = fp32 RGB bytes (out for decode, in for encode) / median time."""
    from univid_amd.wan.vae2_2 import Wan2_2_VAE
    vae = Wan2_2_VAE(device=device, seed=0, precision=precision)
    g = torch.Generator(device=device).manual_seed(7)
    res = {}

    def timed(fn):
        out = fn()                                               # full-size warm-up: arena, kernel load, split weights
        torch.cuda.synchronize()
        a0 = torch.cuda.memory_stats(device).get("num_device_alloc", 0)
        ts = []
        for _ in range(reps):
            del out
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        allocs = torch.cuda.memory_stats(device).get("num_device_alloc", 0) - a0
        ts.sort()
        return out, ts[len(ts) // 2], ts, allocs

    with torch.no_grad():
        z = torch.randn(48, 13, 45, 80, device=device, generator=g)
        v, dt, ts, allocs = timed(lambda: vae.decode([z])[0])
        tf = VAE_DECODE_TFLOP / dt
        res["decode"] = {"metric": "vae_decode_GBps", "value": round(v.numel() * 4 / dt / 1e9, 4), "unit": "GB/s", "seconds": round(dt, 3),
                         "seconds_min": round(ts[0], 3), "seconds_max": round(ts[-1], 3), "max_over_min": round(ts[-1] / ts[0], 4), "timed_calls": reps,
                         "device_allocs_in_timed_calls": allocs,
                         "clip": "49x720x1280 RGB from latent [48,13,45,80]", "dtype": _VAE_DTYPE[precision],
                         "tflops": round(tf, 1), **_vae_bound(precision, tf), "finite": bool(torch.isfinite(v).all().item())}
        if encode:
            video = v.clamp_(-1, 1)                              # a [3,49,720,1280] clip in [-1, 1]: the decode's own output
            zz, dt, ts, allocs = timed(lambda: vae.encode([video])[0])
            tf = VAE_ENCODE_TFLOP / dt
            res["encode"] = {"metric": "vae_encode_GBps", "value": round(video.numel() * 4 / dt / 1e9, 4), "unit": "GB/s", "seconds": round(dt, 3),
                             "seconds_min": round(ts[0], 3), "seconds_max": round(ts[-1], 3), "max_over_min": round(ts[-1] / ts[0], 4), "timed_calls": reps,
                             "device_allocs_in_timed_calls": allocs,
                             "clip": "49x720x1280 RGB -> latent " + str(list(zz.shape)), "dtype": _VAE_DTYPE[precision],
                             "tflops": round(tf, 1), **_vae_bound(precision, tf), "finite": bool(torch.isfinite(zz).all().item())}
    del vae
    torch.cuda.empty_cache()
    return res


def cpu_baseline_vae(budget_s=12.0):
    """VAE leg of the CPU baseline: the oracle's fp32 VAE decoder (oracle/wan_vae.py, bit-identical to the reference module) on
    this host, FULL width, on a BOUNDED sample: the first latent frame of a quarter-area clip (360x640; the full-size first
    frame alone took 78 s on 256 threads), extrapolated to the 49-frame 720x1280 clip by output pixels (the convolutions'
    FLOPs are proportional to them)."""
    from oracle import wan_vae
    cores = _cpu_threads()
    torch.set_num_threads(cores)
    cfg = wan_vae.FULL_CFG
    vae = wan_vae.WanVAE(wan_vae.make_state_dict(cfg, 0), cfg)
    scale = wan_vae.scale_tensors()
    g = torch.Generator().manual_seed(7)
    z = torch.randn(1, 48, 2, 22, 40, generator=g)
    with torch.no_grad():
        t0 = time.time()
        v = vae.decode(z[:, :, :1], scale)
        t1 = time.time() - t0
        frames, secs, nlat = v.shape[2], t1, 1
        if t1 < budget_s / 5:        # fast host: add a steady-state chunk (4 frames per latent frame)
            t0 = time.time()
            v = vae.decode(z, scale)
            secs, frames, nlat = time.time() - t0, v.shape[2], 2
    px = frames * v.shape[3] * v.shape[4]
    full = secs * (49 * 720 * 1280) / px
    return {"value": round(3 * 49 * 720 * 1280 * 4 / full / 1e9, 6), "unit": "GB/s", "cores": cores, "kind": "port",
            "sample": f"oracle/wan_vae.WanVAE.decode (fp32, full width) of {nlat} latent frame(s) [48,{nlat},22,40] -> {frames} frame(s) "
                      f"{v.shape[3]}x{v.shape[4]} in {secs:.1f} s on {cores} threads; 49-frame 720x1280 clip extrapolated by output pixels = {full:.0f} s"}


def _cpu_threads():
    """Threads for the CPU baselines. NOT os.cpu_count(): on the 256-hardware-thread hosts of the GPU pool torch's intra-op pool
    at 256 threads runs the oracle ~80x SLOWER than at 32 (measured: the same VAE chunk 56 s vs 0.7 s; 1.8 s at 64), i.e. the
    container's usable cores are far fewer than the reported count. 32 is the measured best there; `cores` reports what was used."""
    return max(1, min(32, os.cpu_count() or 1))


def cpu_baseline(cfg, budget_s=15.0):
    """Times the CPU restatement (oracle/, validated bit-exact against the reference modules) on this host.

    A full step is 300 TFLOP (hours on a CPU), so a BOUNDED sample is timed: ONE of the 30 DiT blocks at
    L_s tokens (L_s picked so the sample takes roughly `budget_s`), extrapolated to a step by the ratio of the
    SURVEY 8(d) FLOP formula (x num_layers, x 2 forwards)."""
    from oracle import wan_dit
    cores = _cpu_threads()
    torch.set_num_threads(cores)
    c1 = dict(cfg, num_layers=1)
    sd = wan_dit.make_state_dict(c1, 0)
    g = torch.Generator().manual_seed(0)
    ctx = torch.randn(1, cfg["text_len"], cfg["dim"], generator=g).to(torch.bfloat16)
    freqs = wan_dit.rope_table(cfg["dim"] // cfg["num_heads"])

    def run(f, h, w):
        L = f * h * w
        x = torch.randn(1, L, cfg["dim"], generator=g)
        e0 = torch.randn(1, 1, 6, cfg["dim"], generator=g).expand(1, L, 6, cfg["dim"]) * 0.1
        t0 = time.time()
        with torch.no_grad():
            wan_dit.block_forward(sd, "blocks.0.", x, e0, torch.tensor([L]), torch.tensor([[f, h, w]]), freqs, ctx,
                                  cfg["num_heads"], cfg["eps"])
        return L, time.time() - t0

    def block_flops(L):
        return dit_forward_flops(L, c1) - (2 * L * 192 * cfg["dim"] * 2 + 2 * cfg["text_len"] * (cfg["text_dim"] * cfg["dim"] + cfg["dim"] ** 2))

    run(1, 8, 8)                                   # warm the thread pool / oneDNN primitives
    Lp, tp = run(2, 16, 16)                        # probe: 512 tokens
    rate = block_flops(Lp) / tp
    choice = (1, 8, 16)
    for grid in [(13, 22, 40), (13, 15, 26), (4, 22, 40), (2, 22, 40), (1, 22, 40), (1, 16, 16)]:
        L = grid[0] * grid[1] * grid[2]
        if block_flops(L) / rate <= budget_s * 1.5:
            choice = grid
            break
    Ls, ts = run(*choice)
    full_step_s = ts * (2 * dit_forward_flops(L_TOKENS, cfg)) / block_flops(Ls)
    return {"value": 1.0 / full_step_s, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle/wan_dit.block_forward: 1 of {cfg['num_layers']} DiT blocks at L={Ls} tokens "
                      f"(grid {choice}) in {ts:.1f} s on {cores} threads (bf16 autocast dtype flow), extrapolated to one "
                      f"CFG step at L={L_TOKENS} by the SURVEY 8(d) FLOP ratio (self-attention is "
                      f"{100 * self_attn_flops(Ls, cfg['dim']) / block_flops(Ls):.0f} % of the sample's FLOPs and "
                      f"{100 * self_attn_flops(L_TOKENS, cfg['dim']) / block_flops(L_TOKENS):.0f} % of a full-size block's: where they "
                      f"differ the extrapolation assumes the CPU runs attention and GEMMs at the same rate)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto", help="replay the CFG pair's forward from a captured HIP graph (what "
                    "WanTI2V.denoise does for the plain single-process loop): same kernels, same order, bit-identical; the host then issues a handful "
                    "of launches per step instead of ~500. auto = as WanTI2V.denoise decides")
    ap.add_argument("--no-vae", action="store_true", help="skip the (untimed-region) VAE decode measurement")
    ap.add_argument("--no-pipeline-path", action="store_true", help="skip the (untimed-region) generation through CrossAttentionFusionPipeline, UniVid's own entry point")
    ap.add_argument("--no-default-shape", action="store_true", help="skip the (untimed-region) 3-step run at UniVid's default 121-frame latent (L = 27 280)")
    ap.add_argument("--no-stress-shape", action="store_true", help="skip the (untimed-region) full-stack forward pair at the literal 49x90x160 latent (~35 s)")
    ap.add_argument("--stress-blocks", type=int, default=0, help="developer runs: the stress-shape probe on a partial stack of this many blocks")
    ap.add_argument("--no-ranker", action="store_true", help="skip the (untimed-region) SigLIP2 ranker measurement (BASELINE config 5)")
    ap.add_argument("--layers", type=int, default=None, help="debug only: fewer DiT blocks (result is NOT the metric)")
    ap.add_argument("--splitk-strip", action="store_true", help="A/B: ffn.2's leftover rows as one round of 256x256 tiles x split-K 4 (uv_gemm_bf16_nt_ws) instead of "
                    "the 128x128 ring; opt-in because its rows are not bit-identical to the unsplit accumulation (DESIGN 9, round 6)")
    ap.add_argument("--host-sync", choices=["auto", "blocking", "default"], default="auto", help="host scheduling policy of a rank: auto = "
                    "hipDeviceScheduleBlockingSync, the runtime's own dispatch mode; blocking = that plus AMD_DIRECT_DISPATCH=0 (3-5 ms of host CPU per step "
                    "instead of 120+; see the top of this file); default = the runtime's own (a waiting thread spins)")
    ap.add_argument("--host-probe", action="store_true", help="internal: one short generation through the pipeline path under the ranks' host policy; prints "
                    "its host CPU per step (started as a child by the N = 1 run, before that run touches the GPU)")
    ap.add_argument("--kernel-times", action="store_true", help="HIP-event timing of every kernel class (adds ~1%% overhead)")
    ap.add_argument("--shape", choices=["A", "B"], default="A", help="A (default, the metric): 49-frame 704x1280 latent [48,13,44,80], "
                    "L = 11 440. B: the literal '49x90x160 latent' stress shape of BASELINE.json's target, [48,49,90,160], L = 176 400 "
                    "(26 PFLOP per step: use --steps 1 --warmup 1; not the metric)")
    ap.add_argument("--sp", action="store_true", help="N > 1: Ulysses sequence parallelism (ONE sample's tokens sharded over the "
                    "ranks, strong scaling) instead of the default one-sample-per-GPU replicas")
    ap.add_argument("--cfg-parallel", action="store_true", help="N = 2: the cond / uncond forwards of ONE sample on the two ranks, one "
                    "all-gather of the prediction per step (strong scaling; not the metric's replica mode)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # bare `python bench.py --gpus N`: this process has NOT touched the GPU (no torch.cuda / HIP call so far); start the
        # launcher as a child, one rank per GPU, and pass its output (rank 0's JSON line) and exit code through
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=env).returncode)
    if os.environ.get("UV_BENCH_DRYRUN"):      # launcher plumbing check (tests/test_host_logic.py): no GPU is touched
        print(json.dumps({"dryrun": True, "rank": rank, "local_rank": local_rank, "world": world, "gpus": args.gpus}), flush=True)
        return
    if args.host_probe:
        return host_probe_main()
    # (before anything below touches the GPU:) the ranks' host policy measured in a child, for the pipeline_path sub-line of the N = 1 run
    host_probe = run_host_probe_child() if (world == 1 and args.gpus == 1 and args.host_sync == "auto" and not args.no_pipeline_path and not args.layers
                                            and args.shape == "A") else None
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}: start one rank per GPU "
                         f"(`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...`)")
    import torch.distributed as dist
    # host threads: the loop's host side is scalar sampler algebra; never let N ranks open N x 256-thread pools on one host
    torch.set_num_threads(max(1, min(32, (os.cpu_count() or 8) // max(world, 1))))
    # UV_BENCH_SHARE_GPU=1 (test hook, tests/test_gpu_parity.py: the multi-rank branch of this file on a 1-GPU box): every rank uses cuda:0 and the
    # group runs on gloo - RCCL needs one device per rank. Never set by the driver; the line says so ("transport").
    share_gpu = world > 1 and os.environ.get("UV_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)

    global LATENT, L_TOKENS
    if args.shape == "B":
        LATENT, L_TOKENS = (48, 49, 90, 160), 49 * 45 * 80
    from univid_amd.wan.textimage2video import TI2VConfig   # configs/wan_ti2v_5B.py:17-29
    from univid_amd import _lib
    from univid_amd.wan.fm_solvers_unipc import FlowUniPCMultistepScheduler
    TI2V_5B_CFG = {k: v for k, v in TI2VConfig.dit.items() if k not in ("model_type", "window_size", "qk_norm", "cross_attn_norm")}
    cfg = dict(TI2V_5B_CFG)
    if args.layers:
        cfg["num_layers"] = args.layers
    if args.host_sync != "default":
        _lib.host_blocking_sync(True, device)      # before this rank's first synchronize: N ranks must not spin N cores of one host
    _lib.init()
    if args.splitk_strip:
        from univid_amd.wan import model as _wm
        _wm.SPLITK_STRIP = True
    model = build_model(cfg, device, seed=0)

    use_sp = bool(args.sp and world > 1)
    if use_sp:
        model.enable_sequence_parallel()
    cfgp = None
    if args.cfg_parallel:
        if world != 2 or use_sp:
            raise SystemExit("--cfg-parallel needs exactly 2 ranks and excludes --sp")
        from univid_amd.parallel import CfgParallel
        cfgp = CfgParallel()
    shared = use_sp or cfgp is not None
    g = torch.Generator(device=device).manual_seed(42 + (0 if shared else rank))   # SP / CFG pair: every rank holds the same sample
    latent = torch.randn(*LATENT, device=device, generator=g)
    ctx = [torch.randn(77, cfg["text_dim"], device=device, generator=g) * 0.1]
    ctx_null = [torch.randn(12, cfg["text_dim"], device=device, generator=g) * 0.1]
    sched = FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1)
    sched.set_timesteps(SAMPLING_STEPS, device="cpu", shift=SHIFT)
    timesteps = sched.timesteps
    seq_len = L_TOKENS
    assert args.warmup + args.steps <= len(timesteps)

    from univid_amd.wan.textimage2video import _GraphedPair, graph_by_default
    use_graph = (args.graph == "on" or (args.graph == "auto" and graph_by_default(seq_len))) and cfgp is None and not use_sp
    runner = None

    def forward_pair(i, lat):
        if runner is not None:
            return runner(lat, float(timesteps[i]))
        tvec = torch.full((1, seq_len), float(timesteps[i]), device=device)
        # exactly what WanTI2V.denoise does per timestep: the CFG pair as one stacked pass (bit-identical per sample)
        if cfgp is not None:
            return cfgp.exchange(model([lat], t=tvec, context=[ctx[0] if cfgp.rank == 0 else ctx_null[0]], seq_len=seq_len)[0])
        return model([lat, lat], t=torch.cat([tvec, tvec]), context=[ctx[0], ctx_null[0]], seq_len=seq_len)

    def one_step(i, lat):
        cond, uncond = forward_pair(i, lat)
        return sched.step_cfg(cond.unsqueeze(0), uncond.unsqueeze(0), GUIDE, timesteps[i], lat.unsqueeze(0)).squeeze(0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the loop owns its (loop-constant) context tensors, as WanTI2V.denoise does: step-constant context work once per context
    with torch.no_grad(), model.context_cached():
        if use_graph:      # one capture, replayed by every step (WanTI2V.denoise's runner)
            runner = _GraphedPair(model, latent, ctx, ctx_null, seq_len, None)
        for i in range(args.warmup):
            latent = one_step(i, latent)
        barrier()
        # (three HIP events on the launch stream split THIS rank's timed region into its own steps and the collective - read after the
        # region, so one --gpus N run answers "compute, collective or host?" per rank without a second run)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        calls0, cpu0 = _lib.CALL_COUNT, time.process_time()
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(args.warmup, args.warmup + args.steps):
            latent = one_step(i, latent)
        cpu1, calls1 = time.process_time(), _lib.CALL_COUNT     # host work of the steps themselves (launches are asynchronous)
        ev[1].record()
        if world > 1 and not shared:  # the single collective of the path: final latents to every rank (8.8 MB/GPU)
            gathered = [torch.empty_like(latent) for _ in range(world)]
            dist.all_gather(gathered, latent)
        ev[2].record()
        barrier()
        dt = time.perf_counter() - t0
        cpu2 = time.process_time()
        mine = {"rank": rank, "ms_per_step": round(ev[0].elapsed_time(ev[1]) / args.steps, 3), "allgather_us": round(ev[1].elapsed_time(ev[2]) * 1e3, 1) if world > 1 else None,
                "host_cpu_ms_per_step_enqueue": round((cpu1 - cpu0) / args.steps * 1e3, 3),
                "host_cpu_ms_per_step_incl_wait": round((cpu2 - cpu0) / args.steps * 1e3, 3)}
        per_rank = [mine]
        if world > 1:
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
        # Live HIP-event timing of the dominant kernel (self-attention) on the launch stream, in a SEPARATE pass behind the timed region
        # (round-3 verdict: the timed steps record nothing the product does not): one more forward pair of the same loop = the 30
        # self-attention launches of a step, each in its real place between the block's other kernels.
        _lib.PROFILE = {"uv_flash_attn_bf16": []}
        _lib.PROFILE_ALL = bool(args.kernel_times)
        runner = None      # (eager launches: HIP events around individual kernels cannot be recorded inside a graph replay)
        forward_pair(min(args.warmup + args.steps, len(timesteps) - 1), latent)      # the step's two forwards (the sampler update launches no attention)
        barrier()
        prof = _lib.PROFILE
        _lib.PROFILE, _lib.PROFILE_ALL = None, False

    tmax = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt_max = float(tmax.item())
    ok = bool(torch.isfinite(latent).all().item())

    if rank == 0:
        ktimes = {}
        for name, evs in prof.items():
            if evs:
                ms = [s.elapsed_time(e) for s, e, _ in evs]
                ktimes[name] = {"launches": len(ms), "avg_ms": sum(ms) / len(ms), "total_ms": sum(ms),
                                "flops": sum(f for _, _, f in evs)}
        att = ktimes.get("uv_flash_attn_bf16")
        roofline = None
        if att:
            # self-attention launches only (Lk == L); cross-attention launches (Lk = 512) are tagged with fewer flops
            # (block 0's self-attention runs on ONE sample - the CFG pair enters it with identical rows, WanModel.dedup_twins -: the roofline
            # line is the 29 two-sample launches of the other blocks, i.e. the launches with the largest FLOP count)
            fmax = max(f for _, _, f in prof["uv_flash_attn_bf16"])
            self_ev = [(s, e, f) for s, e, f in prof["uv_flash_attn_bf16"] if f >= fmax * 0.99]
            avg_ms = sum(s.elapsed_time(e) for s, e, _ in self_ev) / len(self_ev)
            launch_flops = self_ev[0][2]     # 2 samples (cond + uncond) per launch
            achieved = launch_flops / (avg_ms * 1e-3) / 1e12
            traffic = None   # HBM-side bytes per launch: PMC counters cannot be read in-process; taken from the committed passes
            try:
                traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["self_attention_L11440"]["traffic_bytes_per_launch"]
                traffic *= launch_flops / self_attn_flops(L_TOKENS, cfg["dim"])   # samples per launch of the launches `achieved` is computed from
            except Exception:
                pass
            if args.shape != "A":
                traffic = None    # the PMC passes were taken at shape A
            nsamp = round(launch_flops / self_attn_flops(L_TOKENS, cfg["dim"]))
            roofline = {"kernel": f"{_lib.attn_kernel_name(L_TOKENS, L_TOKENS, cfg['dim'] // cfg['num_heads'], nsamp)} (self-attention, "
                                  f"{nsamp} stacked sample(s) x Lq=Lk={L_TOKENS}, {cfg['num_heads']} heads; name = what the dispatcher selected)",
                        "bound": "mfma",
                        "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                        "traffic_source": None if traffic is None else "committed PMC passes (profiles/pmc_traffic.json <- rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE), not this run",
                        "avg_launch_ms": round(avg_ms, 4), "launches_timed": len(self_ev),
                        "timed": "one extra forward pair of the same loop run right behind the timed steps (HIP events on the launch stream; nothing is recorded inside the timed region)",
                        "flops_per_launch": launch_flops}
        # what the timed step executes: context work cached; block 0's self-attention half (q/k/v/o projections + attention) once for the CFG pair
        step_flops = 2 * dit_forward_flops(L_TOKENS, cfg, executed=True) - (0 if shared else twin_saved_flops(L_TOKENS, cfg))
        step_flops_model = 2 * dit_forward_flops(L_TOKENS, cfg)               # SURVEY 8(d)'s per-step figure (context work included)
        out = {
            "metric": "denoise_steps_per_sec", "value": round((1 if shared else world) * args.steps / dt_max, 4), "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt_max / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "strong" if shared else "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("49-frame 704x1280 latent [48,13,44,80], L=11440 tokens" if args.shape == "A" else
                                    "STRESS SHAPE (not the metric): literal 49x90x160 latent [48,49,90,160], L=176400 tokens") +
                                   "; TI2V-5B DiT (dim 3072, ffn 14336, 24 heads, %d layers); 1 step = cond+uncond forward + CFG + UniPC; "
                                   "one sample per GPU" % cfg["num_layers"],
                       "samples_per_gpu": 1, "guide_scale": GUIDE, "shift": SHIFT, "parallelism": (f"ulysses sequence parallel x{world}, 4 all-to-alls per block" if use_sp else "cfg pair x2, one all-gather of the prediction per step" if cfgp is not None else f"replicas x{world}, all-gather of final latents")},
            "step_tflop": round(step_flops / 1e12, 1),
            "step_tflop_note": f"executed per timed step (block 0's self-attention half runs once for the CFG pair - identical rows until the first cross-attention -: {round(twin_saved_flops(L_TOKENS, cfg) / 1e12, 2)} TFLOP less than two full forwards); SURVEY 8(d)'s {round(step_flops_model / 1e12, 1)} TFLOP also counts text_embedding and the "
                               "cross-attention K/V projections of the (step-constant) context, which run once per context, outside the timed steps",
            "model_tflops_per_gpu": round(step_flops * args.steps / dt_max / 1e12, 1),
            "mfma_frac_whole_step": round(step_flops * args.steps / dt_max / 1e12 / PEAK_BF16_TFLOPS, 4),
            "finite": ok,
            "roofline": roofline,
            # host side of a step (at N ranks on one host, N x this against the host's cores is the scaling risk)
            "host_cpu_ms_per_step": round((cpu1 - cpu0) / args.steps * 1e3, 3),
            "launches_per_step": round((calls1 - calls0) / args.steps, 1),
            "graph": bool(use_graph),
            "per_rank": {"ms_per_step_min": min(r["ms_per_step"] for r in per_rank), "ms_per_step_max": max(r["ms_per_step"] for r in per_rank),
                         "allgather_us_max": max(r["allgather_us"] for r in per_rank) if world > 1 else None,
                         "host_cpu_ms_per_step_incl_wait_max": max(r["host_cpu_ms_per_step_incl_wait"] for r in per_rank), "ranks": per_rank,
                         "note": "per rank, HIP events on its launch stream inside the timed region: its own K steps (GPU time), then the one all-gather "
                                 "(includes waiting for the slowest rank); host CPU = process time of the rank, enqueue only / up to the closing barrier"},
            "host_sync": args.host_sync,
            "host_env": {k: os.environ.get(k) for k in ("AMD_DIRECT_DISPATCH", "HSA_ENABLE_IPC_MODE_LEGACY")},
            "ffn2_strip": "256x256 x split-K 4 (uv_gemm_bf16_nt_ws, --splitk-strip)" if args.splitk_strip else "128x128 ring",
            "host_note": "process CPU time (user + system, all threads of this rank) spent issuing one step's launches, and C-ABI entry-point calls per step",
        }
        if args.kernel_times:
            out["kernel_times"] = {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in ktimes.items()}
        out["rccl_ranks"] = world if (world > 1 and not share_gpu) else 0
        if share_gpu:
            out["transport"] = "gloo, all ranks on cuda:0 (UV_BENCH_SHARE_GPU=1 test hook: NOT a multi-GPU measurement)"
        if world == 1 and not args.no_pipeline_path and args.shape == "A":
            runner = None
            torch.cuda.empty_cache()
            try:
                out["pipeline_path"] = pipeline_path_probe(model, device, cfg, blocking=(args.host_sync != "default"))
                if host_probe is not None:
                    out["pipeline_path"].setdefault("host_sync", {})["rank_policy_child"] = host_probe
                out["pipeline_path"]["vs_headline_ms_per_step"] = round(out["pipeline_path"]["ms_per_step"] / out["ms_per_step"], 4)
            except Exception as ex:      # a side measurement: never fails the bench
                out["pipeline_path"] = {"error": repr(ex)[:300]}
        if world == 1 and not args.no_default_shape and not args.layers and args.shape == "A":
            try:
                out["default_shape"] = default_shape_probe(model, device, cfg)
            except Exception as ex:      # a side measurement: never fails the bench
                out["default_shape"] = {"error": repr(ex)[:300]}
        if world == 1 and not args.no_stress_shape and not args.layers and args.shape == "A":
            runner = None
            torch.cuda.empty_cache()
            try:
                if args.stress_blocks and args.stress_blocks != cfg["num_layers"]:
                    out["stress_shape"] = stress_shape_probe(device, dict(TI2V_5B_CFG), blocks=args.stress_blocks)
                else:
                    out["stress_shape"] = stress_shape_probe(device, dict(TI2V_5B_CFG), blocks=cfg["num_layers"], model=model)
            except Exception as ex:      # a side measurement: never fails the bench
                out["stress_shape"] = {"error": repr(ex)[:300]}
        if world == 1 and not args.no_ranker and args.shape == "A":
            try:
                out["ranker"] = ranker_probe(device)
            except Exception as ex:      # a side measurement: never fails the bench
                out["ranker"] = {"error": repr(ex)[:300]}
        if world == 1 and not args.no_vae and not args.layers and args.shape == "A":
            model = None                                         # the 30 GB of DiT weights are not needed any more
            torch.cuda.empty_cache()
            # headline = the VAE's DEFAULT arithmetic since round 4: f16x3 (f32-grade: as close to an fp64 convolution as the exact f32
            # MFMA, every element of the full clip inside rtol 1e-3 / atol 1e-4 of the fp32 CPU oracle - tests + profiles/r04_vae_full_clip_*);
            # the reference's dtype executed literally (exact f32 MFMA) stays beside it under vae_*_fp32, the other modes under their keys
            x3 = vae_metrics(device, "f16x3")
            out["vae_decode"], out["vae_encode"] = x3["decode"], x3["encode"]
            out["vae_precision_note"] = ("vae_decode / vae_encode = Wan2_2_VAE's default precision 'f16x3'; vae_*_fp32 = the exact-f32 MFMA mode "
                                         "(the reference's dtype, the headline of rounds 1-3); tensors in and out are fp32 in every mode")
            f32 = vae_metrics(device, "fp32", reps=2)
            out["vae_decode_fp32"], out["vae_encode_fp32"] = f32["decode"], f32["encode"]
            x6 = vae_metrics(device, "bf16x6", reps=2)
            out["vae_decode_bf16x6"], out["vae_encode_bf16x6"] = x6["decode"], x6["encode"]
            out["vae_decode_bf16x3"] = vae_metrics(device, "bf16x3", encode=False)["decode"]
        if world == 1 and not args.no_vae and not args.layers and args.shape == "A":
            try:
                out["hipblaslt_ref"] = hipblaslt_ref(device)
            except Exception as ex:      # calibration only: never fails the bench
                out["hipblaslt_ref"] = {"error": repr(ex)[:200]}
            try:
                out["sdpa_ref"] = sdpa_ref(device)
            except Exception as ex:      # calibration only: never fails the bench
                out["sdpa_ref"] = {"error": repr(ex)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dict(TI2V_5B_CFG))
            if not args.no_vae:
                out["cpu_baseline"]["vae_decode"] = cpu_baseline_vae()
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
