"""WanTI2V on MI355X: the denoise loop (2 DiT forwards + CFG + UniPC per step) and VAE decode/encode.

Mirrors /root/reference/models/wan/textimage2video.py (WanTI2V :34-619; generate :162-237, t2v :239-411,
i2v :413-619) with the same method names, keyword arguments and defaults. Differences that are deliberate:

  * the text encoder is injected (`text_encoder(list_of_prompts, device) -> list[Tensor[len, text_dim]]`); the
    umT5-XXL encoder produces the prompt-embeds that this hot path takes as INPUT and is outside its scope
    (SURVEY.md section 8, row #8). `prompt_embeds=` / `negative_prompt_embeds=` can be passed directly.
  * the device is an argument instead of the hard-coded `cuda:{id}`; offload_model is accepted and ignored
    (288 GB HBM: nothing is ever moved to the host).
  * generation errors raise instead of being swallowed.
"""
import collections
import contextlib
import math
import random
import sys
import time
from typing import Callable, List, Optional

import torch

from .fm_solvers import FlowDPMSolverMultistepScheduler, get_sampling_sigmas, retrieve_timesteps
from .fm_solvers_unipc import FlowUniPCMultistepScheduler
from .. import _lib
from .model import WanModel, _ensure_prepared


def masks_like(tensor, zero=False, generator=None, p=0.2):
    """models/wan/utils/utils.py:172-199."""
    assert isinstance(tensor, list)
    out1 = [torch.ones(u.shape, dtype=u.dtype, device=u.device) for u in tensor]
    out2 = [torch.ones(u.shape, dtype=u.dtype, device=u.device) for u in tensor]
    if zero:
        if generator is not None:
            for u, v in zip(out1, out2):
                random_num = torch.rand(1, generator=generator, device=generator.device).item()
                if random_num < p:
                    u[:, 0] = torch.normal(mean=-3.5, std=0.5, size=(1,), device=u.device,
                                           generator=generator).expand_as(u[:, 0]).exp()
                    v[:, 0] = torch.zeros_like(v[:, 0])
        else:
            for u, v in zip(out1, out2):
                u[:, 0] = torch.zeros_like(u[:, 0])
                v[:, 0] = torch.zeros_like(v[:, 0])
    return out1, out2


def best_output_size(w, h, dw, dh, expected_area):
    """models/wan/utils/utils.py:202-225."""
    ratio = w / h
    ow = (expected_area * ratio) ** 0.5
    oh = expected_area / ow
    ow1 = int(ow // dw * dw)
    oh1 = int(expected_area / ow1 // dh * dh)
    assert ow1 % dw == 0 and oh1 % dh == 0 and ow1 * oh1 <= expected_area
    ratio1 = ow1 / oh1
    oh2 = int(oh // dh * dh)
    ow2 = int(expected_area / oh2 // dw * dw)
    assert oh2 % dh == 0 and ow2 % dw == 0 and ow2 * oh2 <= expected_area
    ratio2 = ow2 / oh2
    if max(ratio / ratio1, ratio1 / ratio) < max(ratio / ratio2, ratio2 / ratio):
        return ow1, oh1
    return ow2, oh2


class TI2VConfig:
    """The constants WanTI2V reads from configs/wan_ti2v_5B.py:8-36 + shared_config.py:6-20."""
    num_train_timesteps = 1000
    param_dtype = torch.bfloat16
    text_len = 512
    vae_stride = (4, 16, 16)
    patch_size = (1, 2, 2)
    sample_fps = 24
    sample_shift = 5.0
    sample_steps = 50
    sample_guide_scale = 5.0
    frame_num = 121
    sample_neg_prompt = ""
    # checkpoint file names under checkpoint_dir (configs/wan_ti2v_5B.py:12-16)
    t5_checkpoint = "models_t5_umt5-xxl-enc-bf16.pth"
    t5_tokenizer = "google/umt5-xxl"
    vae_checkpoint = "Wan2.2_VAE.pth"
    vae_kwargs = {}      # extra Wan2_2_VAE constructor arguments (width-reduced test checkpoints); {} = the Wan2.2 VAE
    t5_kwargs = None     # T5Encoder constructor arguments of a non-XXL text-encoder checkpoint; None = umT5-XXL
    dit = dict(model_type="ti2v", patch_size=(1, 2, 2), text_len=512, in_dim=48, dim=3072, ffn_dim=14336, freq_dim=256,
               text_dim=4096, out_dim=48, num_heads=24, num_layers=30, window_size=(-1, -1), qk_norm=True,
               cross_attn_norm=True, eps=1e-6)


def graph_by_default(n_tok):
    """WanTI2V.denoise's automatic choice for the plain single-process loop: replay the CFG pair's forward from a HIP graph when the
    token count allows stacking (a multiple of 8). Small latents gain speed (a step is launch-bound: config 1 runs 3.6 x faster); at
    production sizes the GPU time is the same (kernel time is 99 % of a step) and the gain is the HOST: ~500 entry-point calls per step
    become a handful, and the launching thread no longer spins on a full launch queue for the length of the step (round 4: 212 ms of
    process CPU time per 252 ms step eager) - which is what matters when 8 ranks share one host."""
    return n_tok % 8 == 0


class _GraphedPair:
    """The cond + uncond DiT forward of one denoise step as ONE captured HIP graph, replayed every step.

    What changes from step to step is data, not structure: the latent (copied into a static buffer) and the timestep (written
    into the static table of distinct timesteps - one row for t2v, {0, t} for i2v whose first-frame tokens sit at timestep 0,
    textimage2video.py:573 - which `WanModel.forward(..., t_rows=)` takes instead of the per-token tensor). Every kernel is the
    one the eager path launches, in the same order: outputs are bit-identical.

    What changes from GENERATION to generation is the prompt, and - under UniVid's dynamic text weight - what changes during the
    first steps of a generation is the scale of the prompt's first rows (model_pipeline.py:1699-1810). Both are context work: the
    embedded context and every block's cross-attention K / V^T of it live in buffers the RUNNER owns, which its captured attention
    launches read, and `apply()` recomputes them IN PLACE (two small GEMMs + a norm per block, written straight into the buffers:
    ~3 ms for all 30 blocks at TI2V-5B size) whenever the prompt (`refresh`, at the start of every denoise, unconditionally - an
    identity / version key can be fooled by a freed-and-reallocated address, round-4 advisor finding) or a block's text weight
    differs from what its buffers hold. So one capture per (latent shape, mode, prepared weights) serves every prompt and every text-weight
    schedule: no recapture (0.4 s at 49 frames, 1 s at 121), and once the schedule has reached w = 1 (after
    int(total_sampling_steps * ratio) forwards) a step is the plain replay, bit for bit.
    """
    def __init__(self, model, latent, context, context_null, seq_len, i2v_mask, key=None):
        from . import model as _m
        dev = latent.device
        self.key = key
        L = seq_len
        self.model = model
        self.dev = dev
        # Everything the graph owns is allocated, and the capture is taken, OUTSIDE inference mode: torch's capture_begin updates its
        # own (normal) generator-state tensors in place and raises under torch.inference_mode() ("Inplace update to inference tensor
        # outside InferenceMode" - found by round 4's test of exactly that serving set-up). Reading the caller's inference tensors from
        # here is allowed; the static buffers below are normal tensors, which inference-mode code may write in place.
        # Capture and replay run with the latent's device current: torch's capture stream belongs to the current device, and the
        # kernels follow the device of their tensors (a model on cuda:1 in a process whose current device is cuda:0).
        tw, model._text_weight = model._text_weight, None      # the capture is the PLAIN forward; weights arrive through apply()
        try:
            with torch.inference_mode(False), torch.no_grad(), torch.cuda.device(dev), model.context_cached():
                self.lat = torch.empty(latent.shape, dtype=latent.dtype, device=dev)
                n_t = 1 if i2v_mask is None else 2
                self.tvals = torch.zeros(n_t, dtype=torch.float32, device=dev)
                if i2v_mask is None:
                    tid = None
                else:   # mask 0 (first latent frame) -> row 0 (timestep 0), mask 1 -> row 1 (timestep t); padding tokens never exist here
                    one = i2v_mask.to(torch.int32)
                    tid = torch.cat([one, one]).contiguous()
                self.tid = tid          # the captured kernels read it on every replay: it has to live as long as the graph (until round 5 it was a
                                        # local - freed after the capture, its memory handed to the next allocation; found as a GPU memory fault)
                ctx = [context[0], context_null[0]]
                self.lat.copy_(latent)
                # eager warm-up on the capture inputs: weight preparation, context cache, scratch buffers, function attributes
                model([self.lat, self.lat], None, ctx, L, t_rows=(self.tvals, tid, True))
                torch.cuda.synchronize(dev)
                # runner-owned copies of the context's cross-attention K / V^T, visible to the model's cache lookups during the capture only
                kvk = (model._ctx_gen, (0, 1))
                self.kv, theirs = [], []
                for b in model.blocks:
                    kl, vt = b.cross_attn._kv_cache[kvk]
                    self.kv.append((kl.clone(), vt.clone()))
                    theirs.append((kl, vt))
                    b.cross_attn._kv_cache[kvk] = self.kv[-1]
                emb = model._ctx_cache[2]
                self.emb = torch.cat([emb[0], emb[1]], 0)   # [2 * text_len, C] bf16: what the K / V projections read (own copy)
                self.keep = (model._ctx_cache, ctx)       # (the embedded contexts the captured `cat` reads; its result is unused on a cache hit)
                self.graph = torch.cuda.CUDAGraph()
                # scratch the forward caches per stream (V^T tiles): tensors born during the capture live in THIS graph's memory pool and
                # torch's capture stream is shared by all captures, so they must neither be found by a later capture (which would bake an
                # address of this pool into another graph) nor outlive this runner in a global cache (round-4 advisor finding)
                snap = _m.scratch_snapshot()
                try:
                    with torch.cuda.graph(self.graph):
                        self.out = model([self.lat, self.lat], None, ctx, L, t_rows=(self.tvals, tid, True))
                finally:
                    self.scratch = _m.scratch_take_new(snap)
                    for b, own in zip(model.blocks, theirs):
                        if kvk in b.cross_attn._kv_cache:
                            b.cross_attn._kv_cache[kvk] = own
        finally:
            model._text_weight = tw
        self.state = [(1.0, 1.0)] * len(self.kv)     # per block: the (cond, uncond) text weights its K / V^T buffers were computed with
        self._scaled = None                          # ((w_cond, w_uncond), rows) -> the row-scaled embedded context of the current prompt

    def __del__(self):
        # the graph (and its memory pool) must not be destroyed under a replay that is still running: a generation returns without
        # synchronising, and the next one - another latent shape - drops this runner right away (found by tests/test_pipeline_path.py:
        # the process aborted inside the drop)
        try:
            torch.cuda.synchronize(self.dev)
        except Exception:
            pass

    def refresh(self, context, context_null):
        """New prompt embeddings: the embedded context is recomputed and every block's K / V^T marked stale (recomputed in place by the
        next apply()). Called at the start of every denoise that reuses the runner; there is no 'unchanged' shortcut on purpose."""
        m = self.model
        with torch.inference_mode(False), torch.no_grad(), torch.cuda.device(self.dev):
            emb = m.embed_context([context[0], context_null[0]])          # [2, text_len, C] bf16
            self.emb = torch.cat([emb[0], emb[1]], 0)
        self.state = [None] * len(self.kv)
        self._scaled = None

    def apply(self, text_weight=None):
        """Brings every block's K / V^T buffers to the text weight of the coming forward pair: text_weight = None (plain) or
        ((w_cond, w_uncond), rows, layers | None) as WanModel.set_text_weight takes it. Blocks whose buffers already hold the wanted
        weights are left alone: in the plain loop this is 30 tuple comparisons per step."""
        plain = (1.0, 1.0)
        w, rows, layers = plain, 0, None
        if text_weight is not None:
            w, rows, layers = (float(text_weight[0][0]), float(text_weight[0][1])), int(text_weight[1]), text_weight[2]
        active = w != plain and rows > 0

        def want(li):
            return w if (active and (layers is None or li in layers)) else plain
        stale = [li for li in range(len(self.kv)) if self.state[li] != want(li)]
        if not stale:
            return 0
        m = self.model
        Lc = m.text_len
        with torch.inference_mode(False), torch.no_grad(), torch.cuda.device(self.dev):
            for li in stale:
                hooked = want(li) != plain
                src = self.emb
                if hooked:
                    if self._scaled is None or self._scaled[0] != (w, rows):
                        sc = torch.empty_like(self.emb)
                        for j in range(2):
                            _lib.text_weight_rows(self.emb[j * Lc:(j + 1) * Lc], sc[j * Lc:(j + 1) * Lc], min(rows, Lc) if w[j] != 1.0 else 0, w[j])
                        self._scaled = ((w, rows), sc)
                    src = self._scaled[1]
                m.blocks[li].cross_attn._context_kv(src, Lc, 2, None, out=self.kv[li])
                self.state[li] = w if hooked else plain
        return len(stale)

    def __call__(self, latent, t, text_weight=None):
        with torch.cuda.device(self.dev):
            self.apply(text_weight)
            self.lat.copy_(latent)
            self.tvals[-1:].fill_(t)         # the value travels as a kernel argument (no host buffer to race with); row 0 stays 0 for i2v
            self.graph.replay()
        return self.out[0], self.out[1]      # static outputs: consumed by the sampler update before the next replay


class WanTI2V:
    def __init__(self, config=TI2VConfig, checkpoint_dir=None, device_id=0, rank=0, t5_fsdp=False, dit_fsdp=False,
                 use_sp=False, t5_cpu=False, init_on_cpu=True, convert_model_dtype=False, *, model: WanModel = None,
                 vae=None, text_encoder: Optional[Callable] = None, device=None):
        if t5_fsdp or dit_fsdp:
            raise NotImplementedError("FSDP is not part of this build (UniVid passes False, models/model_pipeline.py:2205-2207; "
                                      "the fp32 masters + bf16 operands of the 5B DiT are 30 GB of a 288 GB GPU)")
        if convert_model_dtype:
            raise NotImplementedError("convert_model_dtype=True (bf16 parameters) is not UniVid's setting")
        self.device = torch.device(device) if device is not None else torch.device(f"cuda:{device_id}")
        self.config = config
        self.rank = rank
        self.t5_cpu = t5_cpu
        self.init_on_cpu = False
        self.num_train_timesteps = config.num_train_timesteps
        self.param_dtype = config.param_dtype
        self.vae_stride = config.vae_stride
        self.patch_size = config.patch_size
        self.sp_size = 1
        self.cfgp = None
        # captured _GraphedPair runners (HIP graph of the CFG pair's forward), most recently used last: a service that alternates t2v and i2v
        # (or two clip lengths) replays both instead of re-capturing on every switch (a capture = one eager pair forward + the capture pass,
        # 0.65 s at L = 11 440 - 13 ms per step of a 50-step generation). Each holds its graph's activation pool; the oldest goes first
        self._runners = collections.OrderedDict()
        self.max_graph_runners = 2
        self._progress, self._steps_issued = None, 0
        self.max_steps_in_flight = 4       # sampler steps the host may queue ahead of the GPU (WanTI2V._steps)
        self.text_weight_schedule = None   # object with next_pair() / rows(text_len) / layers: UniVid's dynamic text weight, native
        self.sample_neg_prompt = config.sample_neg_prompt
        self.text_encoder = text_encoder
        self.vae = vae
        if model is None:
            if checkpoint_dir is None:
                raise ValueError("pass model= (a WanModel) or checkpoint_dir=")
            from .checkpoint import load_wan_model
            model = load_wan_model(checkpoint_dir)
        self.model = model.eval().requires_grad_(False).to(self.device)
        if checkpoint_dir is not None:
            # textimage2video.py:88-103: the VAE and the text encoder come from the same directory. Each is loaded when its
            # file is there and no instance was injected; a missing file leaves the stage to be injected (vae= / text_encoder=)
            import os
            vae_pth = os.path.join(checkpoint_dir, getattr(config, "vae_checkpoint", "Wan2.2_VAE.pth"))
            if self.vae is None and os.path.exists(vae_pth):
                from .vae2_2 import Wan2_2_VAE
                self.vae = Wan2_2_VAE(vae_pth=vae_pth, device=self.device, **dict(getattr(config, "vae_kwargs", {}) or {}))
            t5_pth = os.path.join(checkpoint_dir, getattr(config, "t5_checkpoint", "models_t5_umt5-xxl-enc-bf16.pth"))
            t5_tok = os.path.join(checkpoint_dir, getattr(config, "t5_tokenizer", "google/umt5-xxl"))
            if self.text_encoder is None and os.path.exists(t5_pth) and os.path.isdir(t5_tok):
                from .t5 import T5EncoderModel
                self.text_encoder = T5EncoderModel(text_len=config.text_len, dtype=torch.bfloat16, device=self.device, checkpoint_path=t5_pth,
                                                   tokenizer_path=t5_tok, encoder_kwargs=getattr(config, "t5_kwargs", None))
        if use_sp:
            # textimage2video.py:106-118: Ulysses sequence parallelism over all ranks of the default process group; every rank
            # runs the same sample (same seed) and holds the full result after each forward
            self.model.enable_sequence_parallel()
            self.sp_size = self.model.sp.size

    def _runner_for(self, key, make):
        """The cached runner of `key` = (latent shape, i2v, prepared-weights generation, text_len), moved to the most-recently-used end, or
        `make()` stored under it. Returns (runner, fresh). Before a capture: runners of other prepared weights / another text_len are
        dropped (they can never be asked for again) and the least recently used ones go until there is room - their graphs' pools are
        freed BEFORE the new capture allocates its own. A `make()` that raises leaves nothing behind under `key`."""
        runner = self._runners.pop(key, None)
        fresh = runner is None
        if fresh:
            for stale in [k for k in self._runners if k[2:] != key[2:]]:
                del self._runners[stale]
            while len(self._runners) >= max(1, self.max_graph_runners):
                self._runners.popitem(last=False)
            runner = make()
        self._runners[key] = runner
        return runner, fresh

    @property
    def _runner(self):
        """The runner of the last graph-mode denoise (None before the first)."""
        return next(reversed(self._runners.values())) if self._runners else None

    @_runner.setter
    def _runner(self, value):
        if value is not None:
            raise ValueError("_runner can only be cleared")
        self._runners.clear()

    def enable_cfg_parallel(self, group=None):
        """Extension: the ranks of `group` (exactly 2) split the cond / uncond forwards of every step and exchange
        the predictions with one all-gather; all ranks must pass the same inputs (same seed) and end with the same latent."""
        from ..parallel import CfgParallel
        self.cfgp = CfgParallel(group)
        return self

    # ---- prompt embeds ---------------------------------------------------------------------------------------
    def _encode(self, prompt, embeds):
        if embeds is not None:
            return [e.to(self.device) for e in embeds]
        if self.text_encoder is None:
            raise ValueError("no text_encoder was given: pass prompt_embeds=/negative_prompt_embeds= "
                             "(list of [len<=512, 4096] tensors)")
        return [t.to(self.device) for t in self.text_encoder([prompt], self.device)]

    def generate(self, input_prompt, img=None, size=(1280, 704), max_area=704 * 1280, frame_num=81, shift=5.0,
                 sample_solver="unipc", sampling_steps=50, guide_scale=5.0, n_prompt="", seed=-1, offload_model=True,
                 **kw):
        """textimage2video.py:162-237."""
        if img is not None:
            return self.i2v(input_prompt=input_prompt, img=img, max_area=max_area, frame_num=frame_num, shift=shift,
                            sample_solver=sample_solver, sampling_steps=sampling_steps, guide_scale=guide_scale,
                            n_prompt=n_prompt, seed=seed, offload_model=offload_model, **kw)
        return self.t2v(input_prompt=input_prompt, size=size, frame_num=frame_num, shift=shift,
                        sample_solver=sample_solver, sampling_steps=sampling_steps, guide_scale=guide_scale,
                        n_prompt=n_prompt, seed=seed, offload_model=offload_model, **kw)

    # ---- the hot loop ----------------------------------------------------------------------------------------
    def denoise(self, noise, context, context_null, sampling_steps, shift, guide_scale, z=None, record=None, graph=None,
                sample_solver="unipc"):
        # the loop owns its context tensors for its duration: step-constant context work is computed once (WanModel.context_cached)
        cached = self.model.context_cached() if hasattr(self.model, "context_cached") else contextlib.nullcontext()
        with cached:
            return self._denoise(noise, context, context_null, sampling_steps, shift, guide_scale, z, record, graph, sample_solver)

    def _denoise(self, noise, context, context_null, sampling_steps, shift, guide_scale, z=None, record=None, graph=None,
                 sample_solver="unipc"):
        """Steps of t2v (:356-394) / i2v (:548-601) on a given noise latent [C, f, h, w] (fp32, on device).

        z: first-frame latent [C, 1, h, w] switches on the i2v masking (mask2 zero on frame 0, :550-551, :598).
        record: optional list receiving (noise_pred, latent) per step (costs one extra latent write per step).
        graph: replay the CFG pair's DiT forward from a captured HIP graph (one capture per latent shape, all steps replay it;
            the sampler update stays eager because its coefficients are host scalars that change every step). None = automatic
            (graph_by_default): on for the plain single-process loop whenever the token count allows stacking - small latents gain speed
            (a step is launch-bound), production sizes keep their GPU time (253.7 against 253.4 ms per step at L = 11 440) and free the
            host: 3 instead of 495 entry-point calls and 1 ms instead of 229 ms of process CPU time per step; off whenever a hook or a
            parallel mode owns the forward. Results are bit-identical either way (same kernels, same order).
        sample_solver: 'unipc' (:335-342) or 'dpm++' (:343-351), as in the reference.
        Returns the final latent.
        """
        dev = self.device
        if sample_solver == "unipc":
            sched = FlowUniPCMultistepScheduler(num_train_timesteps=self.num_train_timesteps, shift=1,
                                                use_dynamic_shifting=False)
            sched.set_timesteps(sampling_steps, device="cpu", shift=shift)
            timesteps = sched.timesteps
        elif sample_solver == "dpm++":
            sched = FlowDPMSolverMultistepScheduler(num_train_timesteps=self.num_train_timesteps, shift=1,
                                                    use_dynamic_shifting=False)
            timesteps, _ = retrieve_timesteps(sched, device="cpu", sigmas=get_sampling_sigmas(sampling_steps, shift))
        else:
            raise NotImplementedError("Unsupported solver.")
        latent = noise.to(device=dev, dtype=torch.float32).contiguous()
        i2v = z is not None
        _, mask2 = masks_like([latent], zero=i2v)
        if i2v:
            z = z.to(device=dev, dtype=torch.float32)
            latent = ((1.0 - mask2[0]) * z + mask2[0] * latent).contiguous()
        c, f, h, w = latent.shape
        seq_len = math.ceil((h * w) / (self.patch_size[1] * self.patch_size[2]) * f / self.sp_size) * self.sp_size
        base_mask = mask2[0][0][:, ::2, ::2].flatten()
        foreign = "forward" in self.model.__dict__ or any("forward" in b.cross_attn.__dict__ for b in self.model.blocks)
        plain = self.cfgp is None and self.sp_size == 1 and not foreign
        # the native text-weight schedule (set by univid_amd.model_pipeline.Wan22ContextWrapper.generate for the length of one generation)
        # rides on the plain path; with re-assigned forwards (somebody else's hooks) the closures own the context and it stays out
        tws = None if foreign else self.text_weight_schedule
        n_tok = base_mask.numel()
        if graph is None:
            graph = plain and graph_by_default(n_tok)
        elif graph and not plain:
            raise NotImplementedError("graph=True needs the plain single-process forward (no text-weight hook, CFG pair or SP mode)")
        runner = None
        if graph:
            # one captured graph per (latent shape, mode, contexts, prepared weights): generations that repeat them replay it
            _ensure_prepared(self.model)
            # one captured graph per (latent shape, mode, prepared weights); the prompt travels through the runner's own context buffers,
            # recomputed in place at the start of every call (_GraphedPair.refresh): no recapture for a new prompt, nothing to go stale
            key = (tuple(latent.shape), i2v, self.model._prep_gen, self.model.text_len)
            runner, fresh = self._runner_for(key, lambda: _GraphedPair(self.model, latent, context, context_null, seq_len,
                                                                       base_mask if i2v else None, key))
            if not fresh:
                runner.refresh(context, context_null)
        try:
            return self._steps(sched, timesteps, latent, context, context_null, guide_scale, z, mask2, base_mask, seq_len, record, runner, tws)
        finally:
            if tws is not None and hasattr(self.model, "set_text_weight"):
                self.model.set_text_weight(None)

    def _steps(self, sched, timesteps, latent, context, context_null, guide_scale, z, mask2, base_mask, seq_len, record, runner, tws):
        dev = self.device
        i2v = z is not None
        # The host runs ahead of the GPU (nothing in a step synchronises). Unbounded, it runs into the HIP runtime's launch queue: from about
        # the 23rd queued graph replay on, the launch call blocks by SPINNING (measured at L = 11 440: a 50-step generation burned 13.7 s of
        # CPU time on two threads for 12.6 s of GPU work, while 20 steps cost 9 ms) - one or two busy cores per rank, for nothing. So the loop
        # keeps at most `max_steps_in_flight` steps queued: the GPU bumps a counter in pinned host memory at the end of every step (one scalar
        # fill + one 8-byte asynchronous copy) and the host, when it is that far ahead, naps in 1-ms slices reading that counter as plain
        # memory - no HIP call in the wait (hipEventSynchronize, blocking flag or not, and a hipEventQuery poll both kept a core busy:
        # measured). The GPU never starves: the queue still holds the other steps' work.
        for t in timesteps:
            self._throttle()
            tw = None
            if tws is not None:
                # UniVid's dynamic text weight, native (model_pipeline.py:1699-1810, 1844-1886): the wrapper's counter advances once per
                # DiT forward, the cond forward first (:1856-1864, textimage2video.py:380-385) - the stacked CFG pair carries both values
                wc, wu = tws.next_pair()
                if wc != 1.0 or wu != 1.0:
                    tw = ((wc, wu), tws.rows(self.model.text_len), tws.layers)
            if runner is not None:
                cond, uncond = runner(latent, float(t), tw)
                res = sched.step_cfg(cond.unsqueeze(0), uncond.unsqueeze(0), guide_scale, t, latent.unsqueeze(0),
                                     want_noise_pred=record is not None)
                if record is not None:
                    latent, npred = res[0].squeeze(0), res[1].squeeze(0)
                else:
                    latent = res.squeeze(0)
                if i2v:
                    latent = ((1.0 - mask2[0]) * z + mask2[0] * latent).contiguous()
                if record is not None:
                    record.append((npred, latent.clone()))
                self._step_done()
                continue
            ts = torch.stack([t]).to(dev)
            temp_ts = base_mask * ts                                                   # :373
            temp_ts = torch.cat([temp_ts, temp_ts.new_ones(seq_len - temp_ts.size(0)) * ts])
            tvec = temp_ts.unsqueeze(0)
            if tws is not None:
                if tw is None:
                    self.model.set_text_weight(None)
                else:      # (CFG parallel: this rank runs ONE of the pair's forwards and takes that forward's value of the counter)
                    self.model.set_text_weight(tw[0] if self.cfgp is None else (tw[0][self.cfgp.rank],), tw[1], tw[2])
            if self.cfgp is not None:
                if "forward" in self.model.__dict__:
                    raise NotImplementedError("a re-assigned model.forward (the reference's per-forward text-weight counter, model_pipeline.py"
                                              ":1856-1868) counts two forwards per step on one rank; with CFG parallelism use the native "
                                              "schedule (Wan22ContextWrapper(native=True)): each rank takes its own value of the pair")
                mine = context if self.cfgp.rank == 0 else context_null
                cond, uncond = self.cfgp.exchange(self.model([latent], t=tvec, context=mine, seq_len=seq_len)[0])
            elif "forward" in self.model.__dict__:
                # model.forward was re-assigned (UniVid's per-forward text-weight counter, model_pipeline.py:1856-1868):
                # keep the reference's two calls so the counter advances exactly as there
                cond = self.model([latent], t=tvec, context=context, seq_len=seq_len)[0]
                uncond = self.model([latent], t=tvec, context=context_null, seq_len=seq_len)[0]
            else:
                # cond + uncond as one stacked pass (bit-identical per sample, better occupancy)
                cond, uncond = self.model([latent, latent], t=torch.cat([tvec, tvec]), context=[context[0], context_null[0]],
                                          seq_len=seq_len)
            res = sched.step_cfg(cond.unsqueeze(0), uncond.unsqueeze(0), guide_scale, t, latent.unsqueeze(0),
                                 want_noise_pred=record is not None)
            if record is not None:
                latent, npred = res[0].squeeze(0), res[1].squeeze(0)
            else:
                latent = res.squeeze(0)
            if i2v:
                latent = ((1.0 - mask2[0]) * z + mask2[0] * latent).contiguous()       # :598
            if record is not None:
                record.append((npred, latent.clone()))
            self._step_done()
        return latent

    # ---- host pacing: a step counter the GPU writes into pinned host memory at the end of every step; the host reads it as plain memory.
    # One counter pair per pipeline, written on the stream the loop runs on: a pipeline object serves ONE stream at a time (as the reference's
    # single-threaded, default-stream loop does); two threads driving one WanTI2V on two streams would interleave its counter writes.
    def _step_done(self):
        if self._progress is None:
            with torch.inference_mode(False):      # normal tensors: written in place by later calls inside or outside inference mode
                self._progress = (torch.zeros(1, dtype=torch.int64, device=self.device), torch.zeros(1, dtype=torch.int64).pin_memory())
        dev_ctr, host_ctr = self._progress
        dev_ctr.fill_(self._steps_issued + 1)
        host_ctr.copy_(dev_ctr, non_blocking=True)
        self._steps_issued += 1                    # only once both are enqueued: an exception above must not leave the host a step ahead

    def _throttle(self):
        if self._progress is None:
            return
        host_ctr = self._progress[1]
        waited = 0
        limit = max(1, int(self.max_steps_in_flight))      # (0 or less would nap forever)
        while self._steps_issued - int(host_ctr[0]) >= limit:
            time.sleep(1e-3)
            waited += 1
            if waited % 60000 == 0:      # a minute without progress: let a device fault surface as an error instead of napping forever
                torch.cuda.synchronize(self.device)

    def _noise(self, shape, seed):
        seed = seed if seed >= 0 else random.randint(0, sys.maxsize)
        g = torch.Generator(device=self.device)
        g.manual_seed(seed)
        return torch.randn(*shape, dtype=torch.float32, device=self.device, generator=g)

    def t2v(self, input_prompt, size=(1280, 704), frame_num=121, shift=5.0, sample_solver="unipc", sampling_steps=50,
            guide_scale=5.0, n_prompt="", seed=-1, offload_model=True, *, prompt_embeds=None, negative_prompt_embeds=None,
            noise=None, decode=True):
        """textimage2video.py:239-411. Returns the video [3, N, H, W] in [-1, 1] (or the latent if decode=False)."""
        F = frame_num
        z_dim = self.model.in_dim
        target_shape = (z_dim, (F - 1) // self.vae_stride[0] + 1, size[1] // self.vae_stride[1],
                        size[0] // self.vae_stride[2])
        if n_prompt == "":
            n_prompt = self.sample_neg_prompt
        context = self._encode(input_prompt, prompt_embeds)
        context_null = self._encode(n_prompt, negative_prompt_embeds)
        if noise is None:
            noise = self._noise(target_shape, seed)
        with torch.no_grad():
            x0 = self.denoise(noise, context, context_null, sampling_steps, shift, guide_scale, sample_solver=sample_solver)
            if not decode:
                return x0
            if self.vae is None:
                raise ValueError("no VAE was given: pass vae= or call with decode=False")
            return self.vae.decode([x0])[0]

    def i2v(self, input_prompt, img, max_area=704 * 1280, frame_num=121, shift=5.0, sample_solver="unipc",
            sampling_steps=40, guide_scale=5.0, n_prompt="", seed=-1, offload_model=True, *, prompt_embeds=None,
            negative_prompt_embeds=None, noise=None, decode=True):
        """textimage2video.py:413-619. `img` is a PIL image, or a float tensor [3, H, W] in [-1, 1] that already has
        the output size (the PIL resize/crop of :462-477 is host-side image I/O)."""
        if self.vae is None:
            raise ValueError("i2v needs a VAE (first-frame encode, :512)")
        dh, dw = self.patch_size[1] * self.vae_stride[1], self.patch_size[2] * self.vae_stride[2]
        if torch.is_tensor(img):
            ih, iw = img.shape[-2:]
            ow, oh = best_output_size(iw, ih, dw, dh, max_area)
            if (ow, oh) != (iw, ih):
                raise ValueError(f"tensor image must already be {ow}x{oh} (best_output_size); got {iw}x{ih}")
            img_t = img.to(self.device, torch.float32).unsqueeze(1)
        else:
            import numpy as np
            from PIL import Image
            ih, iw = img.height, img.width
            ow, oh = best_output_size(iw, ih, dw, dh, max_area)
            scale = max(ow / iw, oh / ih)
            img = img.resize((round(iw * scale), round(ih * scale)), Image.LANCZOS)
            x1, y1 = (img.width - ow) // 2, (img.height - oh) // 2
            img = img.crop((x1, y1, x1 + ow, y1 + oh))
            arr = torch.from_numpy(np.asarray(img.convert("RGB"), dtype=np.float32) / 255.0).permute(2, 0, 1)
            img_t = arr.sub_(0.5).div_(0.5).to(self.device).unsqueeze(1)
        F = frame_num
        shape = (self.model.in_dim, (F - 1) // self.vae_stride[0] + 1, oh // self.vae_stride[1], ow // self.vae_stride[2])
        if noise is None:
            noise = self._noise(shape, seed)
        if n_prompt == "":
            n_prompt = self.sample_neg_prompt
        context = self._encode(input_prompt, prompt_embeds)
        context_null = self._encode(n_prompt, negative_prompt_embeds)
        with torch.no_grad():
            z = self.vae.encode([img_t])[0]
            x0 = self.denoise(noise, context, context_null, sampling_steps, shift, guide_scale, z=z, sample_solver=sample_solver)
            return self.vae.decode([x0])[0] if decode else x0
