"""TEST INFRASTRUCTURE ONLY (the checker; `univid_amd` must not import this).

CPU restatement of the WanTI2V denoise loops (/root/reference/models/wan/textimage2video.py: t2v :283-411,
i2v :461-619) with the text encoder and VAE factored out: inputs are (noise, prompt-embeds, negative
prompt-embeds [, first-frame latent z]) and the output is the final latent plus the per-step noise
predictions. Also restates masks_like (models/wan/utils/utils.py:172-199) and UniVid's dynamic
text-weight schedule (models/model_pipeline.py:1699-1735, 1756-1803, 1844-1886).
"""
import math

import torch

from . import dpmpp, unipc, wan_dit


def masks_like(tensors, zero=False):
    """utils.py:172-199 without the random branch (generator=None in both WanTI2V loops)."""
    out1 = [torch.ones(u.shape, dtype=u.dtype, device=u.device) for u in tensors]
    out2 = [torch.ones(u.shape, dtype=u.dtype, device=u.device) for u in tensors]
    if zero:
        for u, v in zip(out1, out2):
            u[:, 0] = 0
            v[:, 0] = 0
    return out1, out2


def text_weight(step, total_steps, ratio=0.4, w_max=1.3, w_min=1.0, schedule="cosine", enabled=True):
    """Wan22ContextWrapper._calculate_text_weight model_pipeline.py:1699-1735.
    `step` counts DiT FORWARD calls (two per sampler step), :1856-1864."""
    if not enabled:
        return 1.0
    transition = int(total_steps * ratio)
    if step >= transition:
        return w_min
    progress = step / max(transition, 1)
    if schedule == "linear":
        return w_max - (w_max - w_min) * progress
    if schedule == "cosine":
        return w_min + (w_max - w_min) * (1 + math.cos(math.pi * progress)) / 2
    if schedule == "exponential":
        return w_min + (w_max - w_min) * math.exp(-5 * progress)
    return 1.0


def context_mask(ctx_shape, w, bagel_sequence_length=128, dtype=torch.bfloat16):
    """The hook's weight_mask (model_pipeline.py:1788-1797): ones, first text_len rows scaled by w."""
    seq = ctx_shape[1]
    text_len = min(bagel_sequence_length, seq // 2)
    m = torch.ones(ctx_shape, dtype=dtype)
    m[:, :text_len, :] *= w
    return m


def seq_len_of(latent_shape, patch=(1, 2, 2)):
    """textimage2video.py:289-291 with sp_size=1."""
    _, f, h, w = latent_shape
    return math.ceil((h * w) / (patch[1] * patch[2]) * f)


def denoise(sd, cfg, noise, context, context_null, steps, shift, guide_scale, z=None, text_weight_cfg=None,
            record=None, sample_solver="unipc"):
    """The hot loop of t2v (:356-394) / i2v (:548-601; enabled by passing the first-frame latent z).

    text_weight_cfg: None, or dict(total_steps, ratio, w_max, w_min, schedule, bagel_sequence_length)
    which activates UniVid's per-layer context scaling with its forward-call counter.
    record: optional list that receives (noise_pred, latent) per step.
    sample_solver: 'unipc' (:335-342) or 'dpm++' (:343-351: sigmas from get_sampling_sigmas through retrieve_timesteps).
    """
    if sample_solver == "unipc":
        sched = unipc.FlowUniPC(num_train_timesteps=1000, shift=1)
        timesteps = sched.set_timesteps(steps, shift=shift)
    elif sample_solver == "dpm++":
        sched = dpmpp.FlowDPMpp(num_train_timesteps=1000, shift=1)
        timesteps = sched.set_timesteps(sigmas=dpmpp.get_sampling_sigmas(steps, shift))
    else:
        raise NotImplementedError("Unsupported solver.")                                # :352-353
    latent = noise
    i2v = z is not None
    _, mask2 = masks_like([noise], zero=i2v)
    if i2v:
        latent = (1.0 - mask2[0]) * z + mask2[0] * latent                             # :551
    seq_len = seq_len_of(noise.shape, cfg["patch_size"])
    fwd_counter = [0]

    def run(ctx, tvec):
        hook = None
        if text_weight_cfg is not None:
            w = text_weight(fwd_counter[0], text_weight_cfg["total_steps"], text_weight_cfg.get("ratio", 0.4),
                            text_weight_cfg.get("w_max", 1.3), text_weight_cfg.get("w_min", 1.0),
                            text_weight_cfg.get("schedule", "cosine"))
            fwd_counter[0] += 1
            if w != 1.0:
                m = context_mask((1, cfg["text_len"], cfg["dim"]), w, text_weight_cfg.get("bagel_sequence_length", 128)).to(noise.device)
                hook = lambda layer: m
        return wan_dit.dit_forward(sd, cfg, [latent], tvec, ctx, seq_len, context_scale_fn=hook)[0]

    for t in timesteps:
        ts = torch.stack([t]).to(noise.device)          # (device move: a no-op on the CPU)
        temp = (mask2[0][0][:, ::2, ::2] * ts).flatten()                                # :373
        temp = torch.cat([temp, temp.new_ones(seq_len - temp.size(0)) * ts])
        tvec = temp.unsqueeze(0)
        cond = run(context, tvec)
        uncond = run(context_null, tvec)
        noise_pred = uncond + guide_scale * (cond - uncond)                             # :385
        latent = sched.step(noise_pred.unsqueeze(0), t, latent.unsqueeze(0)).squeeze(0)
        if i2v:
            latent = (1.0 - mask2[0]) * z + mask2[0] * latent                         # :598
        if record is not None:
            record.append((noise_pred.clone(), latent.clone()))
    return latent
