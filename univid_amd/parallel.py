"""Multi-GPU plan for the denoise hot path: independent diffusion samples shard across ranks; one collective.

The reference never communicates on this path (`use_sp=False, dit_fsdp=False, t5_fsdp=False`,
models/model_pipeline.py:2205-2207; its only multi-GPU mode is manual model placement). Each diffusion sample
(noise, prompt-embeds, seed) is independent for all sampler steps and through VAE decode, so the natural MI355X
mapping is ONE PROCESS PER GPU, one sample (or a contiguous slice of the batch) per rank, weights replicated
(10 GB bf16 + 20 GB fp32 masters of 288 GB HBM), no data-path collective, and a single all-gather of the final
latents (8.8 MB per GPU at 49x704x1280) over xGMI at the end. `torch.distributed` backend "nccl" is RCCL on ROCm;
the same code runs on the gloo backend for the CPU tests.
"""
from typing import List, Sequence

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n_samples: int, rank: int, world_size: int):
    """Contiguous, balanced slice [lo, hi) of the sample batch owned by `rank` (first n % world ranks get one extra)."""
    base, extra = divmod(n_samples, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def sample_seed(base_seed: int, sample_index: int) -> int:
    """Seed of a sample depends on its GLOBAL index only, so results do not depend on the number of GPUs."""
    return base_seed + sample_index


def gather_latents(local: Sequence[torch.Tensor], n_samples: int) -> List[torch.Tensor]:
    """All-gather the final latents of every rank's slice; returns the full list in global sample order on every rank.

    One collective per generation: ranks with a shorter slice pad with a zero latent so that a single fixed-shape
    all_gather suffices (RCCL all-gather over xGMI; ring = per-link bound, irrelevant at 8.8 MB)."""
    rank, ws = world()
    if ws == 1:
        return list(local)
    per = (n_samples + ws - 1) // ws
    if not len(local):
        raise ValueError("every rank must own at least one sample (n_samples >= world_size)")
    ref = local[0]
    buf = torch.zeros(per, *ref.shape, dtype=ref.dtype, device=ref.device)
    for i, t in enumerate(local):
        buf[i].copy_(t)
    out = [torch.empty_like(buf) for _ in range(ws)]
    dist.all_gather(out, buf)
    res = []
    for r in range(ws):
        lo, hi = shard_range(n_samples, r, ws)
        res.extend(out[r][i] for i in range(hi - lo))
    return res


def denoise_batch(pipe, noises: Sequence[torch.Tensor], contexts, contexts_null, sampling_steps, shift, guide_scale,
                  gather=True):
    """Shards `noises` (one latent per sample) over the ranks, denoises the local slice with `pipe.denoise`, and
    (optionally) all-gathers the final latents."""
    rank, ws = world()
    n = len(noises)
    lo, hi = shard_range(n, rank, ws)
    local = [pipe.denoise(noises[i], contexts[i], contexts_null[i], sampling_steps, shift, guide_scale) for i in range(lo, hi)]
    return gather_latents(local, n) if gather else local
