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


_blocking_set = set()

# What a one-rank-per-GPU launcher MAY export before its ranks initialise HIP (the runtime reads its environment then), on hosts whose cores matter:
#   AMD_DIRECT_DISPATCH=0        HIP-graph replays are submitted by the runtime's command thread, which blocks while launches are pending; under
#                                the default (direct dispatch) a runtime thread spins instead - one busy core per rank (bench.py: 123 ms of CPU
#                                per 251 ms step, against 3.4-5.4 ms with this and host_policy() below; same step time; profiles/r06_host_policy.md).
#                                Measured at N = 1 only: RCCL has never run under it here (no multi-GPU box), so bench.py's own N > 1 ranks keep
#                                the runtime's dispatch mode unless started with --host-sync blocking.
#   HSA_ENABLE_IPC_MODE_LEGACY=0 dmabuf IPC for RCCL on this pool's driver
RANK_ENV = {"AMD_DIRECT_DISPATCH": "0", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}


def host_policy(device, blocking_sync=True):
    """One rank per GPU on ONE host: a rank that waits for its GPU must SLEEP (hipDeviceScheduleBlockingSync), not spin - with the runtime's
    default policy every waiting rank keeps a core busy (round 5: 209-235 ms of process CPU per 252 ms step), which at 8 ranks is the host
    contention SURVEY 8(e) names as the scaling risk. Idempotent per device; called by denoise_batch for multi-rank jobs and by bench.py's
    ranks; a single-process user calls it (or not) as it sees fit - it is process-wide policy, so importing the package never sets it."""
    device = torch.device(device)
    if device.type != "cuda" or (device.index, blocking_sync) in _blocking_set:
        return
    from . import _lib
    _lib.host_blocking_sync(blocking_sync, device)
    _blocking_set.discard((device.index, not blocking_sync))
    _blocking_set.add((device.index, blocking_sync))


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
    if n_samples < ws:
        # fewer samples than ranks: the trailing ranks own nothing and learn the latent's shape / dtype from rank 0 (which always owns
        # sample 0), then take part in the one all-gather with an all-zero buffer
        meta = [(tuple(local[0].shape), local[0].dtype) if len(local) else None]
        dist.broadcast_object_list(meta, src=0)
        shape, dtype = meta[0]
        if len(local):
            device = local[0].device
        else:
            device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    else:
        shape, dtype, device = tuple(local[0].shape), local[0].dtype, local[0].device
    buf = torch.zeros(per, *shape, dtype=dtype, device=device)
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
                  gather=True, wrapper=None):
    """Shards `noises` (one latent per sample) over the ranks, denoises the local slice with `pipe.denoise`, and
    (optionally) all-gathers the final latents.

    wrapper: this rank's univid_amd.model_pipeline.Wan22ContextWrapper around `pipe` - UniVid's pipeline path (model_pipeline.py
    :1844-1886): every sample is one generation, i.e. runs inside `wrapper.scheduled()`, whose forward counter (the dynamic text
    weight's clock) belongs to this rank's wrapper and restarts with every sample - so a sample's result does not depend on which rank
    ran it or on what that rank ran before."""
    rank, ws = world()
    n = len(noises)
    lo, hi = shard_range(n, rank, ws)
    if ws > 1 and len(noises) and noises[0].is_cuda:
        host_policy(noises[0].device)

    def one(i):
        if wrapper is not None and wrapper.config.use_dynamic_text_weight:
            with wrapper.scheduled():
                return pipe.denoise(noises[i], contexts[i], contexts_null[i], sampling_steps, shift, guide_scale)
        return pipe.denoise(noises[i], contexts[i], contexts_null[i], sampling_steps, shift, guide_scale)

    local = [one(i) for i in range(lo, hi)]
    return gather_latents(local, n) if gather else local


# ---------------------------------------------------------------------------------------------------------------
# Ulysses sequence parallelism: ONE sample's tokens sharded over the ranks of a group.
#
# Reference: models/wan/distributed/sequence_parallel.py:64-176 (sp_dit_forward / sp_attn_forward) and ulysses.py:9-47
# (distributed_attention = all_to_all(q,k,v: scatter heads, gather tokens) -> flash_attention -> all_to_all back).
# Here every row-wise kernel (LayerNorm, the projections, RMSNorm + RoPE with a global row offset, cross-attention, FFN)
# runs on the rank's contiguous token range [r0, r1); only self-attention needs the other ranks' tokens, and gets them by
# four all-to-alls per block and sample: q and k as [tokens, heads/p], V already TRANSPOSED as [heads/p * D, tokens] (the
# layout the attention kernel reads), the attention output back as [tokens/p, heads]. On MI355X these are RCCL
# all-to-alls over xGMI (backend "nccl"): 8.8 MB per rank and tensor at L = 11 440, p = 8. The gloo backend has no
# all-to-all, so the CPU tests (and the two-processes-on-one-GPU parity test) emulate it with an all-gather through host
# memory - a test transport, not a compute fallback: all arithmetic stays in the HIP kernels.
# ---------------------------------------------------------------------------------------------------------------
class SeqParallel:
    def __init__(self, group=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("sequence parallelism needs an initialised torch.distributed process group")
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.backend = dist.get_backend(group)

    # ---- token partition: equal shards of ceil(L/p) rounded up to 8 tokens (16-byte V^T column alignment); the last
    # ranks may own fewer tokens (or none for very short sequences)
    def shard_len(self, L: int) -> int:
        per = (L + self.size - 1) // self.size
        return (per + 7) // 8 * 8

    def token_range(self, L: int, rank: int = None):
        rank = self.rank if rank is None else rank
        ls = self.shard_len(L)
        r0 = min(rank * ls, L)
        return r0, min(r0 + ls, L)

    def counts(self, L: int):
        return [self.token_range(L, r)[1] - self.token_range(L, r)[0] for r in range(self.size)]

    # ---- transport
    def _all_to_all(self, recv, send):
        """recv[s] <- send[rank] of rank s. All tensors contiguous; sizes are known to every rank by construction."""
        if self.size == 1:
            recv[0].copy_(send[0])
            return
        if self.backend == "nccl":
            dist.all_to_all(recv, send, group=self.group)
            return
        # gloo (tests): all-gather of every rank's concatenated send chunks through host memory, then pick the own column
        nbytes = [t.numel() * t.element_size() for t in send]
        total = torch.tensor([sum(nbytes)], dtype=torch.int64)
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(self.size)]
        dist.all_gather(sizes, total, group=self.group)
        cap = int(max(int(s) for s in sizes))
        flat = torch.zeros(cap, dtype=torch.uint8)
        off = 0
        for t, n in zip(send, nbytes):
            flat[off:off + n] = t.detach().reshape(-1).view(torch.uint8).cpu() if n else flat[off:off]
            off += n
        gathered = [torch.empty(cap, dtype=torch.uint8) for _ in range(self.size)]
        dist.all_gather(gathered, flat, group=self.group)
        # chunk (src -> me) starts after src's chunks for ranks < me; their byte sizes follow from the receive shapes:
        # what src sends to rank d has the same per-source geometry as what I receive, scaled by d's share - so every rank
        # also publishes its chunk offsets
        offs = torch.tensor([sum(nbytes[:d]) for d in range(self.size)], dtype=torch.int64)
        all_offs = [torch.empty(self.size, dtype=torch.int64) for _ in range(self.size)]
        dist.all_gather(all_offs, offs, group=self.group)
        for s in range(self.size):
            n = recv[s].numel() * recv[s].element_size()
            o = int(all_offs[s][self.rank])
            if n:
                recv[s].copy_(gathered[s][o:o + n].view(recv[s].dtype).view(recv[s].shape).to(recv[s].device))

    # ---- the three exchanges of distributed attention
    def heads_to_tokens(self, x_loc: torch.Tensor, L: int, out: torch.Tensor):
        """x_loc [n_me, C] (my tokens, all heads) -> out [L, C/p] (all tokens, my heads). q and k."""
        p, cp = self.size, x_loc.shape[1] // self.size
        n_me = x_loc.shape[0]
        packed = x_loc.view(n_me, p, cp).transpose(0, 1).contiguous()            # [p, n_me, cp]
        send = [packed[d] for d in range(p)]
        recv = [out[self.token_range(L, s)[0]:self.token_range(L, s)[1]] for s in range(p)]
        self._all_to_all(recv, send)
        return out

    def heads_to_tokens_T(self, xt_loc: torch.Tensor, L: int, out_t: torch.Tensor, col0: int = 0):
        """xt_loc [C, n_me] (V^T of my tokens, any column stride) -> out_t[:, col0 : col0 + L] = V^T of all tokens for my
        heads ([C/p, ...])."""
        p, cp = self.size, xt_loc.shape[0] // self.size
        n_me = xt_loc.shape[1]
        send = [xt_loc[d * cp:(d + 1) * cp].contiguous() for d in range(p)]      # [cp, n_me] each
        cnt = self.counts(L)
        recv = [torch.empty(cp, cnt[s], dtype=xt_loc.dtype, device=xt_loc.device) for s in range(p)]
        self._all_to_all(recv, send)
        for s in range(p):
            r0, r1 = self.token_range(L, s)
            out_t[:, col0 + r0:col0 + r1] = recv[s]
        return out_t

    def tokens_to_heads(self, y_full: torch.Tensor, L: int, out_loc: torch.Tensor):
        """y_full [L, C/p] (all tokens, my heads) -> out_loc [n_me, C] (my tokens, all heads). Attention output."""
        p, cp = self.size, y_full.shape[1]
        n_me = out_loc.shape[0]
        send = [y_full[self.token_range(L, d)[0]:self.token_range(L, d)[1]] for d in range(p)]
        recv_buf = torch.empty(p, n_me, cp, dtype=y_full.dtype, device=y_full.device)
        self._all_to_all([recv_buf[s] for s in range(p)], send)
        out_loc.view(n_me, p, cp).copy_(recv_buf.transpose(0, 1))
        return out_loc

    def gather_rows(self, y_loc: torch.Tensor, L: int) -> torch.Tensor:
        """y_loc [n_me, K] -> [L, K] on every rank (head output before unpatchify; gather_forward, util.py:43-51)."""
        if self.size == 1:
            return y_loc
        ls = self.shard_len(L)
        buf = torch.zeros(ls, y_loc.shape[1], dtype=y_loc.dtype, device=y_loc.device)
        buf[:y_loc.shape[0]] = y_loc
        if self.backend == "nccl":
            parts = [torch.empty_like(buf) for _ in range(self.size)]
            dist.all_gather(parts, buf, group=self.group)
        else:
            host = [torch.empty(buf.shape, dtype=buf.dtype) for _ in range(self.size)]
            dist.all_gather(host, buf.cpu(), group=self.group)
            parts = [h.to(y_loc.device) for h in host]
        cnt = self.counts(L)
        return torch.cat([parts[s][:cnt[s]] for s in range(self.size)], 0)


# ---------------------------------------------------------------------------------------------------------------
# CFG parallelism (SURVEY 8(e) "intra-sample sharding"): the conditional and the unconditional DiT forward of ONE sample
# run on the two ranks of a pair; the only traffic is one all-gather of the velocity prediction per step (8.8 MB per rank at
# 49x704x1280 over one xGMI link, ~60 us against a 140 ms forward). The reference has no such mode (it runs the two
# forwards back to back, textimage2video.py:380-385); the arithmetic of each forward is unchanged, so both ranks step the
# sampler on bit-identical (cond, uncond) pairs and hold the same latent after every step.
# ---------------------------------------------------------------------------------------------------------------
class CfgParallel:
    def __init__(self, group=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("CFG parallelism needs an initialised torch.distributed process group")
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        if self.size != 2:
            raise ValueError(f"a CFG pair is exactly 2 ranks (cond, uncond); the group has {self.size}")
        self.backend = dist.get_backend(group)

    @property
    def branch(self) -> str:
        return "cond" if self.rank == 0 else "uncond"

    def exchange(self, pred: torch.Tensor):
        """pred = this rank's forward output -> (cond, uncond) on both ranks."""
        pred = pred.contiguous()
        if self.backend == "nccl":
            parts = [torch.empty_like(pred) for _ in range(2)]
            dist.all_gather(parts, pred, group=self.group)
        else:                                                      # gloo (tests): through host memory
            host = [torch.empty(pred.shape, dtype=pred.dtype) for _ in range(2)]
            dist.all_gather(host, pred.cpu(), group=self.group)
            parts = [h.to(pred.device) for h in host]
        parts[self.rank] = pred
        return parts[0], parts[1]
