"""CPU, world_size 2, gloo: the N > 1 path of the denoise hot path (sample sharding + the one all-gather)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from univid_amd import parallel


class _FakePipe:
    """Stands in for WanTI2V.denoise: a deterministic function of (noise, contexts) so results can be checked."""

    def denoise(self, noise, ctx, ctx_null, steps, shift, gs):
        return noise * 2.0 + ctx[0].sum() - ctx_null[0].sum() + steps


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_samples, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        noises = [torch.randn(4, 2, 3, 3, generator=g) for _ in range(n_samples)]
        ctx = [[torch.full((2, 2), float(i))] for i in range(n_samples)]
        ctxn = [[torch.zeros(1, 2)] for _ in range(n_samples)]
        out = parallel.denoise_batch(_FakePipe(), noises, ctx, ctxn, 3, 5.0, 5.0)
        expect = [n * 2.0 + 4.0 * i + 3 for i, n in enumerate(noises)]
        ok = len(out) == n_samples and all(torch.equal(a, b) for a, b in zip(out, expect))
        lo, hi = parallel.shard_range(n_samples, rank, world)
        q.put((rank, ok, lo, hi))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_samples", [2, 5])
def test_two_rank_sharding_and_gather(n_samples):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_samples, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res)
    assert res[0][2] == 0 and res[0][3] == res[1][2] and res[1][3] == n_samples


@pytest.mark.parametrize("n_samples", [8, 11, 5])
def test_eight_rank_sharding_uneven_shards_and_gather(n_samples):
    """The driver's 8-GPU launch shape on CPU (world 8, gloo): 8 samples (one per rank, the bench's layout), 11 (uneven: three ranks
    own two samples) and 5 (three ranks own NOTHING and still take part in the one all-gather). Every rank must end with all
    latents in global sample order, and the shard ranges must tile [0, n) without gap or overlap."""
    world = 8
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_samples, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r for r, *_ in res] == list(range(world)) and all(ok for _, ok, _, _ in res)
    assert res[0][2] == 0 and res[-1][3] == n_samples
    assert all(res[r][3] == res[r + 1][2] for r in range(world - 1)), "shard ranges must be contiguous"
    sizes = [hi - lo for _, _, lo, hi in res]
    assert max(sizes) - min(sizes) <= 1 and sum(sizes) == n_samples


class WanCrossAttention(torch.nn.Module):          # found by CLASS NAME (model_pipeline.py:1745)
    def forward(self, x, context, context_lens):
        return x


class _SchedPipe:
    """Stands in for WanTI2V on the pipeline path: `denoise` consumes the native text-weight schedule exactly as WanTI2V._steps does
    (one (cond, uncond) pair of weights per step) and folds every weight into the result, so a wrong / shared / not-restarted forward
    counter changes the latent."""

    def __init__(self):
        self.model = torch.nn.Sequential(WanCrossAttention(), WanCrossAttention())
        self.text_weight_schedule = None

    def denoise(self, noise, ctx, ctx_null, steps, shift, gs):
        out = noise * 2.0 + ctx[0].sum()
        for k in range(steps):
            wc, wu = self.text_weight_schedule.next_pair() if self.text_weight_schedule is not None else (1.0, 1.0)
            out = out + (k + 1) * (wc * 10.0 + wu)
        return out


def _pipeline_path_worker(rank, world, port, n_samples, q):
    import logging
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from univid_amd.model_pipeline import CrossAttentionConfig, Wan22ContextWrapper
        pipe = _SchedPipe()
        cfg = CrossAttentionConfig(use_dynamic_text_weight=True, total_sampling_steps=10, text_weight_transition_ratio=0.5, text_weight_schedule="linear")
        wr = Wan22ContextWrapper(pipe, None, logging.getLogger("t"), cfg)
        wr.set_bagel_context(torch.zeros(1, 2, 2))
        g = torch.Generator().manual_seed(0)
        noises = [torch.randn(4, 2, 3, 3, generator=g) for _ in range(n_samples)]
        ctx = [[torch.full((2, 2), float(i))] for i in range(n_samples)]
        ctxn = [[torch.zeros(1, 2)] for _ in range(n_samples)]
        out = parallel.denoise_batch(pipe, noises, ctx, ctxn, 4, 5.0, 5.0, wrapper=wr)
        # single-process expectation: every sample starts its own counter at 0 (w = 1.3 - 0.3 * c / 5 for c < 5, else 1.0)
        w = [wr._calculate_text_weight(c) for c in range(8)]
        bump = sum((k + 1) * (w[2 * k] * 10.0 + w[2 * k + 1]) for k in range(4))
        expect = [n * 2.0 + 4.0 * i + bump for i, n in enumerate(noises)]
        ok = len(out) == n_samples and all(torch.allclose(a, b, rtol=0, atol=1e-4) for a, b in zip(out, expect))
        ok &= w[0] == 1.3 and w[5] == 1.0 and pipe.text_weight_schedule is None and not hasattr(wr, "sampling_step_counter")
        ok &= all("forward" not in m.__dict__ for m in pipe.model)            # native: nothing re-assigned
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_two_rank_denoise_batch_on_the_pipeline_path():
    """parallel.denoise_batch(wrapper=): the replicas run UniVid's pipeline path - each sample one generation under its rank's own
    forward counter, restarted per sample (3 samples on 2 ranks: rank 0 runs two in a row)."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipeline_path_worker, args=(r, 2, port, 3, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def test_seed_depends_on_global_index_only():
    assert parallel.sample_seed(42, 5) == 47
    assert [parallel.sample_seed(42, i) for i in range(*parallel.shard_range(8, 3, 4))] == [48, 49]


# ---------------------------------------------------------------------------------------------------------------
# Ulysses sequence parallelism: the three exchanges of distributed attention against plain slicing (world size 2 and 3,
# gloo; the gloo transport is the all-gather emulation, the layout logic is the one the RCCL path uses)
# ---------------------------------------------------------------------------------------------------------------
def _sp_worker(rank, world, port, L, C, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from univid_amd.parallel import SeqParallel
        par = SeqParallel()
        g = torch.Generator().manual_seed(5)
        full = torch.randn(L, C, generator=g).to(torch.bfloat16)           # the same "all tokens, all heads" tensor on every rank
        r0, r1 = par.token_range(L)
        cp = C // world
        assert sum(par.counts(L)) == L and par.shard_len(L) % 8 == 0
        # q / k: my tokens x all heads -> all tokens x my heads
        out = torch.empty(L, cp, dtype=torch.bfloat16)
        par.heads_to_tokens(full[r0:r1].contiguous(), L, out)
        ok = torch.equal(out, full[:, rank * cp:(rank + 1) * cp])
        # V^T: [C, my tokens] -> [C/p, all tokens] at a column offset
        vt = torch.zeros(cp, L + 72, dtype=torch.bfloat16)
        par.heads_to_tokens_T(full[r0:r1].t(), L, vt, col0=8)
        ok &= torch.equal(vt[:, 8:8 + L], full[:, rank * cp:(rank + 1) * cp].t()) and bool((vt[:, :8] == 0).all())
        # attention output: all tokens x my heads -> my tokens x all heads
        back = torch.empty(r1 - r0, C, dtype=torch.bfloat16)
        par.tokens_to_heads(full[:, rank * cp:(rank + 1) * cp].contiguous(), L, back)
        ok &= torch.equal(back, full[r0:r1])
        # head rows
        rows = par.gather_rows(full[r0:r1].float().contiguous(), L)
        ok &= torch.equal(rows, full.float())
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,L", [(2, 64), (2, 50), (3, 40), (3, 10)])
def test_sequence_parallel_exchanges(world, L):
    """(3, 10): the last rank owns no tokens at all and must still take part in every exchange."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sp_worker, args=(r, world, port, L, 12 * world, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(ok for _, ok in res), res


def _cfgp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if world != 2:
            try:
                parallel.CfgParallel()
                q.put((rank, False))
            except ValueError:
                q.put((rank, True))
            return
        cp = parallel.CfgParallel()
        mine = torch.full((3, 2, 4), float(rank + 1))
        cond, uncond = cp.exchange(mine)
        ok = cp.branch == ("cond", "uncond")[rank] and torch.equal(cond, torch.full((3, 2, 4), 1.0)) and torch.equal(uncond, torch.full((3, 2, 4), 2.0))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_cfg_parallel_exchange(world):
    """A CFG pair is exactly two ranks: rank 0 holds the conditional prediction, rank 1 the unconditional one, both get both."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cfgp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(ok for _, ok in res), res
