"""GPU (-m gpu): UniVid's own entry path - CrossAttentionFusionPipeline -> Wan22ContextWrapper -> WanTI2V - with the dynamic text weight
(models/model_pipeline.py:1699-1810, 1844-1886) as a NATIVE schedule on the fast path (stacked CFG pair, cached context work, fused
residual epilogue, HIP-graph replay), against
  * the reference's own mechanism executed literally by this package (Wan22ContextWrapper(native=False): a closure re-assigned as `forward`
    on every WanCrossAttention + a counting closure as the DiT's forward, run on the model's generic un-fused path): BIT-IDENTICAL, and
  * the pinned CPU oracle's hooked loop (oracle/sampler.py: text_weight_cfg), statistically, as the closure path is tested.
"""
import logging
import math

import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

DEV = "cuda"
BF16 = torch.bfloat16
LOG = logging.getLogger("t")


@pytest.fixture(scope="module", autouse=True)
def _init():
    from univid_amd import _lib
    _lib.init()
    yield


def _tiny(seed=0, **over):
    from oracle import wan_dit
    from univid_amd.wan.model import WanModel
    cfg = dict(wan_dit.TINY_CFG, **over)
    sd = wan_dit.make_state_dict(cfg, seed)
    m = WanModel.from_config(dict(cfg, model_type="ti2v"))
    m.load_state_dict(sd)
    return cfg, sd, m.to(DEV).eval()


def _pipe(m):
    from univid_amd.wan.textimage2video import TI2VConfig, WanTI2V
    return WanTI2V(TI2VConfig, model=m, device=DEV)


def _wrapper(pipe, native, **cfg_over):
    from univid_amd.model_pipeline import CrossAttentionConfig, Wan22ContextWrapper
    ccfg = CrossAttentionConfig(use_dynamic_text_weight=True, **cfg_over)
    wr = Wan22ContextWrapper(pipe, None, LOG, ccfg, native=native)
    wr.set_bagel_context(torch.zeros(1, 4, 8))
    return wr


@pytest.mark.parametrize("w", [1.3, 1.2999999523162842, 0.71, 1.0, 3.0e-3, -2.5])
def test_text_weight_rows_kernel_is_the_hooks_bf16_arithmetic(w):
    """uv_text_weight_rows_bf16 against the hook's own two torch statements on the same bf16 rows (model_pipeline.py:1788-1797):
    weight_mask = ones_like(ctx); weight_mask[:n] *= w; ctx * weight_mask - every bit, including the rows it only copies."""
    from oracle import sampler
    from univid_amd import _lib
    g = torch.Generator().manual_seed(5)
    ctx = (torch.randn(40, 264, generator=g) * 3).to(BF16)
    ctx[3, :8] = torch.tensor([0.0, -0.0, 1e-38, -1e-38, 3e38, -3e38, 1.0, 2.0 ** -126], dtype=torch.float32).to(BF16)
    n = 17
    ref = (ctx.unsqueeze(0) * sampler.context_mask((1, 40, 264), w, bagel_sequence_length=n))[0]
    out = torch.full((40, 264), 7.0, dtype=BF16, device=DEV)
    _lib.text_weight_rows(ctx.to(DEV), out, n, w)
    assert torch.equal(out.cpu().view(torch.int16), ref.view(torch.int16))
    x = ctx.to(DEV)
    _lib.text_weight_rows(x, x, n, w)                                  # in place
    assert torch.equal(x.cpu().view(torch.int16), ref.view(torch.int16))


def test_text_weight_rows_rejects_bad_arguments():
    from univid_amd import _lib
    x = torch.zeros(8, 64, dtype=BF16, device=DEV)
    with pytest.raises(_lib.UnividHipError):
        _lib.text_weight_rows(x, torch.zeros(8, 64, dtype=BF16, device=DEV), 9, 1.1)          # more scaled rows than rows
    with pytest.raises(_lib.UnividHipError):
        _lib.text_weight_rows(x, torch.zeros(8, 64, dtype=BF16, device=DEV), 4, float("inf"))
    with pytest.raises(_lib.UnividHipError):
        _lib.text_weight_rows(x[:, :60], torch.zeros(8, 60, dtype=BF16, device=DEV), 4, 1.1)   # C % 8


def _gen(wr, g, steps, **kw):
    with torch.no_grad():
        return wr.generate(input_prompt="", size=(256, 256), frame_num=13, shift=5.0, sampling_steps=steps, guide_scale=5.0,
                           prompt_embeds=[g["ctx"].to(DEV)], negative_prompt_embeds=[g["ctx_null"].to(DEV)], noise=g["noise"].to(DEV),
                           decode=False, **kw).clone()


@pytest.mark.parametrize("schedule", ["linear", "cosine", "exponential"])
def test_native_schedule_is_bit_identical_to_the_hook_closures_t2v(schedule):
    """t2v, 5 steps, the weight leaving 1.3 over the first 6 forwards (3 steps) and 1.0 afterwards: the native schedule - under the HIP
    graph (the default) and eager - gives the closures' latent bit for bit; it differs from the plain loop; nothing is left re-assigned."""
    g = load_golden("sampler_tiny")
    cfg, sd, m = _tiny(g["seed"])
    pipe = _pipe(m)
    over = dict(total_sampling_steps=10, text_weight_transition_ratio=0.6, text_weight_schedule=schedule)
    wr_c = _wrapper(pipe, native=False, **over)
    assert all("forward" in b.cross_attn.__dict__ for b in m.blocks)
    closures = _gen(wr_c, g, 5)
    wr_c.restore_original_methods()
    assert all("forward" not in b.cross_attn.__dict__ for b in m.blocks) and "forward" not in m.__dict__

    wr_n = _wrapper(pipe, native=True, **over)
    assert len(wr_n.original_forward_methods) == cfg["num_layers"]
    assert all("forward" not in b.cross_attn.__dict__ for b in m.blocks), "the native wrapper must not re-assign anything"
    native_graph = _gen(wr_n, g, 5)
    assert pipe._runner is not None, "the pipeline path must run on the HIP-graph replay"
    assert pipe.text_weight_schedule is None and not hasattr(wr_n, "sampling_step_counter") and m._text_weight is None
    assert wr_n.current_timestep == 9 and wr_n.text_weight_multiplier == 1.0       # the counter saw 2 forwards per step
    assert torch.equal(native_graph, closures), float((native_graph - closures).abs().max())
    # eager native (graph off): same bits
    with torch.no_grad(), wr_n.scheduled():
        native_eager = pipe.denoise(g["noise"].to(DEV), [g["ctx"].to(DEV)], [g["ctx_null"].to(DEV)], 5, 5.0, 5.0, graph=False)
    assert torch.equal(native_eager, closures)
    with torch.no_grad():
        plain = pipe.denoise(g["noise"].to(DEV), [g["ctx"].to(DEV)], [g["ctx_null"].to(DEV)], 5, 5.0, 5.0)
    assert not torch.equal(plain, closures), "the text weight must change the result"
    # a second generation restarts the counter and replays the same graph
    r = pipe._runner
    again = _gen(wr_n, g, 5)
    assert pipe._runner is r and torch.equal(again, closures)


def test_native_schedule_i2v_and_injection_layer_subset_bit_identical():
    """i2v (first latent frame pinned, per-token timesteps {0, t}) and `injection_layers` = a subset of the blocks."""
    g = load_golden("sampler_tiny")
    cfg, sd, m = _tiny(g["seed"], num_layers=3)
    pipe = _pipe(m)
    noise, z = g["noise"].to(DEV), g["z"].to(DEV)
    ctx, ctxn = [g["ctx"].to(DEV)], [g["ctx_null"].to(DEV)]
    over = dict(total_sampling_steps=8, text_weight_transition_ratio=0.5, text_weight_schedule="cosine", bagel_sequence_length=9)
    res = {}
    for layers in (None, [0, 2], [1]):
        for native in (False, True):
            wr = _wrapper(pipe, native=native, **over)
            wr.injection_layers = layers
            with torch.no_grad(), wr.scheduled():
                res[(str(layers), native, "i2v")] = pipe.denoise(noise, ctx, ctxn, 4, 5.0, 5.0, z=z).clone()
            with torch.no_grad(), wr.scheduled():
                res[(str(layers), native, "t2v")] = pipe.denoise(noise, ctx, ctxn, 4, 5.0, 5.0).clone()
            wr.restore_original_methods()
        for mode in ("i2v", "t2v"):
            assert torch.equal(res[(str(layers), True, mode)], res[(str(layers), False, mode)]), (layers, mode)
        assert torch.equal(res[(str(layers), True, "i2v")][:, 0], z[:, 0])
    assert not torch.equal(res[("None", True, "t2v")], res[("[0, 2]", True, "t2v")])
    assert not torch.equal(res[("[1]", True, "t2v")], res[("[0, 2]", True, "t2v")])


def test_native_schedule_weight_one_is_the_plain_loop_and_inactive_conditions():
    """w == 1 everywhere (max = min = 1), no BAGEL context set, or use_dynamic_text_weight off: the hook's own conditions
    (model_pipeline.py:1767-1773) leave the context alone - the result is the plain loop's, bit for bit, and no K / V^T is recomputed."""
    from univid_amd.model_pipeline import CrossAttentionConfig, Wan22ContextWrapper
    g = load_golden("sampler_tiny")
    cfg, sd, m = _tiny(g["seed"])
    pipe = _pipe(m)
    noise, ctx, ctxn = g["noise"].to(DEV), [g["ctx"].to(DEV)], [g["ctx_null"].to(DEV)]
    with torch.no_grad():
        plain = pipe.denoise(noise, ctx, ctxn, 3, 5.0, 5.0).clone()
    wr = _wrapper(pipe, native=True, total_sampling_steps=10, text_weight_max=1.0, text_weight_min=1.0)
    assert torch.equal(_gen(wr, g, 3), plain)
    wr = _wrapper(pipe, native=True, total_sampling_steps=10)
    wr.clear_bagel_context()                                       # hook condition: use_bagel_context
    assert torch.equal(_gen(wr, g, 3), plain)
    wr = Wan22ContextWrapper(pipe, None, LOG, CrossAttentionConfig(use_dynamic_text_weight=False), native=True)
    wr.set_bagel_context(torch.zeros(1, 4, 8))
    assert torch.equal(_gen(wr, g, 3), plain)
    # after the transition the runner's buffers are back to the plain K / V^T: further steps recompute nothing
    wr = _wrapper(pipe, native=True, total_sampling_steps=5, text_weight_transition_ratio=0.4)      # 2 forwards = 1 step weighted
    got = _gen(wr, g, 3)
    r = pipe._runner
    assert all(s == (1.0, 1.0) for s in r.state) and r.apply(None) == 0
    assert not torch.equal(got, plain)


def test_native_schedule_follows_the_hooked_oracle():
    """The native path against the pinned CPU oracle's hooked loop (as test_text_weight_hook_path_matches_oracle does for the wrapper's
    default, which now IS the native path; here with a schedule that stays on for all steps)."""
    from oracle import sampler
    g = load_golden("sampler_tiny")
    cfg, sd, m = _tiny(g["seed"])
    pipe = _pipe(m)
    twc = dict(total_steps=20, ratio=0.5, schedule="linear")
    with torch.no_grad():
        ref = sampler.denoise(sd, cfg, g["noise"], [g["ctx"]], [g["ctx_null"]], 3, 5.0, 5.0, text_weight_cfg=twc)
        plain = sampler.denoise(sd, cfg, g["noise"], [g["ctx"]], [g["ctx_null"]], 3, 5.0, 5.0)
    wr = _wrapper(pipe, native=True, total_sampling_steps=20, text_weight_transition_ratio=0.5, text_weight_schedule="linear")
    got = _gen(wr, g, 3).cpu()
    rel = lambda a, b: float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())
    assert rel(got, ref) < 2e-2 and rel(got, ref) < 0.5 * rel(plain, ref), (rel(got, ref), rel(plain, ref))


def test_graph_runner_is_not_fooled_by_a_recycled_prompt_address():
    """Round-4 advisor finding: t2v re-encodes the prompt for every call and frees it afterwards, so the caching allocator readily hands a
    LATER prompt of equal token length the same address with version 0. The runner refreshes its context buffers at the start of
    every denoise unconditionally: prompt B freed, prompt C allocated in its place -> C's video, not B's."""
    g = load_golden("sampler_tiny")
    cfg, sd, m = _tiny(g["seed"])
    pipe = _pipe(m)
    noise, ctxn = g["noise"].to(DEV), [g["ctx_null"].to(DEV)]
    gen = torch.Generator().manual_seed(1)
    hB, hC = torch.randn(20, cfg["text_dim"], generator=gen), torch.randn(20, cfg["text_dim"], generator=gen)
    with torch.no_grad():
        b = [hB.to(DEV)]
        pipe.denoise(noise, b, ctxn, 2, 5.0, 5.0, graph=True)
        r = pipe._runner
        addr = b[0].data_ptr()
        del b
        c = [hC.to(DEV)]
        same_address = c[0].data_ptr() == addr         # (what the allocator usually does; the test holds either way)
        got = pipe.denoise(noise, c, ctxn, 2, 5.0, 5.0, graph=True).clone()
        assert pipe._runner is r
        want = pipe.denoise(noise, c, ctxn, 2, 5.0, 5.0, graph=False)
        other = pipe.denoise(noise, [hB.to(DEV)], ctxn, 2, 5.0, 5.0, graph=False)
    assert torch.equal(got, want) and not torch.equal(got, other), f"stale context (address recycled: {same_address})"


def test_two_captures_do_not_share_graph_pool_scratch():
    """Round-4 advisor finding: scratch created during a capture lives in that graph's pool and must not be found by the next capture
    (torch's capture stream, part of the scratch key, is shared). t2v -> i2v -> t2v recaptures, dropping the old runner each time;
    every result equals the eager loop's."""
    import gc
    from univid_amd.wan import model as M
    g = load_golden("sampler_tiny")
    cfg, sd, m = _tiny(g["seed"])
    pipe = _pipe(m)
    noise, z, ctx, ctxn = g["noise"].to(DEV), g["z"].to(DEV), [g["ctx"].to(DEV)], [g["ctx_null"].to(DEV)]
    with torch.no_grad():
        e_t = pipe.denoise(noise, ctx, ctxn, 2, 5.0, 5.0, graph=False).clone()
        e_i = pipe.denoise(noise, ctx, ctxn, 2, 5.0, 5.0, z=z, graph=False).clone()
        keys = set(M._zero_cache)
        for _ in range(2):
            assert torch.equal(pipe.denoise(noise, ctx, ctxn, 2, 5.0, 5.0, graph=True), e_t)
            assert set(M._zero_cache) == keys, "a capture left its scratch in the global cache"
            assert pipe._runner.scratch is not None
            assert torch.equal(pipe.denoise(noise, ctx, ctxn, 2, 5.0, 5.0, z=z, graph=True), e_i)
            gc.collect()
            torch.cuda.empty_cache()
            junk = torch.full((1 << 22,), float("nan"), device=DEV)      # anything a released pool gave back gets overwritten
            del junk
        assert torch.equal(pipe.denoise(noise, ctx, ctxn, 2, 5.0, 5.0, graph=True), e_t)


def test_fusion_pipeline_runs_the_fast_path_and_matches_the_closure_pipeline():
    """CrossAttentionFusionPipeline.generate_video_with_bagel_context (model_pipeline.py:2577-2655; what inference.py:311,377 calls) with
    the default native text weight: graph replay, launches per step as the plain loop's; result = the closure pipeline's, bit for bit."""
    import types
    from univid_amd import _lib
    from univid_amd.model_pipeline import CrossAttentionConfig, CrossAttentionFusionPipeline
    g = load_golden("sampler_tiny")
    cfg, sd, m = _tiny(g["seed"])
    pipe = _pipe(m)
    bagel = types.SimpleNamespace(extract_semantic_tokens=lambda text, image: torch.zeros(1, 4, 8))
    ccfg = CrossAttentionConfig(total_sampling_steps=6, text_weight_transition_ratio=0.5, use_dynamic_text_weight=True, use_lora=False)
    kw = dict(steps=6, guidance_scale=5.0, frames=13, size=(256, 256), shift=5.0, decode=False, prompt_embeds=[g["ctx"].to(DEV)],
              negative_prompt_embeds=[g["ctx_null"].to(DEV)], noise=g["noise"].to(DEV))
    res, calls = {}, {}
    for native in (True, False):
        fusion = CrossAttentionFusionPipeline(ccfg, wan_pipeline=pipe, bagel_extractor=bagel, native_text_weight=native)
        with torch.no_grad():
            fusion.generate_video_with_bagel_context("x", **kw)                 # warm-up: capture
            c0 = _lib.CALL_COUNT
            res[native], _ = fusion.generate_video_with_bagel_context("x", **kw)
            calls[native] = _lib.CALL_COUNT - c0
        res[native] = res[native].clone()
        assert fusion.get_fusion_info()["hooked_layers"] == cfg["num_layers"]
        fusion.cleanup_resources()
    assert torch.equal(res[True], res[False])
    # 6 steps: the closures launch every kernel of 12 batch-1 forwards; the native path replays a graph (a handful of entry-point
    # calls per step: the sampler update) + the K / V^T refreshes of the 2 weighted steps + 1 back to plain
    assert calls[True] * 4 < calls[False], calls
