"""TEST INFRASTRUCTURE ONLY. Generates tests/golden/*.npz by IMPORTING THE REFERENCE (build container only).

    python -m oracle.gen_golden            # needs /root/reference; rewrites tests/golden/

The reference ships no tests or golden vectors for this path, so the pins are made here:
every fixture holds INPUTS and the REFERENCE MODULES' OUTPUTS (run on CPU through the shims of
oracle/_refimport.py), and while generating, the CPU restatement in oracle/ is asserted bit-identical to the
reference on the same inputs. Weights are never stored: both sides derive them from univid_amd.detinit
(seed in the fixture). bf16 tensors are stored as their uint16 bit patterns.
"""
import ast
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _refimport, dpmpp, projector, sampler, siglip2, t5, unipc, wan_dit, wan_vae  # noqa: E402
from univid_amd import detinit  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def _np(t):
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().view(np.uint16)
    return t.detach().numpy()


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (_np(v) if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print(f"  wrote {path} ({os.path.getsize(path) >> 10} KiB)")


def ref_dit(ns, cfg, seed):
    m = ns.model.WanModel(model_type="ti2v", patch_size=cfg["patch_size"], text_len=cfg["text_len"], in_dim=cfg["in_dim"],
                          dim=cfg["dim"], ffn_dim=cfg["ffn_dim"], freq_dim=cfg["freq_dim"], text_dim=cfg["text_dim"],
                          out_dim=cfg["out_dim"], num_heads=cfg["num_heads"], num_layers=cfg["num_layers"]).eval()
    detinit.init_module_(m, seed=seed)
    return m


def gen_dit_tiny(ns):
    print("dit_tiny")
    cfg = wan_dit.TINY_CFG
    m = ref_dit(ns, cfg, 0)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(42)
    x = torch.randn(48, 4, 16, 16, generator=g)
    ctx = torch.randn(20, cfg["text_dim"], generator=g)
    L = 4 * 8 * 8
    t_one = torch.full((1, L), 937.0)
    t_two = t_one.clone()
    t_two[0, :64] = 0.0                                # i2v: first latent frame at timestep 0 (textimage2video.py:573)
    outs = {}
    for name, t in (("one", t_one), ("two", t_two)):
        with torch.no_grad(), torch.amp.autocast("cuda", dtype=torch.bfloat16):
            ref = m([x], t=t, context=[ctx], seq_len=L)[0]
        with torch.no_grad():
            mine = wan_dit.dit_forward(sd, cfg, [x], t, [ctx], L)[0]
        assert torch.equal(ref, mine), "oracle restatement != reference (dit_tiny)"
        outs[name] = ref
    # padded sequence: seq_len > L (padding tokens must not change the valid outputs)
    with torch.no_grad(), torch.amp.autocast("cuda", dtype=torch.bfloat16):
        ref_pad = m([x], t=torch.full((1, L + 32), 937.0), context=[ctx], seq_len=L + 32)[0]
    save("dit_tiny", seed=0, x=x, ctx=ctx, t_one=t_one, t_two=t_two, out_one=outs["one"], out_two=outs["two"], out_pad=ref_pad)


def gen_dit_block_3072(ns):
    """One TI2V-5B-width block (dim 3072, ffn 14336, 24 heads) at L = 2*4*6 = 48 tokens, text_len 64."""
    print("dit_block_3072")
    dim, ffn, heads, L, Lc = 3072, 14336, 24, 48, 64
    blk = ns.model.WanAttentionBlock(dim, ffn, heads, (-1, -1), True, True, 1e-6).eval()
    sd_named = {"blocks.0." + k: v for k, v in blk.state_dict(keep_vars=True).items()}
    detinit.init_state_dict_(sd_named, seed=7)
    sd = {k: v.detach().clone() for k, v in sd_named.items()}
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, L, dim, generator=g)
    e_rows = torch.randn(2, 6, dim, generator=g) * 0.3
    tid = (torch.arange(L) >= 24).long()
    e0 = e_rows[tid].unsqueeze(0)
    ctx = (torch.randn(1, Lc, dim, generator=g) * 0.5).to(torch.bfloat16)
    grid = torch.tensor([[2, 4, 6]])
    freqs = wan_dit.rope_table(dim // heads)
    seq_lens = torch.tensor([L])
    outs = {}
    for name, xin in (("f32", x), ("bf16", x.to(torch.bfloat16))):     # bf16 = what block 0 sees (patch embedding output)
        with torch.no_grad(), torch.amp.autocast("cuda", dtype=torch.bfloat16):
            ref = blk(xin, e0, seq_lens, grid, freqs, ctx, None)
        with torch.no_grad():
            mine = wan_dit.block_forward(sd, "blocks.0.", xin, e0, seq_lens, grid, freqs, ctx, heads, 1e-6)
        assert ref.dtype == torch.float32 and torch.equal(ref, mine), "oracle restatement != reference (block)"
        outs[name] = ref
    save("dit_block_3072", seed=7, x=x, e_rows=e_rows, tid=tid, ctx=ctx, grid=grid, out_f32=outs["f32"], out_bf16=outs["bf16"])


def gen_unipc(ns):
    print("unipc")
    S = ns.unipc.FlowUniPCMultistepScheduler
    arrs = {}
    for steps, shift in ((50, 5.0), (10, 5.0), (40, 3.0)):
        sch = S(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        sch.set_timesteps(steps, device="cpu", shift=shift)
        mine = unipc.FlowUniPC(1000, shift=1)
        ts = mine.set_timesteps(steps, shift=shift)
        assert torch.equal(ts, sch.timesteps) and torch.equal(mine.sigmas, sch.sigmas)
        arrs[f"timesteps_{steps}_{shift}"] = sch.timesteps
        arrs[f"sigmas_{steps}_{shift}"] = sch.sigmas
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 48, 2, 6, 8, generator=g)
    outs = torch.randn(10, 1, 48, 2, 6, 8, generator=g)
    sch = S(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
    sch.set_timesteps(10, device="cpu", shift=5.0)
    mine = unipc.FlowUniPC(1000, shift=1)
    mine.set_timesteps(10, shift=5.0)
    lat, latm, traj = x, x, []
    for i, t in enumerate(sch.timesteps):
        with torch.amp.autocast("cuda", dtype=torch.bfloat16):
            lat = sch.step(outs[i], t, lat, return_dict=False)[0]
        latm = mine.step(outs[i], t, latm)
        assert torch.equal(lat, latm), f"oracle UniPC != reference at step {i}"
        traj.append(lat)
    save("unipc", x=x, model_outputs=outs, trajectory=torch.stack(traj), **arrs)


def gen_dpmpp(ns):
    """FlowDPMSolverMultistepScheduler as WanTI2V builds it for sample_solver='dpm++' (textimage2video.py:343-351)."""
    print("dpmpp")
    S = ns.dpm.FlowDPMSolverMultistepScheduler
    arrs = {}
    for steps, shift in ((50, 5.0), (10, 5.0), (20, 3.0), (2, 5.0), (1, 5.0)):
        sch = S(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        sig = ns.dpm.get_sampling_sigmas(steps, shift)
        ts, _ = ns.dpm.retrieve_timesteps(sch, device="cpu", sigmas=sig)
        mine = dpmpp.FlowDPMpp(1000, shift=1)
        assert np.array_equal(sig, dpmpp.get_sampling_sigmas(steps, shift))
        tm = mine.set_timesteps(sigmas=dpmpp.get_sampling_sigmas(steps, shift))
        assert torch.equal(tm, ts) and torch.equal(mine.sigmas, sch.sigmas)
        arrs[f"timesteps_{steps}_{shift}"] = ts
        arrs[f"sigmas_{steps}_{shift}"] = sch.sigmas
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 48, 2, 6, 8, generator=g)
    for steps in (10, 20, 2, 1):              # < 15 steps and >= 15 steps take different lower-order rules (fm_solvers.py:771-777)
        outs = torch.randn(steps, 1, 48, 2, 6, 8, generator=g)
        sch = S(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        ts, _ = ns.dpm.retrieve_timesteps(sch, device="cpu", sigmas=ns.dpm.get_sampling_sigmas(steps, 5.0))
        mine = dpmpp.FlowDPMpp(1000, shift=1)
        mine.set_timesteps(sigmas=dpmpp.get_sampling_sigmas(steps, 5.0))
        lat, latm, traj = x, x, []
        for i, t in enumerate(ts):
            with torch.amp.autocast("cuda", dtype=torch.bfloat16):
                lat = sch.step(outs[i], t, lat, return_dict=False)[0]
            latm = mine.step(outs[i], t, latm)
            assert torch.equal(lat, latm), f"oracle DPM++ != reference at step {i} of {steps}"
            traj.append(lat)
        assert torch.isfinite(lat).all()
        arrs[f"model_outputs_{steps}"] = outs
        arrs[f"trajectory_{steps}"] = torch.stack(traj)
    save("dpmpp", x=x, **arrs)


def gen_sampler(ns, steps=10, keep=(0, 1, 2, 5, 9), name="sampler_tiny", solver="unipc"):
    """t2v + i2v trajectories of the tiny DiT through the reference pieces. `sampler_tiny`: 10 steps; `sampler_tiny_50`: the
    50 flow steps BASELINE config 2 names (textimage2video.py:367-394 with sampling_steps=50), 6 of them kept."""
    print(name)
    cfg = wan_dit.TINY_CFG
    m = ref_dit(ns, cfg, 0)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(42)
    noise = torch.randn(48, 4, 16, 16, generator=g)
    ctx = torch.randn(20, cfg["text_dim"], generator=g)
    ctxn = torch.randn(7, cfg["text_dim"], generator=g)
    z = torch.randn(48, 1, 16, 16, generator=g)
    shift, gs = 5.0, 5.0
    keep = list(keep)
    arrs = {}
    for mode in ("t2v", "i2v"):
        i2v = mode == "i2v"
        rec_ref = []
        # composition of the reference pieces exactly as textimage2video.py:329-394 / 521-601 orders them
        with torch.amp.autocast("cuda", dtype=torch.bfloat16), torch.no_grad():
            if solver == "unipc":                                                  # :335-342
                sch = ns.unipc.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
                sch.set_timesteps(steps, device="cpu", shift=shift)
                timesteps = sch.timesteps
            else:                                                                  # 'dpm++' :343-351
                sch = ns.dpm.FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
                timesteps, _ = ns.dpm.retrieve_timesteps(sch, device="cpu", sigmas=ns.dpm.get_sampling_sigmas(steps, shift))
            latent = noise
            _, mask2 = _refimport.ref_masks_like([noise], zero=i2v)
            if i2v:
                latent = (1. - mask2[0]) * z + mask2[0] * latent
            seq_len = 4 * 8 * 8
            for t in timesteps:
                timestep = torch.stack([t])
                temp_ts = (mask2[0][0][:, ::2, ::2] * timestep).flatten()
                temp_ts = torch.cat([temp_ts, temp_ts.new_ones(seq_len - temp_ts.size(0)) * timestep])
                timestep = temp_ts.unsqueeze(0)
                c = m([latent], t=timestep, context=[ctx], seq_len=seq_len)[0]
                u = m([latent], t=timestep, context=[ctxn], seq_len=seq_len)[0]
                npred = u + gs * (c - u)
                latent = sch.step(npred.unsqueeze(0), t, latent.unsqueeze(0), return_dict=False)[0].squeeze(0)
                if i2v:
                    latent = (1. - mask2[0]) * z + mask2[0] * latent
                rec_ref.append((npred, latent))
        rec = []
        with torch.no_grad():
            mine = sampler.denoise(sd, cfg, noise, [ctx], [ctxn], steps, shift, gs, z=(z if i2v else None), record=rec,
                                   sample_solver=solver)
        for (a, b), (c_, d) in zip(rec_ref, rec):
            assert torch.equal(a, c_) and torch.equal(b, d), "oracle sampler != reference"
        # per-step tensors kept (ALL steps were checked bit-identical above)
        arrs[f"{mode}_noise_pred"] = torch.stack([rec_ref[i][0] for i in keep])
        arrs[f"{mode}_latents"] = torch.stack([rec_ref[i][1] for i in keep])
        arrs["kept_steps"] = torch.tensor(keep)
    save(name, seed=0, steps=steps, shift=shift, guide_scale=gs, noise=noise, ctx=ctx, ctx_null=ctxn, z=z, **arrs)
    return mine


def gen_vae(ns):
    print("vae_small")
    cfg = wan_vae.SMALL_CFG
    m = ns.vae.WanVAE_(dim=cfg["dim"], dec_dim=cfg["dec_dim"], z_dim=cfg["z_dim"], dim_mult=cfg["dim_mult"], num_res_blocks=2,
                       attn_scales=[], temperal_downsample=cfg["temperal_downsample"], dropout=0.0).eval()
    detinit.init_module_(m, seed=1)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    v = wan_vae.WanVAE(sd, cfg)
    scale = wan_vae.scale_tensors()
    g = torch.Generator().manual_seed(3)
    arrs = {}
    for i, shape in enumerate([(3, 9, 32, 48), (3, 1, 32, 32), (3, 5, 48, 32)]):
        vid = torch.tanh(torch.randn(*shape, generator=g))
        with torch.no_grad():
            ref = m.encode(vid.unsqueeze(0), scale)
            mine = v.encode(vid.unsqueeze(0), scale)
        assert torch.equal(ref, mine), "oracle VAE encode != reference"
        arrs[f"enc_in_{i}"], arrs[f"enc_out_{i}"] = vid, ref[0]
    for i, shape in enumerate([(48, 3, 2, 3), (48, 1, 2, 2), (48, 2, 3, 2)]):
        z = torch.randn(*shape, generator=g)
        with torch.no_grad():
            ref = m.decode(z.unsqueeze(0), scale).float().clamp_(-1, 1)      # + the wrapper's clamp (vae2_2.py:1045)
            mine = v.decode(z.unsqueeze(0), scale).clamp(-1, 1)
        assert torch.equal(ref, mine), "oracle VAE decode != reference"
        arrs[f"dec_in_{i}"], arrs[f"dec_out_{i}"] = z, ref[0]
    save("vae_small", seed=1, **arrs)


def gen_text_weight():
    """Known-answer table of Wan22ContextWrapper._calculate_text_weight (models/model_pipeline.py:1699-1735): only
    that method's source is compiled (model_pipeline.py cannot be imported: pip install + file writes at import)."""
    print("text_weight")
    src = open(os.path.join(_refimport.REF_ROOT, "models", "model_pipeline.py")).read()
    tree = ast.parse(src)
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Wan22ContextWrapper"][0]
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "_calculate_text_weight"][0]
    nsx = {}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "model_pipeline.py:_calculate_text_weight", "exec"), nsx)
    rows = []
    for schedule in ("cosine", "linear", "exponential", "other"):
        for total, ratio in ((50, 0.4), (10, 0.4), (30, 0.25), (1, 0.4)):
            for enabled in (True, False):
                cfg = types.SimpleNamespace(use_dynamic_text_weight=enabled, total_sampling_steps=total,
                                            text_weight_transition_ratio=ratio, text_weight_min=1.0, text_weight_max=1.3,
                                            text_weight_schedule=schedule)
                self_ = types.SimpleNamespace(config=cfg)
                for step in range(0, 2 * total + 1):
                    w = nsx["_calculate_text_weight"](self_, step)
                    mine = sampler.text_weight(step, total, ratio, 1.3, 1.0, schedule, enabled)
                    assert w == mine, (schedule, total, ratio, enabled, step, w, mine)
                    rows.append([schedule, total, ratio, int(enabled), step, w])
    with open(os.path.join(OUT, "text_weight.json"), "w") as f:
        json.dump({"columns": ["schedule", "total_steps", "ratio", "enabled", "forward_call", "weight"], "rows": rows}, f)
    print(f"  wrote text_weight.json ({len(rows)} rows)")


def gen_masks():
    print("masks_like")
    x = [torch.zeros(3, 4, 2, 2)]
    a1, a2 = _refimport.ref_masks_like(x, zero=False)
    b1, b2 = _refimport.ref_masks_like(x, zero=True)
    m1, m2 = sampler.masks_like(x, zero=True)
    assert torch.equal(b1[0], m1[0]) and torch.equal(b2[0], m2[0])
    save("masks_like", ones1=a1[0], ones2=a2[0], zero1=b1[0], zero2=b2[0])


def gen_siglip2():
    """BASELINE config 5. The towers' arithmetic is HF transformers' Siglip2Model (the reference only calls it,
    eval_understanding.py:171-206): the oracle is pinned against transformers itself, with the same deterministic weights
    loaded into both; mmr_select against the reference's own function."""
    print("siglip2 (transformers %s)" % __import__("transformers").__version__)
    from transformers import Siglip2Config, Siglip2Model
    cfg = siglip2.TINY_CFG
    seed = 3
    hc = Siglip2Config(vision_config=dict(cfg["vision"], hidden_act="gelu_pytorch_tanh", attention_dropout=0.0),
                       text_config=dict(cfg["text"], hidden_act="gelu_pytorch_tanh", attention_dropout=0.0))
    m = Siglip2Model(hc).eval()
    sd = siglip2.make_state_dict(cfg, seed)
    inc = m.load_state_dict(sd, strict=False)
    assert not inc.unexpected_keys and set(inc.missing_keys) <= {"logit_scale", "logit_bias"}
    g = torch.Generator().manual_seed(11)
    B, N, pdim = 6, cfg["vision"]["num_patches"], 3 * cfg["vision"]["patch_size"] ** 2
    pv = torch.randn(B, N, pdim, generator=g)
    shapes = torch.tensor([[8, 8], [8, 8], [8, 8], [6, 10], [4, 7], [5, 5]])
    mask = torch.zeros(B, N, dtype=torch.int64)
    for b, (h, w) in enumerate(shapes.tolist()):
        mask[b, :h * w] = 1
    ids = torch.randint(0, cfg["text"]["vocab_size"], (2, cfg["text"]["max_position_embeddings"]), generator=g)
    am = torch.ones_like(ids)
    am[1, 9:] = 0
    with torch.no_grad():
        fi = m.get_image_features(pixel_values=pv, pixel_attention_mask=mask, spatial_shapes=shapes)
        fi = fi.pooler_output if hasattr(fi, "pooler_output") else fi
        ft = m.get_text_features(input_ids=ids)
        ft = ft.pooler_output if hasattr(ft, "pooler_output") else ft
        ftm = m.get_text_features(input_ids=ids, attention_mask=am)
        ftm = ftm.pooler_output if hasattr(ftm, "pooler_output") else ftm
    oi, ot, otm = siglip2.image_features(sd, cfg, pv, mask, shapes), siglip2.text_features(sd, cfg, ids), siglip2.text_features(sd, cfg, ids, am)
    for a, b_, n in ((fi, oi, "image"), (ft, ot, "text"), (ftm, otm, "text+mask")):
        err = (a - b_).abs().max().item()
        print(f"   oracle vs transformers {n}: max abs diff {err:.2e}")
        assert err < 5e-6, n
    v = torch.nn.functional.normalize(fi, dim=-1)
    t = torch.nn.functional.normalize(ft[:1], dim=-1)
    idx, vals = siglip2.rank_frames(fi, ft[:1], 4)
    embs = torch.nn.functional.normalize(torch.randn(12, 16, generator=g), dim=-1)
    qe = torch.nn.functional.normalize(torch.randn(1, 16, generator=g), dim=-1)
    mm = [_refimport.ref_mmr_select(embs, qe, K, lam) for K, lam in ((5, 0.5), (12, 0.2), (20, 0.9))]
    assert mm == [siglip2.mmr_select(embs, qe, K, lam) for K, lam in ((5, 0.5), (12, 0.2), (20, 0.9))]
    save("siglip2_tiny", seed=seed, pixel_values=pv, spatial_shapes=shapes, pixel_attention_mask=mask, input_ids=ids, attention_mask=am,
         image_features=fi, text_features=ft, text_features_masked=ftm, rank_idx=np.asarray(idx), rank_vals=np.asarray(vals),
         mmr_embs=embs, mmr_query=qe, mmr_5_05=np.asarray(mm[0]), mmr_12_02=np.asarray(mm[1]), mmr_20_09=np.asarray(mm[2]))


def gen_projector():
    """ContextProjector (model_pipeline.py:1506-1574) at reduced widths: the reference class itself, executed from its source."""
    import contextlib
    import io
    import types
    print("context projector")
    cfg = types.SimpleNamespace(bagel_hidden_dim=128, wan_text_dim=256, wan_text_length=32, use_semantic_alignment=False)
    ref = _refimport.ref_context_projector(cfg).eval()
    sd = projector.make_state_dict(cfg.bagel_hidden_dim, cfg.wan_text_dim, seed=4)
    ref.load_state_dict(sd)
    g = torch.Generator().manual_seed(6)
    outs = {}
    for L in (32, 20, 77):        # equal to / shorter / longer than the target length
        tok = torch.randn(2, L, cfg.bagel_hidden_dim, generator=g)
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            r = ref(tok)
        o = projector.forward(sd, tok, cfg.wan_text_length)
        assert all(torch.equal(a, b) for a, b in zip(r, o)), L
        outs[f"tokens_{L}"] = tok
        outs[f"out_{L}"] = torch.stack(r)
    save("context_projector", seed=4, **outs)


def gen_t5():
    """umT5 encoder (t5.py:267-312) at reduced size: the reference module in bf16 on padded ids + mask, sliced to the prompt length."""
    print("t5 encoder")
    cfg = t5.TINY_CFG
    ref = _refimport.ref_t5_encoder(cfg).to(torch.bfloat16).eval()
    sd = t5.make_state_dict(cfg, seed=9)
    ref.load_state_dict(sd)
    g = torch.Generator().manual_seed(12)
    T = 48
    outs = {}
    for n in (48, 33, 5):
        ids = torch.zeros(1, T, dtype=torch.long)
        ids[0, :n] = torch.randint(1, cfg["vocab_size"], (n,), generator=g)
        mask = (torch.arange(T) < n).long().unsqueeze(0)
        with torch.no_grad():
            r = ref(ids, mask)[0, :n]
        o = t5.encode(sd, cfg, ids[0, :n])
        assert torch.equal(r, o), (n, float((r.float() - o.float()).abs().max()))
        outs[f"ids_{n}"] = ids[0, :n]
        outs[f"out_{n}"] = r
    rel = torch.arange(-200, 201)
    save("t5_tiny", seed=9, rel=rel, buckets=t5.relative_position_bucket(rel), **outs)


def main():
    assert _refimport.available(), "the reference is not mounted; fixtures can only be generated in the build container"
    os.makedirs(OUT, exist_ok=True)
    ns = _refimport.load_reference()
    torch.set_num_threads(8)
    only = sys.argv[1:]
    gens = {"unipc": lambda: gen_unipc(ns), "masks": gen_masks, "text_weight": gen_text_weight, "dit_tiny": lambda: gen_dit_tiny(ns),
            "sampler": lambda: gen_sampler(ns), "dpmpp": lambda: gen_dpmpp(ns),
            "sampler_dpmpp": lambda: gen_sampler(ns, steps=12, keep=(0, 1, 2, 6, 10, 11), name="sampler_tiny_dpmpp", solver="dpm++"),
            "sampler50": lambda: gen_sampler(ns, steps=50, keep=(0, 1, 10, 25, 40, 49), name="sampler_tiny_50"), "vae": lambda: gen_vae(ns), "block": lambda: gen_dit_block_3072(ns),
            "siglip2": gen_siglip2, "projector": gen_projector, "t5": gen_t5}
    for k, fn in gens.items():
        if not only or k in only:
            fn()


if __name__ == "__main__":
    main()
