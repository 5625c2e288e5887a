"""CPU: the oracle (oracle/*.py, the CPU restatement) reproduces the reference's own outputs bit for bit.

The golden vectors were produced by importing the reference modules (oracle/gen_golden.py, build container only);
here only the fixtures and the restatement are needed, so these tests also run on the GPU box.
"""
import json
import os

import pytest
import torch

from oracle import dpmpp, sampler, unipc, wan_dit, wan_vae

from conftest import GOLDEN, load_golden


def test_unipc_schedule_and_trajectory():
    g = load_golden("unipc")
    for steps, shift in ((50, 5.0), (10, 5.0), (40, 3.0)):
        s = unipc.FlowUniPC(1000, shift=1)
        ts = s.set_timesteps(steps, shift=shift)
        assert torch.equal(ts, g[f"timesteps_{steps}_{shift}"])
        assert torch.equal(s.sigmas, g[f"sigmas_{steps}_{shift}"])
    # known prefix/suffix of the 50-step shift-5 schedule
    assert g["timesteps_50_5.0"][:5].tolist() == [999, 995, 991, 987, 982]
    assert g["timesteps_50_5.0"][-3:].tolist() == [241, 172, 92]
    s = unipc.FlowUniPC(1000, shift=1)
    s.set_timesteps(10, shift=5.0)
    lat = g["x"]
    for i, t in enumerate(s.timesteps):
        lat = s.step(g["model_outputs"][i], t, lat)
        assert torch.equal(lat, g["trajectory"][i]), f"step {i}"


def test_dpmpp_schedule_and_trajectory():
    """sample_solver='dpm++' (textimage2video.py:343-351): sigmas / timesteps through get_sampling_sigmas + retrieve_timesteps
    and full trajectories at 10 and 20 steps (the < 15-step rule of fm_solvers.py:771-777 is moot at order 2, both are
    pinned), and the 2- and 1-step corner cases, against the imported reference's outputs."""
    g = load_golden("dpmpp")
    for steps, shift in ((50, 5.0), (10, 5.0), (20, 3.0), (2, 5.0), (1, 5.0)):
        s = dpmpp.FlowDPMpp(1000, shift=1)
        ts = s.set_timesteps(sigmas=dpmpp.get_sampling_sigmas(steps, shift))
        assert torch.equal(ts, g[f"timesteps_{steps}_{shift}"])
        assert torch.equal(s.sigmas, g[f"sigmas_{steps}_{shift}"])
    assert g["timesteps_50_5.0"][0].item() == 1000 and g["sigmas_50_5.0"][-1].item() == 0.0
    for steps in (10, 20, 2, 1):
        s = dpmpp.FlowDPMpp(1000, shift=1)
        s.set_timesteps(sigmas=dpmpp.get_sampling_sigmas(steps, 5.0))
        lat = g["x"]
        for i, t in enumerate(s.timesteps):
            lat = s.step(g[f"model_outputs_{steps}"][i], t, lat)
            assert torch.equal(lat, g[f"trajectory_{steps}"][i]), f"{steps} steps, step {i}"
        # final sigma 0: the last update returns the last x0 prediction itself (lambda_t = +inf, exp(-h) = 0)
        assert torch.isfinite(lat).all()


def test_masks_like():
    g = load_golden("masks_like")
    x = [torch.zeros(3, 4, 2, 2)]
    a1, a2 = sampler.masks_like(x, zero=False)
    b1, b2 = sampler.masks_like(x, zero=True)
    assert torch.equal(a1[0], g["ones1"]) and torch.equal(a2[0], g["ones2"])
    assert torch.equal(b1[0], g["zero1"]) and torch.equal(b2[0], g["zero2"])
    assert b2[0][:, 0].abs().sum() == 0 and b2[0][:, 1:].min() == 1


def test_text_weight_known_answers():
    tab = json.load(open(os.path.join(GOLDEN, "text_weight.json")))
    for schedule, total, ratio, enabled, step, w in tab["rows"]:
        assert sampler.text_weight(step, total, ratio, 1.3, 1.0, schedule, bool(enabled)) == w


def test_dit_tiny_forward():
    g = load_golden("dit_tiny")
    cfg = wan_dit.TINY_CFG
    sd = wan_dit.make_state_dict(cfg, g["seed"])
    L = 4 * 8 * 8
    with torch.no_grad():
        one = wan_dit.dit_forward(sd, cfg, [g["x"]], g["t_one"], [g["ctx"]], L)[0]
        two = wan_dit.dit_forward(sd, cfg, [g["x"]], g["t_two"], [g["ctx"]], L)[0]
        pad = wan_dit.dit_forward(sd, cfg, [g["x"]], torch.full((1, L + 32), 937.0), [g["ctx"]], L + 32)[0]
    assert one.dtype == torch.float32
    assert torch.equal(one, g["out_one"])
    assert torch.equal(two, g["out_two"])
    assert torch.equal(pad, g["out_pad"])
    assert not torch.equal(one, two)
    # sequence padding must not change the valid tokens (k_lens masking, model.py:149)
    assert torch.allclose(pad, one, rtol=0, atol=2e-2)


def test_dit_block_ti2v5b_width():
    g = load_golden("dit_block_3072")
    dim, heads, L = 3072, 24, 48
    shapes = {k: v for k, v in wan_dit.state_dict_shapes(dict(wan_dit.TI2V_5B_CFG, num_layers=1)).items() if k.startswith("blocks.0.")}
    from univid_amd import detinit
    sd = detinit.init_state_dict_({k: torch.empty(v) for k, v in shapes.items()}, g["seed"])
    e0 = g["e_rows"][g["tid"]].unsqueeze(0)
    freqs = wan_dit.rope_table(dim // heads)
    for name, xin in (("f32", g["x"]), ("bf16", g["x"].to(torch.bfloat16))):
        with torch.no_grad():
            out = wan_dit.block_forward(sd, "blocks.0.", xin, e0, torch.tensor([L]), g["grid"], freqs, g["ctx"], heads, 1e-6)
        assert out.dtype == torch.float32
        assert torch.equal(out, g["out_" + name])


@pytest.mark.parametrize("fixture", ["sampler_tiny", "sampler_tiny_50", "sampler_tiny_dpmpp"])
def test_sampler_trajectories(fixture):
    """10-step and 50-step (BASELINE config 2's step count) t2v / i2v trajectories: the oracle loop reproduces the reference
    pieces' latents bit for bit at every kept step."""
    g = load_golden(fixture)
    cfg = wan_dit.TINY_CFG
    sd = wan_dit.make_state_dict(cfg, g["seed"])
    keep = g["kept_steps"].tolist()
    solver = "dpm++" if fixture.endswith("dpmpp") else "unipc"
    for mode in ("t2v", "i2v"):
        rec = []
        with torch.no_grad():
            sampler.denoise(sd, cfg, g["noise"], [g["ctx"]], [g["ctx_null"]], g["steps"], g["shift"], g["guide_scale"],
                            z=(g["z"] if mode == "i2v" else None), record=rec, sample_solver=solver)
        for j, i in enumerate(keep):
            assert torch.equal(rec[i][0], g[f"{mode}_noise_pred"][j]), f"{mode} noise_pred step {i}"
            assert torch.equal(rec[i][1], g[f"{mode}_latents"][j]), f"{mode} latent step {i}"
    # i2v keeps the first latent frame pinned to z (textimage2video.py:598)
    assert torch.equal(g["i2v_latents"][-1][:, 0], g["z"][:, 0])


def test_vae_encode_decode():
    g = load_golden("vae_small")
    cfg = wan_vae.SMALL_CFG
    v = wan_vae.WanVAE(wan_vae.make_state_dict(cfg, g["seed"]), cfg)
    for i in range(3):
        with torch.no_grad():
            assert torch.equal(wan_vae.vae_encode(v, [g[f"enc_in_{i}"]])[0], g[f"enc_out_{i}"])
            assert torch.equal(wan_vae.vae_decode(v, [g[f"dec_in_{i}"]])[0], g[f"dec_out_{i}"])
    # shape contract of the VAE (vae stride 4,16,16; 4n+1 frames)
    assert g["enc_out_0"].shape == (48, 3, 2, 3) and g["dec_out_0"].shape == (3, 9, 32, 48)
    assert g["dec_out_1"].shape == (3, 1, 32, 32)
    assert g["dec_out_0"].abs().max() <= 1.0


def test_text_weight_hook_changes_output():
    """The per-layer context scaling (model_pipeline.py:1787-1797) acts through V (norm_k mostly cancels it on K)."""
    cfg = wan_dit.TINY_CFG
    sd = wan_dit.make_state_dict(cfg, 0)
    g = load_golden("dit_tiny")
    L = 256
    m = sampler.context_mask((1, cfg["text_len"], cfg["dim"]), 1.3, bagel_sequence_length=128)
    assert m.dtype == torch.bfloat16 and float(m[0, 0, 0]) == float(torch.tensor(1.3).to(torch.bfloat16))
    assert float(m[0, cfg["text_len"] // 2, 0]) == 1.0        # text_len = min(128, 32 // 2) = 16 rows scaled
    with torch.no_grad():
        base = wan_dit.dit_forward(sd, cfg, [g["x"]], g["t_one"], [g["ctx"]], L)[0]
        hooked = wan_dit.dit_forward(sd, cfg, [g["x"]], g["t_one"], [g["ctx"]], L, context_scale_fn=lambda i: m)[0]
    assert torch.equal(base, g["out_one"]) and not torch.equal(base, hooked)


# ---------------------------------------------------------------------------------------------------------------
# BASELINE config 5: SigLIP2 ranker. Golden = transformers.Siglip2Model (HF's implementation, which the reference calls) with
# the deterministic weights; mmr_select golden = the reference's own function.
# ---------------------------------------------------------------------------------------------------------------
def test_siglip2_oracle_matches_transformers_golden():
    from oracle import siglip2
    g = load_golden("siglip2_tiny")
    cfg = siglip2.TINY_CFG
    sd = siglip2.make_state_dict(cfg, int(g["seed"]))
    fi = siglip2.image_features(sd, cfg, g["pixel_values"], g["pixel_attention_mask"], g["spatial_shapes"])
    ft = siglip2.text_features(sd, cfg, g["input_ids"])
    ftm = siglip2.text_features(sd, cfg, g["input_ids"], g["attention_mask"])
    for got, ref in ((fi, g["image_features"]), (ft, g["text_features"]), (ftm, g["text_features_masked"])):
        assert (got - ref).abs().max() < 5e-6 * max(1.0, float(ref.abs().max()))
    idx, vals = siglip2.rank_frames(fi, ft[:1], 4)
    assert idx == g["rank_idx"].tolist() and torch.allclose(torch.tensor(vals), g["rank_vals"].float(), atol=1e-6)
    for K, lam, key in ((5, 0.5, "mmr_5_05"), (12, 0.2, "mmr_12_02"), (20, 0.9, "mmr_20_09")):
        assert siglip2.mmr_select(g["mmr_embs"], g["mmr_query"], K, lam) == g[key].tolist()


def test_context_projector_oracle_matches_reference_golden():
    from oracle import projector
    g = load_golden("context_projector")
    sd = projector.make_state_dict(128, 256, int(g["seed"]))
    for L in (32, 20, 77):
        out = torch.stack(projector.forward(sd, g[f"tokens_{L}"], 32))
        assert out.dtype == torch.bfloat16 and torch.equal(out, g[f"out_{L}"])


def test_t5_oracle_matches_reference_golden():
    """oracle/t5.py against the outputs of the reference's own T5Encoder (bf16, padded ids + mask, sliced to the prompt)."""
    from oracle import t5
    g = load_golden("t5_tiny")
    sd = t5.make_state_dict(t5.TINY_CFG, int(g["seed"]))
    for n in (48, 33, 5):
        assert torch.equal(t5.encode(sd, t5.TINY_CFG, g[f"ids_{n}"]), g[f"out_{n}"])
    assert torch.equal(t5.relative_position_bucket(g["rel"]), g["buckets"])
