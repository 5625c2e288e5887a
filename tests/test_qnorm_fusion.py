"""GPU (-m gpu): the q projection's RMSNorm applied inside the attention kernels (round 5): uv_gemm_bf16_nt_ssq (the q GEMM also leaves the output
rows' sums of squares per 32-column group), uv_rms_scale_from_ssq (groups -> 1 / sqrt(mean + eps)) and uv_flash_attn_bf16_qnorm (the Q prologue
scales the RAW projection) against the separate pass they replace (WanRMSNorm, models/wan/utils/modules/model.py:82-85, 138, 169)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
BF16 = torch.bfloat16


@pytest.fixture(scope="module", autouse=True)
def _init():
    from univid_amd import _lib
    _lib.init()
    yield


def L():
    from univid_amd import _lib
    return _lib


@pytest.mark.parametrize("M,N,K,cfgs", [(22880, 3072, 3072, (0, 7, 8, 18, 12)), (300, 256, 256, (0, 12, 1, 5)), (2300, 2048, 640, (0, 7, 12)),
                                        (1120, 3072, 3072, (0, 12)), (70, 64, 128, (0, 12))])
def test_gemm_ssq_epilogue(M, N, K, cfgs):
    """out is UV_EPI_BF16's, bit for bit; every group's sum of squares equals the f64 sum of the stored bf16 values' squares to f32 accuracy;
    and it is THE SAME BITS from every kernel / tile shape / row split (persistent + strip, one tile per workgroup, 128 x 128 ring, 16-wave):
    the summation order is part of the contract."""
    from univid_amd._lib import EPI_BF16
    g = torch.Generator(device=DEV).manual_seed(M + N)
    a = (torch.rand(M, K, device=DEV, generator=g) * 2 - 1).to(BF16)
    w = ((torch.rand(N, K, device=DEV, generator=g) * 2 - 1) * 0.2).to(BF16)
    bias = (torch.rand(N, device=DEV, generator=g) - 0.5).to(BF16)
    ref = torch.zeros(M, N, device=DEV, dtype=BF16)
    L().gemm_bf16(a, w, bias, ref, EPI_BF16)
    first = None
    for cfg in cfgs:
        out = torch.full((M, N), 7.0, device=DEV, dtype=BF16)
        ssq = torch.full((M, N // 32 + 3), -1.0, device=DEV)                  # wider than needed: the tail columns must stay untouched
        L().gemm_bf16_ssq(a, w, bias, out, ssq, tile_cfg=cfg)
        assert torch.equal(out, ref), f"cfg {cfg}: output differs from UV_EPI_BF16"
        assert (ssq[:, N // 32:] == -1.0).all()
        got = ssq[:, :N // 32]
        want = out.double().pow(2).view(M, N // 32, 32).sum(-1)
        assert torch.isfinite(got).all()
        assert ((got.double() - want).abs() <= 4e-7 * want + 1e-30).all(), f"cfg {cfg}: max rel {float(((got.double() - want).abs() / want.clamp_min(1e-30)).max()):.2e}"
        if first is None:
            first = got.clone()
        else:
            assert torch.equal(got, first), f"cfg {cfg}: the group sums depend on the schedule ({int((got != first).sum())} differ)"
    # null bias
    out = torch.empty(M, N, device=DEV, dtype=BF16)
    ssq = torch.empty(M, N // 32, device=DEV)
    L().gemm_bf16_ssq(a, w, None, out, ssq)
    ref0 = torch.empty(M, N, device=DEV, dtype=BF16)
    L().gemm_bf16(a, w, None, ref0, EPI_BF16)
    assert torch.equal(out, ref0)
    assert ((ssq.double() - out.double().pow(2).view(M, N // 32, 32).sum(-1)).abs() <= 4e-7 * ssq.double() + 1e-30).all()


def test_gemm_ssq_rejects_bad_arguments():
    a = torch.zeros(64, 64, dtype=BF16, device=DEV)
    w = torch.zeros(48, 64, dtype=BF16, device=DEV)                           # N % 32 != 0
    with pytest.raises(L().UnividHipError):
        L().gemm_bf16_ssq(a, w, None, torch.zeros(64, 48, dtype=BF16, device=DEV), torch.zeros(64, 2, device=DEV))
    w = torch.zeros(64, 64, dtype=BF16, device=DEV)
    with pytest.raises(L().UnividHipError):
        L().gemm_bf16_ssq(a, w, None, torch.zeros(64, 64, dtype=BF16, device=DEV), torch.zeros(64, 1, device=DEV))      # ld_ssq < N / 32


def _rs_ref(ssq, C, eps):
    """The kernel's order: four ascending quarter sums (f32), then (p0 + p1) + (p2 + p3)."""
    groups = ssq.shape[1]
    per = (groups + 3) // 4
    parts = []
    for k in range(4):
        t = torch.zeros(ssq.shape[0], device=ssq.device)
        for g in range(k * per, min((k + 1) * per, groups)):
            t = t + ssq[:, g]
        parts.append(t)
    t = (parts[0] + parts[1]) + (parts[2] + parts[3])
    return 1.0 / torch.sqrt(t / C + eps)


@pytest.mark.parametrize("M,groups", [(1000, 96), (37, 8), (5, 6)])
def test_rms_scale_from_ssq(M, groups):
    g = torch.Generator(device=DEV).manual_seed(M)
    ssq = torch.rand(M, groups, device=DEV, generator=g) * 40
    rs = torch.empty(M, device=DEV)
    L().rms_scale_from_ssq(ssq, rs, M, groups * 32, 1e-6)
    ref = _rs_ref(ssq, groups * 32, 1e-6)
    assert ((rs - ref).abs() <= 2 ** -22 * ref).all()         # (1 / sqrt: the device's sqrt + reciprocal against torch's; 1-2 ulp)


@pytest.mark.parametrize("Lq,Lk,H,D,B", [(300, 77, 2, 128, 1), (1000, 512, 3, 128, 2), (260, 64, 4, 64, 1), (2304, 2304, 2, 128, 1), (520, 512, 24, 128, 2)])
def test_flash_attn_qnorm_prologue_is_the_separate_pass(Lq, Lk, H, D, B):
    """uv_flash_attn_bf16_qnorm on the raw q == uv_flash_attn_bf16 on q' = bf16( bf16(q * rs) * w ) (the rounding points of uv_rmsnorm_rope), bit for
    bit, in every kernel the dispatcher can pick (cross-attention fwd3, the generic head_dim-64 / -128 forms, the long-key fwd12), stacked samples
    included; and with rs from the fused GEMM path the result is what uv_rmsnorm_rope + uv_flash_attn_bf16 give up to the rare rounding flip of the
    row scale (the two sum the squares in different orders)."""
    C = H * D
    g = torch.Generator(device=DEV).manual_seed(Lq + Lk)
    q = (torch.randn(B * Lq, C, device=DEV, generator=g) * 1.7).to(BF16)
    k = torch.randn(B * Lk, C, device=DEV, generator=g).to(BF16)
    v = torch.randn(B * Lk, C, device=DEV, generator=g).to(BF16)
    w = (1 + 0.2 * torch.randn(C, device=DEV, generator=g))
    vt = torch.zeros(C, (B - 1) * Lk + (Lk + 63) // 64 * 64, dtype=BF16, device=DEV)
    vt[:, :B * Lk] = v.t()
    ssq = q.float().pow(2).view(B * Lq, C // 32, 32).sum(-1)
    rs = torch.empty(B * Lq, device=DEV)
    L().rms_scale_from_ssq(ssq.contiguous(), rs, B * Lq, C, 1e-6)
    qn = ((q.float() * rs[:, None]).to(BF16).float() * w).to(BF16)
    want = torch.empty(B * Lq, C, dtype=BF16, device=DEV)
    L().flash_attn(qn, k, vt, want, Lq, Lk, H, D, 1 / math.sqrt(D), batch=B)
    got = torch.full((B * Lq + 4, C), 3.0, dtype=BF16, device=DEV)
    L().flash_attn(q, k, vt, got, Lq, Lk, H, D, 1 / math.sqrt(D), batch=B, q_rs=rs, q_weight=w)
    assert (got[B * Lq:] == 3.0).all()
    assert torch.equal(got[:B * Lq], want)
    # against the separate pass with ITS sum-of-squares order
    q2 = q.clone()
    L().rmsnorm_rope(q2, q2, w, B * Lq, C, D, 1e-6)
    exact = (q2 == qn).float().mean().item()
    assert exact >= 0.999, exact
    assert ((q2.float() - qn.float()).abs() <= 2 ** -6 * qn.float().abs() + 1e-30).all()      # (a flipped rounding = one bf16 ulp)
