"""Wan2.2 3D causal VAE (`Wan2_2_VAE`) on MI355X: the reference's module tree / state-dict names / list API over
the fp32 implicit-GEMM convolution kernels of libunivid_hip.so.

Mirrors /root/reference/models/wan/utils/modules/vae2_2.py: CausalConv3d :17-42, RMS_norm :45-59, Resample :71-190,
ResidualBlock :193-235, AttentionBlock :238-277, AvgDown3D/DupUp3D :316-412, Down_/Up_ResidualBlock :415-497,
Encoder3d :500-613, Decoder3d :616-723, WanVAE_ :734-860, Wan2_2_VAE :888-1051.

The nn.Modules below only HOLD parameters (same names and shapes as the reference, so `Wan2.2_VAE.pth` loads);
all arithmetic runs in `_Engine`, which walks the tree and launches HIP kernels on channels-last fp32 activations
[T, H, W, C]. The reference's chunked streaming is kept (the first chunk - one frame - on its own, because its special cases
("Rep", DupUp3D(first_chunk)) are part of the function being computed; then passes of `frames_per_pass` latent frames / 4-frame
chunks instead of one: same values, larger launches); the per-convolution feature cache (CACHE_T = 2 frames) is a 2-frame tensor
per convolution that is prepended to each pass's input.
Everything is fp32 like the reference (`dtype=torch.float`, vae2_2.py:897, 1028, 1042). No eager fallback.
"""
import contextlib
import logging
import math
from typing import List

import torch
import torch.nn as nn

from .. import _lib

CACHE_T = 2

_MEAN = [-0.2289, -0.0052, -0.1323, -0.2339, -0.2799, 0.0174, 0.1838, 0.1557, -0.1382, 0.0542, 0.2813, 0.0891, 0.1570,
         -0.0098, 0.0375, -0.1825, -0.2246, -0.1207, -0.0698, 0.5109, 0.2665, -0.2108, -0.2158, 0.2502, -0.2055, -0.0322,
         0.1109, 0.1567, -0.0729, 0.0899, -0.2799, -0.1230, -0.0313, -0.1649, 0.0117, 0.0723, -0.2839, -0.2083, -0.0520,
         0.3748, 0.0152, 0.1957, 0.1433, -0.2944, 0.3573, -0.0548, -0.1681, -0.0667]
_STD = [0.4765, 1.0364, 0.4514, 1.1677, 0.5313, 0.4990, 0.4818, 0.5013, 0.8158, 1.0344, 0.5894, 1.0901, 0.6885, 0.6165,
        0.8454, 0.4978, 0.5759, 0.3523, 0.7135, 0.6804, 0.5833, 1.4146, 0.8986, 0.5659, 0.7069, 0.5338, 0.4889, 0.4917,
        0.4069, 0.4999, 0.6866, 0.4093, 0.5709, 0.6065, 0.6415, 0.4944, 0.5726, 1.2042, 0.5458, 1.6887, 0.3971, 1.0600,
        0.3943, 0.5537, 0.5444, 0.4089, 0.7468, 0.7744]


# ---- parameter containers (names/shapes = reference) ------------------------------------------------------------
class CausalConv3d(nn.Conv3d):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._padding = (self.padding[2], self.padding[2], self.padding[1], self.padding[1], 2 * self.padding[0], 0)
        self.padding = (0, 0, 0)


class RMS_norm(nn.Module):
    def __init__(self, dim, channel_first=True, images=True, bias=False):
        super().__init__()
        broadcastable_dims = (1, 1, 1) if not images else (1, 1)
        self.channel_first = channel_first
        self.scale = dim ** 0.5
        self.gamma = nn.Parameter(torch.ones((dim, *broadcastable_dims) if channel_first else (dim,)))
        self.bias = 0.0


class Resample(nn.Module):
    def __init__(self, dim, mode):
        assert mode in ("none", "upsample2d", "upsample3d", "downsample2d", "downsample3d")
        super().__init__()
        self.dim, self.mode = dim, mode
        if mode in ("upsample2d", "upsample3d"):
            self.resample = nn.Sequential(nn.Identity(), nn.Conv2d(dim, dim, 3, padding=1))
            if mode == "upsample3d":
                self.time_conv = CausalConv3d(dim, dim * 2, (3, 1, 1), padding=(1, 0, 0))
        elif mode in ("downsample2d", "downsample3d"):
            self.resample = nn.Sequential(nn.Identity(), nn.Conv2d(dim, dim, 3, stride=(2, 2)))
            if mode == "downsample3d":
                self.time_conv = CausalConv3d(dim, dim, (3, 1, 1), stride=(2, 1, 1), padding=(0, 0, 0))
        else:
            self.resample = nn.Identity()


class ResidualBlock(nn.Module):
    def __init__(self, in_dim, out_dim, dropout=0.0):
        super().__init__()
        self.in_dim, self.out_dim = in_dim, out_dim
        self.residual = nn.Sequential(RMS_norm(in_dim, images=False), nn.SiLU(), CausalConv3d(in_dim, out_dim, 3, padding=1),
                                      RMS_norm(out_dim, images=False), nn.SiLU(), nn.Dropout(dropout),
                                      CausalConv3d(out_dim, out_dim, 3, padding=1))
        self.shortcut = CausalConv3d(in_dim, out_dim, 1) if in_dim != out_dim else nn.Identity()


class AttentionBlock(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.norm = RMS_norm(dim)
        self.to_qkv = nn.Conv2d(dim, dim * 3, 1)
        self.proj = nn.Conv2d(dim, dim, 1)


class Down_ResidualBlock(nn.Module):
    def __init__(self, in_dim, out_dim, dropout, mult, temperal_downsample=False, down_flag=False):
        super().__init__()
        self.in_dim, self.out_dim = in_dim, out_dim
        self.factor_t = 2 if temperal_downsample else 1
        self.factor_s = 2 if down_flag else 1
        mods = []
        for _ in range(mult):
            mods.append(ResidualBlock(in_dim, out_dim, dropout))
            in_dim = out_dim
        if down_flag:
            mods.append(Resample(out_dim, mode="downsample3d" if temperal_downsample else "downsample2d"))
        self.downsamples = nn.Sequential(*mods)


class Up_ResidualBlock(nn.Module):
    def __init__(self, in_dim, out_dim, dropout, mult, temperal_upsample=False, up_flag=False):
        super().__init__()
        self.in_dim, self.out_dim, self.up_flag = in_dim, out_dim, up_flag
        self.factor_t = 2 if temperal_upsample else 1
        mods = []
        for _ in range(mult):
            mods.append(ResidualBlock(in_dim, out_dim, dropout))
            in_dim = out_dim
        if up_flag:
            mods.append(Resample(out_dim, mode="upsample3d" if temperal_upsample else "upsample2d"))
        self.upsamples = nn.Sequential(*mods)


class Encoder3d(nn.Module):
    def __init__(self, dim=128, z_dim=4, dim_mult=(1, 2, 4, 4), num_res_blocks=2, attn_scales=(),
                 temperal_downsample=(True, True, False), dropout=0.0):
        super().__init__()
        dims = [dim * u for u in [1] + list(dim_mult)]
        self.conv1 = CausalConv3d(12, dims[0], 3, padding=1)
        downs = []
        for i, (in_dim, out_dim) in enumerate(zip(dims[:-1], dims[1:])):
            t_down = temperal_downsample[i] if i < len(temperal_downsample) else False
            downs.append(Down_ResidualBlock(in_dim, out_dim, dropout, num_res_blocks, t_down, i != len(dim_mult) - 1))
        self.downsamples = nn.Sequential(*downs)
        self.middle = nn.Sequential(ResidualBlock(out_dim, out_dim, dropout), AttentionBlock(out_dim),
                                    ResidualBlock(out_dim, out_dim, dropout))
        self.head = nn.Sequential(RMS_norm(out_dim, images=False), nn.SiLU(), CausalConv3d(out_dim, z_dim, 3, padding=1))


class Decoder3d(nn.Module):
    def __init__(self, dim=128, z_dim=4, dim_mult=(1, 2, 4, 4), num_res_blocks=2, attn_scales=(),
                 temperal_upsample=(False, True, True), dropout=0.0):
        super().__init__()
        dims = [dim * u for u in [dim_mult[-1]] + list(dim_mult[::-1])]
        self.conv1 = CausalConv3d(z_dim, dims[0], 3, padding=1)
        self.middle = nn.Sequential(ResidualBlock(dims[0], dims[0], dropout), AttentionBlock(dims[0]),
                                    ResidualBlock(dims[0], dims[0], dropout))
        ups = []
        for i, (in_dim, out_dim) in enumerate(zip(dims[:-1], dims[1:])):
            t_up = temperal_upsample[i] if i < len(temperal_upsample) else False
            ups.append(Up_ResidualBlock(in_dim, out_dim, dropout, num_res_blocks + 1, t_up, i != len(dim_mult) - 1))
        self.upsamples = nn.Sequential(*ups)
        self.head = nn.Sequential(RMS_norm(out_dim, images=False), nn.SiLU(), CausalConv3d(out_dim, 12, 3, padding=1))


# ---- execution engine -----------------------------------------------------------------------------------------
def _pad32(c):
    return (c + 31) // 32 * 32


def f16_weight_scale(max_abs):
    """Power of two that puts the largest weight magnitude of a convolution into [2^13, 2^14) for the f16x3 split (uv_split_weights_f16x3):
    far below fp16's 65 504, and the lo pieces (<= 2^-11 of their hi piece) of all but the smallest weights stay normal fp16 numbers
    (>= 2^-14). 1.0 for an all-zero or non-finite tensor."""
    if not (max_abs > 0 and math.isfinite(max_abs)):
        return 1.0
    return 2.0 ** (13 - math.floor(math.log2(max_abs)))


class _ConvOp:
    """One convolution: channels-last weight [Cout, taps * Cin_pad] and its input ring."""

    def __init__(self, conv, is2d=False):
        w = conv.weight.detach().float()
        if is2d:
            w = w.unsqueeze(2)
        self.cout, self.cin, self.kt, self.kh, self.kw = w.shape
        self.cin_pad = _pad32(self.cin)
        wp = torch.zeros(self.cout, self.kt, self.kh, self.kw, self.cin_pad, dtype=torch.float32, device=w.device)
        wp[..., :self.cin] = w.permute(0, 2, 3, 4, 1)
        self.w = wp.reshape(self.cout, -1).contiguous()
        self.w2d = w.reshape(self.cout, self.cin).contiguous() if (self.kt, self.kh, self.kw) == (1, 1, 1) else None
        self.b = conv.bias.detach().float().contiguous()
        self.w_split = None
        self.w_split6 = None
        self.w_split_f16 = None      # (fp16 hi | lo pieces of w * scale, scale) for the f16x3 kernel
        self.caches = {}     # (H, W, prefix) -> the cached frames [prefix, H, W, cin_pad] (zeros = the causal zero padding)
        self.rings = {}      # private input rings of the few convolutions whose channel count is padded (see _Engine.conv_input)

    def split(self):
        """bf16 hi/lo planes of the weights for the bf16x3 kernel ([Cout][K/32][32 hi | 32 lo])."""
        if self.w_split is None:
            self.w_split = torch.empty(self.w.numel() * 2, dtype=torch.bfloat16, device=self.w.device)
            _lib.call("uv_split_weights_bf16x3", _lib.ptr(self.w), _lib.ptr(self.w_split), self.w.numel(), _lib.stream_ptr())
        return self.w_split

    def split6(self):
        """three bf16 planes of the weights for the bf16x6 kernel ([Cout][K/32][32 p0 | 32 p1 | 32 p2], w = p0 + p1 + p2 exactly)."""
        if self.w_split6 is None:
            self.w_split6 = torch.empty(self.w.numel() * 3, dtype=torch.bfloat16, device=self.w.device)
            _lib.call("uv_split_weights_bf16x6", _lib.ptr(self.w), _lib.ptr(self.w_split6), self.w.numel(), _lib.stream_ptr())
        return self.w_split6

    def split_f16(self):
        """two IEEE fp16 pieces of w * scale ([Cout][K/32][32 hi | 32 lo]) for the f16x3 kernel; scale = the power of two that puts
        max |w| into [2^13, 2^14): all but the smallest weights then have NORMAL lo pieces (fp16 subnormals start below 2^-14), and
        65 504 is far away. The kernel undoes the scale exactly in its epilogue."""
        if self.w_split_f16 is None:
            scale = f16_weight_scale(float(self.w.abs().max()))
            buf = torch.empty(self.w.numel() * 2, dtype=torch.float16, device=self.w.device)
            _lib.call("uv_split_weights_f16x3", _lib.ptr(self.w), _lib.ptr(buf), self.w.numel(), float(scale), _lib.stream_ptr())
            self.w_split_f16 = (buf, float(scale))
        return self.w_split_f16

    def phases(self):
        """The four 2x2 phase kernels of a 2x-nearest-upsampling 3x3 convolution (Resample upsample2d / upsample3d, vae2_2.py:86-96, 153-155)
        as _ConvOp-like objects: on the upsampled image the three taps of a row fall on two source rows (output row parity a = 0:
        {dy = 0} -> y - 1, {1, 2} -> y; a = 1: {0, 1} -> y, {2} -> y + 1; columns alike), so each output phase (a, b) is a 2x2 convolution
        of the SOURCE image with the sums of the collapsed taps - 16 instead of 36 multiply-adds per four output pixels. The sums are
        formed in fp64 and rounded once to f32 (a <= 2^-24 relative change of the summed weights: the same grade as the f32 roundings
        of the three separate products they replace)."""
        if getattr(self, "_phases", None) is None:
            assert (self.kt, self.kh, self.kw) == (1, 3, 3)
            w = self.w.view(self.cout, 3, 3, self.cin_pad).double()
            rows = {0: ([0], [1, 2]), 1: ([0, 1], [2])}
            out = []
            for a in (0, 1):
                for b in (0, 1):
                    wp = torch.zeros(self.cout, 2, 2, self.cin_pad, dtype=torch.float64, device=w.device)
                    for i, dys in enumerate(rows[a]):
                        for j, dxs in enumerate(rows[b]):
                            for dy in dys:
                                for dx in dxs:
                                    wp[:, i, j] += w[:, dy, dx]
                    ph = object.__new__(_ConvOp)
                    ph.cout, ph.cin, ph.kt, ph.kh, ph.kw, ph.cin_pad = self.cout, self.cin, 1, 2, 2, self.cin_pad
                    ph.w = wp.float().reshape(self.cout, -1).contiguous()
                    ph.w2d, ph.b, ph.w_split, ph.w_split6, ph.w_split_f16 = None, self.b, None, None, None
                    ph.caches, ph.rings = {}, {}
                    out.append(ph)
            self._phases = out
        return self._phases

    def cache(self, H, W, prefix=CACHE_T):
        key = (H, W, prefix)
        c = self.caches.get(key)
        if c is None:
            c = self.caches[key] = torch.zeros(prefix, H, W, self.cin_pad, dtype=torch.float32, device=self.w.device)
        return c


class _Engine:
    def __init__(self, model: "WanVAE_", precision="fp32"):
        if precision not in ("fp32", "bf16x6", "f16x3", "bf16x3"):
            raise ValueError("precision must be 'fp32' (f32 MFMA, like the reference), 'bf16x6' (f32-grade: exact 3-way operand "
                             "splitting on the bf16 MFMA), 'f16x3' (f32-grade: 2-way fp16 splitting, 3 passes, for the convolutions "
                             "behind an RMS_norm; bf16x6 elsewhere) or 'bf16x3' (2-way bf16 split, ~1e-5)")
        self.m = model
        self.precision = precision
        self.ops = {}
        for name, mod in model.named_modules():
            if isinstance(mod, nn.Conv3d):
                self.ops[mod] = _ConvOp(mod)
            elif isinstance(mod, nn.Conv2d):
                self.ops[mod] = _ConvOp(mod, is2d=True)
        self.dev = next(model.parameters()).device
        self.scratch = {}    # (H, W, channels) -> shared conv-input buffer [frames, H, W, channels]
        self._fmt = {}       # RMS_norm module -> activation format its output is written in (see split_fmt)
        self.phase_upsample = True      # upsampling 3x3 convolutions as four 2x2 phase convolutions (every mode but 'fp32'); A/B switch

    def split_fmt(self, norm, cin, cout):
        """Format in which RMS_norm (+ SiLU) writes the input of the convolution behind it: 0 = f32 rows, 1 = two bf16 pieces per
        element (bf16x3), 2 = two IEEE fp16 pieces (f16x3). The fp16 form needs |activation| < 65 504: an RMS-normalised row is bounded
        by sqrt(C) * max|gamma| (|x_i| / ||x|| <= 1; SiLU does not grow a magnitude), checked here ONCE per norm; a norm whose bound
        does not hold keeps f32 rows and its convolution runs as bf16x6 (exact splitting, no range limit)."""
        if cin % 32 or cout % 4 or self.precision not in ("bf16x3", "f16x3"):
            return 0
        if self.precision == "bf16x3":
            return 1 if cout % 32 == 0 else 0
        f = self._fmt.get(norm)
        if f is None:
            bound = math.sqrt(norm.gamma.numel()) * float(norm.gamma.detach().abs().max())
            f = self._fmt[norm] = 2 if bound < 6.0e4 else 0
        return f

    def reset(self):
        """WanVAE_.clear_cache (vae2_2.py:853-860): all cached frames back to the causal zero padding."""
        for op in self.ops.values():
            for c in op.caches.values():
                c.zero_()
            for r in op.rings.values():
                r.zero_()

    def conv_input(self, op, H, W, T, fill, prefix=CACHE_T):
        """The input of a cached (time-causal) convolution: [cached frames (prefix) | the T current frames, written by fill(dst)]
        as one [prefix + T, H, W, cin_pad] tensor, with the cache advanced to the last `prefix` frames of it for the next pass
        (the feature-cache update of vae2_2.py:219-232). The reference keeps 2 frames per convolution; so does this: the
        [prefix + T]-frame tensor itself is a scratch buffer SHARED by all convolutions of one (H, W, channels) shape, so its size
        follows the pass length T without multiplying by the number of convolutions. Convolutions whose channel count is padded
        (the first convolution of encoder and decoder) keep a small private ring instead: their pad channels must stay zero.
        Returns (input tensor, None | callable to run after the convolution has been launched)."""
        need = prefix + T
        if op.cin != op.cin_pad:
            key = (H, W, prefix)
            ring = op.rings.get(key)
            if ring is None or ring.shape[0] < need:
                grown = torch.zeros(need, H, W, op.cin_pad, dtype=torch.float32, device=self.dev)
                if ring is not None:
                    grown[:prefix].copy_(ring[:prefix])
                ring = op.rings[key] = grown
            fill(ring[prefix:need])
            nxt = ring[T:need].clone()
            return ring[:need], lambda: ring[:prefix].copy_(nxt)
        key = (H, W, op.cin_pad)
        ring = self.scratch.get(key)
        if ring is None or ring.shape[0] < need:
            self.scratch[key] = None         # release the smaller buffer before allocating the larger one
            ring = self.scratch[key] = torch.empty(need, H, W, op.cin_pad, dtype=torch.float32, device=self.dev)
        cache = op.cache(H, W, prefix)
        ring[:prefix].copy_(cache)
        fill(ring[prefix:need])
        cache.copy_(ring[T:need])
        return ring[:need], None

    # -- kernels --
    def _conv(self, op, src, Tin, Hin, Win, Tout, Hout, Wout, st=1, sh=1, sw=1, t_off=0, ph=0, pw=0, up=0, interleave=0,
              resid=None, out=None, ldo=None, in_split=0, act_scale=None):
        cout = op.cout // 2 if interleave else op.cout
        tt = Tout * 2 if interleave else Tout
        if out is None:
            out = torch.empty(tt, Hout, Wout, cout, dtype=torch.float32, device=self.dev)
            ldo = cout
        flops = 2 * Tout * Hout * Wout * op.cout * op.kt * op.kh * op.kw * op.cin
        geom = (ldo, Tout, Hout, Wout, op.cin_pad, op.cout, op.kt, op.kh, op.kw, st, sh, sw, t_off, ph, pw, up, interleave,
                _lib.ptr(resid), 0 if resid is None else resid.stride(-2))
        if self.precision == "bf16x3":
            _lib.call("uv_conv3d_bf16x3", _lib.ptr(src), src.stride(-2), Tin, Hin, Win, _lib.ptr(op.split()), _lib.ptr(op.b),
                      _lib.ptr(out), *geom, int(in_split == 1), _lib.stream_ptr(), flops=flops)
        elif self.precision == "f16x3" and in_split == 2:
            wsp, wscale = op.split_f16()
            _lib.call("uv_conv3d_f16x3", _lib.ptr(src), src.stride(-2), Tin, Hin, Win, _lib.ptr(wsp), _lib.ptr(op.b), _lib.ptr(out),
                      *geom, wscale, _lib.ptr(act_scale), _lib.stream_ptr(), flops=flops)
        else:
            x6 = self.precision in ("bf16x6", "f16x3")         # f16x3: convolutions whose input is not an RMS_norm output
            _lib.call("uv_conv3d_bf16x6" if x6 else "uv_conv3d_f32", _lib.ptr(src), src.stride(-2), Tin, Hin, Win,
                      _lib.ptr(op.split6() if x6 else op.w), _lib.ptr(op.b), _lib.ptr(out), *geom, _lib.stream_ptr(), flops=flops)
        return out

    def _rms_silu(self, x, gamma, out, silu=True, split=0):
        P = x.numel() // x.shape[-1]
        _lib.call("uv_vae_rms_silu", _lib.ptr(x), x.stride(-2), _lib.ptr(gamma), _lib.ptr(out), out.stride(-2), P, x.shape[-1],
                  int(silu), int(split), _lib.stream_ptr())

    def _split16(self, x):
        """fp16 pieces of a RAW feature map (not an RMS_norm output) for uv_conv3d_f16x3, under the per-tensor power-of-two scale the
        device finds (uv_vae_split_f16: 1 unless max |x| >= 2^15). Returns (split tensor, [1 / s, work] device scalars)."""
        P, C = x.numel() // x.shape[-1], x.shape[-1]
        xs = torch.empty_like(x)
        sc = torch.empty(2, dtype=torch.float32, device=self.dev)
        _lib.call("uv_vae_split_f16", _lib.ptr(x), x.stride(-2), _lib.ptr(xs), xs.stride(-2), P, C, _lib.ptr(sc), _lib.stream_ptr())
        return xs, sc

    def _pointwise(self, op, x, resid=None):
        """1x1(x1) convolution = fp32 GEMM over pixel rows."""
        P = x.numel() // x.shape[-1]
        out = torch.empty(*x.shape[:-1], op.cout, dtype=torch.float32, device=self.dev)
        _lib.call("uv_gemm_f32_nt", _lib.ptr(x), x.stride(-2), _lib.ptr(op.w2d), op.w2d.stride(0), _lib.ptr(op.b), P, op.cout,
                  op.cin, _lib.ptr(out), op.cout, _lib.ptr(resid), 0 if resid is None else resid.stride(-2), _lib.stream_ptr())
        return out

    # -- blocks --
    def causal_conv(self, conv, fill, T, H, W, resid=None, in_split=0):
        """3x3x3 causal conv over [cache(2) | T frames]; `fill(dst)` writes the current frames.
        in_split: the filler writes split activations (1 = bf16 pieces, bf16x3 mode; 2 = fp16 pieces, f16x3 mode; zeros stay zeros,
        so the cache logic is unchanged)."""
        op = self.ops[conv]
        ring, after = self.conv_input(op, H, W, T, fill)
        y = self._conv(op, ring, CACHE_T + T, H, W, T, H, W, t_off=0, ph=1, pw=1, resid=resid, in_split=in_split)
        if after is not None:
            after()
        return y

    def resblock(self, blk, x):
        """ResidualBlock.forward vae2_2.py:214-235."""
        T, H, W, C = x.shape
        res = blk.residual
        if isinstance(blk.shortcut, nn.Identity):
            h = x
        elif self.precision == "f16x3" and C % 32 == 0 and blk.out_dim % 4 == 0 and x.is_contiguous() and H * W >= 16384:     # (per-FRAME size: the choice never depends on the pass length)
            # the 1x1x1 shortcut of the channel-changing blocks (1024 -> 512, 512 -> 256 on 16 large frames: 8 ms each on the exact-f32
            # GEMM) as a three-pass fp16 product too: raw rows, so under the device-found per-tensor scale (uv_vae_split_f16)
            xs, sc = self._split16(x)
            h = self._conv(self.ops[blk.shortcut], xs, T, H, W, T, H, W, in_split=2, act_scale=sc)
        else:
            h = self._pointwise(self.ops[blk.shortcut], x)
        sp1, sp2 = self.split_fmt(res[0], C, blk.out_dim), self.split_fmt(res[3], blk.out_dim, blk.out_dim)
        y = self.causal_conv(res[2], lambda dst: self._rms_silu(x, res[0].gamma, dst, split=sp1), T, H, W, in_split=sp1)
        return self.causal_conv(res[6], lambda dst: self._rms_silu(y, res[3].gamma, dst, split=sp2), T, H, W, resid=h, in_split=sp2)

    def attention(self, blk, x):
        """AttentionBlock.forward vae2_2.py:255-277: per-frame single-head attention, head_dim = C, fp32."""
        T, H, W, C = x.shape
        n = H * W
        n4 = (n + 3) // 4 * 4
        out = torch.empty_like(x)
        xn = torch.zeros(n4, C, dtype=torch.float32, device=self.dev)            # rows >= n stay zero (GEMM N/K padding)
        qkv_op, proj_op = self.ops[blk.to_qkv], self.ops[blk.proj]
        sp = _lib.stream_ptr
        for t in range(T):
            xt = x[t].reshape(n, C)
            self._rms_silu(xt, blk.norm.gamma, xn[:n], silu=False)
            qkv = self._pointwise(qkv_op, xn)                                   # [n4, 3C]
            q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
            s = torch.empty(n, n4, dtype=torch.float32, device=self.dev)
            _lib.call("uv_gemm_f32_nt", _lib.ptr(q), q.stride(0), _lib.ptr(k), k.stride(0), None, n, n4, C, _lib.ptr(s), n4, None, 0, sp())
            _lib.call("uv_softmax_rows_f32", _lib.ptr(s), n4, n, n, 1.0 / math.sqrt(C), sp())
            if n4 > n:
                s[:, n:].zero_()
            vt = torch.zeros(C, n4, dtype=torch.float32, device=self.dev)
            vt[:, :n] = v[:n].t()
            o = torch.empty(n, C, dtype=torch.float32, device=self.dev)
            _lib.call("uv_gemm_f32_nt", _lib.ptr(s), n4, _lib.ptr(vt), n4, None, n, C, n4, _lib.ptr(o), C, None, 0, sp())
            ot = out[t].reshape(n, C)
            _lib.call("uv_gemm_f32_nt", _lib.ptr(o), C, _lib.ptr(proj_op.w2d), C, _lib.ptr(proj_op.b), n, C, C, _lib.ptr(ot), C,
                      _lib.ptr(xt), C, sp())
        return out

    def upsample(self, rs, x, first_chunk):
        """Resample.forward (upsample2d / upsample3d) vae2_2.py:112-155."""
        T, H, W, C = x.shape
        if rs.mode == "upsample3d" and not first_chunk:
            op = self.ops[rs.time_conv]
            ring, after = self.conv_input(op, H, W, T, lambda dst: dst.copy_(x))
            if self.precision == "f16x3" and C % 32 == 0 and ring.is_contiguous():
                # raw residual-stream rows: the whole [cache | pass] input as fp16 pieces under ONE device-found scale (the cache frames
                # stay f32 in the ring, so passes with different scales never mix), three fp16 passes instead of bf16x6's six
                r16, sc = self._split16(ring)
                x = self._conv(op, r16, CACHE_T + T, H, W, T, H, W, t_off=0, interleave=1, in_split=2, act_scale=sc)
            else:
                x = self._conv(op, ring, CACHE_T + T, H, W, T, H, W, t_off=0, interleave=1)
            if after is not None:
                after()
            T = 2 * T
        # first chunk of upsample3d: the "Rep" sentinel - no time conv, cache stays at the zero padding
        op = self.ops[rs.resample[1]]
        if self.precision == "fp32" or not self.phase_upsample:
            return self._conv(op, x, T, H, W, T, 2 * H, 2 * W, ph=1, pw=1, up=1)
        # the f32-grade fast modes: four 2x2 phase convolutions of the source image instead of a 3x3 convolution of the upsampled one
        # (2.25 x fewer multiply-adds; _ConvOp.phases). precision='fp32' keeps the reference's 36 separate products.
        out = torch.empty(T, 2 * H, 2 * W, op.cout, dtype=torch.float32, device=self.dev)
        fmt, sc = 0, None
        if self.precision == "f16x3" and C % 32 == 0 and x.is_contiguous():
            x, sc = self._split16(x)        # raw residual-stream rows: fp16 pieces under a device-found per-tensor scale
            fmt = 2
        for k, ph in enumerate(op.phases()):
            a, b = k >> 1, k & 1
            self._conv(ph, x, T, H, W, T, H, W, ph=1 - a, pw=1 - b, up=2 + k, out=out, ldo=op.cout, in_split=fmt, act_scale=sc)
        return out

    def downsample(self, rs, x, first_chunk):
        """Resample.forward (downsample2d / downsample3d) vae2_2.py:153-169."""
        T, H, W, C = x.shape
        if self.precision == "f16x3" and C % 32 == 0 and x.is_contiguous():
            xs, sc = self._split16(x)
            y = self._conv(self.ops[rs.resample[1]], xs, T, H, W, T, H // 2, W // 2, sh=2, sw=2, in_split=2, act_scale=sc)
        else:
            y = self._conv(self.ops[rs.resample[1]], x, T, H, W, T, H // 2, W // 2, sh=2, sw=2)
        if rs.mode == "downsample3d":
            op = self.ops[rs.time_conv]
            if first_chunk:
                op.cache(H // 2, W // 2, 1)[0].copy_(y[-1])       # feat_cache[idx] = x.clone(); x passes through
            else:
                ring, after = self.conv_input(op, H // 2, W // 2, T, lambda dst: dst.copy_(y), prefix=1)   # cache <- y[-1]
                tout = (1 + T - 3) // 2 + 1
                y = self._conv(op, ring, 1 + T, H // 2, W // 2, tout, H // 2, W // 2, st=2, t_off=0)
                if after is not None:
                    after()
        return y

    # -- encoder / decoder bodies --
    def encoder_chunk(self, vid, f0, T, first_chunk):
        """Frames [f0, f0 + T) of the clip through the encoder (T = 1 for the first chunk, a multiple of 4 afterwards)."""
        enc = self.m.encoder
        F, Hv, Wv = vid.shape[1:]
        H, W = Hv // 2, Wv // 2

        def fill(dst):
            _lib.call("uv_vae_video_in", _lib.ptr(vid), _lib.ptr(dst), dst.stride(-2), F, Hv, Wv, f0, T, _lib.stream_ptr())

        x = self.causal_conv(enc.conv1, fill, T, H, W)
        for stage in enc.downsamples:
            x_copy = x
            for mod in stage.downsamples:
                x = self.resblock(mod, x) if isinstance(mod, ResidualBlock) else self.downsample(mod, x, first_chunk)
            Tc, Hc, Wc, Cc = x_copy.shape
            _lib.call("uv_vae_avgdown_add", _lib.ptr(x_copy), _lib.ptr(x), Tc, Hc, Wc, Cc, stage.out_dim, stage.factor_t,
                      stage.factor_s, _lib.stream_ptr())
        x = self.resblock(enc.middle[0], x)
        x = self.attention(enc.middle[1], x)
        x = self.resblock(enc.middle[2], x)
        T2, H2, W2, _ = x.shape
        sp = self.split_fmt(enc.head[0], x.shape[-1], 32)
        return self.causal_conv(enc.head[2], lambda dst: self._rms_silu(x, enc.head[0].gamma, dst, split=sp), T2, H2, W2, in_split=sp)

    def decoder_chunk(self, xin, first_chunk):
        """xin: [T0, h, w, z] rows of conv2's output for T0 latent frames (T0 = 1 for the first chunk) -> [T, 8h, 8w, 12]."""
        dec = self.m.decoder
        T0, H, W, _ = xin.shape
        x = self.causal_conv(dec.conv1, lambda dst: dst[..., :xin.shape[-1]].copy_(xin), T0, H, W)
        x = self.resblock(dec.middle[0], x)
        x = self.attention(dec.middle[1], x)
        x = self.resblock(dec.middle[2], x)
        for stage in dec.upsamples:
            xm = x
            for mod in stage.upsamples:
                xm = self.resblock(mod, xm) if isinstance(mod, ResidualBlock) else self.upsample(mod, xm, first_chunk)
            if stage.up_flag:
                T, Hh, Ww, C = x.shape
                ft = stage.factor_t
                _lib.call("uv_vae_dupup_add", _lib.ptr(x), _lib.ptr(xm), T, Hh, Ww, C, stage.out_dim, ft,
                          (ft - 1) if first_chunk else 0, _lib.stream_ptr())
            x = xm
        T, Hh, Ww, _ = x.shape
        sp = self.split_fmt(dec.head[0], x.shape[-1], 32)
        return self.causal_conv(dec.head[2], lambda dst: self._rms_silu(x, dec.head[0].gamma, dst, split=sp), T, Hh, Ww, in_split=sp)


class WanVAE_(nn.Module):
    def __init__(self, dim=160, dec_dim=256, z_dim=16, dim_mult=(1, 2, 4, 4), num_res_blocks=2, attn_scales=(),
                 temperal_downsample=(True, True, False), dropout=0.0):
        super().__init__()
        self.dim, self.z_dim, self.dim_mult = dim, z_dim, list(dim_mult)
        self.num_res_blocks = num_res_blocks
        self.temperal_downsample = list(temperal_downsample)
        self.temperal_upsample = self.temperal_downsample[::-1]
        self.encoder = Encoder3d(dim, z_dim * 2, dim_mult, num_res_blocks, attn_scales, self.temperal_downsample, dropout)
        self.conv1 = CausalConv3d(z_dim * 2, z_dim * 2, 1)
        self.conv2 = CausalConv3d(z_dim, z_dim, 1)
        self.decoder = Decoder3d(dec_dim, z_dim, dim_mult, num_res_blocks, attn_scales, self.temperal_upsample, dropout)
        self._engine = None
        self._pool = None             # workspace arena (torch.cuda.MemPool) of the engine: see _arena
        self.use_arena = True
        self.precision = "f16x3"      # default arithmetic of the convolutions (prepare()); 'fp32' = the reference's dtype, exact f32 MFMA
        # Latent frames per decoder pass / 4-frame chunks per encoder pass after the first chunk. The reference streams ONE at a
        # time (vae2_2.py:797-806, 824-835) to bound memory; every layer is time-causal with a 2-frame cache, so longer passes
        # compute the same values (bit-identical here: tested) with fewer, larger launches (grid quantisation on 256 CUs, launch
        # count). 4 keeps the largest decoder tensors of a 720p clip at 4-8 GB; 1 reproduces the reference's streaming granularity.
        self.frames_per_pass = 4
        self.register_load_state_dict_post_hook(lambda m, _k: m.invalidate())

    def invalidate(self):
        self._engine = None
        self._pool = None

    def _arena(self):
        """Workspace arena: everything an encode / decode allocates besides its result - the activations of every layer, the
        convolutions' cached frames and input rings, the split weights - comes from a memory pool this VAE owns (torch's caching
        allocator, routed to a private pool for the length of the call). The first call at a (clip shape, precision, frames_per_pass)
        fills the pool with device allocations; every later call finds its blocks there: no hipMalloc / hipFree inside a decode, whatever
        the rest of the process (the DiT's loop, an empty_cache() elsewhere) did to the shared pool in between - a decode's wall time is its
        kernel time (round-4 verdict: 1.73 -> 3.65 s by allocator state). Memory plumbing only; dropped by invalidate() and by the
        out-of-memory retry."""
        if not self.use_arena or not hasattr(torch.cuda, "MemPool"):
            return contextlib.nullcontext()
        if self._pool is None:
            self._pool = torch.cuda.MemPool()
        return torch.cuda.use_mem_pool(self._pool, device=next(self.parameters()).device)

    def prepare(self, precision=None):
        """precision: 'fp32' = exact f32 MFMA (the reference's dtype) | 'bf16x6' = the same f32 operands, products on the
        bf16 matrix pipe by exact three-way operand splitting (6 passes; as close to an fp64 convolution as the f32 MFMA kernel,
        1.45 x faster) | 'f16x3' (default) = f32-grade too, in 3 passes: the convolutions behind an RMS_norm (ResidualBlocks, heads: ~90 % of
        the FLOPs) take both operands as two IEEE fp16 pieces (22 significant bits; their error against fp64 equals the f32 MFMA's,
        which is accumulation-bound), the others run as bf16x6 | 'bf16x3' = two-way bf16 split, 3 passes (~1e-5 relative error).
        The 1x1 convolutions, norms and the per-frame attention stay on the f32 kernels in every mode."""
        if next(self.parameters()).device.type != "cuda":
            raise _lib.UnividHipError("WanVAE_.prepare: parameters must be on the GPU - there is no CPU path in univid_amd")
        _lib.init()
        if precision is not None:
            self.precision = precision
        self._engine = _Engine(self, self.precision)
        return self

    def _eng(self):
        if self._engine is None:
            self.prepare()
        return self._engine

    def clear_cache(self):
        if self._engine is not None:
            self._engine.reset()

    def _with_pass_length(self, body, *args):
        """Runs encode / decode at `frames_per_pass`; if the device runs out of memory, again with half the pass length (down to the
        reference's 1): the result does not depend on it, only the size of the intermediate tensors does."""
        G = max(1, int(self.frames_per_pass))
        while True:
            try:
                return body(G, *args)
            except torch.cuda.OutOfMemoryError:
                if G == 1:
                    raise
                retry = G // 2
            # outside the handler: the exception's traceback (which holds the failed pass's frame - its engine, rows and intermediates, all
            # arena blocks) is gone, so dropping the engine and the pool really frees the arena before the shorter pass allocates
            logging.warning(f"WanVAE_: out of memory at {G} frames per pass, retrying with {retry}")
            G = retry
            self._engine = None              # (its cached frames and rings live in the arena)
            self._pool = None
            torch.cuda.empty_cache()

    def encode(self, x, scale):
        """WanVAE_.encode vae2_2.py:783-810: x [1, 3, F, H, W] fp32 -> [1, z, (F-1)//4+1, H/16, W/16]."""
        return self._with_pass_length(self._encode, x, scale)

    def _encode(self, G, x, scale):
        vid = x[0].contiguous().float()
        F = vid.shape[1]
        n4 = (F - 1) // 4                                                         # 4-frame chunks after the first frame
        mu = torch.empty(1, self.z_dim, 1 + n4, vid.shape[2] // 16, vid.shape[3] // 16, dtype=torch.float32, device=vid.device)   # the result: the caller's
        with self._arena():
            eng = self._eng()
            eng.reset()
            outs = [eng.encoder_chunk(vid, 0, 1, first_chunk=True)]
            for c0 in range(0, n4, G):
                g = min(G, n4 - c0)
                outs.append(eng.encoder_chunk(vid, 1 + 4 * c0, 4 * g, first_chunk=False))
            out = torch.cat(outs, 0)                                              # [f, h, w, 2z]
            y = eng._pointwise(eng.ops[self.conv1], out)                          # 1x1x1, then chunk(2) -> mu
            f, h, w, _ = y.shape
            if (f, h, w) != tuple(mu.shape[2:]):      # uv_vae_latent_out writes f*h*w*z_dim floats through a raw pointer
                raise ValueError(f"encoder produced {(f, h, w)} latent positions for a result sized {tuple(mu.shape[2:])} "
                                 f"(input {tuple(vid.shape)}: H and W must be multiples of 16)")
            _lib.call("uv_vae_latent_out", _lib.ptr(y), y.stride(-2), _lib.ptr(scale[0]), _lib.ptr(scale[1]), _lib.ptr(mu), self.z_dim,
                      f * h * w, _lib.stream_ptr())
            eng.reset()
            del outs, out, y
        return mu

    def decode(self, z, scale, clamp=True):
        """WanVAE_.decode vae2_2.py:812-839 (+ the wrapper's clamp :1045): z [1, z, f, h, w] -> [1, 3, 4(f-1)+1, 16h, 16w]."""
        return self._with_pass_length(self._decode, z, scale)

    def _decode(self, G, z, scale):
        zz = z[0].contiguous().float()
        Z, f, h, w = zz.shape
        F = 4 * (f - 1) + 1
        vid = torch.empty(1, 3, F, 16 * h, 16 * w, dtype=torch.float32, device=zz.device)      # the result: the caller's, not the arena's
        with self._arena():
            eng = self._eng()
            eng.reset()
            rows = torch.empty(f, h, w, Z, dtype=torch.float32, device=zz.device)
            _lib.call("uv_vae_latent_in", _lib.ptr(zz), _lib.ptr(scale[0]), _lib.ptr(scale[1]), _lib.ptr(rows), Z, Z, f * h * w,
                      _lib.stream_ptr())
            x = eng._pointwise(eng.ops[self.conv2], rows)                         # conv2 (1x1x1) on all frames
            f0 = 0
            for i0, i1 in [(0, 1)] + [(i, min(i + G, f)) for i in range(1, f, G)]:
                y = eng.decoder_chunk(x[i0:i1], first_chunk=(i0 == 0))           # [T, 8h, 8w, 12]
                T = y.shape[0]
                _lib.call("uv_vae_video_out", _lib.ptr(y), y.stride(-2), _lib.ptr(vid), F, 8 * h, 8 * w, f0, T, _lib.stream_ptr())
                f0 += T
            assert f0 == F
            eng.reset()
            del rows, x, y
        return vid

    def init_weights(self, seed=0):
        from .. import detinit
        detinit.init_module_(self, seed)
        self.invalidate()
        return self


class Wan2_2_VAE:
    """List-API wrapper (vae2_2.py:888-1051). `vae_pth=None` builds a randomly initialised model (offline use);
    a path loads the reference checkpoint (same state-dict keys)."""

    def __init__(self, z_dim=48, c_dim=160, vae_pth=None, dim_mult=(1, 2, 4, 4), temperal_downsample=(False, True, True),
                 dtype=torch.float, device="cuda", dec_dim=256, seed=0, precision="f16x3", frames_per_pass=None):
        """precision (extension; WanVAE_.prepare): 'f16x3' - the DEFAULT since round 4: f32-grade arithmetic (measured as close to an
        fp64 convolution as the exact f32 MFMA, every element of a full 49 x 720 x 1280 clip inside rtol 1e-3 / atol 1e-4 of the fp32 CPU
        oracle) at a third of the exact mode's time; 'fp32' = the reference's dtype executed literally on the f32 MFMA; 'bf16x6';
        'bf16x3'. Tensors in and out are fp32 in every mode (vae2_2.py:897)."""
        self.dtype = dtype
        self.device = torch.device(device)
        mean = torch.tensor(_MEAN, dtype=dtype, device=device)
        std = torch.tensor(_STD, dtype=dtype, device=device)
        self.scale = [mean, 1.0 / std]
        with torch.device(device):
            model = WanVAE_(dim=c_dim, dec_dim=dec_dim, z_dim=z_dim, dim_mult=dim_mult, num_res_blocks=2, attn_scales=[],
                            temperal_downsample=temperal_downsample, dropout=0.0)
        if vae_pth is not None:
            logging.info(f"loading {vae_pth}")
            model.load_state_dict(torch.load(vae_pth, map_location=device))
        else:
            model.init_weights(seed)
        self.model = model.eval().requires_grad_(False).to(device)
        self.model.precision = precision
        if frames_per_pass is not None:      # latent frames per decoder pass (WanVAE_.frames_per_pass; 1 = the reference's streaming)
            self.model.frames_per_pass = int(frames_per_pass)

    def encode(self, videos):
        """List in, list out; a non-list argument is logged and answered with None, as the reference does (vae2_2.py:1024-1036:
        its `except TypeError` turns the check into a logged None)."""
        if not isinstance(videos, list):
            logging.warning("Wan2_2_VAE.encode: videos should be a list")
            return None
        return [self.model.encode(u.to(self.device).unsqueeze(0), self.scale).float().squeeze(0) for u in videos]

    def decode(self, zs):
        """vae2_2.py:1038-1051; non-list -> logged None (see encode)."""
        if not isinstance(zs, list):
            logging.warning("Wan2_2_VAE.decode: zs should be a list")
            return None
        return [self.model.decode(u.to(self.device).unsqueeze(0), self.scale).float().squeeze(0) for u in zs]
