"""TEST INFRASTRUCTURE ONLY (the checker; `univid_amd` must not import this).

CPU restatement of the Wan2.2 3D causal VAE (fp32) as UniVid calls it: Wan2_2_VAE.encode / .decode
(/root/reference/models/wan/utils/modules/vae2_2.py:1024-1051) -> WanVAE_.encode :783-810 / .decode :812-839,
including the chunked streaming with the per-convolution feature cache (CACHE_T = 2 frames), the first-chunk
special cases ("Rep" sentinel :117-119,142-151; DupUp3D(first_chunk) :410-411) and the per-channel latent
normalisation (:904-1012). Functional over a state dict with the reference's key names.

Pinned against the reference's own module (bit-exact on CPU) by oracle/gen_golden.py.
"""
import torch
import torch.nn.functional as F

CACHE_T = 2

# vae2_2.py:904-1012
MEAN = [-0.2289, -0.0052, -0.1323, -0.2339, -0.2799, 0.0174, 0.1838, 0.1557, -0.1382, 0.0542, 0.2813, 0.0891, 0.1570,
        -0.0098, 0.0375, -0.1825, -0.2246, -0.1207, -0.0698, 0.5109, 0.2665, -0.2108, -0.2158, 0.2502, -0.2055, -0.0322,
        0.1109, 0.1567, -0.0729, 0.0899, -0.2799, -0.1230, -0.0313, -0.1649, 0.0117, 0.0723, -0.2839, -0.2083, -0.0520,
        0.3748, 0.0152, 0.1957, 0.1433, -0.2944, 0.3573, -0.0548, -0.1681, -0.0667]
STD = [0.4765, 1.0364, 0.4514, 1.1677, 0.5313, 0.4990, 0.4818, 0.5013, 0.8158, 1.0344, 0.5894, 1.0901, 0.6885, 0.6165,
       0.8454, 0.4978, 0.5759, 0.3523, 0.7135, 0.6804, 0.5833, 1.4146, 0.8986, 0.5659, 0.7069, 0.5338, 0.4889, 0.4917,
       0.4069, 0.4999, 0.6866, 0.4093, 0.5709, 0.6065, 0.6415, 0.4944, 0.5726, 1.2042, 0.5458, 1.6887, 0.3971, 1.0600,
       0.3943, 0.5537, 0.5444, 0.4089, 0.7468, 0.7744]

FULL_CFG = dict(dim=160, dec_dim=256, z_dim=48, dim_mult=[1, 2, 4, 4], num_res_blocks=2,
                temperal_downsample=[False, True, True])                      # vae2_2.py:890-899, 1015-1022
SMALL_CFG = dict(dim=32, dec_dim=32, z_dim=48, dim_mult=[1, 2, 4, 4], num_res_blocks=2,
                 temperal_downsample=[False, True, True])                     # width-reduced, same topology (tests)


def scale_tensors(dtype=torch.float32):
    mean = torch.tensor(MEAN, dtype=dtype)
    std = torch.tensor(STD, dtype=dtype)
    return [mean, 1.0 / std]                                                  # vae2_2.py:1012


def causal_conv3d(x, w, b, stride=(1, 1, 1), pad=(0, 0, 0), cache_x=None):
    """CausalConv3d.forward :34-42: 2*pad_t frames on the left (cache frames first), none on the right."""
    padding = [pad[2], pad[2], pad[1], pad[1], 2 * pad[0], 0]
    if cache_x is not None and padding[4] > 0:
        x = torch.cat([cache_x, x], dim=2)
        padding[4] -= cache_x.shape[2]
    return F.conv3d(F.pad(x, padding), w, b, stride=stride)


def rms_norm(x, gamma):
    """RMS_norm :57-59 (channel_first): normalize over C, * sqrt(C) * gamma."""
    return F.normalize(x, dim=1) * (x.shape[1] ** 0.5) * gamma


def _cache_tail(x, prev):
    """last CACHE_T frames of x, topped up with the last frame of the previous cache (:219-229)."""
    c = x[:, :, -CACHE_T:].clone()
    if c.shape[2] < 2 and prev is not None:
        c = torch.cat([prev[:, :, -1:].to(c.device), c], dim=2)
    return c


class WanVAE:
    def __init__(self, sd, cfg):
        self.sd, self.cfg = sd, cfg
        self.z_dim = cfg["z_dim"]

    # ---- building blocks ----------------------------------------------------------------------------------
    def _conv_cached(self, key, x, cache, idx, pad=(1, 1, 1)):
        i = idx[0]
        c = _cache_tail(x, cache[i])
        y = causal_conv3d(x, self.sd[key + ".weight"], self.sd[key + ".bias"], pad=pad, cache_x=cache[i])
        cache[i] = c
        idx[0] += 1
        return y

    def _resblock(self, p, x, cache, idx):
        """ResidualBlock.forward :214-235."""
        sd = self.sd
        h = x
        if (p + "shortcut.weight") in sd:
            h = causal_conv3d(x, sd[p + "shortcut.weight"], sd[p + "shortcut.bias"])
        x = F.silu(rms_norm(x, sd[p + "residual.0.gamma"]))
        x = self._conv_cached(p + "residual.2", x, cache, idx)
        x = F.silu(rms_norm(x, sd[p + "residual.3.gamma"]))
        x = self._conv_cached(p + "residual.6", x, cache, idx)
        return x + h

    def _attn(self, p, x):
        """AttentionBlock.forward :255-277: per-frame single-head attention over h*w tokens."""
        sd = self.sd
        b, c, t, h, w = x.shape
        y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
        y = F.normalize(y, dim=1) * (c ** 0.5) * sd[p + "norm.gamma"]
        qkv = F.conv2d(y, sd[p + "to_qkv.weight"], sd[p + "to_qkv.bias"])
        q, k, v = qkv.reshape(b * t, 1, c * 3, -1).permute(0, 1, 3, 2).contiguous().chunk(3, dim=-1)
        y = F.scaled_dot_product_attention(q, k, v)
        y = y.squeeze(1).permute(0, 2, 1).reshape(b * t, c, h, w)
        y = F.conv2d(y, sd[p + "proj.weight"], sd[p + "proj.bias"])
        y = y.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4)
        return y + x

    def _resample(self, p, x, mode, cache, idx):
        """Resample.forward :112-169."""
        sd = self.sd
        b, c, t, h, w = x.shape
        if mode == "upsample3d":
            i = idx[0]
            if cache[i] is None:
                cache[i] = "Rep"                                                       # first chunk: no time conv
                idx[0] += 1
            else:
                cx = x[:, :, -CACHE_T:].clone()
                if cx.shape[2] < 2:
                    if isinstance(cache[i], str):                                      # "Rep": pad with zeros
                        cx = torch.cat([torch.zeros_like(cx), cx], dim=2)
                    else:
                        cx = torch.cat([cache[i][:, :, -1:], cx], dim=2)
                tw, tb = sd[p + "time_conv.weight"], sd[p + "time_conv.bias"]
                if isinstance(cache[i], str):
                    x = causal_conv3d(x, tw, tb, pad=(1, 0, 0))
                else:
                    x = causal_conv3d(x, tw, tb, pad=(1, 0, 0), cache_x=cache[i])
                cache[i] = cx
                idx[0] += 1
                x = x.reshape(b, 2, c, t, h, w)
                x = torch.stack((x[:, 0], x[:, 1]), 3).reshape(b, c, t * 2, h, w)     # interleave the two halves
        t = x.shape[2]
        y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
        if mode in ("upsample2d", "upsample3d"):
            y = F.interpolate(y.float(), scale_factor=(2.0, 2.0), mode="nearest-exact").type_as(y)
            y = F.conv2d(y, sd[p + "resample.1.weight"], sd[p + "resample.1.bias"], padding=1)
        else:
            y = F.conv2d(F.pad(y, (0, 1, 0, 1)), sd[p + "resample.1.weight"], sd[p + "resample.1.bias"], stride=(2, 2))
        x = y.reshape(b, t, c, y.shape[2], y.shape[3]).permute(0, 2, 1, 3, 4)
        if mode == "downsample3d":
            i = idx[0]
            if cache[i] is None:
                cache[i] = x.clone()
                idx[0] += 1
            else:
                cx = x[:, :, -1:].clone()
                x = causal_conv3d(torch.cat([cache[i][:, :, -1:], x], 2), sd[p + "time_conv.weight"],
                                  sd[p + "time_conv.bias"], stride=(2, 1, 1))
                cache[i] = cx
                idx[0] += 1
        return x

    @staticmethod
    def avg_down3d(x, out_channels, factor_t, factor_s):
        """AvgDown3D.forward :335-367."""
        pad_t = (factor_t - x.shape[2] % factor_t) % factor_t
        x = F.pad(x, (0, 0, 0, 0, pad_t, 0))
        B, C, T, H, W = x.shape
        factor = factor_t * factor_s * factor_s
        x = x.view(B, C, T // factor_t, factor_t, H // factor_s, factor_s, W // factor_s, factor_s)
        x = x.permute(0, 1, 3, 5, 7, 2, 4, 6).contiguous()
        x = x.view(B, C * factor, T // factor_t, H // factor_s, W // factor_s)
        x = x.view(B, out_channels, C * factor // out_channels, T // factor_t, H // factor_s, W // factor_s)
        return x.mean(dim=2)

    @staticmethod
    def dup_up3d(x, out_channels, factor_t, factor_s, first_chunk):
        """DupUp3D.forward :390-412."""
        factor = factor_t * factor_s * factor_s
        x = x.repeat_interleave(out_channels * factor // x.shape[1], dim=1)
        x = x.view(x.size(0), out_channels, factor_t, factor_s, factor_s, x.size(2), x.size(3), x.size(4))
        x = x.permute(0, 1, 5, 2, 6, 3, 7, 4).contiguous()
        x = x.view(x.size(0), out_channels, x.size(2) * factor_t, x.size(4) * factor_s, x.size(6) * factor_s)
        if first_chunk:
            x = x[:, :, factor_t - 1:]
        return x

    # ---- encoder / decoder bodies -------------------------------------------------------------------------
    def _encoder(self, x, cache, idx):
        """Encoder3d.forward :559-613."""
        cfg, sd = self.cfg, self.sd
        dims = [cfg["dim"] * u for u in [1] + cfg["dim_mult"]]
        x = self._conv_cached("encoder.conv1", x, cache, idx)
        nstage = len(cfg["dim_mult"])
        for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
            t_down = cfg["temperal_downsample"][i] if i < len(cfg["temperal_downsample"]) else False
            down = i != nstage - 1
            p = f"encoder.downsamples.{i}."
            x_copy = x.clone()
            for j in range(cfg["num_res_blocks"]):
                x = self._resblock(p + f"downsamples.{j}.", x, cache, idx)
            if down:
                x = self._resample(p + f"downsamples.{cfg['num_res_blocks']}.", x,
                                   "downsample3d" if t_down else "downsample2d", cache, idx)
            x = x + self.avg_down3d(x_copy, cout, 2 if t_down else 1, 2 if down else 1)       # Down_ResidualBlock :447-452
        x = self._resblock("encoder.middle.0.", x, cache, idx)
        x = self._attn("encoder.middle.1.", x)
        x = self._resblock("encoder.middle.2.", x, cache, idx)
        x = F.silu(rms_norm(x, sd["encoder.head.0.gamma"]))
        return self._conv_cached("encoder.head.2", x, cache, idx)

    def _decoder(self, x, cache, idx, first_chunk):
        """Decoder3d.forward :672-723."""
        cfg, sd = self.cfg, self.sd
        dm = cfg["dim_mult"]
        dims = [cfg["dec_dim"] * u for u in [dm[-1]] + dm[::-1]]
        t_up_flags = cfg["temperal_downsample"][::-1]
        x = self._conv_cached("decoder.conv1", x, cache, idx)
        x = self._resblock("decoder.middle.0.", x, cache, idx)
        x = self._attn("decoder.middle.1.", x)
        x = self._resblock("decoder.middle.2.", x, cache, idx)
        for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
            t_up = t_up_flags[i] if i < len(t_up_flags) else False
            up = i != len(dm) - 1
            p = f"decoder.upsamples.{i}."
            xm = x.clone()
            for j in range(cfg["num_res_blocks"] + 1):
                xm = self._resblock(p + f"upsamples.{j}.", xm, cache, idx)
            if up:
                xm = self._resample(p + f"upsamples.{cfg['num_res_blocks'] + 1}.", xm,
                                    "upsample3d" if t_up else "upsample2d", cache, idx)
                x = xm + self.dup_up3d(x, cout, 2 if t_up else 1, 2, first_chunk)             # Up_ResidualBlock :489-497
            else:
                x = xm
        x = F.silu(rms_norm(x, sd["decoder.head.0.gamma"]))
        return self._conv_cached("decoder.head.2", x, cache, idx)

    def _count_convs(self, prefix):
        """count_conv3d :726-731: CausalConv3d modules = 5-D weights under the prefix."""
        return sum(1 for k, v in self.sd.items() if k.startswith(prefix) and k.endswith(".weight") and v.dim() == 5)

    # ---- public API (batch of one, like Wan2_2_VAE) -------------------------------------------------------
    def encode(self, x, scale):
        """WanVAE_.encode :783-810. x [1, 3, F, H, W] -> mu [1, z, (F-1)/4+1, H/16, W/16]."""
        b, c, f, h, w = x.shape
        x = x.view(b, c, f, h // 2, 2, w // 2, 2).permute(0, 1, 6, 4, 2, 3, 5).reshape(b, c * 4, f, h // 2, w // 2)  # patchify :285-292
        cache = [None] * self._count_convs("encoder.")
        outs = []
        for i in range(1 + (f - 1) // 4):
            idx = [0]
            chunk = x[:, :, :1] if i == 0 else x[:, :, 1 + 4 * (i - 1):1 + 4 * i]
            outs.append(self._encoder(chunk, cache, idx))
        out = torch.cat(outs, 2)
        mu, _ = causal_conv3d(out, self.sd["conv1.weight"], self.sd["conv1.bias"]).chunk(2, dim=1)
        return (mu - scale[0].view(1, self.z_dim, 1, 1, 1)) * scale[1].view(1, self.z_dim, 1, 1, 1)

    def decode(self, z, scale):
        """WanVAE_.decode :812-839. z [1, z, f, h, w] -> [1, 3, 4(f-1)+1, 16h, 16w] (not clamped)."""
        z = z / scale[1].view(1, self.z_dim, 1, 1, 1) + scale[0].view(1, self.z_dim, 1, 1, 1)
        x = causal_conv3d(z, self.sd["conv2.weight"], self.sd["conv2.bias"])
        cache = [None] * self._count_convs("decoder.")
        outs = []
        for i in range(z.shape[2]):
            idx = [0]
            outs.append(self._decoder(x[:, :, i:i + 1], cache, idx, first_chunk=(i == 0)))
        out = torch.cat(outs, 2)
        b, c, f, h, w = out.shape                                                        # unpatchify :306-312
        out = out.view(b, c // 4, 2, 2, f, h, w).permute(0, 1, 4, 5, 3, 6, 2).reshape(b, c // 4, f, h * 2, w * 2)
        return out


def vae_encode(vae, videos, scale=None):
    """Wan2_2_VAE.encode :1024-1036: list of [3, F, H, W] -> list of [z, f, h, w] fp32."""
    scale = scale or scale_tensors()
    return [vae.encode(u.unsqueeze(0), scale).float().squeeze(0) for u in videos]


def vae_decode(vae, zs, scale=None):
    """Wan2_2_VAE.decode :1038-1051: list of latents -> list of [3, F, H, W] fp32 clamped to [-1, 1]."""
    scale = scale or scale_tensors()
    return [vae.decode(u.unsqueeze(0), scale).float().clamp_(-1, 1).squeeze(0) for u in zs]


def state_dict_shapes(cfg):
    """Key -> shape of a WanVAE_ state dict (vae2_2.py:734-776)."""
    s = {}
    dim, dec, z, dm, nres = cfg["dim"], cfg["dec_dim"], cfg["z_dim"], cfg["dim_mult"], cfg["num_res_blocks"]

    def conv3(k, co, ci, kt=3, ks=3):
        s[k + ".weight"] = (co, ci, kt, ks, ks)
        s[k + ".bias"] = (co,)

    def res(p, ci, co):
        s[p + "residual.0.gamma"] = (ci, 1, 1, 1)
        conv3(p + "residual.2", co, ci)
        s[p + "residual.3.gamma"] = (co, 1, 1, 1)
        conv3(p + "residual.6", co, co)
        if ci != co:
            conv3(p + "shortcut", co, ci, 1, 1)

    def attn(p, c):
        s[p + "norm.gamma"] = (c, 1, 1)
        s[p + "to_qkv.weight"] = (3 * c, c, 1, 1)
        s[p + "to_qkv.bias"] = (3 * c,)
        s[p + "proj.weight"] = (c, c, 1, 1)
        s[p + "proj.bias"] = (c,)

    dims = [dim * u for u in [1] + dm]
    conv3("encoder.conv1", dims[0], 12)
    td = cfg["temperal_downsample"]
    for i, (ci, co) in enumerate(zip(dims[:-1], dims[1:])):
        p = f"encoder.downsamples.{i}.downsamples."
        c = ci
        for j in range(nres):
            res(p + f"{j}.", c, co)
            c = co
        if i != len(dm) - 1:
            s[p + f"{nres}.resample.1.weight"] = (co, co, 3, 3)
            s[p + f"{nres}.resample.1.bias"] = (co,)
            if i < len(td) and td[i]:
                conv3(p + f"{nres}.time_conv", co, co, 3, 1)
    co = dims[-1]
    res("encoder.middle.0.", co, co)
    attn("encoder.middle.1.", co)
    res("encoder.middle.2.", co, co)
    s["encoder.head.0.gamma"] = (co, 1, 1, 1)
    conv3("encoder.head.2", 2 * z, co)
    conv3("conv1", 2 * z, 2 * z, 1, 1)
    conv3("conv2", z, z, 1, 1)
    ddims = [dec * u for u in [dm[-1]] + dm[::-1]]
    conv3("decoder.conv1", ddims[0], z)
    res("decoder.middle.0.", ddims[0], ddims[0])
    attn("decoder.middle.1.", ddims[0])
    res("decoder.middle.2.", ddims[0], ddims[0])
    tu = td[::-1]
    for i, (ci, co) in enumerate(zip(ddims[:-1], ddims[1:])):
        p = f"decoder.upsamples.{i}.upsamples."
        c = ci
        for j in range(nres + 1):
            res(p + f"{j}.", c, co)
            c = co
        if i != len(dm) - 1:
            s[p + f"{nres + 1}.resample.1.weight"] = (co, co, 3, 3)
            s[p + f"{nres + 1}.resample.1.bias"] = (co,)
            if i < len(tu) and tu[i]:
                conv3(p + f"{nres + 1}.time_conv", 2 * co, co, 3, 1)
    s["decoder.head.0.gamma"] = (ddims[-1], 1, 1, 1)
    conv3("decoder.head.2", 12, ddims[-1])
    return s


def make_state_dict(cfg, seed=0):
    from univid_amd import detinit
    sd = {k: torch.empty(v, dtype=torch.float32) for k, v in state_dict_shapes(cfg).items()}
    return detinit.init_state_dict_(sd, seed)
