// HBM-bound glue of the Wan2.2 VAE on channels-last fp32 activations ([T, H, W, C], one wave per pixel row).
//
// Replaces (reference, PyTorch eager, fp32):
//   RMS_norm + SiLU                      models/wan/utils/modules/vae2_2.py:45-59, 201-206
//   AvgDown3D / DupUp3D shortcuts        vae2_2.py:316-412
//   softmax of the single-head AttentionBlock (F.scaled_dot_product_attention)   vae2_2.py:255-277
//   patchify / unpatchify, latent (de)normalisation, clamp    vae2_2.py:280-313, 803-808, 814-818, 1045
#include "common.h"

// y = x / max(||x||_2, 1e-12) * sqrt(C) * gamma  [-> SiLU]      (F.normalize(x, dim=C) * scale * gamma)
// split_out: write each value as bf16 hi + bf16 lo in the [C/32][32 hi | 32 lo] layout uv_conv3d_bf16x3(in_split=1) reads
// (same bytes per pixel as f32).
// One wave per RPW consecutive rows (pixels): the loads of all RPW rows are issued before the first reduction, so a wave keeps RPW rows
// in flight - at C = 160 a row is 640 bytes, and one row per wave left the kernel latency-bound at 2.9 TB/s (32 waves x 640 B per CU).
// MAXV = ceil(C / 256) exactly (chosen by the launcher): every load is issued unconditionally from a clamped column (lanes beyond the
// row re-read its last chunk and contribute zero) - guarded loads compile to one basic block and one vmcnt(0) EACH (the C = 1024
// rows of the decoder's first stage ran at 2.3 TB/s behind eight guarded loads).
template <int MAXV, int RPW>
__global__ __launch_bounds__(256) void vae_rms_silu_kernel(const float* in, long ld_in, const float* gamma, float* out,
                                                          long ld_out, long P, int C, int do_silu, int split_out) {
    const int lane = threadIdx.x & 63;
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= P) return;
    const int nv = C >> 2;  // float4 count
    f32x4 v[RPW][MAXV];
    float ss[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const float* xr = in + min(row0 + r, P - 1) * ld_in;      // rows beyond P: a valid row re-read, never stored
#pragma unroll
        for (int i = 0; i < MAXV; ++i) v[r][i] = *(const f32x4*)(xr + min(i * 64 + lane, nv - 1) * 4);
    }
    f32x4 g[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) g[i] = *(const f32x4*)(gamma + min(i * 64 + lane, nv - 1) * 4);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const float q = v[r][i][0] * v[r][i][0] + v[r][i][1] * v[r][i][1] + v[r][i][2] * v[r][i][2] + v[r][i][3] * v[r][i][3];
            a += (i * 64 + lane < nv) ? q : 0.f;
        }
        ss[r] = a;
    }
    const float scale = sqrtf((float)C);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const long row = row0 + r;
        const float denom = fmaxf(sqrtf(wave_sum(ss[r])), 1e-12f);
        if (row >= P) continue;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = i * 64 + lane;
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = __fmul_rn(__fmul_rn(__fdiv_rn(v[r][i][e], denom), scale), g[i][e]);
                if (do_silu) t = silu_f32(t);
                y[e] = t;
            }
            if (c4 < nv) {
                if (split_out == 2) {        // two IEEE fp16 pieces (uv_conv3d_f16x3): hi = fp16(y), lo = fp16(y - hi)
                    const int c = c4 * 4;
                    bf16_t* ob = (bf16_t*)(out + row * ld_out) + (c >> 5) * 64 + (c & 31);
                    _Float16 hi[4], lo[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hi[e] = (_Float16)y[e];
                        lo[e] = (_Float16)(y[e] - (float)hi[e]);
                    }
                    auto pk = [](_Float16 a, _Float16 b) { return (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16); };
                    *(u32x2*)ob = (u32x2){pk(hi[0], hi[1]), pk(hi[2], hi[3])};
                    *(u32x2*)(ob + 32) = (u32x2){pk(lo[0], lo[1]), pk(lo[2], lo[3])};
                } else if (split_out) {
                    const int c = c4 * 4;
                    bf16_t* ob = (bf16_t*)(out + row * ld_out) + (c >> 5) * 64 + (c & 31);
                    float lo[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) lo[e] = y[e] - round_bf(y[e]);
                    *(u32x2*)ob = (u32x2){pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3])};
                    *(u32x2*)(ob + 32) = (u32x2){pack_bf2(lo[0], lo[1]), pack_bf2(lo[2], lo[3])};
                } else {
                    *(f32x4*)(out + row * ld_out + c4 * 4) = y;
                }
            }
        }
    }
}

extern "C" int uv_vae_rms_silu(const float* in, long ld_in, const float* gamma, float* out, long ld_out, long P, int C,
                               int do_silu, int split_out, void* stream) {
    UV_CHECK_ARG(in && gamma && out && P > 0, "uv_vae_rms_silu: bad arguments");
    UV_CHECK_ARG(split_out >= 0 && split_out <= 2, "uv_vae_rms_silu: split_out must be 0 (f32 rows), 1 (bf16 pieces) or 2 (fp16 pieces), got %d", split_out);
    UV_CHECK_ARG(!split_out || C % 32 == 0, "uv_vae_rms_silu: split output needs C %% 32 == 0 (C=%d)", C);
    UV_CHECK_ARG(C % 4 == 0 && C <= 2048 && ld_in % 4 == 0 && ld_out % 4 == 0, "uv_vae_rms_silu: C=%d unsupported", C);
    const dim3 block(256);
    const int maxv = (C + 255) / 256;                    // float4 chunks per lane
#define UV_RMS_LAUNCH(MV, RPW)                                                                                                          \
    hipLaunchKernelGGL((vae_rms_silu_kernel<MV, RPW>), dim3((unsigned)((P + 4 * RPW - 1) / (4 * RPW))), block, 0, (hipStream_t)stream, in, \
                       ld_in, gamma, out, ld_out, P, C, do_silu, split_out)
    switch (maxv) {
        case 1: UV_RMS_LAUNCH(1, 4); break;              // rows of at most 1 KiB: four per wave
        case 2: UV_RMS_LAUNCH(2, 2); break;
        case 3: UV_RMS_LAUNCH(3, 1); break;
        case 4: UV_RMS_LAUNCH(4, 1); break;
        default: UV_RMS_LAUNCH(8, 1); break;
    }
#undef UV_RMS_LAUNCH
    UV_CHECK_LAUNCH("uv_vae_rms_silu");
    return 0;
}

// in-place softmax over rows of x * scale (fp32), one wave per row
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* x, long ld, int R, int n, float scale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    float* xr = x + (long)row * ld;
    float m = -INFINITY;
    for (int i = lane; i < n; i += 64) m = fmaxf(m, xr[i] * scale);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float s = 0.f;
    for (int i = lane; i < n; i += 64) {
        const float e = expf(xr[i] * scale - m);
        xr[i] = e;
        s += e;
    }
    s = wave_sum(s);
    const float inv = 1.0f / s;
    for (int i = lane; i < n; i += 64) xr[i] *= inv;
}

extern "C" int uv_softmax_rows_f32(float* x, long ld, int R, int n, float scale, void* stream) {
    UV_CHECK_ARG(x && R > 0 && n > 0, "uv_softmax_rows_f32: bad arguments");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ld, R, n, scale);
    UV_CHECK_LAUNCH("uv_softmax_rows_f32");
    return 0;
}

// DupUp3D (vae2_2.py:390-412) added onto the main path: out[t',h',w',co] += x[t'/ft, h'/2, w'/2, j / repeats],
// j = ((co*ft + t'%ft)*2 + h'%2)*2 + w'%2, repeats = Cout*ft*4 / Cin; `drop` leading frames are skipped
// (first_chunk: ft-1).  x: [T, H, W, Cin]; out: [T*ft - drop, 2H, 2W, Cout].
// One thread per 4 consecutive output channels (Cout % 4 == 0: a 16-byte read-modify-write of `out`; the four source channels are
// j / repeats for four different j, gathered from a row that stays in cache) - or per channel (VEC = 1).
template <int VEC>
__global__ void dupup_add_kernel(const float* x, float* out, int T, int H, int W, int Cin, int Cout, int ft, int drop) {
    const int To = T * ft - drop, Ho = 2 * H, Wo = 2 * W;
    const int cv = Cout / VEC;
    const long total = (long)To * Ho * Wo * cv;
    const int repeats = Cout * ft * 4 / Cin;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i % cv) * VEC;
        long r = i / cv;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho);
        const int to = (int)(r / Ho) + drop;
        const float* xr = x + (((long)(to / ft) * H + (ho >> 1)) * W + (wo >> 1)) * Cin;
        const int sub = (to % ft) * 4 + (ho & 1) * 2 + (wo & 1);          // j = co * ft * 4 + sub
        if constexpr (VEC == 4) {
            f32x4 o = *(f32x4*)(out + i * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += xr[((co + e) * ft * 4 + sub) / repeats];
            *(f32x4*)(out + i * 4) = o;
        } else {
            out[i] += xr[(co * ft * 4 + sub) / repeats];
        }
    }
}

extern "C" int uv_vae_dupup_add(const float* x, float* out, int T, int H, int W, int Cin, int Cout, int ft, int drop,
                                void* stream) {
    UV_CHECK_ARG(x && out && (Cout * ft * 4) % Cin == 0, "uv_vae_dupup_add: bad arguments");
    const long total = (long)(T * ft - drop) * 2 * H * 2 * W * Cout;
    if (Cout % 4 == 0 && ((uintptr_t)out & 15) == 0)
        hipLaunchKernelGGL(dupup_add_kernel<4>, dim3((unsigned)min((total / 4 + 255) / 256, (long)16384)), dim3(256), 0,
                           (hipStream_t)stream, x, out, T, H, W, Cin, Cout, ft, drop);
    else
        hipLaunchKernelGGL(dupup_add_kernel<1>, dim3((unsigned)min((total + 255) / 256, (long)8192)), dim3(256), 0,
                           (hipStream_t)stream, x, out, T, H, W, Cin, Cout, ft, drop);
    UV_CHECK_LAUNCH("uv_vae_dupup_add");
    return 0;
}

// AvgDown3D (vae2_2.py:335-367) added onto the main path: out[t,h,w,co] += mean_g x[cin, t*ft+a-pad, h*fs+b, w*fs+c],
// (cin,a,b,c) = unravel(co*group + g, (Cin, ft, fs, fs)); frames with negative index (front padding) read zero.
__global__ void avgdown_add_kernel(const float* x, float* out, int T, int H, int W, int Cin, int Cout, int ft, int fs) {
    const int pad = (ft - T % ft) % ft;
    const int To = (T + pad) / ft, Ho = H / fs, Wo = W / fs;
    const int factor = ft * fs * fs, group = Cin * factor / Cout;
    const long total = (long)To * Ho * Wo * Cout;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout);
        long r = i / Cout;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho);
        const int to = (int)(r / Ho);
        float s = 0.f;
        for (int g = 0; g < group; ++g) {
            const int j = co * group + g;
            const int c = j % fs, b = (j / fs) % fs, a = (j / (fs * fs)) % ft, cin = j / factor;
            const int ti = to * ft + a - pad;
            if (ti >= 0) s += x[(((long)ti * H + ho * fs + b) * W + wo * fs + c) * Cin + cin];
        }
        out[i] += s / (float)group;
    }
}

extern "C" int uv_vae_avgdown_add(const float* x, float* out, int T, int H, int W, int Cin, int Cout, int ft, int fs,
                                  void* stream) {
    UV_CHECK_ARG(x && out && (Cin * ft * fs * fs) % Cout == 0 && H % fs == 0 && W % fs == 0, "uv_vae_avgdown_add: bad arguments");
    const int pad = (ft - T % ft) % ft;
    const long total = (long)((T + pad) / ft) * (H / fs) * (W / fs) * Cout;
    hipLaunchKernelGGL(avgdown_add_kernel, dim3((unsigned)min((total + 255) / 256, (long)8192)), dim3(256), 0,
                       (hipStream_t)stream, x, out, T, H, W, Cin, Cout, ft, fs);
    UV_CHECK_LAUNCH("uv_vae_avgdown_add");
    return 0;
}

// decode entry: z [Z, f, h, w] -> channels-last rows [f*h*w, ld] of z / inv_std + mean   (vae2_2.py:814-816)
__global__ void latent_in_kernel(const float* z, const float* mean, const float* inv_std, float* out, long ld, int Z, long P) {
    const long total = P * Z;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Z);
        const long pix = i / Z;
        out[pix * ld + c] = __fadd_rn(__fdiv_rn(z[(long)c * P + pix], inv_std[c]), mean[c]);
    }
}

extern "C" int uv_vae_latent_in(const float* z, const float* mean, const float* inv_std, float* out, long ld, int Z, long P,
                                void* stream) {
    UV_CHECK_ARG(z && mean && inv_std && out && ld >= Z, "uv_vae_latent_in: bad arguments");
    hipLaunchKernelGGL(latent_in_kernel, dim3((unsigned)min((P * Z + 255) / 256, (long)4096)), dim3(256), 0,
                       (hipStream_t)stream, z, mean, inv_std, out, ld, Z, P);
    UV_CHECK_LAUNCH("uv_vae_latent_in");
    return 0;
}

// encode exit: rows [P, ld] (first Z channels = mu) -> [Z, f, h, w] of (mu - mean) * inv_std   (vae2_2.py:803-806)
__global__ void latent_out_kernel(const float* mu, long ld, const float* mean, const float* inv_std, float* out, int Z, long P) {
    const long total = P * Z;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i % P;
        const int c = (int)(i / P);
        out[i] = __fmul_rn(__fsub_rn(mu[pix * ld + c], mean[c]), inv_std[c]);
    }
}

extern "C" int uv_vae_latent_out(const float* mu, long ld, const float* mean, const float* inv_std, float* out, int Z, long P,
                                 void* stream) {
    UV_CHECK_ARG(mu && mean && inv_std && out && ld >= Z, "uv_vae_latent_out: bad arguments");
    hipLaunchKernelGGL(latent_out_kernel, dim3((unsigned)min((P * Z + 255) / 256, (long)4096)), dim3(256), 0,
                       (hipStream_t)stream, mu, ld, mean, inv_std, out, Z, P);
    UV_CHECK_LAUNCH("uv_vae_latent_out");
    return 0;
}

// encode entry: video [3, F, H, W] frames [f0, f0+T) -> patchified channels-last [T, H/2, W/2, ld]
// channel = c*4 + r*2 + q  with q = row parity, r = column parity   ("b c f (h q) (w r) -> b (c r q) f h w")
__global__ void video_in_kernel(const float* vid, float* out, long ld, int F, int H, int W, int f0, int T) {
    const int Hp = H / 2, Wp = W / 2;
    const long total = (long)T * Hp * Wp * 12;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % 12);
        long r = i / 12;
        const int w = (int)(r % Wp); r /= Wp;
        const int h = (int)(r % Hp);
        const int t = (int)(r / Hp);
        const int c = ch >> 2, rr = (ch >> 1) & 1, q = ch & 1;
        out[(((long)t * Hp + h) * Wp + w) * ld + ch] = vid[(((long)c * F + f0 + t) * H + 2 * h + q) * W + 2 * w + rr];
    }
}

extern "C" int uv_vae_video_in(const float* vid, float* out, long ld, int F, int H, int W, int f0, int T, void* stream) {
    UV_CHECK_ARG(vid && out && H % 2 == 0 && W % 2 == 0 && ld >= 12 && f0 >= 0 && f0 + T <= F, "uv_vae_video_in: bad arguments");
    const long total = (long)T * (H / 2) * (W / 2) * 12;
    hipLaunchKernelGGL(video_in_kernel, dim3((unsigned)min((total + 255) / 256, (long)4096)), dim3(256), 0,
                       (hipStream_t)stream, vid, out, ld, F, H, W, f0, T);
    UV_CHECK_LAUNCH("uv_vae_video_in");
    return 0;
}

// decode exit: head output [T, Hp, Wp, ld] (12 channels) -> video [3, F, 2Hp, 2Wp] frames [f0, f0+T), clamped to
// [-1, 1]   ("b (c r q) f h w -> b c f (h q) (w r)", vae2_2.py:306-312, clamp :1045)
__global__ void video_out_kernel(const float* y, long ld, float* vid, int F, int Hp, int Wp, int f0, int T) {
    const int H = 2 * Hp, W = 2 * Wp;
    const long total = (long)3 * T * H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        long r = i / W;
        const int h = (int)(r % H); r /= H;
        const int t = (int)(r % T);
        const int c = (int)(r / T);
        const int ch = c * 4 + (w & 1) * 2 + (h & 1);
        const float v = y[(((long)t * Hp + (h >> 1)) * Wp + (w >> 1)) * ld + ch];
        vid[(((long)c * F + f0 + t) * H + h) * W + w] = fminf(fmaxf(v, -1.0f), 1.0f);
    }
}

extern "C" int uv_vae_video_out(const float* y, long ld, float* vid, int F, int Hp, int Wp, int f0, int T, void* stream) {
    UV_CHECK_ARG(y && vid && ld >= 12 && f0 >= 0 && f0 + T <= F, "uv_vae_video_out: bad arguments");
    const long total = (long)3 * T * 4 * Hp * Wp;
    hipLaunchKernelGGL(video_out_kernel, dim3((unsigned)min((total + 255) / 256, (long)8192)), dim3(256), 0,
                       (hipStream_t)stream, y, ld, vid, F, Hp, Wp, f0, T);
    UV_CHECK_LAUNCH("uv_vae_video_out");
    return 0;
}

// ---- f16x3 operands for convolutions whose input is NOT an RMS_norm output (Resample's 3x3 convolutions: raw residual-stream rows) -----------
// uv_conv3d_f16x3 needs |x| < 65 504. A raw feature map has no static bound, so the split is taken under a PER-TENSOR power-of-two scale found
// on the device: hi / lo fp16 pieces of x * s with s = 1 while max |x| < 2^15 (every realistic activation: then this is exactly what
// uv_vae_rms_silu(split_out=2) writes; the maximum is found in the same pass, see split_f16_scaled_kernel)
// and s = 2^(14 - floor(log2 max)) above. 1 / s goes to scale[0]; the convolution multiplies its result by it (exact). No host round trip.
// SPECULATIVE = the first launch: pieces under s = 1 AND the tensor's max |x| in the same pass over x (one atomicMax on the f32 bits per block: non-negative
// floats order like their bit patterns; fmaxf drops NaNs - a NaN input leaves the scale at that of the finite values and propagates through the
// products). !SPECULATIVE = the second launch: writes 1 / s and, only if s != 1 (max |x| >= 2^15: no realistic activation), splits again under s;
// otherwise every block returns after one load. (Round 4: the separate max pass in front of the split cost a second read of every raw tensor.)
template <bool SPECULATIVE>
__global__ __launch_bounds__(256) void split_f16_scaled_kernel(const float* x, long ld, float* out, long ld_out, long P, int C, float* scale) {
    float s = 1.f;
    if constexpr (!SPECULATIVE) {
        const float mx = __builtin_bit_cast(float, *(const unsigned*)(scale + 1));
        if (mx >= 32768.f && mx < INFINITY) {
            int ex;
            frexpf(mx, &ex);                                            // mx = f * 2^ex, f in [0.5, 1)
            s = ldexpf(1.f, 15 - ex);                                   // mx * s in [2^14, 2^15)
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) scale[0] = 1.f / s;    // exact (power of two); only this thread writes it, everyone reads scale[1]
        if (s == 1.f) return;                                           // the speculative pieces stand
    }
    const int nv = C >> 2;
    const long total = P * nv;
    float m = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long row = i / nv;
        const int c = (int)(i - row * nv) * 4;
        const f32x4 v = *(const f32x4*)(x + row * ld + c);
        if constexpr (SPECULATIVE) m = fmaxf(fmaxf(fmaxf(m, fabsf(v[0])), fmaxf(fabsf(v[1]), fabsf(v[2]))), fabsf(v[3]));
        _Float16 hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float y = v[e] * s;
            hi[e] = (_Float16)y;
            lo[e] = (_Float16)(y - (float)hi[e]);
        }
        auto pk = [](_Float16 a, _Float16 b) { return (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16); };
        bf16_t* ob = (bf16_t*)(out + row * ld_out) + (c >> 5) * 64 + (c & 31);
        *(u32x2*)ob = (u32x2){pk(hi[0], hi[1]), pk(hi[2], hi[3])};
        *(u32x2*)(ob + 32) = (u32x2){pk(lo[0], lo[1]), pk(lo[2], lo[3])};
    }
    if constexpr (SPECULATIVE) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        __shared__ float part[4];
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
            atomicMax((unsigned*)(scale + 1), __builtin_bit_cast(unsigned, m));
        }
    }
}

extern "C" int uv_vae_split_f16(const float* x, long ld, float* out, long ld_out, long P, int C, float* scale, void* stream) {
    UV_CHECK_ARG(x && out && scale && P > 0 && C > 0 && C % 32 == 0 && ld % 4 == 0 && ld_out % 4 == 0 && ld >= C && ld_out >= C,
                 "uv_vae_split_f16: bad arguments (C=%d must be a multiple of 32)", C);
    UV_CHECK_ARG((const void*)x != (const void*)out, "uv_vae_split_f16: out must not alias x (the second launch may re-read x)");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(scale, 0, 2 * sizeof(float), st) != hipSuccess) {
        uv_set_error("uv_vae_split_f16: memset failed");
        return -2;
    }
    const unsigned blocks = (unsigned)min((P * (C >> 2) + 255) / 256, (long)8192);
    hipLaunchKernelGGL(split_f16_scaled_kernel<true>, dim3(blocks), dim3(256), 0, st, x, ld, out, ld_out, P, C, scale);
    hipLaunchKernelGGL(split_f16_scaled_kernel<false>, dim3(blocks), dim3(256), 0, st, x, ld, out, ld_out, P, C, scale);
    UV_CHECK_LAUNCH("uv_vae_split_f16");
    return 0;
}
