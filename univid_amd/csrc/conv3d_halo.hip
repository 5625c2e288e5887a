// Causal 3x3(x3) stride-1 convolution of the Wan2.2 VAE as an LDS-HALO'd implicit GEMM (the large ResidualBlock convolutions:
// ~90 % of the decoder's FLOPs), channels-last f32 activations.
//
// Replaces (reference, fp32 nn.Conv3d through cuDNN): CausalConv3d.forward  models/wan/utils/modules/vae2_2.py:17-42 as used by
// ResidualBlock (vae2_2.py:193-235); same arithmetic contract as conv3d_f32_kernel (conv3d_f32.hip), which keeps every other
// geometry (1x1x1, strided, time_conv, Cout not a multiple of 16 x 8). Resample's "nearest-exact 2x + Conv2d 3x3" (vae2_2.py:86-96,
// 153-155) runs here too: the halo image is filled through the upsampling map, the MFMA loop does not know.
//
// conv3d_f32_kernel gathers a 128-pixel x 32-channel A tile from L2 for EVERY tap (27 gathers of the same pixels per channel block).
// Here a workgroup owns an 8 x 32 pixel patch of one frame and stages, once per (input frame dt, 32-channel block), the patch with
// its one-pixel halo - 10 x 34 pixels - in LDS; the nine spatial taps then read their A fragments from that image at shifted
// addresses (halo pixel = (y + dh) * 34 + x + dw): 9x less L2 -> LDS traffic and LDS-write work for the activations (x 340 / 256
// for the halo), half the weight traffic per output (256 instead of 128 pixels per weight tile).
//   * 8 waves: 4 (pixels) x 2 (output channels), a wave owns 64 pixels x 64 channels (the MFMA tile of conv3d_f32_kernel).
//   * the halo goes global -> registers -> LDS (six 16-byte loads per lane, issued a whole tap group ahead of their use) because the
//     f32-grade bf16 arithmetic (PREC 3, "bf16x6") SPLITS on the way: every f32 activation is written as three exact bf16 planes
//     ONCE per halo element instead of once per (element, tap, wave column) in the MFMA loop - 13x less splitting work, which was
//     the limiter of that mode. PREC 0 (exact-f32 MFMA) writes the f32 image unchanged.
//   * weights stream as before: one k-tile (tap, 32 channels) per step, LDS-DMA double buffer, one barrier per k-tile.
#include "conv_args.h"
#include <stdlib.h>

typedef __attribute__((address_space(3))) void lds_void_h;

// TW = 32: 8 waves on an 8 x 32 patch, one workgroup per CU (the bf16x6 form). TW = 16: 4 waves on an 8 x 16 patch, 55 KiB of LDS, TWO
// independent workgroups per CU that cover each other's barrier stalls (what the exact-f32 arithmetic, bound by its MFMA, needs).
// BN = output channels per workgroup: 128, or 160 for the encoder's 160 / 320-channel stages (exact f32 only).
// WBUF = weight buffers: 2 (next tap's weights land while this tap computes), or 1 - 39 KiB of LDS and <= 168 registers, THREE
// workgroups per CU that cover each other's weight waits (exact f32, 128-wide tile).
template <int PREC, int TW = 32, int BN = 128, int WBUF = 2>
__global__ __launch_bounds__(TW * 16) __attribute__((amdgpu_waves_per_eu(WBUF == 1 ? 3 : 2, WBUF == 1 ? 3 : 2))) void conv3d_halo_kernel(ConvArgs p) {
    static_assert(PREC == 0 || PREC == 3, "exact-f32 MFMA or f32-grade bf16x6");
    static_assert(TW == 32 || TW == 16, "patch width");
    constexpr int TH = 8, HW_ = TW + 2, NHP = (TH + 2) * HW_;                   // 340 / 180 halo pixels
    static_assert(BN == 128 || (BN == 160 && PREC == 0 && TW == 16), "output-channel tile");
    constexpr int NW = TW / 4, NT = NW * 64, TM = 4, TN = BN / 32;
    constexpr int HALO_ROW = PREC == 3 ? 64 : 128;                              // bytes per halo pixel (per plane for PREC 3)
    constexpr int NHP8 = (NHP + 7) / 8 * 8;                                      // padded to whole 8-pixel groups
    constexpr int HALO_PLANE = NHP8 * HALO_ROW;
    constexpr int HALO_BYTES = (PREC == 3 ? 3 : 1) * HALO_PLANE;
    constexpr int W_BYTES = PREC == 3 ? BN * 192 : BN * 128;
    constexpr int W_PIECES = W_BYTES / 1024, W_INSTR = W_PIECES / NW;            // 16 / 24 pieces -> 2 / 3 per wave
    constexpr int NLD = (NHP * 8 + NT - 1) / NT;                                // sixteen-byte chunks of the halo per lane (6)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wbuf = smem + HALO_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_w = (p.Wout + TW - 1) / TW, tiles_h = (p.Hout + TH - 1) / TH;
    const int tile_n = blockIdx.x % p.tiles_n;
    int mt = blockIdx.x / p.tiles_n;
    const int tx0 = (mt % tiles_w) * TW;
    mt /= tiles_w;
    const int ty0 = (mt % tiles_h) * TH;
    const int tf = mt / tiles_h;                                                 // output frame
    const int n0 = tile_n * BN;
    const int K = p.kt * 9 * p.Cin;
    const long frame = (long)p.Hin * p.Win * p.ld_in;

    // ---- halo staging map of this lane: slot s = it * 512 + tid -> halo pixel s / 8, 16-byte chunk (4 channels) s % 8
    long h_off[NLD];             // element offset of the chunk inside a frame (channel block 0)
    unsigned h_ok = 0;           // bit it: inside the frame
    unsigned h_lds[NLD];         // LDS byte offset the chunk (its first plane) is written to
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        const int s = it * NT + tid, hp = s >> 3, ch = s & 7;
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int y = ty0 + hy - 1, x = tx0 + hx - 1;                          // coordinates of the convolution's input image
        const bool ok = hp < NHP && (unsigned)y < (unsigned)p.Hout && (unsigned)x < (unsigned)p.Wout;
        h_ok |= (unsigned)ok << it;
        // up: that image is the nearest-exact 2x upsampling of the stored one (vae2_2.py:86-96): pixel (y, x) <- (y >> 1, x >> 1)
        const int sy = p.up ? y >> 1 : y, sx = p.up ? x >> 1 : x;
        h_off[it] = ok ? ((long)sy * p.Win + sx) * p.ld_in + ch * 4 : 0;
        if constexpr (PREC == 3) h_lds[it] = hp * 64 + ((((ch >> 1) ^ ((0x1320 >> (4 * ((hp >> 2) & 3))) & 3)) << 4) | ((ch & 1) << 3));
        else h_lds[it] = hp * 128 + ((ch ^ ((hp >> 1) & 7)) << 4);
        if (hp >= NHP8) h_lds[it] = 0xffffffffu;                               // beyond the padded image: never written
    }
    f32x4 hreg[NLD];
    // group g = (dt, channel block cb): dt = g / ncb, cb = g % ncb
    const int ncb = p.Cin >> 5, ngroups = p.kt * ncb;
    auto load_halo = [&](int g) __attribute__((always_inline)) {
        const int dt = g / ncb, cb = g - dt * ncb;
        const int fi = tf * p.st + p.t_off + dt;
        const bool fok = (unsigned)fi < (unsigned)p.Tin;
        const float* base = p.in + (long)(fok ? fi : 0) * frame + cb * 32;
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const bool ok = fok && ((h_ok >> it) & 1u);
            hreg[it] = *(const f32x4*)(ok ? base + h_off[it] : p.zeros);
        }
    };
    auto write_halo = [&](char* halo) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            if (h_lds[it] == 0xffffffffu) continue;
            if constexpr (PREC == 3) {
                float r[4] = {hreg[it][0], hreg[it][1], hreg[it][2], hreg[it][3]};
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    bf16x4 q;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const __bf16 hq = (__bf16)r[e];
                        q[e] = hq;
                        if (pl < 2) r[e] = r[e] - (float)hq;
                    }
                    *(bf16x4*)(halo + pl * HALO_PLANE + h_lds[it]) = q;
                }
            } else {
                *(f32x4*)(halo + h_lds[it]) = hreg[it];
            }
        }
    };

    // ---- weight staging (LDS-DMA, as conv3d_f32_kernel): one k-tile = (tap, 32 input channels)
    const int srow = lane >> 3, pchunk = lane & 7;
    const float* w_src[W_INSTR];
    int w_dst[W_INSTR];
#pragma unroll
    for (int i = 0; i < W_INSTR; ++i) {
        const int pi = i * NW + wave;
        if constexpr (PREC == 3) {
            const int plane = pi / (BN / 16), rblk = pi % (BN / 16);
            const int row = rblk * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ ((0x1320 >> (4 * ((row >> 2) & 3))) & 3);
            w_dst[i] = plane * (BN * 64) + rblk * 1024;
            w_src[i] = p.w + ((long)min(n0 + row, p.Cout - 1) * (K / 32) * 48 + plane * 16 + chunk * 4);
        } else {
            const int row = pi * 8 + srow;
            w_dst[i] = pi * 1024;
            w_src[i] = p.w + (long)min(n0 + row, p.Cout - 1) * K + (pchunk ^ ((row >> 1) & 7)) * 4;
        }
    }
    // (the source operand is cast to const void*: given a `const float*` rvalue here, hipcc's host pass silently drops the kernel's
    // launch stub from the object - undefined symbol at load time)
#define UV_HALO_STAGE_W(G, TAP, BUF)                                                                                        \
    do {                                                                                                                    \
        const int dt_ = (G) / ncb, cb_ = (G) - dt_ * ncb;                                                                    \
        const int ktile_ = (dt_ * 9 + (TAP)) * ncb + cb_;      /* k order of the weight matrix: tap-major, channels minor */ \
        const int koff_ = ktile_ * (PREC == 3 ? 48 : 32);                                                                   \
        _Pragma("unroll") for (int i = 0; i < W_INSTR; ++i)                                                                 \
            __builtin_amdgcn_global_load_lds((const void*)(w_src[i] + koff_), (lds_void_h*)(wbuf + (BUF) * W_BYTES + w_dst[i]), 16, 0, 0);  \
    } while (0)

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    int hpb[TM];                 // halo pixel of this lane's output pixel of fragment j for tap (0, 0)
#pragma unroll
    for (int j = 0; j < TM; ++j) hpb[j] = TW == 32 ? (wm * 2 + (j >> 1)) * HW_ + (j & 1) * 16 + frow : (wm * 4 + j) * HW_ + frow;
    int w_off[TN], w_key[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int row = wn * (BN / 2) + i * 16 + frow;
        if constexpr (PREC == 3) {
            w_off[i] = row * 64 + ((fq ^ ((0x1320 >> (4 * ((row >> 2) & 3))) & 3)) << 4);
            w_key[i] = 0;
        } else {
            w_off[i] = row * 128;
            w_key[i] = (row >> 1) & 7;
        }
    }

    auto compute = [&](int tap, const char* wb, const char* halo) __attribute__((always_inline)) {
        const int dh = tap / 3, dw = tap - dh * 3;
        const int toff = dh * HW_ + dw;
        if constexpr (PREC == 0) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f32x4 af[TM], wf[TN];
                const int c = ks * 4 + fq;
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const int hp = hpb[j] + toff;
                    af[j] = *(const f32x4*)(halo + hp * 128 + ((c ^ ((hp >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < TN; ++i) wf[i] = *(const f32x4*)(wb + w_off[i] + ((c ^ w_key[i]) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TN; ++i)
#pragma unroll
                        for (int j = 0; j < TM; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][e], af[j][e], acc[i][j], 0, 0, 0);
            }
        } else {
            bf16x8 ap[TM][3], wp[TN][3];
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int hp = hpb[j] + toff;
                const int o = hp * 64 + ((fq ^ ((0x1320 >> (4 * ((hp >> 2) & 3))) & 3)) << 4);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) ap[j][pl] = *(const bf16x8*)(halo + pl * HALO_PLANE + o);
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wp[i][pl] = *(const bf16x8*)(wb + w_off[i] + pl * (BN * 64));
            // the six terms with plane(i) + plane(j) <= 2, smallest first (conv3d_f32_kernel PREC 3: same order, same values)
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[i][2], ap[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[i][1], ap[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[i][0], ap[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[i][1], ap[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[i][0], ap[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp[i][0], ap[j][0], acc[i][j], 0, 0, 0);
                }
        }
    };

    // ---- pipeline: halo of group g+1 in registers while the nine taps of group g run; weights double-buffered per tap
    load_halo(0);
    UV_HALO_STAGE_W(0, 0, 0);
    write_halo(smem);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    if constexpr (WBUF == 1) {
        for (int g = 0; g < ngroups; ++g) {
            const bool more = g + 1 < ngroups;
            if (more) load_halo(g + 1);
            for (int tap = 0; tap < 9; ++tap) {
                compute(tap, wbuf, smem);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __syncthreads();                              // every wave has read this tap's weights (and, at tap 8, the halo)
                if (tap < 8) UV_HALO_STAGE_W(g, tap + 1, 0);
                else if (more) { UV_HALO_STAGE_W(g + 1, 0, 0); write_halo(smem); }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
    } else
    for (int g = 0; g < ngroups; ++g) {
        const bool more = g + 1 < ngroups;
        if (more) load_halo(g + 1);
        for (int tap = 0; tap < 9; ++tap) {
            if (tap < 8) UV_HALO_STAGE_W(g, tap + 1, buf ^ 1);
            else if (more) UV_HALO_STAGE_W(g + 1, 0, buf ^ 1);
            compute(tap, wbuf + buf * W_BYTES, smem);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            buf ^= 1;
        }
        if (more) {   // every wave has left the halo image of group g behind (barrier above)
            write_halo(smem);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    // ---- epilogue: bias, optional residual, f32 store (pixel m = (frame, y, x) of the output tensor)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int py = TW == 32 ? wm * 2 + (j >> 1) : wm * 4 + j, px = TW == 32 ? (j & 1) * 16 + frow : frow;
        const int y = ty0 + py, x = tx0 + px;
        if (y >= p.Hout || x >= p.Wout) continue;
        const long m = ((long)tf * p.Hout + y) * p.Wout + x;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int n = n0 + wn * (BN / 2) + i * 16 + 4 * fq;
            if (n >= p.Cout) continue;
            f32x4 v = acc[i][j];
            if (p.bias) {
                const f32x4 b = *(const f32x4*)(p.bias + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += b[e];
            }
            if (p.resid) {
                const f32x4 rr = *(const f32x4*)(p.resid + m * p.ldr + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += rr[e];
            }
            *(f32x4*)(p.out + m * p.ldo + n) = v;
        }
    }
}


// Which convolutions take the halo kernel: 3x3 spatial taps, stride 1, padding 1 (plain or behind the 2x upsampling), no interleave, whole 32-channel input
// blocks (all callers pad), whole 128-wide output-channel tiles, and enough tiles per frame to fill the chip at four frames per pass.
// uv_set_option(UV_OPT_CONV_HALO, v) (developer A/B switch and test hook; never the environment): 0 = never, 1 = whenever the geometry
// fits (also launches too small to fill the chip, which the tests use), -1 = automatic (default; both arithmetics).
// output-channel tile of the halo kernel for this convolution (0: none fits): whole 128-wide tiles, or - exact f32 - whole 160-wide ones
static int uv_conv3d_halo_bn(const ConvArgs& a, int prec) {
    if (a.Cout % 128 == 0) return 128;
    if (prec == 0 && a.Cout % 160 == 0) return 160;
    return 0;
}

bool uv_conv3d_halo_eligible(const ConvArgs& a, int prec) {
    const int force = uv_option(UV_OPT_CONV_HALO);
    if (force == 0) return false;
    if (!(prec == 0 || prec == 3)) return false;
    if (a.kh != 3 || a.kw != 3 || (a.kt != 3 && a.kt != 1)) return false;
    if (a.st != 1 || a.sh != 1 || a.sw != 1 || a.ph != 1 || a.pw != 1 || a.interleave) return false;
    const int mul = a.up ? 2 : 1;                            // up: the halo image is filled through the 2x nearest-exact map
    const int bn = uv_conv3d_halo_bn(a, prec);
    if (a.Hin * mul != a.Hout || a.Win * mul != a.Wout || bn == 0 || a.Cin % 32 != 0) return false;
    // per-FRAME tile count: the choice must not depend on how many frames a pass carries (the pass length is a memory / speed knob
    // that leaves results bit-identical, and the two kernels sum their k-tiles in different orders)
    // (exact f32: 8 x 16 patches on 4-wave workgroups, two per CU - 6.28 s against 6.43 s per 49 x 720 x 1280 decode on the gather kernel,
    // same process, interleaved; the 8-wave form of round 3's first version lost to it, 6.49 s, with one workgroup per CU)
    const long tiles = (long)((a.Hout + 7) / 8) * ((a.Wout + (prec == 3 ? 31 : 15)) / (prec == 3 ? 32 : 16)) * (a.Cout / bn);
    return force == 1 || 4 * tiles >= (prec == 3 ? 1 : 2) * uv_num_cus();
}

int uv_launch_conv3d_halo(ConvArgs& a, int prec, hipStream_t stream) {
    const int bn = uv_conv3d_halo_bn(a, prec);
    a.tiles_n = a.Cout / bn;
    if (prec == 3) {
        a.tiles_m = a.Tout * ((a.Hout + 7) / 8) * ((a.Wout + 31) / 32);
        const size_t lds = 3 * 344 * 64 + 2 * 128 * 192;
        UV_ONCE_PER_DEVICE(hipFuncSetAttribute((const void*)conv3d_halo_kernel<3, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((conv3d_halo_kernel<3, 32>), dim3(a.tiles_m * a.tiles_n), dim3(512), lds, stream, a);
    } else if (bn == 160) {
        a.tiles_m = a.Tout * ((a.Hout + 7) / 8) * ((a.Wout + 15) / 16);
        const size_t lds = 184 * 128 + 2 * 160 * 128;          // 63 KiB: two workgroups per CU
        UV_ONCE_PER_DEVICE(hipFuncSetAttribute((const void*)conv3d_halo_kernel<0, 16, 160>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((conv3d_halo_kernel<0, 16, 160>), dim3(a.tiles_m * a.tiles_n), dim3(256), lds, stream, a);
    } else {
        a.tiles_m = a.Tout * ((a.Hout + 7) / 8) * ((a.Wout + 15) / 16);
        // single weight buffer: 39 KiB of LDS and 168 registers, three workgroups per CU (6.19-6.21 s against 6.26 s per decode with two
        // double-buffered workgroups per CU, same process, interleaved, bit-identical: the k order is the same)
        const size_t lds = 184 * 128 + 128 * 128;
        hipLaunchKernelGGL((conv3d_halo_kernel<0, 16, 128, 1>), dim3(a.tiles_m * a.tiles_n), dim3(256), lds, stream, a);
    }
    return 0;
}
