// Causal 3D / 2D convolution of the Wan2.2 VAE as an implicit GEMM on the exact-f32 MFMA
// (v_mfma_f32_16x16x4_f32; opt-in: the same f32 products on the bf16 MFMA, PREC 3 / PREC 1-2 below), channels-last activations.
//
//   out[pixel, co] = bias[co] + sum_{tap, ci} in[pixel + tap, ci] * w[co, tap, ci]   (+ residual)
//   M = output pixels (t, h, w), N = Cout, K = taps * Cin; A rows are GATHERED straight from the input
//   ring by the global_load_lds source address (one 16-byte chunk of one tap per lane), out-of-range taps
//   (spatial zero padding, causal time padding) read a zero page instead.
//
// Replaces (reference, fp32 nn.Conv3d / nn.Conv2d through cuDNN):
//   CausalConv3d.forward            models/wan/utils/modules/vae2_2.py:34-42  (time left-pad / cache prepend)
//   Resample: Upsample(nearest-exact 2x) + Conv2d 3x3           vae2_2.py:86-96, 153-155   (up=1: the 2x
//       upsample is folded into the gather, the upsampled tensor is never materialised)
//   Resample: ZeroPad2d(0,1,0,1) + Conv2d 3x3 stride 2          vae2_2.py:99-108
//   Resample.time_conv 3x1x1 (C -> 2C, then frame interleave)   vae2_2.py:97-98, 143-151  (interleave=1:
//       the reshape/stack of :148-151 is done by the epilogue's store addresses)
//
// The reference's per-convolution feature cache (CACHE_T = 2 frames, vae2_2.py:219-232) is realised by the
// caller as an input RING [2 + T, H, W, C]: frames 0..1 hold the cached frames (zeros before the first
// chunk = the causal zero padding), the producer writes frames 2.., and `t_off` selects the first ring frame
// a kernel tap reads.
#include "conv_args.h"
#include <stdlib.h>
#include <math.h>

typedef __attribute__((address_space(3))) void lds_void_c;

// PREC 0: exact-f32 MFMA (v_mfma_f32_16x16x4_f32), weights f32 [Cout][K].
// PREC 1: "bf16x3": every f32 operand x is split as hi + lo (two bf16), and x*w ~ xh*wh + xh*wl + xl*wh on the bf16 MFMA
//         (v_mfma_f32_16x16x32_bf16, f32 accumulate). Relative error per product ~1e-5 (the dropped xl*wl term and the
//         bf16 rounding of lo), i.e. ~100x below the 1e-3 tolerance, at up to 16/3 x the f32-MFMA rate. Weights are
//         pre-split on the host into [Cout][K/32][32 hi | 32 lo] bf16 (same bytes per row as f32, so the staging code
//         is shared); activations stay f32 in HBM/LDS and are split in registers.
// PREC 2: bf16x3 with the ACTIVATIONS pre-split too (the producer, uv_vae_rms_silu, writes [C/32][32 hi | 32 lo] bf16 into
//         the input ring: same bytes per pixel as f32): no conversion work in the MFMA loop at all.
// PREC 4: "f16x3", f32-GRADE on HALF the passes of PREC 3: both operands pre-split into TWO IEEE fp16 pieces (hi = fp16(x), lo =
//         fp16(x - hi): 11 + 11 significand bits and a sign = x to 2^-22 relative, worst case; same memory formats as PREC 2 with fp16 in
//         place of bf16), x*w ~ xh*wh + xh*wl + xl*wh on the fp16 MFMA (v_mfma_f32_16x16x32_f16), f32 accumulate. The dropped xl*wl is
//         2^-22 |x w|. Against an fp64 convolution the result is as close as the exact-f32 MFMA kernel's, because at K = 27 C both are
//         dominated by the f32 ACCUMULATION error (1.4e-6 .. 2.8e-6 relative rms at K = 6 912 .. 27 648; the 22-bit operands add
//         ~1.5e-7): test_conv3d_f16x3_is_f32_grade. fp16's range is what the caller must respect: the WEIGHTS are pre-scaled by a power
//         of two (uv_split_weights_f16x3: max |w| * scale in [2^13, 2^14), so that the lo pieces of all but the smallest weights are
//         normal fp16 numbers; undone exactly in the epilogue: out = acc * out_scale + bias), the ACTIVATIONS are taken as they are
//         and must stay below 65 504 in magnitude - uv_vae_rms_silu(split_out = 2) writes them, and an RMS-normalised row is bounded by
//         sqrt(C) max|gamma| (the host checks that bound per convolution and keeps PREC 3 where it does not hold). Activation lo pieces
//         of |x| < 2^-3 are subnormal fp16: an ABSOLUTE error <= 2^-25 there, i.e. below f32's own rounding of the row's O(1) elements.
// PREC 3: "bf16x6", f32-GRADE: every f32 operand is three bf16 planes (x = x0 + x1 + x2 exactly) and the product keeps the six
//         terms with i + j <= 2 (error < 2^-26 |x w|, under f32's own rounding), f32 accumulate. Activations f32 in HBM/LDS, split in
//         registers; weights pre-split on the host into [Cout][K/32][32 p0 | 32 p1 | 32 p2] bf16 and staged as three plane sub-tiles.
template <int BM, int BN, int WM, int WN, int PREC>
__global__ __launch_bounds__(WM* WN * 64) void conv3d_f32_kernel(ConvArgs p) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    // PREC 3 stages the weights as three bf16 planes (64-byte rows, one [BN][64 B] sub-tile per plane) instead of f32 rows
    constexpr int A_BYTES = BM * 128, W_BYTES = PREC == 3 ? BN * 192 : BN * 128, STAGE = A_BYTES + W_BYTES;
    constexpr int W_PIECES = PREC == 3 ? 3 * (BN / 16) : BN / 8;             // 1-KiB LDS-DMA pieces per W tile and k-tile
    constexpr int A_INSTR = BM / 8 / NW, W_INSTR = (W_PIECES + NW - 1) / NW;  // a narrow W tile (BN = 16) is staged by the first waves only
    static_assert(BM % (8 * NW) == 0, "A tile rows must divide over the waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // n tiles fastest: blocks that share an A panel (the expensive gather) are adjacent
    const int tile_n = blockIdx.x % p.tiles_n, tile_m = blockIdx.x / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int K = p.kt * p.kh * p.kw * p.Cin;

    const int srow = lane >> 3, pchunk = lane & 7;
    int a_t[A_INSTR], a_h[A_INSTR], a_w[A_INSTR], a_c[A_INSTR];
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int row = (i * NW + wave) * 8 + srow;
        a_c[i] = (pchunk ^ ((row >> 1) & 7)) * 4;
        const int m = min(m0 + row, p.M - 1);
        a_w[i] = m % p.Wout;
        const int r = m / p.Wout;
        a_h[i] = r % p.Hout;
        a_t[i] = r / p.Hout;
    }
    const float* w_src[W_INSTR];
    int w_c[W_INSTR];
    int w_dst[W_INSTR];            // LDS byte offset of the piece inside the W tile
#pragma unroll
    for (int i = 0; i < W_INSTR; ++i) {
        const int pi = i * NW + wave;
        if constexpr (PREC == 3) {
            // piece = 16 rows x 64 B of ONE plane; lane (row = lane / 4, slot = lane % 4) fetches chunk slot ^ swz(row) so that the
            // fragment reads below (16 lanes of a ds_read_b128 group span rows 4 apart at a 64-byte row stride) are conflict-free
            const int plane = pi / (BN / 16), rblk = pi % (BN / 16);
            const int row = rblk * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ ((0x1320 >> (4 * ((row >> 2) & 3))) & 3);          // swz(q) = {0, 2, 3, 1}[q]
            w_c[i] = 0;
            w_dst[i] = plane * (BN * 64) + rblk * 1024;
            // split weights: [Cout][K / 32][32 p0 | 32 p1 | 32 p2] bf16 = 48 f32-sized units per 32 k; p.w is typed float*
            w_src[i] = p.w + ((long)min(n0 + row, p.Cout - 1) * (K / 32) * 48 + plane * 16 + chunk * 4);
        } else {
            const int row = pi * 8 + srow;
            w_c[i] = (pchunk ^ ((row >> 1) & 7)) * 4;
            w_dst[i] = pi * 1024;
            w_src[i] = p.w + (long)min(n0 + row, p.Cout - 1) * K + w_c[i];
        }
    }
    const int Hlim = p.up ? 2 * p.Hin : p.Hin, Wlim = p.up ? 2 * p.Win : p.Win;
    const long frame = (long)p.Hin * p.Win * p.ld_in;

    // Fast gather addressing (every convolution except the 2x-upsampling ones): a piece's source for tap (dt, dh, dw) is its
    // pointer for tap (0, 0, 0) plus an offset that is THE SAME for every lane, and whether the tap falls inside the input is a
    // property of (piece, tap) that does not change along K. So per piece ONE base pointer and ONE bit mask over the <= 27 taps
    // are computed up front, and staging a k-tile costs a bit test, a select against the zero page and a 64-bit add per piece -
    // instead of re-deriving (t, h, w), three bounds checks and a 64-bit multiply-add behind an exec-mask branch, per piece and
    // k-tile (the compiled loop spent ~150 VALU / SALU instructions and 4 divergent branches there per 128 MFMAs).
    // The 2x-upsampling convolutions (nearest-exact upsample folded into the gather: source pixel = upsampled coordinate >> 1) use
    // the same scheme with one more per-piece fact: the source offset of tap dh is ((oh + dh - ph) >> 1) - ((oh - ph) >> 1) rows,
    // which depends on the parity of the piece's own upsampled row (and column) only: (dh + par_h) >> 1.
    const bool fast = p.kt * p.kh * p.kw <= 32;
    const float* a_base[A_INSTR];
    unsigned a_mask[A_INSTR];
    int a_par[A_INSTR];              // up: bit 0 = parity of the upsampled row of tap 0, bit 1 = of its column
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int it0 = a_t[i] * p.st + p.t_off, ih0 = a_h[i] * p.sh - p.ph, iw0 = a_w[i] * p.sw - p.pw;
        // (arithmetic shift: ih0 = -1 -> source row -1, whose taps are masked off below)
        const int sh0 = p.up ? ih0 >> 1 : ih0, sw0 = p.up ? iw0 >> 1 : iw0;
        a_par[i] = p.up ? ((ih0 & 1) | ((iw0 & 1) << 1)) : 0;
        a_base[i] = p.in + (long)it0 * frame + ((long)sh0 * p.Win + sw0) * p.ld_in + a_c[i];
        unsigned mk = 0;
        if (fast) {
            for (int dt = 0, tap = 0; dt < p.kt; ++dt)
                for (int dh = 0; dh < p.kh; ++dh)
                    for (int dw = 0; dw < p.kw; ++dw, ++tap) {
                        const bool ok = (unsigned)(it0 + dt) < (unsigned)p.Tin && (unsigned)(ih0 + dh) < (unsigned)Hlim &&
                                        (unsigned)(iw0 + dw) < (unsigned)Wlim;
                        mk |= (unsigned)ok << tap;
                    }
        }
        a_mask[i] = mk;
    }

    // running (tap, channel) position of the NEXT k-tile to stage; a k-tile (32 channels) never straddles taps
    int s_ci = 0, s_dt = 0, s_dh = 0, s_dw = 0, s_tap = 0;
    long s_off = 0;                      // element offset of the current tap from tap (0, 0, 0): dt * frame + (dh * Win + dw) * ld_in
#define UV_CONV_STAGE(KT, BUF)                                                                                        \
    do {                                                                                                              \
        char* sbase = smem + (BUF) * STAGE;                                                                           \
        if (fast && !p.up) {                                                                                          \
            _Pragma("unroll") for (int i = 0; i < A_INSTR; ++i) {                                                     \
                const float* src = ((a_mask[i] >> s_tap) & 1u) ? a_base[i] + (s_off + s_ci) : p.zeros;                \
                __builtin_amdgcn_global_load_lds(src, (lds_void_c*)(sbase + (i * NW + wave) * 1024), 16, 0, 0);       \
            }                                                                                                         \
        } else if (fast) {                                                                                            \
            _Pragma("unroll") for (int i = 0; i < A_INSTR; ++i) {                                                     \
                const int rh = (s_dh + (a_par[i] & 1)) >> 1, rw = (s_dw + (a_par[i] >> 1)) >> 1;                      \
                const long off = (long)s_dt * frame + (long)(rh * p.Win + rw) * p.ld_in + s_ci;                       \
                const float* src = ((a_mask[i] >> s_tap) & 1u) ? a_base[i] + off : p.zeros;                           \
                __builtin_amdgcn_global_load_lds(src, (lds_void_c*)(sbase + (i * NW + wave) * 1024), 16, 0, 0);       \
            }                                                                                                         \
        } else {                                                                                                      \
            _Pragma("unroll") for (int i = 0; i < A_INSTR; ++i) {                                                     \
                const int it = a_t[i] * p.st + s_dt + p.t_off;                                                        \
                int ih = a_h[i] * p.sh + s_dh - p.ph;                                                                 \
                int iw = a_w[i] * p.sw + s_dw - p.pw;                                                                 \
                const bool ok = (unsigned)it < (unsigned)p.Tin && (unsigned)ih < (unsigned)Hlim &&                    \
                                (unsigned)iw < (unsigned)Wlim;                                                        \
                if (p.up) { ih >>= 1; iw >>= 1; }                                                                     \
                const float* src = ok ? p.in + it * frame + ((long)ih * p.Win + iw) * p.ld_in + s_ci + a_c[i] : p.zeros; \
                __builtin_amdgcn_global_load_lds(src, (lds_void_c*)(sbase + (i * NW + wave) * 1024), 16, 0, 0);       \
            }                                                                                                         \
        }                                                                                                             \
        const int koff = (KT) * (PREC == 3 ? 48 : 32);                                                                \
        _Pragma("unroll") for (int i = 0; i < W_INSTR; ++i) {                                                         \
            const float* wsrc = w_src[i] + koff;                                                                      \
            if (W_PIECES % NW == 0 || i * NW + wave < W_PIECES)                                                       \
                __builtin_amdgcn_global_load_lds(wsrc, (lds_void_c*)(sbase + A_BYTES + w_dst[i]), 16, 0, 0);          \
        }                                                                                                             \
        s_ci += 32;                                                                                                   \
        if (s_ci >= p.Cin) {                                                                                          \
            s_ci = 0;                                                                                                 \
            ++s_tap;                                                                                                  \
            if (++s_dw == p.kw) { s_dw = 0; if (++s_dh == p.kh) { s_dh = 0; ++s_dt; } }                               \
            s_off = (long)s_dt * frame + ((long)s_dh * p.Win + s_dw) * p.ld_in;                                       \
        }                                                                                                             \
    } while (0)

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    int a_off[TM], a_key[TM], w_off[TN], w_key[TN];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int row = wm * (BM / WM) + j * 16 + frow;
        a_off[j] = row * 128; a_key[j] = (row >> 1) & 7;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int row = wn * (BN / WN) + i * 16 + frow;
        if constexpr (PREC == 3) {   // plane 0 address of this lane's 16-byte k-chunk (planes follow at BN * 64 bytes)
            w_off[i] = A_BYTES + row * 64 + ((fq ^ ((0x1320 >> (4 * ((row >> 2) & 3))) & 3)) << 4);
            w_key[i] = 0;
        } else {
            w_off[i] = A_BYTES + row * 128; w_key[i] = (row >> 1) & 7;
        }
    }

    const int nk = K / 32;
    UV_CONV_STAGE(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) UV_CONV_STAGE(kt + 1, buf ^ 1);
        const char* base = smem + buf * STAGE;
        if (PREC == 0) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f32x4 af[TM], wf[TN];
                const int c = ks * 4 + fq;
#pragma unroll
                for (int j = 0; j < TM; ++j) af[j] = *(const f32x4*)(base + a_off[j] + ((c ^ a_key[j]) << 4));
#pragma unroll
                for (int i = 0; i < TN; ++i) wf[i] = *(const f32x4*)(base + w_off[i] + ((c ^ w_key[i]) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TN; ++i)
#pragma unroll
                        for (int j = 0; j < TM; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][e], af[j][e], acc[i][j], 0, 0, 0);
            }
        } else if (PREC == 3) {
            // "bf16x6": every f32 operand is three bf16 planes, x = x0 + x1 + x2 with round-to-nearest at every level (|x1| <= 2^-9 |x|,
            // |x2| <= 2^-18 |x|, and the three planes hold all 24 significand bits: the decomposition is exact). The ACTIVATIONS stay
            // f32 in HBM and LDS (staging identical to PREC 0) and are split in registers; the WEIGHTS are split once on the host
            // (uv_split_weights_bf16x6) and staged as planes - splitting both in registers made the kernel VALU-bound (11 vector
            // instructions per operand pair: 74 ms for the 16 x 360 x 640, 256 -> 256 layer; 60 ms with the weight half gone).
            // The product keeps the six terms with
            // i + j <= 2; the dropped ones (x1 w2, x2 w1, x2 w2) are below 2^-26 |x w|, i.e. under the f32 rounding of the product
            // itself. Terms are accumulated smallest first on the f32 accumulators of the bf16 MFMA.
            bf16x8 ap[TM][3], wp[TN][3];
            auto split8 = [](const f32x4& x0, const f32x4& x1, bf16x8* out3) {
                float r[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { r[e] = x0[e]; r[4 + e] = x1[e]; }
#pragma unroll
                for (int lvl = 0; lvl < 3; ++lvl)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const __bf16 hq = (__bf16)r[e];
                        out3[lvl][e] = hq;
                        if (lvl < 2) r[e] = r[e] - (float)hq;
                    }
            };
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const f32x4 x0 = *(const f32x4*)(base + a_off[j] + (((2 * fq) ^ a_key[j]) << 4));
                const f32x4 x1 = *(const f32x4*)(base + a_off[j] + (((2 * fq + 1) ^ a_key[j]) << 4));
                split8(x0, x1, ap[j]);
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wp[i][pl] = *(const bf16x8*)(base + w_off[i] + pl * (BN * 64));
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
        } else {
            // one K = 32 MFMA step per k-tile: lane (row, fq) owns k = 8*fq .. 8*fq+7 of both operands
            bf16x8 ah[TM], al[TM], wh[TN], wl[TN];
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                if (PREC == 2 || PREC == 4) {
                    ah[j] = *(const bf16x8*)(base + a_off[j] + ((fq ^ a_key[j]) << 4));
                    al[j] = *(const bf16x8*)(base + a_off[j] + (((4 + fq) ^ a_key[j]) << 4));
                    continue;
                }
                const f32x4 x0 = *(const f32x4*)(base + a_off[j] + (((2 * fq) ^ a_key[j]) << 4));
                const f32x4 x1 = *(const f32x4*)(base + a_off[j] + (((2 * fq + 1) ^ a_key[j]) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const __bf16 h0 = (__bf16)x0[e], h1 = (__bf16)x1[e];
                    ah[j][e] = h0;
                    ah[j][4 + e] = h1;
                    al[j][e] = (__bf16)(x0[e] - (float)h0);
                    al[j][4 + e] = (__bf16)(x1[e] - (float)h1);
                }
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                wh[i] = *(const bf16x8*)(base + w_off[i] + ((fq ^ w_key[i]) << 4));
                wl[i] = *(const bf16x8*)(base + w_off[i] + (((4 + fq) ^ w_key[i]) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    acc[i][j] = mfma_16x16x32<PREC == 4>(wl[i], ah[j], acc[i][j]);
                    acc[i][j] = mfma_16x16x32<PREC == 4>(wh[i], al[j], acc[i][j]);
                    acc[i][j] = mfma_16x16x32<PREC == 4>(wh[i], ah[j], acc[i][j]);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    const int hw = p.Hout * p.Wout;
    const int chalf = p.Cout >> 1;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + wm * (BM / WM) + j * 16 + frow;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int n = n0 + wn * (BN / WN) + i * 16 + 4 * fq;
            if (n >= p.Cout) continue;
            f32x4 v = acc[i][j];
            if constexpr (PREC == 4) {
                const float sc = p.act_scale ? p.out_scale * *p.act_scale : p.out_scale;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= sc;                 // exact: a power of two
            }
            if (p.bias) {
                const f32x4 b = *(const f32x4*)(p.bias + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += b[e];
            }
            if (p.interleave) {
                // channel halves become consecutive frames: out frame = 2*t + (n >= C/2)   vae2_2.py:148-151
                const int t = m / hw, pix = m - t * hw;
                const int half = n >= chalf;
                *(f32x4*)(p.out + ((long)(2 * t + half) * hw + pix) * p.ldo + (n - half * chalf)) = v;
            } else if (p.ophase >= 0) {
                // one phase of a 2x-upsampling convolution: pixel (t, y, x) of the source-resolution grid -> (t, 2y + a, 2x + b)
                const int t = m / hw, pix = m - t * hw, y = pix / p.Wout, x = pix - y * p.Wout;
                const long orow = ((long)t * 2 * p.Hout + 2 * y + (p.ophase >> 1)) * (2 * p.Wout) + 2 * x + (p.ophase & 1);
                *(f32x4*)(p.out + orow * p.ldo + n) = v;
            } else {
                if (p.resid) {
                    const f32x4 rr = *(const f32x4*)(p.resid + (long)m * p.ldr + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += rr[e];
                }
                *(f32x4*)(p.out + (long)m * p.ldo + n) = v;
            }
        }
    }
}

const float* uv_zero_page();

template <int BM, int BN, int WM, int WN, int PREC>
static void launch_conv(ConvArgs& a, hipStream_t stream) {
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    auto kern = conv3d_f32_kernel<BM, BN, WM, WN, PREC>;
    const size_t lds = 2 * (BM * 128 + BN * (PREC == 3 ? 192 : 128));
    UV_ONCE_PER_DEVICE(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(WM * WN * 64), lds, stream, a);
}

static int conv_common(const float* in, long ld_in, int Tin, int Hin, int Win, const void* w, const float* bias, float* out,
                       long ldo, int Tout, int Hout, int Wout, int Cin, int Cout, int kt, int kh, int kw, int st, int sh,
                       int sw, int t_off, int ph, int pw, int up, int interleave, const float* resid, long ldr, int prec,
                       void* stream, float out_scale = 1.0f, const float* act_scale = nullptr) {
    UV_CHECK_ARG(in && w && out, "uv_conv3d: null pointer");
    UV_CHECK_ARG(Cin % 32 == 0, "uv_conv3d: Cin=%d must be a multiple of 32 (pad channels with zeros)", Cin);
    UV_CHECK_ARG(Cout % 4 == 0, "uv_conv3d: Cout=%d must be a multiple of 4", Cout);
    UV_CHECK_ARG(ld_in % 4 == 0 && ldo % 4 == 0 && ldr % 4 == 0 && ld_in >= Cin, "uv_conv3d: bad leading dimensions");
    UV_CHECK_ARG(Tout > 0 && Hout > 0 && Wout > 0 && kt > 0 && kh > 0 && kw > 0, "uv_conv3d: bad geometry");
    UV_CHECK_ARG(!interleave || (Cout % 8 == 0 && !resid), "uv_conv3d: interleave needs Cout %% 8 == 0 and no residual");
    UV_CHECK_ARG(up >= 0 && up <= 5, "uv_conv3d: up=%d (0 plain, 1 nearest-2x folded into the gather, 2..5 one output phase of it)", up);
    UV_CHECK_ARG(up < 2 || (!resid && !interleave && sh == 1 && sw == 1 && st == 1), "uv_conv3d: an output-phase launch (up >= 2) takes no residual, interleave or stride");
    UV_CHECK_ARG((((uintptr_t)in | (uintptr_t)w | (uintptr_t)out | (uintptr_t)bias | (uintptr_t)resid) & 15) == 0,
                 "uv_conv3d: pointers must be 16-byte aligned");
    ConvArgs a;
    a.in = in; a.w = (const float*)w; a.bias = bias; a.resid = resid; a.out = out; a.zeros = uv_zero_page();
    UV_CHECK_ARG(a.zeros, "uv_conv3d: zero page missing (call uv_init)");
    a.ld_in = ld_in; a.ldo = ldo; a.ldr = ldr;
    a.Tout = Tout; a.Hout = Hout; a.Wout = Wout; a.Tin = Tin; a.Hin = Hin; a.Win = Win;
    a.Cin = Cin; a.Cout = Cout; a.kt = kt; a.kh = kh; a.kw = kw; a.st = st; a.sh = sh; a.sw = sw;
    a.t_off = t_off; a.ph = ph; a.pw = pw; a.up = up == 1; a.ophase = up >= 2 ? up - 2 : -1; a.interleave = interleave;
    a.M = Tout * Hout * Wout;
    a.out_scale = out_scale;
    a.act_scale = act_scale;
    hipStream_t s = (hipStream_t)stream;
    // the large 3x3(x3) stride-1 convolutions (ResidualBlocks): LDS-halo kernel (conv3d_halo.hip), exact f32 and bf16x6
    if (uv_conv3d_halo_eligible(a, prec)) {
        uv_launch_conv3d_halo(a, prec, s);
        UV_CHECK_LAUNCH("uv_conv3d (halo)");
        return 0;
    }
    if (prec == 4 && uv_conv3d_halo16_eligible(a)) {      // f16x3: its own LDS-halo kernel (pre-split operands, halo filled by LDS-DMA)
        uv_launch_conv3d_halo16(a, s);
        UV_CHECK_LAUNCH("uv_conv3d_f16x3 (halo)");
        return 0;
    }
    // tile choice: 256x256 (16 waves, 4 per SIMD) when Cout >= 256 and the grid still fills the chip: halves the A gather
    // per output; 256x128 (8 waves) next; the 4-wave 128x128 tile for the low-resolution stages
    const long t256 = (long)((a.M + 255) / 256) * ((Cout + 255) / 256);
    const long t128 = (long)((a.M + 255) / 256) * ((Cout + 127) / 128);
    if (prec == 0) {
        // exact f32: 128-wide column tiles, except where they would mostly compute padding:
        //   Cout a multiple of 160 but not of 128 (the encoder's 160 / 320 channel stages): 160-wide tiles, no padded columns
        //     (128-wide ones compute 256 columns for 160: 37.5 % of the MFMA work wasted; 384 for 320: 17 %);
        //   Cout <= 16 (the decoder's last convolution, 256 -> 12 channels on full-resolution frames): 256 x 16 tiles
        //     (a 128-wide tile computes 128 columns for 12).
        if (Cout <= 16) launch_conv<256, 16, 4, 1, 0>(a, s);
        else if (Cout % 160 == 0 && Cout % 128 != 0) launch_conv<128, 160, 2, 2, 0>(a, s);
        else launch_conv<128, 128, 2, 2, 0>(a, s);
    } else if (prec == 3) {
        if (Cout <= 16) launch_conv<256, 16, 4, 1, 3>(a, s);
        // 160-wide tiles carry 30 KiB of weight planes per stage: 64 rows keep two workgroups per CU (2 x 76 KiB of LDS)
        else if (Cout % 160 == 0 && Cout % 128 != 0) launch_conv<64, 160, 2, 2, 3>(a, s);
        else launch_conv<128, 128, 2, 2, 3>(a, s);
    } else if (prec == 4) {
        if (Cout <= 16) launch_conv<256, 16, 4, 1, 4>(a, s);
        else if (Cout >= 256 && t256 >= 256) launch_conv<256, 256, 4, 4, 4>(a, s);
        else if (t128 >= 256) launch_conv<256, 128, 4, 2, 4>(a, s);
        else launch_conv<128, 128, 2, 2, 4>(a, s);
    } else if (prec == 1) {
        if (Cout >= 256 && t256 >= 256) launch_conv<256, 256, 4, 4, 1>(a, s);
        else if (t128 >= 256) launch_conv<256, 128, 4, 2, 1>(a, s);
        else launch_conv<128, 128, 2, 2, 1>(a, s);
    } else {
        if (Cout >= 256 && t256 >= 256) launch_conv<256, 256, 4, 4, 2>(a, s);
        else if (t128 >= 256) launch_conv<256, 128, 4, 2, 2>(a, s);
        else launch_conv<128, 128, 2, 2, 2>(a, s);
    }
    UV_CHECK_LAUNCH("uv_conv3d");
    return 0;
}

extern "C" int uv_conv3d_f32(const float* in, long ld_in, int Tin, int Hin, int Win, const float* w, const float* bias,
                             float* out, long ldo, int Tout, int Hout, int Wout, int Cin, int Cout, int kt, int kh,
                             int kw, int st, int sh, int sw, int t_off, int ph, int pw, int up, int interleave,
                             const float* resid, long ldr, void* stream) {
    return conv_common(in, ld_in, Tin, Hin, Win, w, bias, out, ldo, Tout, Hout, Wout, Cin, Cout, kt, kh, kw, st, sh, sw, t_off,
                       ph, pw, up, interleave, resid, ldr, 0, stream);
}

// Same convolution with f32-grade arithmetic on the bf16 matrix pipe (PREC 3 above: exact three-way operand splitting, six bf16 MFMA
// passes per product, f32 accumulate; per-product error below 2^-26, under f32's own rounding). Activations, bias, residual and output
// are the f32 tensors of uv_conv3d_f32; w_split6 = the weights split once by uv_split_weights_bf16x6:
// [Cout][K / 32][32 p0 | 32 p1 | 32 p2] bf16.
extern "C" int uv_conv3d_bf16x6(const float* in, long ld_in, int Tin, int Hin, int Win, const void* w_split6, const float* bias,
                                float* out, long ldo, int Tout, int Hout, int Wout, int Cin, int Cout, int kt, int kh,
                                int kw, int st, int sh, int sw, int t_off, int ph, int pw, int up, int interleave,
                                const float* resid, long ldr, void* stream) {
    return conv_common(in, ld_in, Tin, Hin, Win, w_split6, bias, out, ldo, Tout, Hout, Wout, Cin, Cout, kt, kh, kw, st, sh, sw,
                       t_off, ph, pw, up, interleave, resid, ldr, 3, stream);
}

// w [rows][K] f32 (K % 32 == 0) -> [rows][K/32][32 p0 | 32 p1 | 32 p2] bf16 with p0 = bf16(w), p1 = bf16(w - p0), p2 = bf16(w - p0 - p1)
// (round to nearest; w = p0 + p1 + p2 exactly)
__global__ void split_weights6_kernel(const float* w, bf16_t* out, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float r = w[i];
        const long blk = i >> 5, e = i & 31;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const bf16_t h = f2bf(r);
            out[blk * 96 + pl * 32 + e] = h;
            r = r - bf2f(h);
        }
    }
}

extern "C" int uv_split_weights_bf16x6(const float* w, void* out, long n, void* stream) {
    UV_CHECK_ARG(w && out && n > 0 && n % 32 == 0, "uv_split_weights_bf16x6: n must be a positive multiple of 32");
    hipLaunchKernelGGL(split_weights6_kernel, dim3((unsigned)min((n + 255) / 256, (long)4096)), dim3(256), 0, (hipStream_t)stream, w,
                       (bf16_t*)out, n);
    UV_CHECK_LAUNCH("uv_split_weights_bf16x6");
    return 0;
}

// Same convolution, f32-grade, in THREE fp16 MFMA passes (PREC 4 above). `in` holds PRE-SPLIT activations ([C/32][32 hi | 32 lo] IEEE fp16 per
// pixel, written by uv_vae_rms_silu with split_out = 2; ld_in counted in f32-sized elements = channels; |x| < 65 504), w_split = the
// weights split by uv_split_weights_f16x3 with the power-of-two `w_scale`; out = acc / w_scale (* *act_scale, a device scalar, when the
// activations were split by uv_vae_split_f16 under its own power-of-two scale) + bias (+ residual), f32.
extern "C" int uv_conv3d_f16x3(const float* in, long ld_in, int Tin, int Hin, int Win, const void* w_split, const float* bias,
                               float* out, long ldo, int Tout, int Hout, int Wout, int Cin, int Cout, int kt, int kh, int kw, int st,
                               int sh, int sw, int t_off, int ph, int pw, int up, int interleave, const float* resid, long ldr,
                               float w_scale, const float* act_scale, void* stream) {
    int ex = 0;
    UV_CHECK_ARG(w_scale > 0.f && frexpf(w_scale, &ex) == 0.5f, "uv_conv3d_f16x3: w_scale=%g must be a power of two", (double)w_scale);
    return conv_common(in, ld_in, Tin, Hin, Win, w_split, bias, out, ldo, Tout, Hout, Wout, Cin, Cout, kt, kh, kw, st, sh, sw, t_off,
                       ph, pw, up, interleave, resid, ldr, 4, stream, 1.0f / w_scale, act_scale);
}

// w [rows][K] f32 (K % 32 == 0) -> [rows][K/32][32 hi | 32 lo] IEEE fp16 with hi = fp16(w * scale), lo = fp16(w * scale - hi); scale is a
// power of two chosen by the caller so that max |w| * scale < 65 504 (round to nearest; hi + lo = w * scale to 2^-22 relative)
__global__ void split_weights_f16_kernel(const float* w, bf16_t* out, long n, float scale) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float x = w[i] * scale;
        const _Float16 h = (_Float16)x;
        const _Float16 l = (_Float16)(x - (float)h);
        const long blk = i >> 5, e = i & 31;
        out[blk * 64 + e] = __builtin_bit_cast(bf16_t, h);
        out[blk * 64 + 32 + e] = __builtin_bit_cast(bf16_t, l);
    }
}

extern "C" int uv_split_weights_f16x3(const float* w, void* out, long n, float scale, void* stream) {
    int ex = 0;
    UV_CHECK_ARG(w && out && n > 0 && n % 32 == 0, "uv_split_weights_f16x3: n must be a positive multiple of 32");
    UV_CHECK_ARG(scale > 0.f && frexpf(scale, &ex) == 0.5f, "uv_split_weights_f16x3: scale=%g must be a power of two", (double)scale);
    hipLaunchKernelGGL(split_weights_f16_kernel, dim3((unsigned)min((n + 255) / 256, (long)4096)), dim3(256), 0, (hipStream_t)stream, w,
                       (bf16_t*)out, n, scale);
    UV_CHECK_LAUNCH("uv_split_weights_f16x3");
    return 0;
}

// Same convolution with split-bf16 (3-pass) arithmetic. w_split: [Cout][K/32][32 hi | 32 lo] bf16 (uv_split_weights_bf16x3).
// in_split = 1: `in` already holds split activations ([C/32][32 hi | 32 lo] bf16 per pixel, written by uv_vae_rms_silu with
// split_out = 1; ld_in still counted in f32-sized elements = channels).
extern "C" int uv_conv3d_bf16x3(const float* in, long ld_in, int Tin, int Hin, int Win, const void* w_split,
                                const float* bias, float* out, long ldo, int Tout, int Hout, int Wout, int Cin, int Cout,
                                int kt, int kh, int kw, int st, int sh, int sw, int t_off, int ph, int pw, int up,
                                int interleave, const float* resid, long ldr, int in_split, void* stream) {
    return conv_common(in, ld_in, Tin, Hin, Win, w_split, bias, out, ldo, Tout, Hout, Wout, Cin, Cout, kt, kh, kw, st, sh, sw,
                       t_off, ph, pw, up, interleave, resid, ldr, in_split ? 2 : 1, stream);
}

// w [rows][K] f32 (K % 32 == 0) -> [rows][K/32][32 hi | 32 lo] bf16, hi = bf16(w), lo = bf16(w - hi)
__global__ void split_weights_kernel(const float* w, bf16_t* out, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float x = w[i];
        const bf16_t h = f2bf(x);
        const long blk = i >> 5, e = i & 31;
        out[blk * 64 + e] = h;
        out[blk * 64 + 32 + e] = f2bf(x - bf2f(h));
    }
}

extern "C" int uv_split_weights_bf16x3(const float* w, void* out, long n, void* stream) {
    UV_CHECK_ARG(w && out && n > 0 && n % 32 == 0, "uv_split_weights_bf16x3: n must be a positive multiple of 32");
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)min((n + 255) / 256, (long)4096)), dim3(256), 0, (hipStream_t)stream, w,
                       (bf16_t*)out, n);
    UV_CHECK_LAUNCH("uv_split_weights_bf16x3");
    return 0;
}
