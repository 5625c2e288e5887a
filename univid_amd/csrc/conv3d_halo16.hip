// The LDS-halo implicit GEMM of conv3d_halo.hip for the f32-grade THREE-pass fp16 arithmetic ("f16x3", PREC 4 of conv3d_f32.hip): the 3x3(x3)
// stride-1 convolutions behind an RMS_norm (ResidualBlocks: ~75 % of the decoder's time in that mode), both operands PRE-SPLIT into two
// IEEE fp16 pieces per element.
//
// Replaces (reference, fp32 nn.Conv3d through cuDNN): CausalConv3d.forward  models/wan/utils/modules/vae2_2.py:17-42 as used by
// ResidualBlock (vae2_2.py:193-235); arithmetic contract = conv3d_f32_kernel<.., 4> (x*w ~ xh*wh + xh*wl + xl*wh on
// v_mfma_f32_16x16x32_f16, f32 accumulate, out = acc * out_scale + bias (+ residual)); only the k ORDER of the sum differs (taps inside a
// channel block here, channel blocks inside a tap there), so the two agree to f32 summation noise, not bit for bit - the choice between
// them depends on per-FRAME geometry only, never on the pass length (pass-length independence is bit-exact and tested).
//
// Against conv3d_halo_kernel<3, 32> (bf16x6):
//   * the activations arrive pre-split ([C/32][32 hi | 32 lo] fp16 per pixel = the 128 bytes of 32 f32 channels, written by
//     uv_vae_rms_silu(split_out = 2)), so the halo image (8 x 32 patch + one-pixel rim = 340 pixels x 128 B) is filled by LDS-DMA straight
//     from global memory - no register staging, no conversion work in this kernel at all; out-of-frame pixels read the zero page;
//   * two halo images: the pieces of channel group g+1 are issued ONE PER TAP during taps 1..6 of group g;
//   * weights: [Cout][K/32][32 hi | 32 lo] fp16 of w * 2^s (uv_split_weights_f16x3), one k-tile = (tap, 32 channels) = 128 B per output
//     channel; the loop runs in STEPS of two taps (96 MFMAs per wave) with one barrier per step and the next step's two weight tiles in
//     flight behind them (four 16-KiB tiles of LDS);
//   * 8 waves = 4 (pixel rows) x 2 (output-channel halves), a wave owns 64 pixels x 64 channels: 16 accumulator fragments, 3 MFMAs each
//     per tap; fragment reads: 16 ds_read_b128 per 48 MFMAs and wave (a third of the LDS bandwidth), bank-conflict-free by the XOR
//     swizzle chunk ^= (row >> 1) & 7 on 128-byte rows (applied on the DMA's source address and on the read).
#include "conv_args.h"
#include <stdlib.h>

typedef __attribute__((address_space(3))) void lds_void_g;

// BN = output channels per workgroup: 128 with two-tap steps (four 16-KiB weight tiles), or 160 for the encoder's 160 / 320-channel stages with
// ONE tap per step (two 20-KiB tiles: 126 KiB of LDS; two-tap steps would need 166 KiB) - the step length was measured not to matter (§9 round 4).
// TW = patch width: 32 (8 x 32 patches) or 16 (16 x 16: for frames such as 45 x 80 that cut into fewer 16 x 16 than 8 x 32 patches - 15 against
// 18; a pixel's sum runs over (channel block, tap) in the same order whatever the patch, so the results do not depend on it).
template <int BN = 128, int TAPS = 2, int TW = 32>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3d_halo_f16_kernel(ConvArgs p) {
    constexpr int TH = 256 / TW, HW_ = TW + 2, NHP = (TH + 2) * HW_;         // 340 / 324 halo pixels
    constexpr int FPR = TW / 16, RPW = TH / 4;                               // 16-pixel fragments per patch row; patch rows per wave row group
    static_assert(TW == 32 || TW == 16, "patch width");
    constexpr int NW = 8, TM = 4, TN = BN / 32;
    constexpr int H_PIECES = (NHP + 7) / 8;                                  // 43 one-KiB pieces (8 pixels x 128 B)
    constexpr int HALO_BYTES = H_PIECES * 1024;                              // 44 032
    constexpr int H_INSTR = (H_PIECES + NW - 1) / NW;                        // 6 (waves 0..2) / 5
    constexpr int W_BYTES = BN * 128, W_PIECES = W_BYTES / 1024, W_INSTR = (W_PIECES + NW - 1) / NW;   // 16 / 20 KiB: 2 / 2-3 pieces per wave
    static_assert(BN == 128 || BN == 160, "output-channel tile");
    static_assert(H_INSTR <= 6, "one halo piece per tap during taps 1..6");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wbuf = smem + 2 * HALO_BYTES;         // 2 * TAPS weight tiles: two steps x TAPS taps

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_w = (p.Wout + TW - 1) / TW, tiles_h = (p.Hout + TH - 1) / TH;
    const int tile_n = blockIdx.x % p.tiles_n;
    int mt = blockIdx.x / p.tiles_n;
    const int tx0 = (mt % tiles_w) * TW;
    mt /= tiles_w;
    const int ty0 = (mt % tiles_h) * TH;
    const int tf = mt / tiles_h;                                             // output frame
    const int n0 = tile_n * BN;
    const int K = p.kt * 9 * p.Cin;
    const long frame = (long)p.Hin * p.Win * p.ld_in;

    // ---- halo staging map: piece pi = it * 8 + wave covers halo pixels 8 pi .. 8 pi + 7; lane -> pixel lane >> 3, physical chunk lane & 7
    long h_off[H_INSTR];             // f32-sized element offset of this lane's chunk inside a frame (channel block 0)
    unsigned h_ok = 0;               // bit it: inside the frame
#pragma unroll
    for (int it = 0; it < H_INSTR; ++it) {
        const int hp = (it * NW + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((hp >> 1) & 7);                          // logical 16-byte chunk stored at physical chunk lane & 7
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int y = ty0 + hy - 1, x = tx0 + hx - 1;                        // coordinates in the convolution's input image
        const bool ok = hp < NHP && (unsigned)y < (unsigned)p.Hout && (unsigned)x < (unsigned)p.Wout;
        h_ok |= (unsigned)ok << it;
        const int sy = p.up ? y >> 1 : y, sx = p.up ? x >> 1 : x;            // up: nearest-exact 2x (vae2_2.py:86-96)
        h_off[it] = ok ? ((long)sy * p.Win + sx) * p.ld_in + c * 4 : 0;
    }
    const int ncb = p.Cin >> 5, ngroups = p.kt * ncb;                        // group g = (dt, channel block cb)
    auto stage_halo_piece = [&](int g, int it, int hb) __attribute__((always_inline)) {
        if (it * NW + wave >= H_PIECES) return;
        const int dt = g / ncb, cb = g - dt * ncb;
        const int fi = tf * p.st + p.t_off + dt;
        const bool ok = (unsigned)fi < (unsigned)p.Tin && ((h_ok >> it) & 1u);
        const float* src = ok ? p.in + (long)fi * frame + cb * 32 + h_off[it] : p.zeros;
        __builtin_amdgcn_global_load_lds((const void*)src, (lds_void_g*)(smem + hb * HALO_BYTES + (it * NW + wave) * 1024), 16, 0, 0);
    };

    // ---- weight staging: piece = 8 output channels x 128 B of one k-tile
    const int srow = lane >> 3, pchunk = lane & 7;
    const float* w_src[W_INSTR];
#pragma unroll
    for (int i = 0; i < W_INSTR; ++i) {
        const int row = (i * NW + wave) * 8 + srow;
        w_src[i] = p.w + (long)min(n0 + row, p.Cout - 1) * K + (pchunk ^ ((row >> 1) & 7)) * 4;
    }
    auto stage_w = [&](int g, int tap, int wb) __attribute__((always_inline)) {
        const int dt = g / ncb, cb = g - dt * ncb;
        const int koff = ((dt * 9 + tap) * ncb + cb) * 32;                   // k order of the weight matrix: tap-major, channels minor
#pragma unroll
        for (int i = 0; i < W_INSTR; ++i)
            if (W_PIECES % NW == 0 || i * NW + wave < W_PIECES)
                __builtin_amdgcn_global_load_lds((const void*)(w_src[i] + koff), (lds_void_g*)(wbuf + wb * W_BYTES + (i * NW + wave) * 1024), 16, 0, 0);
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    int hpb[TM];                     // halo pixel of this lane's output pixel of fragment j for tap (0, 0)
#pragma unroll
    for (int j = 0; j < TM; ++j) hpb[j] = (wm * RPW + j / FPR) * HW_ + (j % FPR) * 16 + frow;
    int w_off[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int row = wn * (BN / 2) + i * 16 + frow;
        w_off[i] = row * 128 + ((fq ^ ((row >> 1) & 7)) << 4);               // hi piece; lo piece = the same address ^ 64
    }

    auto compute = [&](int tap, const char* wb, const char* halo) __attribute__((always_inline)) {
        const int dh = tap / 3, dw = tap - dh * 3;
        const int toff = dh * HW_ + dw;
        bf16x8 ah[TM], al[TM], wh[TN], wl[TN];
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int hp = hpb[j] + toff;
            const int o = hp * 128 + ((fq ^ ((hp >> 1) & 7)) << 4);
            ah[j] = *(const bf16x8*)(halo + o);
            al[j] = *(const bf16x8*)(halo + (o ^ 64));
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            wh[i] = *(const bf16x8*)(wb + w_off[i]);
            wl[i] = *(const bf16x8*)(wb + (w_off[i] ^ 64));
        }
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                acc[i][j] = mfma_16x16x32<true>(wl[i], ah[j], acc[i][j]);    // smallest terms first (as conv3d_f32_kernel PREC 4)
                acc[i][j] = mfma_16x16x32<true>(wh[i], al[j], acc[i][j]);
                acc[i][j] = mfma_16x16x32<true>(wh[i], ah[j], acc[i][j]);
            }
    };

    // ---- pipeline: STEPS of two taps, one barrier per step (a tap is only 48 MFMAs per wave: with a barrier and a weight wait per tap
    // ~225 of ~1 000 cycles per tap were overhead - bf16x6's 96-MFMA taps amortise the same cost twice as well). The taps of two
    // consecutive channel groups form one unrolled sequence of 18 taps = 9 steps (the host sends only even group counts here); the
    // weights of the NEXT step (2 x 16 KiB) and, during taps 1..6 of a group, one halo piece of the next group are in flight behind
    // the 96 MFMAs of a step.
    auto stage_w2 = [&](int g, int q, int sb) __attribute__((always_inline)) {      // the TAPS taps q .. of the 9 * TAPS-tap sequence starting at group g
#pragma unroll
        for (int u = 0; u < TAPS; ++u) {
            const int qq = q + u, gg = g + qq / 9;
            if (gg < ngroups) stage_w(gg, qq % 9, sb * TAPS + u);
        }
    };
#pragma unroll
    for (int it = 0; it < H_INSTR; ++it) stage_halo_piece(0, it, 0);
    stage_w2(0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int sb = 0;
    for (int g = 0; g < ngroups; g += TAPS) {
#pragma unroll
        for (int st = 0; st < 9; ++st) {
#pragma unroll
            for (int u = 0; u < TAPS; ++u) {
                // halo pieces of the group after each tap's own group, piece `tap - 1` at taps 1..6: the image they go to was last
                // read by tap 8 of the group BEFORE this one, which shares its step with this group's tap 0 when the group is the
                // second of the pair - so nothing is issued at tap 0, and every piece lands behind a barrier that follows that read
                const int q = TAPS * st + u, gg = g + q / 9, tap = q % 9;
                if (tap >= 1 && tap - 1 < H_INSTR && gg + 1 < ngroups) stage_halo_piece(gg + 1, tap - 1, (gg + 1) & 1);
            }
            stage_w2(g, TAPS * st + TAPS, sb ^ 1);     // (crosses into the next block of groups at st = 8)
#pragma unroll
            for (int u = 0; u < TAPS; ++u) {
                const int q = TAPS * st + u, gg = g + q / 9, tap = q % 9;
                compute(tap, wbuf + (sb * TAPS + u) * W_BYTES, smem + (gg & 1) * HALO_BYTES);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            sb ^= 1;
        }
    }

    // ---- epilogue: exact power-of-two descale, bias, optional residual, f32 store
    const float out_scale = p.act_scale ? p.out_scale * *p.act_scale : p.out_scale;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int py = wm * RPW + j / FPR, px = (j % FPR) * 16 + frow;
        const int y = ty0 + py, x = tx0 + px;
        if (y >= p.Hout || x >= p.Wout) continue;
        const long m = ((long)tf * p.Hout + y) * p.Wout + x;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int n = n0 + wn * (BN / 2) + i * 16 + 4 * fq;
            if (n >= p.Cout) continue;
            f32x4 v = acc[i][j];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= out_scale;
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

// ------------------------------------------------------------------------------------------------------------------------------------
// The same arithmetic for Cout <= 16: the decoder's head convolution (256 -> 12 channels on full-resolution frames, vae2_2.py:720-723).
// On the gather kernel that layer re-reads its 3.8 GB input once per tap through L2 (27 x: 102 GB per launch, 14 ms); here a workgroup
// stages the 8 x 32 patch + rim once per (frame tap, 32-channel block) and ALL NINE spatial taps' weights (9 x 16 rows x 128 B = 18 KiB)
// with it, so a group costs ONE barrier. 8 waves x one patch row each (32 pixels = 2 fragments x 16 output channels): 6 MFMAs and 6
// ds_read_b128 per tap and wave - the kernel is bound by the LDS reads of the activation fragments (each is used for one 16-channel
// fragment only), ~1 us per workgroup and group, an order of magnitude under the gather form's L2 traffic.
// ------------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void conv3d_halo_f16_n16_kernel(ConvArgs p) {
    constexpr int TW = 32, TH = 8, HW_ = TW + 2, NHP = (TH + 2) * HW_;
    constexpr int NW = 8;
    constexpr int H_PIECES = (NHP + 7) / 8, HALO_BYTES = H_PIECES * 1024, H_INSTR = (H_PIECES + NW - 1) / NW;
    constexpr int W_BYTES = 9 * 16 * 128, W_PIECES = 18, W_INSTR = (W_PIECES + NW - 1) / NW;     // 18 KiB per group, 3 pieces for waves 0..1
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wbuf = smem + 2 * HALO_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_w = (p.Wout + TW - 1) / TW, tiles_h = (p.Hout + TH - 1) / TH;
    int mt = blockIdx.x;
    const int tx0 = (mt % tiles_w) * TW;
    mt /= tiles_w;
    const int ty0 = (mt % tiles_h) * TH;
    const int tf = mt / tiles_h;
    const int K = p.kt * 9 * p.Cin;
    const long frame = (long)p.Hin * p.Win * p.ld_in;

    long h_off[H_INSTR];
    unsigned h_ok = 0;
#pragma unroll
    for (int it = 0; it < H_INSTR; ++it) {
        const int hp = (it * NW + wave) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((hp >> 1) & 7);
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int y = ty0 + hy - 1, x = tx0 + hx - 1;
        const bool ok = hp < NHP && (unsigned)y < (unsigned)p.Hout && (unsigned)x < (unsigned)p.Wout;
        h_ok |= (unsigned)ok << it;
        h_off[it] = ok ? ((long)y * p.Win + x) * p.ld_in + c * 4 : 0;
    }
    const int ncb = p.Cin >> 5, ngroups = p.kt * ncb;
    // weight pieces: piece pi = (tap, half): rows 8 half .. 8 half + 7 of the tap's 16-row tile
    const float* w_src[W_INSTR];
#pragma unroll
    for (int i = 0; i < W_INSTR; ++i) {
        const int pi = i * NW + wave, tap = pi >> 1, row = (pi & 1) * 8 + (lane >> 3);
        w_src[i] = p.w + (long)min(row, p.Cout - 1) * K + (long)tap * ncb * 32 + ((lane & 7) ^ ((row >> 1) & 7)) * 4;
    }
    auto stage = [&](int g, int b) __attribute__((always_inline)) {
        const int dt = g / ncb, cb = g - dt * ncb;
        const int fi = tf * p.st + p.t_off + dt;
        const bool fok = (unsigned)fi < (unsigned)p.Tin;
#pragma unroll
        for (int it = 0; it < H_INSTR; ++it) {
            if (it * NW + wave >= H_PIECES) continue;
            const bool ok = fok && ((h_ok >> it) & 1u);
            const float* src = ok ? p.in + (long)fi * frame + cb * 32 + h_off[it] : p.zeros;
            __builtin_amdgcn_global_load_lds((const void*)src, (lds_void_g*)(smem + b * HALO_BYTES + (it * NW + wave) * 1024), 16, 0, 0);
        }
        const int koff = (dt * 9 * ncb + cb) * 32;
#pragma unroll
        for (int i = 0; i < W_INSTR; ++i)
            if (i * NW + wave < W_PIECES)
                __builtin_amdgcn_global_load_lds((const void*)(w_src[i] + koff), (lds_void_g*)(wbuf + b * W_BYTES + (i * NW + wave) * 1024), 16, 0, 0);
    };

    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    const int frow = lane & 15, fq = lane >> 4;
    const int hpb0 = wave * HW_ + frow;                                       // fragment j: + 16 j
    const int w_off = frow * 128 + ((fq ^ ((frow >> 1) & 7)) << 4);

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int g = 0; g < ngroups; ++g) {
        const int b = g & 1;
        if (g + 1 < ngroups) stage(g + 1, b ^ 1);      // the other buffers were last read in group g-1 (barrier below)
        const char* halo = smem + b * HALO_BYTES;
        const char* wb = wbuf + b * W_BYTES;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = (tap / 3) * HW_ + tap % 3;
            const bf16x8 wh = *(const bf16x8*)(wb + tap * 2048 + w_off);
            const bf16x8 wl = *(const bf16x8*)(wb + tap * 2048 + (w_off ^ 64));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int hp = hpb0 + 16 * j + toff;
                const int o = hp * 128 + ((fq ^ ((hp >> 1) & 7)) << 4);
                const bf16x8 ah = *(const bf16x8*)(halo + o);
                const bf16x8 al = *(const bf16x8*)(halo + (o ^ 64));
                acc[j] = mfma_16x16x32<true>(wl, ah, acc[j]);
                acc[j] = mfma_16x16x32<true>(wh, al, acc[j]);
                acc[j] = mfma_16x16x32<true>(wh, ah, acc[j]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }

    const float out_scale = p.act_scale ? p.out_scale * *p.act_scale : p.out_scale;
    const int y = ty0 + wave, n = 4 * fq;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int x = tx0 + 16 * j + frow;
        if (y >= p.Hout || x >= p.Wout || n >= p.Cout) continue;
        const long m = ((long)tf * p.Hout + y) * p.Wout + x;
        f32x4 v = acc[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= out_scale;
        if (p.bias) {
            const f32x4 bb = *(const f32x4*)(p.bias + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bb[e];
        }
        if (p.resid) {
            const f32x4 rr = *(const f32x4*)(p.resid + m * p.ldr + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += rr[e];
        }
        *(f32x4*)(p.out + m * p.ldo + n) = v;
    }
}

// Which f16x3 convolutions take this kernel: the geometry rule of uv_conv3d_halo_eligible (3x3 spatial taps, stride 1, padding 1, plain or
// behind the 2x upsampling, whole 32-channel input blocks and 128-wide output tiles) and enough tiles PER FRAME to fill the chip at four
// frames per pass. uv_set_option(UV_OPT_CONV_HALO, 0 | 1) forces never / whenever the geometry fits, as for the other arithmetics.
// 16 x 16 patches where a frame cuts into FEWER of them than 8 x 32 ones (45 x 80: 3 x 5 = 15 against 6 x 3 = 18; 360 x 640: 920 against 900). A
// per-frame rule, and the results do not depend on the patch shape anyway (the kernel's note on TW).
static bool halo16_square_patches(const ConvArgs& a) {
    return (long)((a.Hout + 15) / 16) * ((a.Wout + 15) / 16) < (long)((a.Hout + 7) / 8) * ((a.Wout + 31) / 32);
}

bool uv_conv3d_halo16_eligible(const ConvArgs& a) {
    const int force = uv_option(UV_OPT_CONV_HALO);
    if (force == 0) return false;
    if (a.kh != 3 || a.kw != 3 || (a.kt != 3 && a.kt != 1)) return false;
    if (a.st != 1 || a.sh != 1 || a.sw != 1 || a.ph != 1 || a.pw != 1 || a.interleave) return false;
    const int mul = a.up ? 2 : 1;
    if (a.Hin * mul != a.Hout || a.Win * mul != a.Wout || a.Cin % 32 != 0) return false;
    if (a.Cout <= 16) {      // the narrow-output kernel (the decoder's head): plain geometry only, Cout a multiple of 4
        const long tiles = (long)((a.Hout + 7) / 8) * ((a.Wout + 31) / 32);
        return !a.up && a.Cout % 4 == 0 && (force == 1 || 4 * tiles >= uv_num_cus());
    }
    // 128-wide output tiles (two-tap steps pair the channel groups: even counts only), or whole 160-wide ones (the encoder's 160 / 320-channel stages)
    const int bn = a.Cout % 128 == 0 ? 128 : a.Cout % 160 == 0 ? 160 : 0;
    if (bn == 0 || (bn == 128 && ((a.kt * (a.Cin >> 5)) & 1))) return false;
    const long patches = bn == 128 && halo16_square_patches(a) ? (long)((a.Hout + 15) / 16) * ((a.Wout + 15) / 16) : (long)((a.Hout + 7) / 8) * ((a.Wout + 31) / 32);
    return force == 1 || 4 * patches * (a.Cout / bn) >= uv_num_cus();
}

int uv_launch_conv3d_halo16(ConvArgs& a, hipStream_t stream) {
    if (a.Cout <= 16) {
        a.tiles_n = 1;
        a.tiles_m = a.Tout * ((a.Hout + 7) / 8) * ((a.Wout + 31) / 32);
        const size_t lds = 2 * 43 * 1024 + 2 * 9 * 16 * 128;                  // two halo images + two 9-tap weight sets = 122 KiB
        UV_ONCE_PER_DEVICE(hipFuncSetAttribute((const void*)conv3d_halo_f16_n16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(conv3d_halo_f16_n16_kernel, dim3(a.tiles_m), dim3(512), lds, stream, a);
        return 0;
    }
    a.tiles_m = a.Tout * ((a.Hout + 7) / 8) * ((a.Wout + 31) / 32);
    if (a.Cout % 128 != 0) {
        a.tiles_n = a.Cout / 160;
        const size_t lds160 = 2 * 43 * 1024 + 2 * 160 * 128;                 // two halo images + two 20-KiB weight tiles = 126 KiB
        UV_ONCE_PER_DEVICE(hipFuncSetAttribute((const void*)conv3d_halo_f16_kernel<160, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds160));
        hipLaunchKernelGGL((conv3d_halo_f16_kernel<160, 1>), dim3(a.tiles_m * a.tiles_n), dim3(512), lds160, stream, a);
        return 0;
    }
    a.tiles_n = a.Cout / 128;
    const size_t lds = 2 * 43 * 1024 + 4 * 128 * 128;                        // two halo images + four weight tiles = 150 KiB
    if (halo16_square_patches(a)) {
        a.tiles_m = a.Tout * ((a.Hout + 15) / 16) * ((a.Wout + 15) / 16);
        UV_ONCE_PER_DEVICE(hipFuncSetAttribute((const void*)conv3d_halo_f16_kernel<128, 2, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((conv3d_halo_f16_kernel<128, 2, 16>), dim3(a.tiles_m * a.tiles_n), dim3(512), lds, stream, a);
        return 0;
    }
    UV_ONCE_PER_DEVICE(hipFuncSetAttribute((const void*)conv3d_halo_f16_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((conv3d_halo_f16_kernel<128>), dim3(a.tiles_m * a.tiles_n), dim3(512), lds, stream, a);
    return 0;
}
