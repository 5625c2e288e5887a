// HBM-bound glue kernels of the Wan DiT forward, each a single fused pass (one read + one write of
// the [L, C] activation), vectorised 16 B/lane, one 64-lane wave per token row.
//
// Replaces (reference, all PyTorch eager):
//   WanLayerNorm + AdaLN modulate   models/wan/utils/modules/model.py:93-98, 239-245, 253, 287-290
//   WanRMSNorm (QK-norm over dim)   models/wan/utils/modules/model.py:82-85, 138-139, 170-171
//   rope_apply (complex128 RoPE)    models/wan/utils/modules/model.py:38-66
//   patch_embedding im2col / unpatchify   model.py:448-451, 499-522
//   sinusoidal_embedding_1d + time MLPs   model.py:14-24, 384-386, 460-469
#include "common.h"
#include <type_traits>
#ifndef UV_LN_NT
#define UV_LN_NT 1
#endif

// ------------------------------------------------------------------------------------------------
// LayerNorm (no affine, eps) over C, then one of:
//   mode 0: y                                  (plain)
//   mode 1: y * (1 + scale[t]) + shift[t]      (AdaLN; t = tid[row], rows of a [n_t, tab_stride] table)
//   mode 2: y * w + b                          (elementwise affine, norm3)
// `round_ln` rounds y to bf16 first (block 0: the residual stream is still bf16 there, and
// WanLayerNorm.forward does .type_as(x), model.py:98). Output bf16 or f32.
// Every product/sum keeps the reference's separate roundings (no FMA contraction).
// ------------------------------------------------------------------------------------------------
struct LnArgs {
    const float* x; long ldx;
    void* out; long ldo;
    const float* tab; long tab_stride; int shift_off, scale_off;  // mode 1
    const int32_t* tid;                                           // mode 1 (nullptr => row 0)
    const float* w; const float* b;                               // mode 2
    int L, C; float eps;
    int mode, round_ln, out_bf16;
};

// A 4-wave block walks 4*RPW consecutive token rows, one row per wave at a time. The modulation (or affine) parameters of
// the block's first row are staged in LDS once and read from there by every row with the same table index (the usual
// case: a sample's tokens share one timestep); the parameter reads are otherwise 2x the x reads in load instructions.
// EXACT: C == MAXV * 256, every lane owns MAXV float4 chunks: no per-chunk predicates, so a row's loads are ONE burst behind ONE wait
// (behind the `i < nv` predicates hipcc emitted a load and an s_waitcnt vmcnt(0) per chunk: twelve dependent HBM round trips per row).
// SMODE >= 0: the mode is a compile-time constant, the output is bf16 and the normalised value is not rounded before the modulation (the
// DiT's per-block calls): the output pass is straight-line code - with mode / output type / rounding as run-time branches inside the
// chunk loop every chunk was its own basic block with its own waits.
template <int MAXV, int RPW, bool EXACT = false, int SMODE = -1>
__global__ __launch_bounds__(256) void layernorm_mod_kernel(LnArgs p) {
    if constexpr (SMODE >= 0) { p.mode = SMODE; p.out_bf16 = 1; p.round_ln = 0; }
    extern __shared__ __attribute__((aligned(16))) char ln_smem[];
    f32x4* sp = (f32x4*)ln_smem;                 // [2][nv*64]: scale|shift (mode 1) or w|b (mode 2)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nv = EXACT ? MAXV : p.C >> 8;       // float4 per lane
    const int brow0 = blockIdx.x * 4 * RPW;
    int t_blk = 0;
    if (p.mode != 0) {
        const float *a0, *b0;
        if (p.mode == 1) {
            t_blk = p.tid ? p.tid[brow0] : 0;
            a0 = p.tab + (long)t_blk * p.tab_stride + p.scale_off;
            b0 = p.tab + (long)t_blk * p.tab_stride + p.shift_off;
        } else {
            a0 = p.w;
            b0 = p.b;
        }
        for (int i = threadIdx.x; i < nv * 64; i += 256) {
            sp[i] = *(const f32x4*)(a0 + i * 4);
            sp[nv * 64 + i] = *(const f32x4*)(b0 + i * 4);
        }
        __syncthreads();
    }
    for (int r = 0; r < RPW; ++r) {
        const int row = brow0 + r * 4 + wave;
        if (row >= p.L) break;
        const float* xr = p.x + (long)row * p.ldx;
        f32x4 v[MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
            if (EXACT || i < nv) {
                v[i] = UV_LN_NT ? __builtin_nontemporal_load((const f32x4*)(xr + (i * 64 + lane) * 4)) : *(const f32x4*)(xr + (i * 64 + lane) * 4);
                s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
            }
        const int t = (p.mode == 1 && p.tid) ? __builtin_amdgcn_readfirstlane(p.tid[row]) : 0;
        const bool from_lds = p.mode == 2 || t == t_blk;
        const float* scale = p.tab + (long)t * p.tab_stride + p.scale_off;   // used only when the row's index differs
        const float* shift = p.tab + (long)t * p.tab_stride + p.shift_off;
        const float mean = wave_sum(s) / (float)p.C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
            if (EXACT || i < nv) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[i][e] - mean;
                    q += d * d;
                }
            }
        const float var = wave_sum(q) / (float)p.C;
        const float rstd = 1.0f / sqrtf(var + p.eps);
        // the output pass, once per parameter source: with the LDS / global choice INSIDE the chunk loop every chunk carried a possible
        // global load, and hipcc waited vmcnt(0) - i.e. for the previous chunk's STORE - in front of each one
        auto emit = [&](auto lds_tag) __attribute__((always_inline)) {
            constexpr bool FROM_LDS = decltype(lds_tag)::value;
#pragma unroll
            for (int i = 0; i < MAXV; ++i)
                if (EXACT || i < nv) {
                    const int c = (i * 64 + lane) * 4;
                    f32x4 y;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float tt = __fmul_rn(v[i][e] - mean, rstd);
                        if (p.round_ln) tt = round_bf(tt);
                        y[e] = tt;
                    }
                    if (p.mode != 0) {
                        f32x4 pa, pb;
                        if constexpr (FROM_LDS) {
                            pa = sp[i * 64 + lane];
                            pb = sp[nv * 64 + i * 64 + lane];
                        } else {
                            pa = *(const f32x4*)(scale + c);
                            pb = *(const f32x4*)(shift + c);
                        }
                        if (p.mode == 1) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) y[e] = __fadd_rn(__fmul_rn(y[e], __fadd_rn(1.0f, pa[e])), pb[e]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) y[e] = __fadd_rn(__fmul_rn(y[e], pa[e]), pb[e]);
                        }
                    }
                    if (p.out_bf16 == 2) {   // IEEE fp16 output (SigLIP2 ranker in its reference dtype)
                        u32x2 o = {pack16_2<true>(y[0], y[1]), pack16_2<true>(y[2], y[3])};
                        *(u32x2*)((bf16_t*)p.out + (long)row * p.ldo + c) = o;
                    } else if (p.out_bf16) {
                        u32x2 o = {pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3])};
                        if (UV_LN_NT) __builtin_nontemporal_store(o, (u32x2*)((bf16_t*)p.out + (long)row * p.ldo + c));
                        else *(u32x2*)((bf16_t*)p.out + (long)row * p.ldo + c) = o;
                    } else {
                        *(f32x4*)((float*)p.out + (long)row * p.ldo + c) = y;
                    }
                }
        };
        if (from_lds) emit(std::true_type{});
        else emit(std::false_type{});
    }
}

extern "C" int uv_layernorm_mod(const float* x, long ldx, void* out, long ldo, int L, int C, float eps,
                                int mode, const float* tab, long tab_stride, int shift_off, int scale_off,
                                const int32_t* tid, const float* w, const float* b, int round_ln,
                                int out_bf16, void* stream) {
    UV_CHECK_ARG(x && out && L > 0, "uv_layernorm_mod: null pointer / empty");
    UV_CHECK_ARG(C % 256 == 0 && C <= 8192, "uv_layernorm_mod: C=%d must be a multiple of 256 and <= 8192", C);
    UV_CHECK_ARG(ldx % 4 == 0 && ldo % 4 == 0, "uv_layernorm_mod: ldx/ldo must be multiples of 4");
    UV_CHECK_ARG(mode >= 0 && mode <= 2, "uv_layernorm_mod: bad mode %d", mode);
    if (mode == 1) UV_CHECK_ARG(tab && tab_stride % 4 == 0 && shift_off % 4 == 0 && scale_off % 4 == 0,
                                "uv_layernorm_mod: modulation table missing / misaligned");
    if (mode == 2) UV_CHECK_ARG(w && b, "uv_layernorm_mod: affine weight/bias missing");
    LnArgs a{x, ldx, out, ldo, tab, tab_stride, shift_off, scale_off, tid, w, b, L, C, eps, mode, round_ln, out_bf16};
    // one row per wave: measured 4.2-4.4 TB/s; 2 or 4 rows per wave run the waves of a CU in lockstep (all loading, then
    // all storing) and lose 15-30 %, so the LDS staging only trims the parameter reads to one copy per 4 rows
    const dim3 grid((L + 3) / 4), block(256);
    const size_t lds = mode == 0 ? 0 : (size_t)2 * (C / 4) * 16;
    hipStream_t st = (hipStream_t)stream;
    const bool plain_bf16 = out_bf16 == 1 && !round_ln;
    if (C == 3072 && plain_bf16 && mode == 1) hipLaunchKernelGGL((layernorm_mod_kernel<12, 1, true, 1>), grid, block, lds, st, a);   // the DiT:
    else if (C == 3072 && plain_bf16 && mode == 2) hipLaunchKernelGGL((layernorm_mod_kernel<12, 1, true, 2>), grid, block, lds, st, a);   // AdaLN / norm3
    else if (C == 3072) hipLaunchKernelGGL((layernorm_mod_kernel<12, 1, true>), grid, block, lds, st, a);
    else if (C == 768) hipLaunchKernelGGL((layernorm_mod_kernel<3, 1, true>), grid, block, lds, st, a);       // the SigLIP2 towers
    else if (C == 1024) hipLaunchKernelGGL((layernorm_mod_kernel<4, 1, true>), grid, block, lds, st, a);
    else if (C <= 1024) hipLaunchKernelGGL((layernorm_mod_kernel<4, 1>), grid, block, lds, st, a);
    else if (C <= 3072) hipLaunchKernelGGL((layernorm_mod_kernel<12, 1>), grid, block, lds, st, a);
    else if (C <= 4096) hipLaunchKernelGGL((layernorm_mod_kernel<16, 1>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((layernorm_mod_kernel<32, 1>), grid, block, lds, st, a);
    UV_CHECK_LAUNCH("uv_layernorm_mod");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// QK RMSNorm over the whole projection width C (= heads*head_dim), then optional 3-axis RoPE.
//   y  = bf16( x * rsqrt(mean(x^2) + eps) )            WanRMSNorm._norm(x.float()).type_as(x)
//   y  = float(y) * weight                             (fp32 parameter promotes to fp32)
//   RoPE: pairs (y[2i], y[2i+1]) of every head times freqs[pos][i] in complex128, result -> f32
//   out = bf16(...)                                    flash_attention's half() cast
// freqs: [max_pos, D/2] complex128 as (re, im) doubles; the first nf complex columns rotate with the
// frame index, the next nh with the row index, the last nw with the column index.
// Rows >= F*Hh*Ww (sequence padding) pass through un-rotated, as rope_apply does (model.py:62).
// ------------------------------------------------------------------------------------------------
struct RmsRopeArgs {
    const bf16_t* x; long ldx;
    bf16_t* out; long ldo;
    const float* weight;
    const bf16_t* x2;     // second tensor of the launch (blockIdx.y == 1): k beside q, same geometry; nullptr = none
    bf16_t* out2;
    const float* weight2;
    const double* freqs;  // nullptr => no rope
    int L, C, D;          // D = head_dim
    int Ls;               // rows per sample: stacked samples restart their RoPE positions every Ls rows
    int F, Hh, Ww;        // token grid
    int nf, nh, nw;       // complex columns per axis
    int row0;             // global token index of row 0 (sequence-parallel shards), RoPE positions only
    float eps;
};

// One wave walks RPW consecutive token rows; the norm weight of its columns stays in registers. A lane's 8-element
// chunks sit 512 columns apart, so when head_dim divides 512 they all map to the same 4 complex RoPE columns of a head:
// the factors are fetched once per row instead of once per chunk.
// EXACT: C == MAXV * 512, every lane owns MAXV full chunks: no per-chunk predicates, so a row's loads are ONE burst behind ONE wait
// (with the predicates hipcc branches around every chunk's load and waits vmcnt(0) chunk by chunk).
template <int MAXV, int RPW, bool EXACT = false>
__global__ __launch_bounds__(256) void rmsnorm_rope_kernel(RmsRopeArgs p) {
    const int lane = threadIdx.x & 63;
    // the wave index as a provably uniform value: the row, its RoPE position (three integer divisions) and the rope / no-rope branch
    // then live on the scalar unit instead of costing vector instructions and registers in every lane
    const int row0 = (blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * RPW;
    if (row0 >= p.L) return;
    if (blockIdx.y == 1) { p.x = p.x2; p.out = p.out2; p.weight = p.weight2; }
    const int nv = EXACT ? MAXV : p.C >> 9;      // 8-element chunks per lane (64 lanes * 8 = 512)
    const int rem = EXACT ? 0 : (p.C & 511) >> 3;   // leftover chunks (C % 512 != 0, e.g. C = 256)
    const int half = p.D >> 1;
    const bool same_cols = EXACT || (512 % p.D) == 0;      // EXACT launches require it: no per-chunk factor fetches, no branches
    f32x4 wv[MAXV][2];
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (EXACT || (i < nv) || (i == nv && lane < rem)) {
            wv[i][0] = *(const f32x4*)(p.weight + (i * 64 + lane) * 8);
            wv[i][1] = *(const f32x4*)(p.weight + (i * 64 + lane) * 8 + 4);
        }
    typedef __attribute__((ext_vector_type(2))) double f64x2;
    for (int rr = 0; rr < RPW; ++rr) {
        const int row = row0 + rr;
        if (row >= p.L) break;
        const bf16_t* xr = p.x + (long)row * p.ldx;
        const int grow = row % p.Ls + p.row0;        // token index in its sample's whole sequence
        const bool do_rope = p.freqs != nullptr && grow < p.F * p.Hh * p.Ww;
        int pf = 0, ph = 0, pw = 0;
        if (do_rope) {
            pw = grow % p.Ww;
            const int t = grow / p.Ww;
            ph = t % p.Hh;
            pf = t / p.Hh;
        }
        // this lane's 4 complex128 RoPE factors (valid for every chunk when same_cols): requested FIRST, as one burst with the row's
        // chunks behind them (fetched one by one next to their conversion they were four dependent L2 round trips per row)
        f64x2 fraw[4];
        if (do_rope && same_cols) {
            const int pair0 = ((lane * 8) % p.D) >> 1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ci = pair0 + e;
                const int pos = ci < p.nf ? pf : (ci < p.nf + p.nh ? ph : pw);
                fraw[e] = *(const f64x2*)(p.freqs + ((long)pos * half + ci) * 2);
            }
        }
        u32x4 raw[MAXV];                 // the row stays packed (bf16 pairs) between the two passes: half the registers of f32
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const bool on = EXACT || (i < nv) || (i == nv && lane < rem);
            if (on) {
                raw[i] = *(const u32x4*)(xr + (i * 64 + lane) * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = bf2f((bf16_t)(raw[i][e] & 0xffff)), hi = bf2f((bf16_t)(raw[i][e] >> 16));
                    ss += lo * lo;
                    ss += hi * hi;
                }
            }
        }
        // each double factor split into an f32 head and tail (cr = crh + crl to ~2^-49)
        float fcrh[4], fcrl[4], fcih[4], fcil[4];
        auto split_factor = [](const f64x2 f, float& rh, float& rl, float& ih, float& il) {
            const double cr = f[0], ci = f[1];
            rh = (float)cr; rl = (float)(cr - (double)rh);
            ih = (float)ci; il = (float)(ci - (double)ih);
        };
        if (do_rope && same_cols) {
#pragma unroll
            for (int e = 0; e < 4; ++e) split_factor(fraw[e], fcrh[e], fcrl[e], fcih[e], fcil[e]);
        }
        const float mean = wave_sum(ss) / (float)p.C;
        const float rs = 1.0f / sqrtf(mean + p.eps);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const bool on = EXACT || (i < nv) || (i == nv && lane < rem);
            if (on) {
                const int c0 = (i * 64 + lane) * 8;
                float y[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float ve = bf2f((bf16_t)((e & 1) ? raw[i][e >> 1] >> 16 : raw[i][e >> 1] & 0xffff));
                    y[e] = __fmul_rn(round_bf(__fmul_rn(ve, rs)), wv[i][e >> 2][e & 3]);
                }
                if (do_rope) {
                    const int pair0 = (c0 % p.D) >> 1;  // first complex index of this chunk inside its head
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float rh, rl, ih, il;
                        if (same_cols) {
                            rh = fcrh[e]; rl = fcrl[e]; ih = fcih[e]; il = fcil[e];
                        } else {
                            const int ci = pair0 + e;
                            const int pos = ci < p.nf ? pf : (ci < p.nf + p.nh ? ph : pw);
                            split_factor(*(const f64x2*)(p.freqs + ((long)pos * half + ci) * 2), rh, rl, ih, il);
                        }
                        // (a + i b)(cr + i ci) of the reference's complex128 product, then .float(): evaluated in f32 with the factor
                        // tails carried, innermost (smallest) terms first. a and b are f32 exactly. Error bound (round-3 advisor): every
                        // fused multiply-add rounds its running sum to f32, so the ABSOLUTE error of a component is <= 2^-24 (|a cr| + |b ci|)
                        // (+ second-order terms) - NOT "1 f32 ulp of float(fp64 product)": under cancellation (a cr ~ b ci) that is many
                        // ulps of the small result. For the bf16 rounding that follows what matters is the absolute error against the bf16
                        // spacing of the OUTPUT, and the output's rms is that of the inputs (a unit rotation): the rounded value differs
                        // from the reference's complex128 path only when the exact result lies within ~2^-16 relative to the pair's larger
                        // element of a bf16 rounding boundary - measured 1.3e-5 of the elements (tests/manual/rope_ulp_stats.py; plain f32
                        // factors without the tails: 1.9e-5; fp64 arithmetic: 1.0e-6, the sum-of-squares order alone), never more than one
                        // bf16 ulp of the rotated pair's larger element, which is the test's gate (rare=(2e-5, 2 ulp)). Was: four f64
                        // multiplies + two f64 adds + four conversions per pair.
                        const float a = y[2 * e], b = y[2 * e + 1];
                        y[2 * e] = __builtin_fmaf(a, rh, __builtin_fmaf(-b, ih, __builtin_fmaf(a, rl, -__fmul_rn(b, il))));
                        y[2 * e + 1] = __builtin_fmaf(a, ih, __builtin_fmaf(b, rh, __builtin_fmaf(a, il, __fmul_rn(b, rl))));
                    }
                }
                u32x4 o = {pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3]), pack_bf2(y[4], y[5]), pack_bf2(y[6], y[7])};
                *(u32x4*)(p.out + (long)row * p.ldo + c0) = o;
            }
        }
    }
}

static int rmsnorm_rope_launch(const char* name, const void* x, void* out, const float* weight, const void* x2, void* out2,
                               const float* weight2, long ldx, long ldo, int L, int Ls, int C, int head_dim, float eps,
                               const double* freqs, int F, int Hh, int Ww, int row0, void* stream) {
    UV_CHECK_ARG(x && out && weight && L > 0, "%s: null pointer / empty", name);
    UV_CHECK_ARG(C % 8 == 0 && C <= 8192, "%s: C=%d must be a multiple of 8 and <= 8192", name, C);
    UV_CHECK_ARG(head_dim % 8 == 0 && C % head_dim == 0, "%s: bad head_dim %d", name, head_dim);
    UV_CHECK_ARG(ldx % 8 == 0 && ldo % 8 == 0, "%s: ldx/ldo must be multiples of 8", name);
    UV_CHECK_ARG(row0 >= 0 && Ls > 0, "%s: negative row offset / bad rows per sample", name);
    if (freqs) UV_CHECK_ARG(F > 0 && Hh > 0 && Ww > 0 && F <= 1024 && Hh <= 1024 && Ww <= 1024,
                            "%s: grid (%d,%d,%d) outside the 1024-row RoPE table", name, F, Hh, Ww);
    RmsRopeArgs a;
    a.x = (const bf16_t*)x; a.ldx = ldx; a.out = (bf16_t*)out; a.ldo = ldo; a.weight = weight; a.freqs = freqs;
    a.x2 = (const bf16_t*)x2; a.out2 = (bf16_t*)out2; a.weight2 = weight2;
    a.L = L; a.Ls = Ls; a.C = C; a.D = head_dim; a.F = F; a.Hh = Hh; a.Ww = Ww; a.eps = eps; a.row0 = row0;
    const int c = head_dim / 2;  // model.py:43  split [c - 2*(c//3), c//3, c//3]
    a.nh = c / 3; a.nw = c / 3; a.nf = c - 2 * (c / 3);
    const int rpw = L >= 4096 ? 4 : 1;
    const dim3 grid((L + 4 * rpw - 1) / (4 * rpw), x2 ? 2 : 1), block(256);
    const int chunks = (C + 511) / 512;
    hipStream_t st = (hipStream_t)stream;
    // whole 512-column chunks and RoPE factors shared by a lane's chunks (or no RoPE): the predicate-free instantiations
    const bool exact = C % 512 == 0 && (!freqs || 512 % head_dim == 0);
    if (rpw == 4) {
        if (exact && chunks == 2) hipLaunchKernelGGL((rmsnorm_rope_kernel<2, 4, true>), grid, block, 0, st, a);
        else if (exact && chunks == 6) hipLaunchKernelGGL((rmsnorm_rope_kernel<6, 4, true>), grid, block, 0, st, a);   // the DiT's 3072
        else if (exact && chunks == 8) hipLaunchKernelGGL((rmsnorm_rope_kernel<8, 4, true>), grid, block, 0, st, a);   // umT5's 4096
        else if (chunks <= 2) hipLaunchKernelGGL((rmsnorm_rope_kernel<2, 4>), grid, block, 0, st, a);
        else if (chunks <= 6) hipLaunchKernelGGL((rmsnorm_rope_kernel<6, 4>), grid, block, 0, st, a);
        else if (chunks <= 8) hipLaunchKernelGGL((rmsnorm_rope_kernel<8, 4>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((rmsnorm_rope_kernel<16, 4>), grid, block, 0, st, a);
    } else {
        if (exact && chunks == 6) hipLaunchKernelGGL((rmsnorm_rope_kernel<6, 1, true>), grid, block, 0, st, a);
        else if (exact && chunks == 8) hipLaunchKernelGGL((rmsnorm_rope_kernel<8, 1, true>), grid, block, 0, st, a);
        else if (chunks <= 2) hipLaunchKernelGGL((rmsnorm_rope_kernel<2, 1>), grid, block, 0, st, a);
        else if (chunks <= 8) hipLaunchKernelGGL((rmsnorm_rope_kernel<8, 1>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((rmsnorm_rope_kernel<16, 1>), grid, block, 0, st, a);
    }
    UV_CHECK_LAUNCH(name);
    return 0;
}

extern "C" int uv_rmsnorm_rope(const void* x, long ldx, void* out, long ldo, const float* weight, int L, int C,
                               int head_dim, float eps, const double* freqs, int F, int Hh, int Ww, int row0,
                               void* stream) {
    return rmsnorm_rope_launch("uv_rmsnorm_rope", x, out, weight, nullptr, nullptr, nullptr, ldx, ldo, L, L, C, head_dim, eps, freqs,
                               F, Hh, Ww, row0, stream);
}

// q and k of a self-attention in ONE launch (same geometry, each with its own norm weight), over `L` rows holding L / Ls stacked
// samples whose RoPE positions restart every Ls rows: 4x the rows per launch of the per-tensor, per-sample form.
extern "C" int uv_rmsnorm_rope_qk(const void* q, void* q_out, const float* q_weight, const void* k, void* k_out,
                                  const float* k_weight, long ldx, long ldo, int L, int Ls, int C, int head_dim, float eps,
                                  const double* freqs, int F, int Hh, int Ww, int row0, void* stream) {
    UV_CHECK_ARG(k && k_out && k_weight, "uv_rmsnorm_rope_qk: null pointer");
    UV_CHECK_ARG(Ls > 0 && L % Ls == 0, "uv_rmsnorm_rope_qk: L=%d must be a whole number of samples of Ls=%d rows", L, Ls);
    return rmsnorm_rope_launch("uv_rmsnorm_rope_qk", q, q_out, q_weight, k, k_out, k_weight, ldx, ldo, L, Ls, C, head_dim, eps,
                               freqs, F, Hh, Ww, row0, stream);
}

// ------------------------------------------------------------------------------------------------
// patch_embedding im2col: latent [Cin, F, H, W] f32 -> rows [L, Kpad] bf16, column = (c, kt, kh, kw)
// flattened like Conv3d.weight.flatten(1) (model.py:378-379, 448-451); token = (f', h', w') row-major.
// Columns >= Cin*pt*ph*pw (K padding to the GEMM's 64 granularity) are zero.
// ------------------------------------------------------------------------------------------------
__global__ void patchify_kernel(const float* x, bf16_t* out, long ldo, int Cin, int F, int H, int W,
                                int pt, int ph, int pw, int Kpad) {
    const int Fp = F / pt, Hp = H / ph, Wp = W / pw;
    const long total = (long)Fp * Hp * Wp * Kpad;
    const int K = Cin * pt * ph * pw;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kpad);
        const long tok = i / Kpad;
        float v = 0.f;
        if (k < K) {
            const int kw = k % pw, kh = (k / pw) % ph, kt = (k / (pw * ph)) % pt, c = k / (pw * ph * pt);
            const int wq = (int)(tok % Wp), hq = (int)((tok / Wp) % Hp), fq = (int)(tok / ((long)Wp * Hp));
            v = x[(((long)c * F + fq * pt + kt) * H + hq * ph + kh) * W + wq * pw + kw];
        }
        out[tok * ldo + k] = f2bf(v);
    }
}

extern "C" int uv_patchify_bf16(const float* x, void* out, long ldo, int Cin, int F, int H, int W, int pt,
                                int ph, int pw, int Kpad, void* stream) {
    UV_CHECK_ARG(x && out, "uv_patchify_bf16: null pointer");
    UV_CHECK_ARG(pt > 0 && ph > 0 && pw > 0 && F % pt == 0 && H >= ph && W >= pw, "uv_patchify_bf16: bad patch");
    UV_CHECK_ARG(Kpad >= Cin * pt * ph * pw && ldo >= Kpad, "uv_patchify_bf16: Kpad/ldo too small");
    const long total = (long)(F / pt) * (H / ph) * (W / pw) * Kpad;
    const int blocks = (int)min((total + 255) / 256, (long)4096);
    hipLaunchKernelGGL(patchify_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)out, ldo,
                       Cin, F, H, W, pt, ph, pw, Kpad);
    UV_CHECK_LAUNCH("uv_patchify_bf16");
    return 0;
}

// unpatchify: head output [L, pt*ph*pw*Cout] f32 -> [Cout, Fp*pt, Hp*ph, Wp*pw] f32
// einsum 'fhwpqrc->cfphqwr' (model.py:518-520): column = ((p*ph + q)*pw + r)*Cout + c
__global__ void unpatchify_kernel(const float* in, long ldi, float* out, int Cout, int Fp, int Hp, int Wp, int pt,
                                  int ph, int pw) {
    const int F = Fp * pt, H = Hp * ph, W = Wp * pw;
    const long total = (long)Cout * F * H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int w = (int)(i % W), hh = (int)((i / W) % H), f = (int)((i / ((long)W * H)) % F);
        const int c = (int)(i / ((long)W * H * F));
        const int wq = w / pw, r = w % pw, hq = hh / ph, q = hh % ph, fq = f / pt, pp = f % pt;
        const long tok = ((long)fq * Hp + hq) * Wp + wq;
        out[i] = in[tok * ldi + ((pp * ph + q) * pw + r) * Cout + c];
    }
}

extern "C" int uv_unpatchify_f32(const float* in, long ldi, float* out, int Cout, int Fp, int Hp, int Wp, int pt,
                                 int ph, int pw, void* stream) {
    UV_CHECK_ARG(in && out, "uv_unpatchify_f32: null pointer");
    const long total = (long)Cout * Fp * pt * Hp * ph * Wp * pw;
    const int blocks = (int)min((total + 255) / 256, (long)4096);
    hipLaunchKernelGGL(unpatchify_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, ldi, out, Cout, Fp,
                       Hp, Wp, pt, ph, pw);
    UV_CHECK_LAUNCH("uv_unpatchify_f32");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// sinusoidal_embedding_1d (model.py:14-24): fp64 cos||sin of t * 10000^(-i/half), stored f32.
// ------------------------------------------------------------------------------------------------
__global__ void sinusoid_kernel(const float* t, float* out, int n, int dim) {
    const int half = dim >> 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * half) return;
    const int r = i / half, c = i % half;
    const double freq = pow(10000.0, -((double)c / (double)half));
    const double a = (double)t[r] * freq;
    out[(long)r * dim + c] = (float)cos(a);
    out[(long)r * dim + half + c] = (float)sin(a);
}

extern "C" int uv_sinusoid_f32(const float* t, float* out, int n, int dim, void* stream) {
    UV_CHECK_ARG(t && out && n > 0 && dim % 2 == 0, "uv_sinusoid_f32: bad arguments");
    const int total = n * (dim / 2);
    hipLaunchKernelGGL(sinusoid_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, out, n, dim);
    UV_CHECK_LAUNCH("uv_sinusoid_f32");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// fp32 Linear on a handful of rows (the <=2 distinct timesteps of a forward): out[r][n] =
// act_in(x[r]) . W[n] + b[n]; one wave per output column, W streamed once (HBM-bound).
// time_embedding / time_projection  model.py:384-386, 465-468.  act_in: 0 none, 1 SiLU.
// ------------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void linear_rows_f32_kernel(const float* x, long ldx, const float* W,
                                                               const float* b, float* out, long ldo, int r0,
                                                               int N, int K, int act_in) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* wr = W + (long)n * K;
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const f32x4 w = *(const f32x4*)(wr + k);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            f32x4 xv = *(const f32x4*)(x + (long)(r0 + r) * ldx + k);
            if (act_in == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) xv[e] = silu_f32(xv[e]);
            }
            acc[r] += xv[0] * w[0] + xv[1] * w[1] + xv[2] * w[2] + xv[3] * w[3];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float s = wave_sum(acc[r]);
        if (lane == 0) out[(long)(r0 + r) * ldo + n] = s + (b ? b[n] : 0.f);
    }
}

extern "C" int uv_linear_rows_f32(const float* x, long ldx, const float* W, const float* b, float* out, long ldo,
                                  int R, int N, int K, int act_in, void* stream) {
    UV_CHECK_ARG(x && W && out && R > 0 && N > 0, "uv_linear_rows_f32: bad arguments");
    UV_CHECK_ARG(K % 4 == 0 && ldx % 4 == 0, "uv_linear_rows_f32: K and ldx must be multiples of 4");
    const dim3 grid((N + 3) / 4), block(256);
    int r0 = 0;
    for (; r0 + 4 <= R; r0 += 4)
        hipLaunchKernelGGL(linear_rows_f32_kernel<4>, grid, block, 0, (hipStream_t)stream, x, ldx, W, b, out, ldo, r0, N, K, act_in);
    for (; r0 + 2 <= R; r0 += 2)
        hipLaunchKernelGGL(linear_rows_f32_kernel<2>, grid, block, 0, (hipStream_t)stream, x, ldx, W, b, out, ldo, r0, N, K, act_in);
    for (; r0 < R; ++r0)
        hipLaunchKernelGGL(linear_rows_f32_kernel<1>, grid, block, 0, (hipStream_t)stream, x, ldx, W, b, out, ldo, r0, N, K, act_in);
    UV_CHECK_LAUNCH("uv_linear_rows_f32");
    return 0;
}

// out[r][j][c] = mod[j][c] + e0[r][j][c]   (fp32; (self.modulation.unsqueeze(0) + e) model.py:239, 287)
__global__ void add_rows_kernel(const float* mod, const float* e0, float* out, int R, long n) {
    const long total = (long)R * n;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        out[i] = __fadd_rn(mod[i % n], e0[i]);
}

extern "C" int uv_add_rows_f32(const float* mod, const float* e0, float* out, int R, long n, void* stream) {
    UV_CHECK_ARG(mod && e0 && out && R > 0 && n > 0, "uv_add_rows_f32: bad arguments");
    const int blocks = (int)min(((long)R * n + 255) / 256, (long)2048);
    hipLaunchKernelGGL(add_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mod, e0, out, R, n);
    UV_CHECK_LAUNCH("uv_add_rows_f32");
    return 0;
}

// f32 -> bf16 cast (weights once, contexts), optional zero row padding handled by the caller's memset
__global__ void cast_f32_bf16_kernel(const float* in, bf16_t* out, long n) {
    const long n4 = n >> 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = *(const f32x4*)(in + i * 4);
        u32x2 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        *(u32x2*)(out + i * 4) = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) out[n4 * 4 + threadIdx.x] = f2bf(in[n4 * 4 + threadIdx.x]);
}

extern "C" int uv_cast_f32_bf16(const float* in, void* out, long n, void* stream) {
    UV_CHECK_ARG(in && out && n > 0, "uv_cast_f32_bf16: bad arguments");
    UV_CHECK_ARG((((uintptr_t)in & 15) | ((uintptr_t)out & 7)) == 0, "uv_cast_f32_bf16: misaligned pointers");
    const int blocks = (int)min((n / 4 + 255) / 256 + 1, (long)4096);
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)out, n);
    UV_CHECK_LAUNCH("uv_cast_f32_bf16");
    return 0;
}

// x_f32[m][n] += float(y_bf16[m][n])   (un-fused cross-attention residual for the hooked path, model.py:251)
__global__ void add_bf16_resid_kernel(float* x, long ldx, const bf16_t* y, long ldy, int L, int C) {
    const int c4 = C >> 2;
    const long total = (long)L * c4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c4;
        const int c = (int)(i % c4) * 4;
        f32x4 xv = *(f32x4*)(x + r * ldx + c);
        const u32x2 yv = *(const u32x2*)(y + r * ldy + c);
        xv[0] = __fadd_rn(xv[0], bf2f((bf16_t)(yv[0] & 0xffff)));
        xv[1] = __fadd_rn(xv[1], bf2f((bf16_t)(yv[0] >> 16)));
        xv[2] = __fadd_rn(xv[2], bf2f((bf16_t)(yv[1] & 0xffff)));
        xv[3] = __fadd_rn(xv[3], bf2f((bf16_t)(yv[1] >> 16)));
        *(f32x4*)(x + r * ldx + c) = xv;
    }
}

extern "C" int uv_add_bf16_resid(float* x, long ldx, const void* y, long ldy, int L, int C, void* stream) {
    UV_CHECK_ARG(x && y && L > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "uv_add_bf16_resid: bad arguments");
    const int blocks = (int)min(((long)L * (C / 4) + 255) / 256, (long)4096);
    hipLaunchKernelGGL(add_bf16_resid_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       (const bf16_t*)y, ldy, L, C);
    UV_CHECK_LAUNCH("uv_add_bf16_resid");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// UniVid's dynamic text weight (models/model_pipeline.py:1787-1797): `context * weight_mask` on the bf16 embedded context, where
// weight_mask = ones_like(context) with its first text_len rows multiplied IN PLACE by the python float w - i.e. the mask holds
// bf16(w), and every scaled element is bf16(float(c) * float(bf16(w))) (the product of two bf16 values is exact in f32: one rounding).
// Rows >= n_scaled are copied (x * bf16(1) = x). One launch per sample; the result feeds the K / V projections.
// ------------------------------------------------------------------------------------------------
__global__ void text_weight_rows_kernel(const bf16_t* in, long ldi, bf16_t* out, long ldo, int R, int n_scaled, int C, float wb) {
    const int c8 = C >> 3;
    const long total = (long)R * c8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c8;
        const int c = (int)(i % c8) * 8;
        u32x4 v = *(const u32x4*)(in + r * ldi + c);
        if (r < n_scaled) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[j] = pack_bf2(__fmul_rn(bf2f((bf16_t)(v[j] & 0xffff)), wb), __fmul_rn(bf2f((bf16_t)(v[j] >> 16)), wb));
        }
        *(u32x4*)(out + r * ldo + c) = v;
    }
}

extern "C" int uv_text_weight_rows_bf16(const void* in, long ldi, void* out, long ldo, int R, int n_scaled, int C, float w, void* stream) {
    UV_CHECK_ARG(in && out && R > 0 && n_scaled >= 0 && n_scaled <= R && C > 0 && C % 8 == 0 && ldi % 8 == 0 && ldo % 8 == 0,
                 "uv_text_weight_rows_bf16: bad arguments (C and the leading dimensions must be multiples of 8)");
    UV_CHECK_ARG((((uintptr_t)in | (uintptr_t)out) & 15) == 0, "uv_text_weight_rows_bf16: misaligned pointers");
    // the mask element torch builds: ones(bf16) *= w  ->  bf16(1.0f * (float)w)
    const uint32_t wbits = __builtin_bit_cast(uint32_t, w);
    uint32_t rb = wbits + 0x7fffu + ((wbits >> 16) & 1u);          // round-to-nearest-even to the top 16 bits (w is finite: checked)
    UV_CHECK_ARG((wbits & 0x7f800000u) != 0x7f800000u, "uv_text_weight_rows_bf16: w must be finite");
    rb &= 0xffff0000u;
    const float wb = __builtin_bit_cast(float, rb);
    const int blocks = (int)min(((long)R * (C / 8) + 255) / 256, (long)2048);
    hipLaunchKernelGGL(text_weight_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, ldi, (bf16_t*)out, ldo,
                       R, n_scaled, C, wb);
    UV_CHECK_LAUNCH("uv_text_weight_rows_bf16");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// WanRMSNorm's per-row scale from the sums of squares a q projection's GEMM epilogue left per 32-column group (uv_gemm_bf16_nt_ssq):
// rs[m] = 1 / sqrt(sum_g ssq[m][g] / C + eps) (f32; the formula of rmsnorm_rope_kernel: mean = sum / C; 1.0f / sqrtf(mean + eps)), the groups
// added in a fixed order. Consumed by uv_flash_attn_bf16_qnorm's Q prologue.
// ------------------------------------------------------------------------------------------------
// Four lanes per row: lane part p adds its quarter of the row's groups in ascending order (16-byte loads where the quarter allows), the four
// quarter sums are combined as (p0 + p1) + (p2 + p3). 64 rows per 256-thread block, a row's loads contiguous across its four lanes.
__global__ __launch_bounds__(256) void rms_scale_from_ssq_kernel(const float* ssq, long ld, int M, int groups, float c, float eps, float* rs) {
    const int m = blockIdx.x * 64 + (threadIdx.x >> 2), part = threadIdx.x & 3;
    const int per = (groups + 3) >> 2;
    const int g0 = part * per, g1 = min(g0 + per, groups);
    float t = 0.f;
    if (m < M) {
        const float* row = ssq + (long)m * ld;
        if ((per & 3) == 0 && (ld & 3) == 0 && g1 - g0 == per) {
            for (int g = g0; g < g1; g += 4) {
                const f32x4 v = *(const f32x4*)(row + g);
                t = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(t, v[0]), v[1]), v[2]), v[3]);
            }
        } else {
            for (int g = g0; g < g1; ++g) t = __fadd_rn(t, row[g]);
        }
    }
    t = __fadd_rn(t, __shfl_xor(t, 1, 64));
    t = __fadd_rn(t, __shfl_xor(t, 2, 64));
    if (m < M && part == 0) rs[m] = 1.0f / sqrtf(t / c + eps);
}

extern "C" int uv_rms_scale_from_ssq(const float* ssq, long ld_ssq, int M, int groups, int C, float eps, float* rs, void* stream) {
    UV_CHECK_ARG(ssq && rs && M > 0 && groups > 0 && C > 0 && ld_ssq >= groups, "uv_rms_scale_from_ssq: bad arguments");
    UV_CHECK_ARG((((uintptr_t)ssq) & 15) == 0, "uv_rms_scale_from_ssq: ssq must be 16-byte aligned");
    hipLaunchKernelGGL(rms_scale_from_ssq_kernel, dim3((M + 63) / 64), dim3(256), 0, (hipStream_t)stream, ssq, ld_ssq, M, groups, (float)C, eps, rs);
    UV_CHECK_LAUNCH("uv_rms_scale_from_ssq");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Row-wise L2 normalisation: out[r] = x[r] / max(||x[r]||_2, eps)  (torch.nn.functional.normalize(dim=-1), the cosine
// scoring of the SigLIP2 ranker: models/BAGEL/eval_understanding.py:185,195). One wave per row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2_normalize_rows_kernel(const float* x, long ldx, float* out, long ldo, int R, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const float* xr = x + (long)row * ldx;
    float ss = 0.f;
    for (int c = lane; c < C; c += 64) ss += xr[c] * xr[c];
    const float nrm = fmaxf(sqrtf(wave_sum(ss)), eps);
    for (int c = lane; c < C; c += 64) out[(long)row * ldo + c] = xr[c] / nrm;
}

extern "C" int uv_l2_normalize_rows_f32(const float* x, long ldx, float* out, long ldo, int R, int C, float eps, void* stream) {
    UV_CHECK_ARG(x && out && R > 0 && C > 0, "uv_l2_normalize_rows_f32: bad arguments");
    hipLaunchKernelGGL(l2_normalize_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, out, ldo, R, C, eps);
    UV_CHECK_LAUNCH("uv_l2_normalize_rows_f32");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// ContextProjector glue (models/model_pipeline.py:1506-1574, bf16 module):
//   uv_gelu_erf_bf16          nn.GELU() (exact, erf) on a bf16 tensor: f32 evaluation, one rounding
//   uv_interp_linear_rows_bf16 F.interpolate(x^T, size=Lout, mode='linear', align_corners=False)^T on [Lin, C] bf16 rows:
//                             src = (i + 0.5) * Lin / Lout - 0.5 (clamped at 0), out = (1-w) x[i0] + w x[min(i0+1, Lin-1)] in f32
// ------------------------------------------------------------------------------------------------
__global__ void gelu_erf_bf16_kernel(const bf16_t* in, bf16_t* out, long n) {
    for (long i = (blockIdx.x * (long)blockDim.x + threadIdx.x) * 2; i + 1 < n + 1; i += (long)gridDim.x * blockDim.x * 2) {
        if (i + 1 < n) {
            const uint32_t v = *(const uint32_t*)(in + i);
            const float a = bf2f((bf16_t)(v & 0xffff)), b = bf2f((bf16_t)(v >> 16));
            const float ga = 0.5f * a * (1.0f + erff(a * 0.70710678118654752440f));
            const float gb = 0.5f * b * (1.0f + erff(b * 0.70710678118654752440f));
            *(uint32_t*)(out + i) = pack_bf2(ga, gb);
        } else if (i < n) {
            const float a = bf2f(in[i]);
            out[i] = f2bf(0.5f * a * (1.0f + erff(a * 0.70710678118654752440f)));
        }
    }
}

extern "C" int uv_gelu_erf_bf16(const void* in, void* out, long n, void* stream) {
    UV_CHECK_ARG(in && out && n > 0 && (((uintptr_t)in | (uintptr_t)out) & 3) == 0, "uv_gelu_erf_bf16: bad arguments");
    const int blocks = (int)min((n / 2 + 255) / 256 + 1, (long)4096);
    hipLaunchKernelGGL(gelu_erf_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, (bf16_t*)out, n);
    UV_CHECK_LAUNCH("uv_gelu_erf_bf16");
    return 0;
}

__global__ void interp_linear_rows_bf16_kernel(const bf16_t* in, long ldi, bf16_t* out, long ldo, int Lin, int Lout, int C) {
    const int row = blockIdx.x;
    const float scale = (float)Lin / (float)Lout;
    float src = scale * ((float)row + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    const int i0 = (int)src;
    const int i1 = i0 + (i0 < Lin - 1 ? 1 : 0);
    const float w1 = src - (float)i0, w0 = 1.0f - w1;
    const bf16_t* a = in + (long)i0 * ldi;
    const bf16_t* b = in + (long)i1 * ldi;
    for (int c = threadIdx.x; c < C; c += blockDim.x) out[(long)row * ldo + c] = f2bf(w0 * bf2f(a[c]) + w1 * bf2f(b[c]));
}

extern "C" int uv_interp_linear_rows_bf16(const void* in, long ldi, void* out, long ldo, int Lin, int Lout, int C, void* stream) {
    UV_CHECK_ARG(in && out && Lin > 0 && Lout > 0 && C > 0, "uv_interp_linear_rows_bf16: bad arguments");
    hipLaunchKernelGGL(interp_linear_rows_bf16_kernel, dim3(Lout), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, ldi,
                       (bf16_t*)out, ldo, Lin, Lout, C);
    UV_CHECK_LAUNCH("uv_interp_linear_rows_bf16");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// umT5 encoder glue (models/wan/utils/modules/t5.py, bf16 module: every torch op on a bf16 tensor computes in fp32 and rounds once)
//   uv_add_bf16            out = bf16(x + y)                                  residual adds, t5.py:175-176
//   uv_t5_gated_gelu_bf16  out = bf16(fc1 * GELU(gate)) with the reference's op-by-op GELU (t5.py:46-50, 138):
//                          0.5 * x * (1.0 + tanh(sqrt(2/pi) * (x + 0.044715 * pow(x, 3)))), each op rounded to bf16
// ------------------------------------------------------------------------------------------------
__global__ void add_bf16_kernel(const bf16_t* x, const bf16_t* y, bf16_t* out, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = f2bf(bf2f(x[i]) + bf2f(y[i]));
}

extern "C" int uv_add_bf16(const void* x, const void* y, void* out, long n, void* stream) {
    UV_CHECK_ARG(x && y && out && n > 0, "uv_add_bf16: bad arguments");
    hipLaunchKernelGGL(add_bf16_kernel, dim3((int)min((n + 255) / 256, (long)4096)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (const bf16_t*)y, (bf16_t*)out, n);
    UV_CHECK_LAUNCH("uv_add_bf16");
    return 0;
}

__global__ void t5_gated_gelu_bf16_kernel(const bf16_t* gate, const bf16_t* fc1, bf16_t* out, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float x = bf2f(gate[i]);
        const float t1 = round_bf(round_bf(x * x) * x);                     // torch.pow(x, 3.0) on a bf16 tensor = (x * x) * x, each product rounded
        const float t2 = round_bf(0.044715f * t1);
        const float t3 = round_bf(x + t2);
        const float t4 = round_bf(0.7978845608028654f * t3);
        const float t5 = round_bf(tanhf(t4));
        const float t6 = round_bf(1.0f + t5);
        const float t7 = round_bf(0.5f * x);
        const float g = round_bf(t7 * t6);
        out[i] = f2bf(bf2f(fc1[i]) * g);
    }
}

extern "C" int uv_t5_gated_gelu_bf16(const void* gate, const void* fc1, void* out, long n, void* stream) {
    UV_CHECK_ARG(gate && fc1 && out && n > 0, "uv_t5_gated_gelu_bf16: bad arguments");
    hipLaunchKernelGGL(t5_gated_gelu_bf16_kernel, dim3((int)min((n + 255) / 256, (long)4096)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)gate, (const bf16_t*)fc1, (bf16_t*)out, n);
    UV_CHECK_LAUNCH("uv_t5_gated_gelu_bf16");
    return 0;
}
