// Kernels, fused epilogues and launch helpers of the bf16 / fp16 "NT" GEMM (shared by gemm_bf16.hip = the product entry points and
// gemm_bf16_diag.hip = tile configurations only tests and developer tools select).
#pragma once
// bf16 "NT" GEMM on gfx950 MFMA:  C[M,N] = A[M,K] . W[N,K]^T (+ bias), fp32 accumulate,
// with the fused epilogues the Wan DiT block needs.
//
// Replaces (reference, PyTorch ops under autocast-bf16):
//   nn.Linear q/k/v/o      models/wan/utils/modules/model.py:119-122,138-140,154,170-172,179
//   ffn Linear/GELU/Linear models/wan/utils/modules/model.py:212-214,252-253
//   gated fp32 residuals   models/wan/utils/modules/model.py:247,251,255
//   patch / text embedding models/wan/utils/modules/model.py:378-382,448,473
//
// Layout: both operands are K-contiguous (activations [M,K], nn.Linear weight [N,K]), so both
// MFMA fragments are 16-byte row reads. Tiles go HBM -> LDS with global_load_lds_dwordx4
// (1 KiB per wave-instruction = 8 rows x 128 B), LDS image is lane-linear and the XOR swizzle
// (chunk ^= (row>>1)&7) is applied on the SOURCE address and again on the ds_read_b128, which makes
// the 16x16x32 fragment reads bank-conflict-free on 128-byte rows.
// The MFMA is issued as D = Wfrag x Afrag so that a lane ends up with 4 CONSECUTIVE n for one m:
// epilogue stores are 8 B (bf16) / 16 B (fp32) per lane instead of 2 B scatters.
//
// Kernels in this file:
//   gemm_bf16_8ph_kernel  256x256 tiles, 8 waves in two ping-pong groups, LDS-DMA in flight across raw barriers: every
//                         large projection (tile_cfg 7; what tile_cfg 0 picks for M >= 2048, N % 256 == 0, K % 128 == 0)
//   gemm_bf16_nt_kernel   generic BM x BN tile, NS LDS stages: 128x128 / 8 waves / 4-stage ring for leftover-row strips
//                         and small-M projections (tile_cfg 12), 2-stage 4-wave and 16-wave forms for everything else
//   uv_gemm_bf16_nt       shape-based choice, leftover-row split (launch_m_split)
#include "common.h"
#include <stdlib.h>

const float* uv_zero_page();   // gemm_f32.hip: one 4-KiB page of zeros per device

#define UV_BK 64  // k elements per LDS tile (128-byte rows)

enum {
    UV_EPI_BF16 = 0,           // out_bf16 = bf16(acc + bias)
    UV_EPI_GELU_BF16 = 1,      // out_bf16 = bf16(gelu_tanh(bf16(acc + bias)))
    UV_EPI_F32_FROM_BF16 = 2,  // out_f32  = float(bf16(acc + bias))
    UV_EPI_RESID_F32 = 3,      // x_f32   += float(bf16(acc + bias))
    UV_EPI_GATE_RESID_F32 = 4, // x_f32    = x + float(bf16(acc + bias)) * gate[tid[m]][n]
    UV_EPI_BF16_T = 5,         // outT_bf16[n][m] = bf16(acc + bias)   (V^T for attention)
    UV_EPI_BF16_SSQ = 6,       // UV_EPI_BF16 + ssq[m][n / 32] = sum of float(out_bf16)^2 over each aligned group of 32 columns (uv_gemm_bf16_nt_ssq)
};

struct GemmArgs {
    const bf16_t* A;
    const bf16_t* W;
    const bf16_t* bias;  // [N] bf16 or nullptr
    void* out;
    const float* gate;      // [n_t, gate_stride]  (EPI 4)
    const int32_t* gate_tid;  // [M] row -> gate row (EPI 4), may be nullptr => row 0
    const float* zeros;       // the library's 4-KiB zero page: what a null bias reads through in the branch-free epilogue
    long lda, ldw, ldo, gate_stride;
    int M, N, K;
    int tiles_m, tiles_n;
    int gm;                   // persistent kernel: height (in tiles) of the column groups the tile walk is made of
    float* ssq;               // UV_EPI_BF16_SSQ: [M, ld_ssq] f32 partial sums of squares, one per 32-column group
    long ld_ssq;
    float* ws_slab;           // split-K launch (gemm_bf16_8ph_kernel<.., SK > 1>): f32 partial tiles [tile][slice][32 fragments][512 threads][4]
    int* ws_cnt;              //   and one arrival counter per output tile (zeroed by a memset node in front of every launch)
};

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void glds16(const void* g, lds_void* l) {
    __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
}

// One 16x16 accumulator fragment through the fused epilogue. Plain product (D = Wfrag x Afrag): the lane holds
// n = nb + 4*fq + {0..3} for m = mb + frow. Transposed product (EPI_BF16_T, D = Afrag x Wfrag): m = mb + 4*fq + {0..3}
// for n = nb + frow.
// FULL: the fragment lies inside the matrix for sure (whole 256x256 tiles of the persistent kernel) - no bounds checks, and a
// null bias reads zeros instead of branching, so the 32 fragments of a wave form ONE basic block: their loads, transcendental
// latencies and stores overlap instead of running fragment by fragment behind exec-mask branches.
template <int EPI, bool F16 = false, bool FULL = false>
__device__ __forceinline__ void epi_frag(const GemmArgs& p, int mb, int nb, const f32x4& a, int frow, int fq) {
    if (EPI != UV_EPI_BF16_T) {
        const int m = mb + frow, n = nb + 4 * fq;
        if (!FULL && (m >= p.M || n >= p.N)) return;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = a[e];
        if (FULL || p.bias) {
            const u32x2 bb = *(const u32x2*)((FULL && !p.bias) ? (const bf16_t*)p.zeros : p.bias + n);
            v[0] += in16<F16>((bf16_t)(bb[0] & 0xffff));
            v[1] += in16<F16>((bf16_t)(bb[0] >> 16));
            v[2] += in16<F16>((bf16_t)(bb[1] & 0xffff));
            v[3] += in16<F16>((bf16_t)(bb[1] >> 16));
        }
        if (EPI == UV_EPI_BF16) {
            u32x2 o = {pack16_2<F16>(v[0], v[1]), pack16_2<F16>(v[2], v[3])};
            *(u32x2*)((bf16_t*)p.out + (long)m * p.ldo + n) = o;
        } else if (EPI == UV_EPI_GELU_BF16) {
#pragma unroll
            for (int e = 0; e < 4; e += 2) {      // (pairwise rounding + packed f32 GELU: see epi_pair16)
                const uint32_t pk = pack16_2<F16>(v[e], v[e + 1]);
                const f32x2 y = gelu_tanh_f32x2((f32x2){in16<F16>((bf16_t)(pk & 0xffff)), in16<F16>((bf16_t)(pk >> 16))});
                v[e] = y[0];
                v[e + 1] = y[1];
            }
            u32x2 o = {pack16_2<F16>(v[0], v[1]), pack16_2<F16>(v[2], v[3])};
            *(u32x2*)((bf16_t*)p.out + (long)m * p.ldo + n) = o;
        } else if (EPI == UV_EPI_F32_FROM_BF16) {
            f32x4 o = {round16<F16>(v[0]), round16<F16>(v[1]), round16<F16>(v[2]), round16<F16>(v[3])};
            *(f32x4*)((float*)p.out + (long)m * p.ldo + n) = o;
        } else if (EPI == UV_EPI_RESID_F32) {
            float* xp = (float*)p.out + (long)m * p.ldo + n;
            f32x4 x = *(const f32x4*)xp;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = __fadd_rn(x[e], round16<F16>(v[e]));
            *(f32x4*)xp = x;
        } else if (EPI == UV_EPI_GATE_RESID_F32) {
            const int t = p.gate_tid ? p.gate_tid[m] : 0;
            const f32x4 g = *(const f32x4*)(p.gate + (long)t * p.gate_stride + n);
            float* xp = (float*)p.out + (long)m * p.ldo + n;
            f32x4 x = *(const f32x4*)xp;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = __fadd_rn(x[e], __fmul_rn(round16<F16>(v[e]), g[e]));
            *(f32x4*)xp = x;
        }
    } else {
        const int n = nb + frow, m = mb + 4 * fq;
        if (!FULL && (n >= p.N || m >= p.M)) return;  // M is padded to a multiple of 4 by the caller's ldo
        const float b = FULL ? in16<F16>(*(p.bias ? p.bias + n : (const bf16_t*)p.zeros)) : (p.bias ? in16<F16>(p.bias[n]) : 0.f);
        bf16_t* op = (bf16_t*)p.out + (long)n * p.ldo + m;
        if (FULL || m + 3 < p.M) {
            u32x2 o = {pack16_2<F16>(a[0] + b, a[1] + b), pack16_2<F16>(a[2] + b, a[3] + b)};
            *(u32x2*)op = o;
        } else {
            for (int e = 0; e < 4 && m + e < p.M; ++e) op[e] = out16<F16>(a[e] + b);
        }
    }
}


// Sum of squares of the 16-bit OUTPUT values of one aligned 32-column group (columns nb .. nb+31, nb % 32 == 0) of row mb + frow, from the
// group's two fragments in the MFMA's layout (lane fq holds columns nb + 16 f + 4 fq + e of fragment f). ONE summation order, whatever
// kernel and tile shape the group is computed in - so the value does not depend on the GEMM's schedule (tested bit for bit):
//   quad sums   s_k = ((x0^2 + x1^2) + x2^2) + x3^2 over the 4 consecutive columns of quad k = 4 f + fq   (plain f32 multiplies and adds)
//   lane        u_fq = s_fq + s_(4+fq)
//   group       G = (u_0 + u_1) + (u_2 + u_3)        (two butterfly exchanges: every lane of the row ends with the same G)
// v0 / v1: the fragment values AFTER the 16-bit rounding of the output, as f32. Returned in every lane.
__device__ __forceinline__ float ssq_group32(const float (&v0)[4], const float (&v1)[4]) {
    const float s0 = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(v0[0], v0[0]), __fmul_rn(v0[1], v0[1])), __fmul_rn(v0[2], v0[2])), __fmul_rn(v0[3], v0[3]));
    const float s1 = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(v1[0], v1[0]), __fmul_rn(v1[1], v1[1])), __fmul_rn(v1[2], v1[2])), __fmul_rn(v1[3], v1[3]));
    const float u = __fadd_rn(s0, s1);
    const float t = __fadd_rn(u, __shfl_xor(u, 16, 64));
    return __fadd_rn(t, __shfl_xor(t, 32, 64));
}

// UV_EPI_BF16_SSQ on two column-adjacent fragments, fragment-wise stores (any kernel, with bounds): the bf16 output of UV_EPI_BF16 and the
// group's sum of squares. nb % 32 == 0. Rows / columns outside the matrix store nothing (their lanes still take part in the exchange).
template <bool F16>
__device__ __forceinline__ void epi_pair_ssq(const GemmArgs& p, int mb, int nb, const f32x4& a0, const f32x4& a1, int frow, int fq) {
    const int m = mb + frow;
    float v[2][4];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int n = min(nb + 16 * f + 4 * fq, p.N - 4);
        const u32x2 bb = p.bias ? *(const u32x2*)(p.bias + n) : (u32x2){0u, 0u};
        const f32x4& a = f ? a1 : a0;
        v[f][0] = round16<F16>(a[0] + in16<F16>((bf16_t)(bb[0] & 0xffff)));
        v[f][1] = round16<F16>(a[1] + in16<F16>((bf16_t)(bb[0] >> 16)));
        v[f][2] = round16<F16>(a[2] + in16<F16>((bf16_t)(bb[1] & 0xffff)));
        v[f][3] = round16<F16>(a[3] + in16<F16>((bf16_t)(bb[1] >> 16)));
        if (m < p.M && nb + 16 * f + 4 * fq < p.N) {
            u32x2 o = {pack16_2<F16>(v[f][0], v[f][1]), pack16_2<F16>(v[f][2], v[f][3])};
            *(u32x2*)((bf16_t*)p.out + (long)m * p.ldo + nb + 16 * f + 4 * fq) = o;
        }
    }
    const float g = ssq_group32(v[0], v[1]);
    if (fq == 0 && m < p.M && nb < p.N) p.ssq[(long)m * p.ld_ssq + (nb >> 5)] = g;
}

// Two column-adjacent 16x16 fragments (columns nb .. nb+15 and nb+16 .. nb+31 of the same 16 rows) through the bf16 / GELU
// epilogue with 16-BYTE stores: in the MFMA's layout a lane holds 4 consecutive columns (8 bytes of bf16) of each fragment; two
// v_permlane16_swap exchange the halves between the lane groups fq = 0 <-> 1 and 2 <-> 3 so that every lane ends up with 8
// CONSECUTIVE columns (fq 0: 0-7, fq 2: 8-15 of the first fragment; fq 1: 16-23, fq 3: 24-31 of the second). One
// global_store_dwordx4 then writes 64 contiguous bytes per row instead of two dwordx2 stores of 32 bytes each: half the store
// instructions, twice the segment. Whole tiles only (the persistent kernel); values are those of epi_frag.
template <int EPI, bool F16>
__device__ __forceinline__ void epi_pair16(const GemmArgs& p, int mb, int nb, const f32x4& a0, const f32x4& a1, int frow, int fq) {
    static_assert(EPI == UV_EPI_BF16 || EPI == UV_EPI_GELU_BF16 || EPI == UV_EPI_BF16_SSQ, "16-bit row-major outputs only");
    const int m = mb + frow;
    uint32_t w[2][2];
    float r[2][4];      // UV_EPI_BF16_SSQ: the rounded outputs as f32
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int n = nb + 16 * f + 4 * fq;
        const u32x2 bb = *(const u32x2*)(p.bias ? p.bias + n : (const bf16_t*)p.zeros);
        const f32x4& a = f ? a1 : a0;
        float v[4] = {a[0] + in16<F16>((bf16_t)(bb[0] & 0xffff)), a[1] + in16<F16>((bf16_t)(bb[0] >> 16)),
                      a[2] + in16<F16>((bf16_t)(bb[1] & 0xffff)), a[3] + in16<F16>((bf16_t)(bb[1] >> 16))};
        if (EPI == UV_EPI_GELU_BF16) {
            // the pre-activation rounded pairwise (one packed conversion per pair, the two 16-bit values widened again) and GELU evaluated with
            // packed f32 arithmetic: the epilogue of a 256 x 256 tile was ~1 600 vector instructions per wave, every one of them with the matrix
            // pipe idle; this form issues ~25 % fewer and the rest two elements per slot. Same operations per element: bit-identical values.
            // (two-library same-process A/B, round 5: ffn.0 at 23 040 rows 1 576.1 -> 1 509.9 us; the plain epilogue 1 486.9 / 1 488.4)
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                const uint32_t pk = pack16_2<F16>(v[e], v[e + 1]);
                const f32x2 r = {in16<F16>((bf16_t)(pk & 0xffff)), in16<F16>((bf16_t)(pk >> 16))};
                const f32x2 y = gelu_tanh_f32x2(r);
                v[e] = y[0];
                v[e + 1] = y[1];
            }
        }
        if (EPI == UV_EPI_BF16_SSQ) {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[f][e] = v[e] = round16<F16>(v[e]);
        }
        w[f][0] = pack16_2<F16>(v[0], v[1]);
        w[f][1] = pack16_2<F16>(v[2], v[3]);
    }
    if (EPI == UV_EPI_BF16_SSQ) {
        const float g = ssq_group32(r[0], r[1]);
        if (fq == 0) p.ssq[(long)m * p.ld_ssq + (nb >> 5)] = g;
    }
    // swap lanes 16-31 / 48-63 of the first fragment's words with lanes 0-15 / 32-47 of the second's
    const auto s0 = __builtin_amdgcn_permlane16_swap(w[0][0], w[1][0], false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(w[0][1], w[1][1], false, false);
    const u32x4 o = {(uint32_t)s0[0], (uint32_t)s1[0], (uint32_t)s0[1], (uint32_t)s1[1]};
    const int col = ((fq & 1) << 4) + ((fq >> 1) << 3);
    *(u32x4*)((bf16_t*)p.out + (long)m * p.ldo + nb + col) = o;
}

// The transposed 16-bit epilogue (V^T) with 16-byte stores, same exchange as epi_pair16: in the transposed product a lane holds 4
// consecutive ROWS m of one column n, and the wave's row fragments j and j+1 (rows mb .. mb+15 and mb+16 .. mb+31) are adjacent
// along the contiguous axis of the transposed output; after two v_permlane16_swap every lane owns 8 consecutive m of its n:
// 64 contiguous bytes per output row and instruction instead of 32. Whole tiles only.
template <bool F16>
__device__ __forceinline__ void epi_pair16_T(const GemmArgs& p, int mb, int nb, const f32x4& a0, const f32x4& a1, int frow, int fq) {
    const int n = nb + frow;
    const float b = in16<F16>(*(p.bias ? p.bias + n : (const bf16_t*)p.zeros));
    uint32_t w0[2] = {pack16_2<F16>(a0[0] + b, a0[1] + b), pack16_2<F16>(a0[2] + b, a0[3] + b)};
    uint32_t w1[2] = {pack16_2<F16>(a1[0] + b, a1[1] + b), pack16_2<F16>(a1[2] + b, a1[3] + b)};
    const auto s0 = __builtin_amdgcn_permlane16_swap(w0[0], w1[0], false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(w0[1], w1[1], false, false);
    const u32x4 o = {(uint32_t)s0[0], (uint32_t)s1[0], (uint32_t)s0[1], (uint32_t)s1[1]};
    const int row = ((fq & 1) << 4) + ((fq >> 1) << 3);
    *(u32x4*)((bf16_t*)p.out + (long)n * p.ldo + mb + row) = o;
}

// The fp32 read-modify-write epilogues (x += y, x += y*gate) as a D-deep software pipeline over a wave's NF fragments:
// the token->gate-row indices of all rows are fetched first, then the x / gate loads of fragment f+D are issued before
// fragment f is stored. Written in this order by hand because the compiler must assume the x stores alias the later
// loads and otherwise serialises {index load -> gate/x load -> store} per row block (4-8 dependent HBM round trips per tile).
template <int EPI, int NF, int D, bool F16 = false>
__device__ __forceinline__ void epi_rmw_pipe(const GemmArgs& p, const int (&mb)[NF], const int (&nb)[NF], const f32x4 (&acc)[NF],
                                             int frow, int fq) {
    static_assert(EPI == UV_EPI_RESID_F32 || EPI == UV_EPI_GATE_RESID_F32, "rmw epilogues only");
    constexpr bool GATE = EPI == UV_EPI_GATE_RESID_F32;
    const float* grow[NF];
    if (GATE) {
        int t[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) t[f] = p.gate_tid ? p.gate_tid[min(mb[f] + frow, p.M - 1)] : 0;
#pragma unroll
        for (int f = 0; f < NF; ++f) grow[f] = p.gate + (long)t[f] * p.gate_stride;
    }
    f32x4 xb[D], gb[D];
    u32x2 bb[D];
    auto issue = [&](int f, int slot) {
        const int m = min(mb[f] + frow, p.M - 1), n = min(nb[f] + 4 * fq, p.N - 4);
        xb[slot] = *(const f32x4*)((const float*)p.out + (long)m * p.ldo + n);
        if (GATE) gb[slot] = *(const f32x4*)(grow[f] + n);
        bb[slot] = p.bias ? *(const u32x2*)(p.bias + n) : (u32x2){0u, 0u};
    };
#pragma unroll
    for (int f = 0; f < D && f < NF; ++f) issue(f, f);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int slot = f % D;
        const int m = mb[f] + frow, n = nb[f] + 4 * fq;
        float v[4];
        v[0] = acc[f][0] + in16<F16>((bf16_t)(bb[slot][0] & 0xffff));
        v[1] = acc[f][1] + in16<F16>((bf16_t)(bb[slot][0] >> 16));
        v[2] = acc[f][2] + in16<F16>((bf16_t)(bb[slot][1] & 0xffff));
        v[3] = acc[f][3] + in16<F16>((bf16_t)(bb[slot][1] >> 16));
        f32x4 x = xb[slot];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            x[e] = GATE ? __fadd_rn(x[e], __fmul_rn(round16<F16>(v[e]), gb[slot][e])) : __fadd_rn(x[e], round16<F16>(v[e]));
        if (m < p.M && n < p.N) *(f32x4*)((float*)p.out + (long)m * p.ldo + n) = x;
        if (f + D < NF) issue(f + D, slot);
    }
}

template <int BM, int BN, int WM, int WN, int EPI, int NS = 2, bool F16 = false>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_nt_kernel(GemmArgs p) {
    constexpr int NW = WM * WN;
    constexpr int NT = NW * 64;
    constexpr int TM = BM / WM / 16;  // 16-row m tiles per wave
    constexpr int TN = BN / WN / 16;
    constexpr int A_BYTES = BM * 128;
    constexpr int W_BYTES = BN * 128;
    constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
    constexpr int A_INSTR = (BM / 8 + NW - 1) / NW;  // glds wave-instructions per wave for the A tile
    constexpr int W_INSTR = (BN / 8 + NW - 1) / NW;  // (the last pass may cover only part of the waves: guarded below)
    static_assert(BM % 8 == 0 && BN % 8 == 0 && (BM / WM) % 16 == 0 && (BN / WN) % 16 == 0, "tile/wave mismatch");
    constexpr bool TRANS = (EPI == UV_EPI_BF16_T);

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- XCD-aware tile mapping: blocks b, b+8, ... share an XCD's L2; give each XCD a
    // contiguous range of logical tiles, and walk logical tiles in GM-tall column groups so the
    // ~32 blocks resident on one XCD share few A / W panels.
    const int nblk = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int GM = 4;
    const int group_sz = GM * p.tiles_n;
    const int group = bid / group_sz;
    const int first_m = group * GM;
    const int gm = min(GM, p.tiles_m - first_m);
    const int in_group = bid - group * group_sz;
    const int tile_m = first_m + in_group % gm;
    const int tile_n = in_group / gm;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-lane staging source pointers (row clamp keeps every load in bounds)
    const int srow = lane >> 3;  // row inside the 8-row glds piece
    const int pchunk = lane & 7; // physical 16-B chunk inside the 128-B row
    const bf16_t* a_src[A_INSTR];
    const bf16_t* w_src[W_INSTR];
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int row = (i * NW + wave) * 8 + srow;
        const int c = pchunk ^ ((row >> 1) & 7);
        const int gr = min(m0 + row, p.M - 1);
        a_src[i] = p.A + (long)gr * p.lda + c * 8;
    }
#pragma unroll
    for (int i = 0; i < W_INSTR; ++i) {
        const int row = (i * NW + wave) * 8 + srow;
        const int c = pchunk ^ ((row >> 1) & 7);
        const int gr = min(n0 + row, p.N - 1);
        w_src[i] = p.W + (long)gr * p.ldw + c * 8;
    }

    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * STAGE_BYTES;
        const int koff = kt * UV_BK;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i)
            if ((i * NW + wave) * 8 < BM) glds16(a_src[i] + koff, (lds_void*)(base + (i * NW + wave) * 1024));
#pragma unroll
        for (int i = 0; i < W_INSTR; ++i)
            if ((i * NW + wave) * 8 < BN) glds16(w_src[i] + koff, (lds_void*)(base + A_BYTES + (i * NW + wave) * 1024));
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: row = base + (lane&15), logical chunk = ks*4 + (lane>>4)
    const int frow = lane & 15;
    const int fq = lane >> 4;
    int a_off[TM], w_off[TN];  // byte offset of the row start; swizzle key per row
    int a_key[TM], w_key[TN];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int row = wm * (BM / WM) + j * 16 + frow;
        a_off[j] = row * 128;
        a_key[j] = (row >> 1) & 7;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int row = wn * (BN / WN) + i * 16 + frow;
        w_off[i] = A_BYTES + row * 128;
        w_key[i] = (row >> 1) & 7;
    }

    const int nk = p.K / UV_BK;
    auto compute = [&](const char* base) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[TM], wf[TN];
            const int c = ks * 4 + fq;
#pragma unroll
            for (int j = 0; j < TM; ++j)
                af[j] = *(const bf16x8*)(base + a_off[j] + ((c ^ a_key[j]) << 4));
#pragma unroll
            for (int i = 0; i < TN; ++i)
                wf[i] = *(const bf16x8*)(base + w_off[i] + ((c ^ w_key[i]) << 4));
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    if (TRANS)
                        acc[i][j] = mfma_16x16x32<F16>(af[j], wf[i], acc[i][j]);
                    else
                        acc[i][j] = mfma_16x16x32<F16>(wf[i], af[j], acc[i][j]);
                }
        }
    };
    if constexpr (NS == 2) {
        // two LDS stages: the DMA of tile kt+1 overlaps the MFMAs of tile kt, one vmcnt(0) + barrier per K tile
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
            compute(smem + buf * STAGE_BYTES);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else {
        // NS-deep ring for launches that run ~one workgroup per CU (the leftover-row strip of a split GEMM, small-M
        // projections): NS-1 K tiles stay in flight across raw barriers, the only VMEM wait is a counted vmcnt.
        // After barrier kt every wave has finished the MFMAs of tile kt-1, so its buffer can take tile kt+NS-1.
        static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "ring variant needs unguarded staging");
        constexpr int LPS = A_INSTR + W_INSTR;  // DMA instructions per thread per K tile
        static_assert((NS - 2) * LPS < 64, "vmcnt range");
#pragma unroll
        for (int s0 = 0; s0 < NS - 1; ++s0)
            if (s0 < nk) stage(s0, s0);
        int buf = 0, nbuf = NS - 1;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + NS - 2 < nk) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPS) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (kt + NS - 1 < nk) stage(kt + NS - 1, nbuf);
            compute(smem + buf * STAGE_BYTES);
            buf = buf + 1 == NS ? 0 : buf + 1;
            nbuf = nbuf + 1 == NS ? 0 : nbuf + 1;
        }
    }

    // ---- epilogue
    if constexpr (EPI == UV_EPI_RESID_F32 || EPI == UV_EPI_GATE_RESID_F32) {
        constexpr int NF = TM * TN;
        int mb[NF], nb[NF];
        f32x4 av[NF];
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                mb[j * TN + i] = m0 + wm * (BM / WM) + j * 16;
                nb[j * TN + i] = n0 + wn * (BN / WN) + i * 16;
                av[j * TN + i] = acc[i][j];
            }
        epi_rmw_pipe<EPI, NF, (NF >= 8 ? 4 : 2), F16>(p, mb, nb, av, frow, fq);
    } else if constexpr (EPI == UV_EPI_BF16_SSQ) {
        static_assert(TN % 2 == 0 && (BN / WN) % 32 == 0, "UV_EPI_BF16_SSQ needs whole 32-column groups per wave");
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int i = 0; i < TN; i += 2)
                epi_pair_ssq<F16>(p, m0 + wm * (BM / WM) + j * 16, n0 + wn * (BN / WN) + i * 16, acc[i][j], acc[i + 1][j], frow, fq);
    } else {
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int i = 0; i < TN; ++i)
                epi_frag<EPI, F16>(p, m0 + wm * (BM / WM) + j * 16, n0 + wn * (BN / WN) + i * 16, acc[i][j], frow, fq);
    }
}


// The fp32 read-modify-write epilogues of the 256x256 ping-pong kernel, ROW-COALESCED through LDS. In the MFMA's own layout a
// lane holds 4 consecutive columns of one row and a 16x16 fragment spans 16 rows x 64 B: every x load / store instruction of
// the fragment-wise epilogue touches 16 different 12-KB-strided rows with 64 B each. Here the workgroup's (now idle) staging
// LDS takes the bf16-rounded product y as fp32 [128 rows][256 cols] (one half of the tile at a time, 128 KiB, 16-byte chunks
// XOR-swizzled by row & 7 so that neither the fragment-shaped writes nor the row-shaped reads conflict), and then each wave
// walks whole rows: one ds_read_b128, one 1-KiB-contiguous x load, one 1-KiB-contiguous store per row - the access shape HBM and
// the address path like best. Arithmetic per element is unchanged: x + float(bf16(acc + bias)) [* gate], so the result is
// bit-identical to the fragment-wise epilogue.
template <int EPI, bool F16>
__device__ __forceinline__ void epi_rmw_rows_lds(const GemmArgs& p, char* smem, const f32x4 (&acc)[2][2][2][4], int m0, int n0,
                                                 int wave, int lane) {
    static_assert(EPI == UV_EPI_RESID_F32 || EPI == UV_EPI_GATE_RESID_F32, "rmw epilogues only");
    constexpr bool GATE = EPI == UV_EPI_GATE_RESID_F32;
    constexpr int D = 8;                       // rows in flight per wave
    const int wr = wave >> 2, wc = wave & 3;
    const int frow = lane & 15, fq = lane >> 4;
    // bias of this lane's 4 columns in each of its 4 column fragments (hn, i)
    float bv[2][2][4];
#pragma unroll
    for (int hn = 0; hn < 2; ++hn)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = min(n0 + hn * 128 + wc * 32 + i * 16 + 4 * fq, p.N - 4);
            const u32x2 bb = p.bias ? *(const u32x2*)(p.bias + n) : (u32x2){0u, 0u};
            bv[hn][i][0] = in16<F16>((bf16_t)(bb[0] & 0xffff));
            bv[hn][i][1] = in16<F16>((bf16_t)(bb[0] >> 16));
            bv[hn][i][2] = in16<F16>((bf16_t)(bb[1] & 0xffff));
            bv[hn][i][3] = in16<F16>((bf16_t)(bb[1] >> 16));
        }
    const int n = n0 + 4 * lane;               // this lane's 4 columns in the row phase
    const bool n_ok = n < p.N;
    const int nc = min(n, p.N - 4);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        // ---- y of rows 128 h .. 128 h + 127 -> LDS (fragment-shaped writes)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int hn = 0; hn < 2; ++hn)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = wr * 64 + j * 16 + frow;
                    const int chunk = hn * 32 + wc * 8 + i * 4 + fq;
                    f32x4 y;
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = round16<F16>(acc[hn][h][i][j][e] + bv[hn][i][e]);
                    *(f32x4*)(smem + row * 1024 + ((chunk ^ (row & 7)) << 4)) = y;
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- rows 16 wave .. 16 wave + 15 of the half: x (+)= y (* gate), D rows in flight
        const int mbase = m0 + h * 128 + wave * 16;
        f32x4 xb[D], gb[D];
        // token -> gate row of this wave's 16 rows: ONE load (lane r holds row r's index), broadcast per row by readlane, so
        // the gate loads below do not sit behind a dependent index load each
        int tid16 = 0;
        if (GATE && p.gate_tid) tid16 = p.gate_tid[min(mbase + (lane & 15), p.M - 1)];
        auto issue = [&](int rr, int slot) {
            const int m = min(mbase + rr, p.M - 1);
            xb[slot] = *(const f32x4*)((const float*)p.out + (long)m * p.ldo + nc);
            if (GATE) {
                const int t = __builtin_amdgcn_readlane(tid16, rr);
                gb[slot] = *(const f32x4*)(p.gate + (long)t * p.gate_stride + nc);
            }
        };
#pragma unroll
        for (int rr = 0; rr < D; ++rr) issue(rr, rr);
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int slot = rr % D;
            const int row = wave * 16 + rr;
            const f32x4 y = *(const f32x4*)(smem + row * 1024 + ((lane ^ (row & 7)) << 4));
            f32x4 x = xb[slot];
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = GATE ? __fadd_rn(x[e], __fmul_rn(y[e], gb[slot][e])) : __fadd_rn(x[e], y[e]);
            if (mbase + rr < p.M && n_ok) *(f32x4*)((float*)p.out + (long)(mbase + rr) * p.ldo + n) = x;
            if (rr + D < 16) issue(rr + D, slot);
        }
        if (h == 0) {   // the second half overwrites the staging area: every wave must have read its rows
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 256x256x64 tile, 8 waves in two groups that run half a phase apart ("ping-pong"): while one group issues its 16 MFMAs
// of a phase, the other group (the second wave of every SIMD) reads its fragments from LDS and issues the LDS-DMA of a
// later half-tile. The K tile is staged as four 16-KiB half-tiles (A rows 0-127 / 128-255, W rows 0-127 / 128-255), one
// per phase, and stays in flight across the barriers: the only VMEM wait in the loop is a counted vmcnt(6) once per K
// tile. A wave owns the output rows {128*h + 64*wr + 0..63} and columns {128*h + 32*wc + 0..31}, h = 0,1, so each
// half-tile is read in exactly one phase and can be restaged right after it:
//     phase   LDS reads (this K tile)        MFMA quadrant      DMA issued
//       1     W[0] sub (4) + A[0] sub (8)    (A0, W0)           A[1] of K tile t+1
//       2     W[1] sub (4)                   (A0, W1)           W[0] of K tile t+2
//       3     A[1] sub (8)                   (A1, W1)           A[0] of K tile t+2
//       4     -                              (A1, W0)           W[1] of K tile t+2, then vmcnt(6): K tile t+1 landed
// That is the 4-phase schedule (VAR 0, kept as the A/B and race-screen reference, tile_cfg 14). The default (VAR 5) merges the
// phases pairwise - 32 MFMAs per cluster, half as many barriers and cluster ramps per MFMA: +2-4 % measured.
// Needs N % 256 == 0 rows of W to exist (clamped like A otherwise) and an even K/64 >= 4.
#define UV_SB() __builtin_amdgcn_s_barrier()
#ifndef UV_REBAL
#define UV_REBAL 1   // VAR 5: W[0] of K tile t+1 is staged in phase A of K tile t (into the other buffer, beside A[1]) instead of as W[0] of t+2 in phase B of t-1:
                     // 4 + 4 LDS-DMA pieces per phase instead of 2 + 6. Same arithmetic, bit-identical
#endif
#define UV_SCHED() __builtin_amdgcn_sched_barrier(0)

// SK > 1 (VAR 5 only): SPLIT-K form for the leftover-row strips of the long-K projections (ffn.2: 1 120 rows x 3 072 columns x K 14 336 =
// 60 tiles, a quarter of a round). The grid is tiles x SK workgroups - ONE round of the chip - and workgroup (tile, slice) runs the
// same ping-pong K loop over K / SK. Placement (speed only, never correctness): the SK slices of a tile run on ONE XCD, so the last
// arriver reads the other slabs from its own L2 (measured 120.8 us on the ffn.2 strip; with one K range per XCD - operand panels fetched
// once per XCD, slabs crossing the fabric - 134.2; the 128x128 ring 144.1 in the same run: the 63 MB of partial tiles, not the 120 MB of
// operands, are what a one-round split-K launch waits for). Every slice leaves its f32 partial tile in a slab (fragment-major,
// 1 KiB per wave-instruction), publishes it (vmcnt(0) -> barrier -> one agent-scope release fence -> relaxed ticket on the tile's
// counter) and ends; the workgroup that draws the last ticket acquires once, sums the SK slabs IN SLICE ORDER ((s0 + s1) + s2) + s3 -
// its own included, read back from the slab, so the result does not depend on who arrives last - and runs the ordinary epilogue.
// Deterministic run to run; NOT bit-identical to the unsplit accumulation (four f32 partial sums instead of one chain: <= 1 ulp of
// the 16-bit rounding in the epilogue on a tiny fraction of the elements).
template <int EPI, int VAR = 0, bool F16 = false, int SK = 1>
__global__ __launch_bounds__(512) void gemm_bf16_8ph_kernel(GemmArgs p) {
    static_assert(SK == 1 || (VAR == 5 && (8 % SK) == 0), "split-K: VAR 5, SK in {2, 4, 8}");
    constexpr bool TRANS = (EPI == UV_EPI_BF16_T);
    constexpr int HALF = 16384;      // one half-tile: 128 rows x 128 B
    constexpr int BUF = 4 * HALF;    // A[0] A[1] W[0] W[1]
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int nblk = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    int slice = 0;
    if constexpr (SK == 1) {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    } else {
        const int xcd = bid & 7, idx = bid >> 3, per = gridDim.x >> 3;
        if (p.gm == 0) {        // the SK slices of a tile on ONE XCD: per = ceil(nblk * SK / 8) (tile, slice) units per XCD
            const int u = xcd * per + idx;
            slice = u % SK;
            bid = u / SK;
        } else {                // A/B (tile_cfg 21): XCD x takes slice x % SK of the tiles of part x / SK; per = ceil(nblk / (8 / SK)) tiles per XCD
            slice = xcd % SK;
            bid = (xcd / SK) * per + idx;
        }
        if (bid >= nblk) return;
    }
    constexpr int GM = 4;
    const int group_sz = GM * p.tiles_n;
    const int group = bid / group_sz;
    const int first_m = group * GM;
    const int gm = min(GM, p.tiles_m - first_m);
    const int in_group = bid - group * group_sz;
    const int m0 = (first_m + in_group % gm) * 256, n0 = (in_group / gm) * 256;
    const int kbeg = SK == 1 ? 0 : slice * (p.K / SK);                       // this slice's first k

    // staging sources: half-tile h, wave-instruction i covers rows (i*8 + wave)*8 + srow of the half-tile
    const int srow = lane >> 3, pchunk = lane & 7;
    const bf16_t* a_src[2][2];
    const bf16_t* w_src[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (i * 8 + wave) * 8 + srow;
            const int c = pchunk ^ ((row >> 1) & 7);
            a_src[h][i] = p.A + (long)min(m0 + h * 128 + row, p.M - 1) * p.lda + c * 8 + kbeg;
            w_src[h][i] = p.W + (long)min(n0 + h * 128 + row, p.N - 1) * p.ldw + c * 8 + kbeg;
        }
    char* const my_dst = smem + wave * 1024;
    const int nk = p.K / SK / UV_BK;
#define UV_STAGE(SRC, KT, DSTOFF)                                         \
    {                                                                     \
        const bf16_t* g0_ = SRC[0] + (long)(KT) * UV_BK;                  \
        const bf16_t* g1_ = SRC[1] + (long)(KT) * UV_BK;                  \
        glds16(g0_, (lds_void*)(my_dst + (DSTOFF)));                      \
        glds16(g1_, (lds_void*)(my_dst + (DSTOFF) + 8192));              \
    }

    // fragment reads: row = 64*wr (A) / 32*wc (W) + 16*frag + frow, chunk (ks*4 + fq) ^ (frow >> 1)
    const int frow = lane & 15, fq = lane >> 4;
    const int lx = (fq ^ (frow >> 1)) << 4;
    const char* const la = smem + (wr * 64 + frow) * 128;
    const char* const lw = smem + 2 * HALF + (wc * 32 + frow) * 128;

    f32x4 acc[2][2][2][4];  // [hn][hm][i][j]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[a][b][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 af[4][2], w0[2][2], w1[2][2];

#define UV_RD_A(B, H)                                                                             \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                     \
        af[j][0] = *(const bf16x8*)(la + (B) * BUF + (H) * HALF + j * 2048 + lx);                 \
        af[j][1] = *(const bf16x8*)(la + (B) * BUF + (H) * HALF + j * 2048 + (lx ^ 64));          \
    }
#define UV_RD_W(B, H, WF)                                                                         \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                     \
        WF[i][0] = *(const bf16x8*)(lw + (B) * BUF + (H) * HALF + i * 2048 + lx);                 \
        WF[i][1] = *(const bf16x8*)(lw + (B) * BUF + (H) * HALF + i * 2048 + (lx ^ 64));          \
    }
#define UV_MFMA_H(HM, HN, WF, KS)                                                                 \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                 \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
        if (TRANS)                                                                                \
            acc[HN][HM][i][j] = mfma_16x16x32<F16>(af[j][KS], WF[i][KS], acc[HN][HM][i][j]); \
        else                                                                                      \
            acc[HN][HM][i][j] = mfma_16x16x32<F16>(WF[i][KS], af[j][KS], acc[HN][HM][i][j]); \
    }
#define UV_MFMA_Q(HM, HN, WF)                                                                     \
    __builtin_amdgcn_s_setprio(1);                                                                \
    UV_MFMA_H(HM, HN, WF, 0) UV_MFMA_H(HM, HN, WF, 1)                                             \
    __builtin_amdgcn_s_setprio(0);
#define UV_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// one K tile T living in buffer B (the other buffer is O); ST1: K tile T+1 exists, ST2: K tile T+2 exists
#define UV_KTILE0(T, B, O, ST1, ST2)                                                              \
    UV_RD_W(B, 0, w0) UV_SCHED(); UV_RD_A(B, 0)                                                   \
    if (ST1) UV_STAGE(a_src[1], (T) + 1, (O) * BUF + HALF)                                        \
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");                                            \
    UV_SB(); UV_LGKM0(); UV_SCHED();                                                              \
    UV_MFMA_Q(0, 0, w0) UV_SCHED(); UV_SB();                                                      \
    UV_RD_W(B, 1, w1)                                                                             \
    if (ST2) UV_STAGE(w_src[0], (T) + 2, (B) * BUF + 2 * HALF)                                    \
    UV_SB(); UV_LGKM0(); UV_SCHED();                                                              \
    UV_MFMA_Q(0, 1, w1) UV_SCHED(); UV_SB();                                                      \
    UV_RD_A(B, 1)                                                                                 \
    if (ST2) UV_STAGE(a_src[0], (T) + 2, (B) * BUF)                                               \
    UV_SB(); UV_LGKM0(); UV_SCHED();                                                              \
    UV_MFMA_Q(1, 1, w1) UV_SCHED(); UV_SB();                                                      \
    if (ST2) {                                                                                    \
        UV_STAGE(w_src[1], (T) + 2, (B) * BUF + 3 * HALF)                                         \
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                          \
    } else {                                                                                      \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                          \
    }                                                                                             \
    UV_SB(); UV_SCHED();                                                                          \
    UV_MFMA_Q(1, 0, w0) UV_SCHED(); UV_SB();
// VAR 5: two phases of 32 MFMAs per K tile instead of four of 16 (half as many barriers and cluster ramps per MFMA):
//     phase A   LDS reads W[0] W[1] A[0] (16)   MFMA (A0,W0) (A0,W1)   DMA W[0] A[1] of K tile t+1 (other buffer)
//     phase B   LDS reads A[1] (8)              MFMA (A1,W1) (A1,W0)   DMA W[1] A[0] of K tile t+2 (this buffer)
// (4 + 4 LDS-DMA pieces per wave; UV_REBAL 0 = the schedule of rounds 2-5 with 2 + 6: A[1] of t+1 | W[0] W[1] A[0] of t+2.)
// A half-tile is restaged only after BOTH wave groups have retired their reads of it: the second group runs one barrier behind the
// first, so a piece staged in front of a phase's first barrier may only overwrite what was last read a whole phase earlier - W[0] of the
// OTHER buffer (read in phase A of the previous K tile) qualifies in phase A, W[0] of this buffer does not. Each phase waits on a counted
// vmcnt after its own DMA issue (A: 8, B: 6), which retires exactly the half-tiles the NEXT phase reads.
// UV_KTILE5X names the K-tile indices of its two staging groups separately (K1: the W[0] A[1] halves staged in phase A, K2: W[1]
// A[0] staged in phase B): the persistent kernel stages the NEXT output tile's first K tiles through the same code at a tile's end.
#define UV_KTILE5(T, B, O, ST1, ST2) UV_KTILE5X((T) + 1, (T) + 2, B, O, ST1, ST2)
#define UV_KTILE5X(K1, K2, B, O, ST1, ST2)                                                        \
    UV_RD_W(B, 0, w0) UV_RD_W(B, 1, w1) UV_RD_A(B, 0)                                             \
    if (ST1) {                                                                                    \
        if (UV_REBAL) UV_STAGE(w_src[0], K1, (O) * BUF + 2 * HALF)                                \
        UV_STAGE(a_src[1], K1, (O) * BUF + HALF)                                                  \
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                          \
    } else {                                                                                      \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                          \
    }                                                                                             \
    UV_LGKM0(); UV_SB(); UV_SCHED();                                                              \
    __builtin_amdgcn_s_setprio(1);                                                                \
    UV_MFMA_H(0, 0, w0, 0) UV_MFMA_H(0, 0, w0, 1) UV_MFMA_H(0, 1, w1, 0) UV_MFMA_H(0, 1, w1, 1)   \
    __builtin_amdgcn_s_setprio(0); UV_SCHED(); UV_SB();                                           \
    UV_RD_A(B, 1)                                                                                 \
    if (ST2) {                                                                                    \
        if (!UV_REBAL) UV_STAGE(w_src[0], K2, (B) * BUF + 2 * HALF)                               \
        UV_STAGE(w_src[1], K2, (B) * BUF + 3 * HALF)                                              \
        UV_STAGE(a_src[0], K2, (B) * BUF)                                                         \
        if (UV_REBAL) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                            \
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                     \
    } else if (ST1) {                                                                             \
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                          \
    } else {                                                                                      \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                          \
    }                                                                                             \
    UV_LGKM0(); UV_SB(); UV_SCHED();                                                              \
    __builtin_amdgcn_s_setprio(1);                                                                \
    UV_MFMA_H(1, 1, w1, 0) UV_MFMA_H(1, 1, w1, 1) UV_MFMA_H(1, 0, w0, 0) UV_MFMA_H(1, 0, w0, 1)   \
    __builtin_amdgcn_s_setprio(0); UV_SCHED(); UV_SB();
#define UV_KTILE(T, B, O, ST1, ST2)                                                               \
    if constexpr (VAR == 5) { UV_KTILE5(T, B, O, ST1, ST2) } else { UV_KTILE0(T, B, O, ST1, ST2) }

    if constexpr (VAR == 5) {
        // prologue: W0 W1 A0 A1 of K tile 0, W0 W1 A0 of K tile 1; vmcnt(8) = W0 W1 A0 of K tile 0 landed
        UV_STAGE(w_src[0], 0, 2 * HALF) UV_STAGE(w_src[1], 0, 3 * HALF) UV_STAGE(a_src[0], 0, 0) UV_STAGE(a_src[1], 0, HALF)
        if (!UV_REBAL) UV_STAGE(w_src[0], 1, BUF + 2 * HALF)
        UV_STAGE(w_src[1], 1, BUF + 3 * HALF) UV_STAGE(a_src[0], 1, BUF)
        if (UV_REBAL) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        // prologue: K tile 0 (W0 A0 W1 A1) and W0 A0 W1 of K tile 1; vmcnt(6) = K tile 0 landed
        UV_STAGE(w_src[0], 0, 2 * HALF) UV_STAGE(a_src[0], 0, 0) UV_STAGE(w_src[1], 0, 3 * HALF) UV_STAGE(a_src[1], 0, HALF)
        UV_STAGE(w_src[0], 1, BUF + 2 * HALF) UV_STAGE(a_src[0], 1, BUF) UV_STAGE(w_src[1], 1, BUF + 3 * HALF)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    UV_SB();
    if (wr == 1) UV_SB();  // the second group runs one barrier behind the first

    int t = 0;
    for (; t + 2 < nk; t += 2) {
        UV_KTILE(t, 0, 1, true, true)
        UV_KTILE(t + 1, 1, 0, true, true)
    }
    UV_KTILE(t, 0, 1, true, false)
    UV_KTILE(t + 1, 1, 0, false, false)
    if (wr == 0) UV_SB();

    if constexpr (SK > 1) {
        // ---- publish this slice's partial tile: slab[tile][slice][f][tid] (f = fragment index in the accumulator array's own order).
        // Buffer addressing (one descriptor for the tile's SK slabs, the thread's 16-byte column as the VGPR offset, slab / fragment as
        // the scalar offset): flat addressing would hold one 64-bit address pair per store / load - 300 spilled registers in hipcc's build
        f32x4* const accf = &acc[0][0][0][0];
        const auto slabs = __builtin_amdgcn_make_buffer_rsrc(p.ws_slab + (long)bid * SK * 65536, 0, SK * 262144, 0x00020000);
        const int voff = tid * 16;
        // fragment f = ((hn * 2 + hm) * 2 + i) * 4 + j covers rows m0 + 128 hm + 64 wr + 16 j ..+15: fragments wholly below the matrix (the
        // last row tile of a strip: 96 of 256 rows at 1 120) are neither published nor summed
        const int row_lim = p.M - m0 - wr * 64;      // fragment (hm, j) is live iff 128 hm + 16 j < row_lim (wave-uniform)
#pragma unroll
        for (int f = 0; f < 32; ++f)
            if (((f >> 3) & 1) * 128 + (f & 3) * 16 < row_lim)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, accf[f]), slabs, voff, slice * 262144 + f * 8192, 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // every wave's slab stores are complete (and its K-loop LDS reads retired)
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (keep: the fence's own wait can be dropped by the compiler)
            const int ticket = __hip_atomic_fetch_add(p.ws_cnt + bid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ticket == SK - 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            *(volatile int*)smem = ticket;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int ticket = __builtin_amdgcn_readfirstlane(*(volatile int*)smem);
        if (ticket != SK - 1) return;
        // ---- last arriver: the tile = the slabs summed in slice order, G fragments x SK slabs in flight per thread
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // (the epilogues below reuse smem: everyone has read the ticket)
        constexpr int G = 4;
#pragma unroll
        for (int f0 = 0; f0 < 32; f0 += G) {
            f32x4 v[G][SK];
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int sl = 0; sl < SK; ++sl) {
                    v[g][sl] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if ((((f0 + g) >> 3) & 1) * 128 + ((f0 + g) & 3) * 16 < row_lim)
                        v[g][sl] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(slabs, voff, sl * 262144 + (f0 + g) * 8192, 0));
                }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                f32x4 sum = v[g][0];
#pragma unroll
                for (int sl = 1; sl < SK; ++sl)
#pragma unroll
                    for (int e = 0; e < 4; ++e) sum[e] = __fadd_rn(sum[e], v[g][sl][e]);
                // the sum must EXIST here: left alone, hipcc sinks all 32 x (SK - 1) x 4 additions behind the last load (their only use is
                // the epilogue) and keeps 32 x SK loaded fragments alive meanwhile - 280 spilled registers
                asm volatile("" : "+v"(sum));
                accf[f0 + g] = sum;
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    if constexpr ((EPI == UV_EPI_RESID_F32 || EPI == UV_EPI_GATE_RESID_F32) && VAR != 0) {
        // every wave has left the K loop's last LDS reads behind (the loop ends on a barrier both groups take)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        epi_rmw_rows_lds<EPI, F16>(p, smem, acc, m0, n0, wave, lane);
    } else if constexpr (EPI == UV_EPI_RESID_F32 || EPI == UV_EPI_GATE_RESID_F32) {
        // VAR 0 (tile_cfg 14, the A/B reference): the fragment-wise read-modify-write epilogue
        int mb[32], nb[32];
        f32x4 av[32];
#pragma unroll
        for (int hm = 0; hm < 2; ++hm)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int hn = 0; hn < 2; ++hn)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int f = ((hm * 4 + j) * 2 + hn) * 2 + i;
                        mb[f] = m0 + hm * 128 + wr * 64 + j * 16;
                        nb[f] = n0 + hn * 128 + wc * 32 + i * 16;
                        av[f] = acc[hn][hm][i][j];
                    }
        epi_rmw_pipe<EPI, 32, 8, F16>(p, mb, nb, av, frow, fq);
    } else if constexpr (EPI == UV_EPI_BF16_SSQ) {
#pragma unroll
        for (int hm = 0; hm < 2; ++hm)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int hn = 0; hn < 2; ++hn)
                    epi_pair_ssq<F16>(p, m0 + hm * 128 + wr * 64 + j * 16, n0 + hn * 128 + wc * 32, acc[hn][hm][0][j], acc[hn][hm][1][j], frow, fq);
    } else {
#pragma unroll
        for (int hm = 0; hm < 2; ++hm)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int hn = 0; hn < 2; ++hn)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        epi_frag<EPI, F16>(p, m0 + hm * 128 + wr * 64 + j * 16, n0 + hn * 128 + wc * 32 + i * 16, acc[hn][hm][i][j], frow, fq);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// PERSISTENT form of the ping-pong kernel (the default for the large projections): one workgroup per CU walks a list of
// 256x256 output tiles instead of one workgroup per tile. In the one-tile-per-workgroup launch a CU pays, per tile, the
// dispatch of the next workgroup (3.7 us between a workgroup's end and its successor's entry, in-kernel stamps) and the
// pipeline fill (2.0 us until the first K tile has landed) on top of 70 us of K loop and 5.4 us of epilogue. Here
//   * the tile list of a workgroup is fixed by its id: XCD x (workgroups b with b % 8 == x) owns the same contiguous range of
//     logical tiles as in the dynamic launch and its G = gridDim / 8 workgroups take them round-robin, so the tiles in flight
//     on an XCD at any time are G consecutive ones of the 4-tall column walk - the L2 footprint is unchanged;
//   * the last two K tiles of an output tile stage the FIRST two K tiles of the next one through the same schedule (the
//     source pointers move to the next tile exactly where the schedule stops needing the old ones), so the next K loop starts
//     with its operands landed: no fill, no prologue. The read-modify-write epilogues stage y through the same LDS and
//     therefore restart with a prologue instead (they still save the dispatch).
// Per-tile arithmetic (K order, MFMA order, epilogue) is the one of gemm_bf16_8ph_kernel<EPI, 5>: results are bit-identical.
// Needs M % 256 == 0 and N % 256 == 0 (no clamped rows: the host splits ragged rows off to the small-tile kernel), an even
// K / 64 >= 4 and gridDim % 8 == 0.
// (Tried and dropped: waiting for the prefetched K tiles ahead of the epilogue and skipping the first K tile's waits, so that the
// epilogue's stores drain behind the next K loop - as a peeled first K tile or as a run-time predicate on the waits it costs
// 40-60 spilled registers in hipcc's allocation of this kernel and 10-50 % of its speed.)
template <int EPI, bool F16 = false>
__global__ __launch_bounds__(512) void gemm_bf16_8ph_persist_kernel(GemmArgs p) {
    constexpr bool TRANS = (EPI == UV_EPI_BF16_T);
    constexpr bool RMW = (EPI == UV_EPI_RESID_F32 || EPI == UV_EPI_GATE_RESID_F32);
    constexpr int VAR = 5;           // the macros below are shared with gemm_bf16_8ph_kernel
    constexpr int HALF = 16384;
    constexpr int BUF = 4 * HALF;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- this workgroup's tile list
    const int nblk = p.tiles_m * p.tiles_n;
    const int G = gridDim.x >> 3;
    int first, cnt;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7;
        first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        cnt = q + (xcd < r ? 1 : 0);
    }
    int cur = blockIdx.x >> 3;
    if (cur >= cnt) return;
    const int GM = p.gm;
    const int group_sz = GM * p.tiles_n;
    auto origin = [&](int bid, int& m0, int& n0) {
        const int group = bid / group_sz;
        const int first_m = group * GM;
        const int gm = min(GM, p.tiles_m - first_m);
        const int in_group = bid - group * group_sz;
        m0 = (first_m + in_group % gm) * 256;
        n0 = (in_group / gm) * 256;
    };
    int m0, n0;
    origin(first + cur, m0, n0);

    const int srow = lane >> 3, pchunk = lane & 7;
    const bf16_t* a_src[2][2];
    const bf16_t* w_src[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (i * 8 + wave) * 8 + srow;
            const int c = pchunk ^ ((row >> 1) & 7);
            a_src[h][i] = p.A + (long)(m0 + h * 128 + row) * p.lda + c * 8;
            w_src[h][i] = p.W + (long)(n0 + h * 128 + row) * p.ldw + c * 8;
        }
    char* const my_dst = smem + wave * 1024;
    const int nk = p.K / UV_BK;
    const int frow = lane & 15, fq = lane >> 4;
    const int lx = (fq ^ (frow >> 1)) << 4;
    const char* const la = smem + (wr * 64 + frow) * 128;
    const char* const lw = smem + 2 * HALF + (wc * 32 + frow) * 128;
    f32x4 acc[2][2][2][4];
    bf16x8 af[4][2], w0[2][2], w1[2][2];

    auto move_ptrs = [&](const bf16_t* (&src)[2], long d) { src[0] += d; src[1] += d; };
    bool fresh = true;               // the next K loop needs the prologue (first tile; every tile of the LDS-staged epilogues)
    for (;;) {
        if (fresh) {
            // prologue: W0 W1 A0 A1 of K tile 0, W0 W1 A0 of K tile 1; vmcnt(8) = W0 W1 A0 of K tile 0 landed
            UV_STAGE(w_src[0], 0, 2 * HALF) UV_STAGE(w_src[1], 0, 3 * HALF) UV_STAGE(a_src[0], 0, 0) UV_STAGE(a_src[1], 0, HALF)
            if (!UV_REBAL) UV_STAGE(w_src[0], 1, BUF + 2 * HALF)
            UV_STAGE(w_src[1], 1, BUF + 3 * HALF) UV_STAGE(a_src[0], 1, BUF)
            if (UV_REBAL) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            UV_SB();
        }
        fresh = RMW;
        if (wr == 1) UV_SB();        // the second group runs one barrier behind the first
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[a][b][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        const int nxt = cur + G;
        const bool has_next = nxt < cnt;
        int m1 = m0, n1 = n0;
        long dA = 0, dW = 0;         // element offsets from this tile's operand panels to the next tile's (0: the last tile)
        if (has_next) {
            origin(first + nxt, m1, n1);
            dA = (long)(m1 - m0) * p.lda;
            dW = (long)(n1 - n0) * p.ldw;
        }
        int t = 0;
        for (; t + 2 < nk; t += 2) {
            UV_KTILE5(t, 0, 1, true, true)
            UV_KTILE5(t + 1, 1, 0, true, true)
        }
        if constexpr (!RMW) {
            // K tiles nk-2 and nk-1 of this output tile, ONE code path for every tile (a second copy of the MFMA schedule behind
            // a branch makes hipcc shuttle all 128 accumulators through copies and spill): staged meanwhile are A[1] of K tile
            // nk-1 (old pointer) and then K tiles 0 and 1 of the next output tile. The workgroup's last tile "prefetches" its own
            // first K tiles again (dA = dW = 0: 128 KiB of harmless loads, drained before the workgroup ends).
            if (!UV_REBAL) move_ptrs(w_src[0], dW);
            move_ptrs(w_src[1], dW); move_ptrs(a_src[0], dA);
            UV_KTILE5X(nk - 1, 0, 0, 1, true, true)
            if (UV_REBAL) move_ptrs(w_src[0], dW);      // (W[0] travels with A[1]: phase A's pieces belong to the K tile after this one)
            move_ptrs(a_src[1], dA);
            UV_KTILE5X(0, 1, 1, 0, true, true)
        } else {
            UV_KTILE5(t, 0, 1, true, false)
            UV_KTILE5(t + 1, 1, 0, false, false)
            move_ptrs(w_src[0], dW); move_ptrs(w_src[1], dW); move_ptrs(a_src[0], dA); move_ptrs(a_src[1], dA);
        }
        if (wr == 0) UV_SB();        // both groups have left the K loop

        if constexpr (RMW) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            epi_rmw_rows_lds<EPI, F16>(p, smem, acc, m0, n0, wave, lane);
            if (has_next) {          // the next prologue overwrites the staging area: every wave must have read its rows
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        } else if constexpr (EPI == UV_EPI_BF16 || EPI == UV_EPI_GELU_BF16 || EPI == UV_EPI_BF16_SSQ) {
#pragma unroll
            for (int hm = 0; hm < 2; ++hm)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int hn = 0; hn < 2; ++hn)      // the wave's two column fragments (i = 0, 1) are adjacent: one 16-byte store per lane
                        epi_pair16<EPI, F16>(p, m0 + hm * 128 + wr * 64 + j * 16, n0 + hn * 128 + wc * 32, acc[hn][hm][0][j], acc[hn][hm][1][j], frow, fq);
        } else if constexpr (EPI == UV_EPI_BF16_T) {
#pragma unroll
            for (int hm = 0; hm < 2; ++hm)
#pragma unroll
                for (int j = 0; j < 4; j += 2)          // row fragments j, j+1 are adjacent along the transposed output's rows
#pragma unroll
                    for (int hn = 0; hn < 2; ++hn)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            epi_pair16_T<F16>(p, m0 + hm * 128 + wr * 64 + j * 16, n0 + hn * 128 + wc * 32 + i * 16, acc[hn][hm][i][j], acc[hn][hm][i][j + 1], frow, fq);
        } else {
#pragma unroll
            for (int hm = 0; hm < 2; ++hm)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int hn = 0; hn < 2; ++hn)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            epi_frag<EPI, F16, true>(p, m0 + hm * 128 + wr * 64 + j * 16, n0 + hn * 128 + wc * 32 + i * 16, acc[hn][hm][i][j], frow, fq);
        }
        if (!has_next) break;
        m0 = m1; n0 = n1; cur = nxt;
    }
    // the last tile's self-prefetch (and every store) must have landed before the workgroup's LDS is handed on
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int VAR = 0, bool F16 = false>
static int launch_8ph(const GemmArgs& a0, int epi, hipStream_t stream) {
    GemmArgs a = a0;
    a.tiles_m = (a.M + 255) / 256;
    a.tiles_n = (a.N + 255) / 256;
    const dim3 grid(a.tiles_m * a.tiles_n), block(512);
    const size_t lds = 128 * 1024;
#define UV_LAUNCH8(E)                                                                              \
    case E: {                                                                                      \
        auto kern = gemm_bf16_8ph_kernel<E, VAR, F16>;                                               \
        UV_ONCE_PER_DEVICE(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));  \
        hipLaunchKernelGGL(kern, grid, block, lds, stream, a);                                     \
        break;                                                                                     \
    }
    switch (epi) {
        UV_LAUNCH8(UV_EPI_BF16)
        UV_LAUNCH8(UV_EPI_GELU_BF16)
        UV_LAUNCH8(UV_EPI_F32_FROM_BF16)
        UV_LAUNCH8(UV_EPI_RESID_F32)
        UV_LAUNCH8(UV_EPI_GATE_RESID_F32)
        UV_LAUNCH8(UV_EPI_BF16_T)
        UV_LAUNCH8(UV_EPI_BF16_SSQ)
        default:
            uv_set_error("uv_gemm_bf16_nt: unknown epilogue %d", epi);
            return -1;
    }
#undef UV_LAUNCH8
    UV_CHECK_LAUNCH("uv_gemm_bf16_nt");
    return 0;
}

// Split-K launch of the one-tile-per-workgroup ping-pong kernel (see gemm_bf16_8ph_kernel<.., SK>): workspace = [4 KiB of tile counters]
// [tiles x SK slabs of 256 KiB]. The counters are zeroed by a memset node on the stream in front of EVERY launch (a capture records it).
static inline long splitk_ws_bytes(int M, int N, int sk) { return 4096 + (long)((M + 255) / 256) * ((N + 255) / 256) * sk * 262144L; }

template <int SK, bool F16 = false>
static int launch_8ph_splitk(const GemmArgs& a0, int epi, hipStream_t stream, void* ws, long ws_bytes) {
    GemmArgs a = a0;
    a.tiles_m = (a.M + 255) / 256;
    a.tiles_n = (a.N + 255) / 256;
    const int tiles = a.tiles_m * a.tiles_n;
    UV_CHECK_ARG(a.K % (SK * 128) == 0 && a.K / SK >= 256, "uv_gemm_bf16_nt_ws: split-K %d needs K %% %d == 0 and K / %d >= 256 (K=%d)", SK, SK * 128, SK, a.K);
    UV_CHECK_ARG(tiles <= 1024, "uv_gemm_bf16_nt_ws: split-K serves leftover strips (at most 1024 tiles; %d here)", tiles);
    UV_CHECK_ARG(ws && ((uintptr_t)ws & 255) == 0 && ws_bytes >= splitk_ws_bytes(a.M, a.N, SK),
                 "uv_gemm_bf16_nt_ws: workspace of %ld bytes (256-byte aligned) needed, %ld given", splitk_ws_bytes(a.M, a.N, SK), ws_bytes);
    a.ws_cnt = (int*)ws;
    a.ws_slab = (float*)((char*)ws + 4096);
    if (hipMemsetAsync(a.ws_cnt, 0, (size_t)tiles * sizeof(int), stream) != hipSuccess) {
        uv_set_error("uv_gemm_bf16_nt_ws: hipMemsetAsync of the tile counters failed");
        return -1;
    }
    const int per = a.gm == 0 ? (tiles * SK + 7) / 8 : (tiles + 8 / SK - 1) / (8 / SK);
    const dim3 grid(8 * per), block(512);
    const size_t lds = 128 * 1024;
#define UV_LAUNCH8S(E)                                                                             \
    case E: {                                                                                      \
        auto kern = gemm_bf16_8ph_kernel<E, 5, F16, SK>;                                           \
        UV_ONCE_PER_DEVICE(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));  \
        hipLaunchKernelGGL(kern, grid, block, lds, stream, a);                                     \
        break;                                                                                     \
    }
    switch (epi) {
        UV_LAUNCH8S(UV_EPI_BF16)
        UV_LAUNCH8S(UV_EPI_RESID_F32)
        UV_LAUNCH8S(UV_EPI_GATE_RESID_F32)
        default:
            uv_set_error("uv_gemm_bf16_nt_ws: the split-K strip is built for the bf16 and the residual epilogues (0, 3, 4), not %d", epi);
            return -1;
    }
#undef UV_LAUNCH8S
    UV_CHECK_LAUNCH("uv_gemm_bf16_nt_ws");
    return 0;
}

template <bool F16 = false>
static int launch_8ph_persist(const GemmArgs& a0, int epi, hipStream_t stream) {
    GemmArgs a = a0;
    UV_CHECK_ARG(a.M % 256 == 0 && a.N % 256 == 0, "uv_gemm_bf16_nt: the persistent kernel needs whole 256x256 tiles (M=%d N=%d)", a.M, a.N);
    UV_CHECK_ARG(a.ldo % 8 == 0 || (epi != UV_EPI_BF16_T && epi != UV_EPI_BF16 && epi != UV_EPI_GELU_BF16 && epi != UV_EPI_BF16_SSQ),
                 "uv_gemm_bf16_nt: the persistent kernel stores 16 bytes per lane: ldo=%ld must be a multiple of 8 elements", a.ldo);
    a.tiles_m = a.M / 256;
    a.tiles_n = a.N / 256;
    // height of the tile walk's column groups (same-process A/B on the DiT shapes, tools/gemm_gm_ab.py: N = K = 3072 325 -> 319 us at 8,
    // N = 14336 best at 4, K = 14336 1438 -> 1428 us at 2; the results do not depend on it)
    a.gm = (a.N <= 4096 && a.K <= 4096) ? 8 : (a.K >= 8192 ? 2 : 4);
    if (const int gm = uv_option(UV_OPT_GEMM_GM); gm > 0) a.gm = gm;                  // developer A/B switch (uv_set_option)
    const int tiles = a.tiles_m * a.tiles_n;
    int wgs = uv_num_cus() & ~7;
    if (wgs > tiles) wgs = tiles & ~7;
    UV_CHECK_ARG(wgs >= 8, "uv_gemm_bf16_nt: too few tiles (%d) for the persistent kernel", tiles);
    const dim3 grid(wgs), block(512);
    const size_t lds = 128 * 1024;
#define UV_LAUNCH8P(E)                                                                             \
    case E: {                                                                                      \
        auto kern = gemm_bf16_8ph_persist_kernel<E, F16>;                                          \
        UV_ONCE_PER_DEVICE(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));  \
        hipLaunchKernelGGL(kern, grid, block, lds, stream, a);                                     \
        break;                                                                                     \
    }
    switch (epi) {
        UV_LAUNCH8P(UV_EPI_BF16)
        UV_LAUNCH8P(UV_EPI_GELU_BF16)
        UV_LAUNCH8P(UV_EPI_F32_FROM_BF16)
        UV_LAUNCH8P(UV_EPI_RESID_F32)
        UV_LAUNCH8P(UV_EPI_GATE_RESID_F32)
        UV_LAUNCH8P(UV_EPI_BF16_T)
        UV_LAUNCH8P(UV_EPI_BF16_SSQ)
        default:
            uv_set_error("uv_gemm_bf16_nt: unknown epilogue %d", epi);
            return -1;
    }
#undef UV_LAUNCH8P
    UV_CHECK_LAUNCH("uv_gemm_bf16_nt");
    return 0;
}

template <int BM, int BN, int WM, int WN, int NS = 2, bool F16 = false>
static int launch_cfg(const GemmArgs& a0, int epi, hipStream_t stream) {
    GemmArgs a = a0;
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.N + BN - 1) / BN;
    const dim3 grid(a.tiles_m * a.tiles_n), block(WM * WN * 64);
    const size_t lds = NS * (BM + BN) * 128;
#define UV_LAUNCH(E)                                                                               \
    case E: {                                                                                      \
        auto kern = gemm_bf16_nt_kernel<BM, BN, WM, WN, E, NS, F16>;                                \
        UV_ONCE_PER_DEVICE(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));  \
        hipLaunchKernelGGL(kern, grid, block, lds, stream, a);                                     \
        break;                                                                                     \
    }
    switch (epi) {
        UV_LAUNCH(UV_EPI_BF16)
        UV_LAUNCH(UV_EPI_GELU_BF16)
        UV_LAUNCH(UV_EPI_F32_FROM_BF16)
        UV_LAUNCH(UV_EPI_RESID_F32)
        UV_LAUNCH(UV_EPI_GATE_RESID_F32)
        UV_LAUNCH(UV_EPI_BF16_T)
        case UV_EPI_BF16_SSQ:
            if constexpr ((BN / WN) % 32 == 0) {
                auto kern = gemm_bf16_nt_kernel<BM, BN, WM, WN, UV_EPI_BF16_SSQ, NS, F16>;
                UV_ONCE_PER_DEVICE(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
                break;
            } else {
                uv_set_error("uv_gemm_bf16_nt_ssq: this tile shape has no whole 32-column groups per wave");
                return -1;
            }
        default:
            uv_set_error("uv_gemm_bf16_nt: unknown epilogue %d", epi);
            return -1;
    }
#undef UV_LAUNCH
    UV_CHECK_LAUNCH("uv_gemm_bf16_nt");
    return 0;
}

