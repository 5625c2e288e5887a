// Argument block and shared constants of the flash-attention kernels (attention.hip; tools/diag/attn_pw4.hip).
#pragma once
#include "common.h"

#define UV_ATT_QW 32     // queries per 32x32 MFMA block
#define UV_ATT_KV 64     // keys per staged tile
#define UV_ATT_DEFER 8.0f   // log2 of the largest P allowed before the reference maximum is moved

struct AttnArgs {
    const bf16_t* q;   // [Lq, ldq]   head h at column h*128
    const bf16_t* k;   // [Lk, ldk]
    const bf16_t* vt;  // [H*128, ldvt]  (V transposed; ldvt >= roundup(Lk, 64), pad finite)
    bf16_t* out;       // [Lq, ldo]
    long ldq, ldk, ldvt, ldo;
    int Lq, Lk, H, q_blocks, batch;
    int n12;           // flash_attn_fwd12_kernel: query blocks per head that own 12 units; the other q_blocks - n12 own 8
    float scale_log2;  // softmax_scale * log2(e)
    // uv_flash_attn_bf16_qnorm: q holds the RAW projection; the kernels' Q prologue applies WanRMSNorm to it (model.py:82-85, 138 / 169):
    // q_rs [batch * Lq] f32 = 1 / sqrt(mean(q_row^2) + eps) over ALL heads' columns (uv_rms_scale_from_ssq), q_w [H * 128] f32 = norm_q.weight
    const float* q_rs;
    const float* q_w;
};

// Q fragments of one wave: lane (r, h) holds Q[qrow][hcol + 16 kk + 8 h .. + 7], kk = 0 .. NKK-1. QN = 1: q is the RAW projection and the row's
// RMSNorm is applied by attn_apply_qnorm with the rounding points of rmsnorm_rope_kernel (dit_glue.hip): bf16( bf16(q * rs) * w ), products in
// f32 - bit for bit what that kernel writes for the same rs. Two steps on purpose: the loads (q, the row scale, the lane's 8 NKK weights) are
// ISSUED with the Q loads at the top of the kernel, the arithmetic runs behind the first K / V tile's LDS-DMA issue - applied right at the load it
// made the prologue wait for the Q data before staging anything (measured: + 3 us per cross-attention workgroup).
template <int NKK, int QN>
struct AttnQNorm {
    float rs;
    f32x4 w[QN ? 2 * NKK : 1];
};

template <int NKK, int QN>
__device__ __forceinline__ void attn_load_q(const AttnArgs& p, bf16x8 (&qf)[NKK], AttnQNorm<NKK, QN>& qn, int qrow, long hcol, int h) {
    const bf16_t* qp = p.q + (long)qrow * p.ldq + hcol + 8 * h;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) qf[kk] = *(const bf16x8*)(qp + 16 * kk);
    if constexpr (QN >= 1) {
        qn.rs = p.q_rs[qrow];
        const float* wp = p.q_w + hcol + 8 * h;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            qn.w[2 * kk] = *(const f32x4*)(wp + 16 * kk);
            qn.w[2 * kk + 1] = *(const f32x4*)(wp + 16 * kk + 4);
        }
    }
}

template <int NKK, int QN>
__device__ __forceinline__ void attn_apply_qnorm(bf16x8 (&qf)[NKK], const AttnQNorm<NKK, QN>& qn) {
    if constexpr (QN >= 1) {
        __builtin_amdgcn_sched_barrier(0);       // (not hoisted above the staging that precedes the call)
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const u32x4 raw = __builtin_bit_cast(u32x4, qf[kk]);
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = bf2f((bf16_t)(raw[e] & 0xffff)), hi = bf2f((bf16_t)(raw[e] >> 16));
                const f32x4& wv = qn.w[2 * kk + (e >> 1)];
                o[e] = pack_bf2(__fmul_rn(round_bf(__fmul_rn(lo, qn.rs)), wv[(2 * e) & 3]), __fmul_rn(round_bf(__fmul_rn(hi, qn.rs)), wv[(2 * e + 1) & 3]));
            }
            qf[kk] = __builtin_bit_cast(bf16x8, o);
        }
    }
}

__device__ __forceinline__ int perm23(int i) {  // swap bits 2 and 3
    return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1);
}
