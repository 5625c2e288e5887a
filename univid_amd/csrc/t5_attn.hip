// umT5 text-encoder attention for gfx950 (reference models/wan/utils/modules/t5.py:93-120, bf16 module).
//
// One prompt is at most 512 tokens (text_len) and runs once per generation, so this kernel is written for FIDELITY to the
// reference's rounding points, not for the MFMA roofline (umT5-XXL at 512 tokens: 4.3 GFLOP of attention per layer):
//     attn = einsum(q, k)            -> bf16            (fp32 accumulate, one rounding)
//     attn = attn + pos_bias         -> bf16
//     attn = softmax(attn.float())   -> bf16            (normalised BEFORE the rounding - a flash-style kernel rounds the
//                                                        unnormalised exp and divides at the end, which the tiny golden model
//                                                        amplifies to 1.7e-2 relative rms at the encoder output)
//     x    = einsum(attn, v)         -> bf16
// One wave per (query, head): lanes over keys for the scores and the softmax (wave reductions), the probabilities go through
// LDS, lanes over the 64 head dimensions for P.V (one coalesced 128-byte V row per key).
#include "common.h"

#define UV_T5_MAXN 1024

__global__ __launch_bounds__(256) void t5_attention_kernel(const bf16_t* q, long ldq, const bf16_t* k, long ldk, const bf16_t* v, long ldv,
                                                           bf16_t* out, long ldo, int n, int H, const float* rel_bias, int span) {
    __shared__ float p_sh[4][UV_T5_MAXN];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int head = blockIdx.y;
    const int qi = blockIdx.x * 4 + wave;
    if (qi >= n) return;
    const long hc = (long)head * 64;
    // the query row, all 64 values in every lane
    float qv[64];
    {
        const bf16_t* qp = q + (long)qi * ldq + hc;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const u32x4 raw = *(const u32x4*)(qp + c * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                qv[c * 8 + 2 * e] = bf2f((bf16_t)(raw[e] & 0xffff));
                qv[c * 8 + 2 * e + 1] = bf2f((bf16_t)(raw[e] >> 16));
            }
        }
    }
    const float* brow = rel_bias + (long)head * (2 * span - 1) + (span - 1) - qi;
    // scores of this lane's keys (key = lane + 64*i)
    float s[UV_T5_MAXN / 64];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < UV_T5_MAXN / 64; ++i) {
        const int key = lane + 64 * i;
        s[i] = -INFINITY;
        if (key < n) {
            const bf16_t* kp = k + (long)key * ldk + hc;
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const u32x4 raw = *(const u32x4*)(kp + c * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc = fmaf(qv[c * 8 + 2 * e], bf2f((bf16_t)(raw[e] & 0xffff)), acc);
                    acc = fmaf(qv[c * 8 + 2 * e + 1], bf2f((bf16_t)(raw[e] >> 16)), acc);
                }
            }
            s[i] = round_bf(round_bf(acc) + brow[key]);
            m = fmaxf(m, s[i]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < UV_T5_MAXN / 64; ++i) {
        const int key = lane + 64 * i;
        if (key < n) {
            s[i] = expf(s[i] - m);
            l += s[i];
        }
    }
    l = wave_sum(l);
#pragma unroll
    for (int i = 0; i < UV_T5_MAXN / 64; ++i) {
        const int key = lane + 64 * i;
        if (key < n) p_sh[wave][key] = round_bf(s[i] / l);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's own LDS writes are visible to its own reads below
    // out[d] = sum_j P[j] * V[j][d], lane = d
    float acc = 0.f;
    const bf16_t* vp = v + hc + lane;
    for (int j = 0; j < n; ++j) acc = fmaf(p_sh[wave][j], bf2f(vp[(long)j * ldv]), acc);
    out[(long)qi * ldo + hc + lane] = f2bf(acc);
}

extern "C" int uv_t5_attention_bf16(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* out, long ldo,
                                    int n, int H, const float* rel_bias, int span, void* stream) {
    UV_CHECK_ARG(q && k && v && out && rel_bias, "uv_t5_attention_bf16: null pointer");
    UV_CHECK_ARG(n > 0 && n <= UV_T5_MAXN && H > 0 && span >= n, "uv_t5_attention_bf16: bad shape n=%d H=%d span=%d (n <= %d)", n, H, span,
                 UV_T5_MAXN);
    UV_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && (((uintptr_t)q | (uintptr_t)k) & 15) == 0, "uv_t5_attention_bf16: q / k must be 16-byte aligned rows");
    hipLaunchKernelGGL(t5_attention_kernel, dim3((n + 3) / 4, H), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk,
                       (const bf16_t*)v, ldv, (bf16_t*)out, ldo, n, H, rel_bias, span);
    UV_CHECK_LAUNCH("uv_t5_attention_bf16");
    return 0;
}
