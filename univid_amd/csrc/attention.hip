// Flash attention forward (non-causal, head_dim 128, bf16 in / fp32 softmax+accumulate / bf16 out)
// for gfx950. Serves both the DiT spatiotemporal self-attention (Lq = Lk = L) and the text
// cross-attention (Lk = 512).
//
// Replaces: flash_attention()  models/wan/utils/modules/attention.py:24-130  (FA2/FA3 varlen call,
//           q/k/v cast to bf16 :59-83, result cast back :130), called from
//           WanSelfAttention.forward model.py:145-150 and WanCrossAttention.forward model.py:175.
//
// Structure (one workgroup = 8 waves = 256 queries of one head; KV tile = 64 keys):
//   * swapped product S^T = K.Q^T with v_mfma_f32_32x32x16_bf16: the query sits on the lane, its
//     32 keys of a tile sit in the 16 accumulator registers of both half-waves, so the softmax row
//     max / sum are register-local plus ONE exchange with lane^32.
//   * the S^T accumulator is reused in place as the B operand of O^T = V^T.P^T (no LDS round trip);
//     O^T keeps the query on the lane too, so the online-softmax rescale is a plain per-lane multiply.
//   * K rows are written to LDS in an order with bits 2 and 3 of the key index swapped; with that
//     permutation the 8 keys a lane needs from V^T for one PV k-step are CONTIGUOUS (one
//     ds_read_b128). Softmax is invariant to the key order, masks use the true key index.
//   * V arrives already transposed ([H*128, Lk_pad], written by the V-projection GEMM epilogue).
//   * K tile [64][128] (256-B rows) swizzle chunk ^= row&15; V^T tile [128][64] (128-B rows) swizzle
//     chunk ^= (row>>1)&7: both make the ds_read_b128 fragment reads bank-conflict-free.
//   * register-staged double buffer: tile t+1 is fetched to VGPRs before the MFMAs of tile t and
//     written to the other LDS buffer after them; one barrier per tile.
#include "common.h"

#define UV_ATT_QW 32     // queries per wave
#define UV_ATT_WAVES 8
#define UV_ATT_QB (UV_ATT_QW * UV_ATT_WAVES)
#define UV_ATT_KV 64

struct AttnArgs {
    const bf16_t* q;   // [Lq, ldq]   head h at column h*128
    const bf16_t* k;   // [Lk, ldk]
    const bf16_t* vt;  // [H*128, ldvt]  (V transposed; ldvt >= roundup(Lk, 64), pad finite)
    bf16_t* out;       // [Lq, ldo]
    long ldq, ldk, ldvt, ldo;
    int Lq, Lk, H, q_blocks;
    float scale_log2;  // softmax_scale * log2(e)
};

__device__ __forceinline__ int perm23(int i) {  // swap bits 2 and 3
    return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1);
}

// D = head_dim (128 for TI2V-5B; 64 for the reference's CPU-runnable tiny config).
template <int D>
__global__ __launch_bounds__(UV_ATT_WAVES * 64) void flash_attn_fwd_kernel(AttnArgs p) {
    constexpr int KROW = 2 * D;                 // bytes per K row in LDS (256 or 128)
    constexpr int KCH = D / 8;                  // 16-B chunks per K row
    constexpr int NKK = D / 16;                 // MFMA k-steps over the head dim
    constexpr int ND = D / 32;                  // 32-row d tiles of O^T
    constexpr int NCH = D / 64;                 // staging chunks per thread per operand
    constexpr int K_BYTES = UV_ATT_KV * KROW;   // 16 KiB at D=128
    constexpr int V_BYTES = D * 128;            // 16 KiB at D=128
    constexpr int STAGE = K_BYTES + V_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    // head-major block order: consecutive block ids walk the q-blocks of one head
    const int head = blockIdx.x / p.q_blocks;
    const int qb = blockIdx.x - head * p.q_blocks;
    const int q0w = qb * UV_ATT_QB + wave * UV_ATT_QW;
    const long hcol = (long)head * D;

    // ---- Q fragments (B operand of S^T = K.Q^T): lane (r,h) holds Q[q0w+r][16kk+8h .. +7]
    bf16x8 qf[NKK];
    {
        const int qrow = min(q0w + r, p.Lq - 1);
        const bf16_t* qp = p.q + (long)qrow * p.ldq + hcol + 8 * h;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) qf[kk] = *(const bf16x8*)(qp + 16 * kk);
    }

    // ---- staging maps: NCH K chunks + NCH V^T chunks (16 B each) per thread per tile
    const bf16_t* ksrc[NCH];
    int kdst[NCH], krow[NCH];
    const bf16_t* vsrc[NCH];
    int vdst[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int id = tid + 512 * i;
        const int row = id / KCH, c = id % KCH;  // key row in tile, 16-B chunk of the K row
        const int lrow = perm23(row);
        krow[i] = row;
        ksrc[i] = p.k + hcol + c * 8;
        kdst[i] = lrow * KROW + ((c ^ (D == 128 ? (lrow & 15) : ((lrow >> 1) & 7))) << 4);
        const int drow = id >> 3, vc = id & 7;  // d row, 16-B chunk (8 keys) of the 128-B row
        vsrc[i] = p.vt + (hcol + drow) * p.ldvt + vc * 8;
        vdst[i] = K_BYTES + drow * 128 + ((vc ^ ((drow >> 1) & 7)) << 4);
    }
    u32x4 kreg[NCH], vreg[NCH];
    auto fetch = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int kr = min(kv0 + krow[i], p.Lk - 1);
            kreg[i] = *(const u32x4*)(ksrc[i] + (long)kr * p.ldk);
            vreg[i] = *(const u32x4*)(vsrc[i] + kv0);
        }
    };
    auto commit = [&](int buf) {
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            *(u32x4*)(base + kdst[i]) = kreg[i];
            *(u32x4*)(base + vdst[i]) = vreg[i];
        }
    };

    // fragment read offsets
    //   K  : LDS row 32T + r, logical chunk 2kk + h, phys = chunk ^ (row & 15) on 256-B rows (D=128),
    //        chunk ^ ((row>>1)&7) on 128-B rows (D=64); both keys depend on r only
    //   V^T: LDS row 32dt + r, logical chunk 4T + 2s + h, phys = chunk ^ ((row>>1)&7); ((32dt+r)>>1)&7 == (r>>1)&7
    const int k_row_off = r * KROW;
    const int k_key = (D == 128) ? (r & 15) : ((r >> 1) & 7);
    const int v_row_off = K_BYTES + r * 128;
    const int v_key = (r >> 1) & 7;

    f32x16 oacc[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
    float m_run = -INFINITY;  // running max of raw scores (both half-waves hold the same value)
    float l_run = 0.f;        // this half-wave's partial row sum

    const int nt = (p.Lk + UV_ATT_KV - 1) / UV_ATT_KV;
    fetch(0);
    commit(0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int kv0 = t * UV_ATT_KV;
        const char* base = smem + (t & 1) * STAGE;
        if (t + 1 < nt) fetch(kv0 + UV_ATT_KV);

        // ---- S^T = K . Q^T  (two 32-key tiles)
        f32x16 sacc[2];
#pragma unroll
        for (int T = 0; T < 2; ++T) {
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[T][e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const bf16x8 kf =
                    *(const bf16x8*)(base + T * 32 * KROW + k_row_off + (((2 * kk + h) ^ k_key) << 4));
                sacc[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], sacc[T], 0, 0, 0);
            }
        }

        // ---- mask the ragged last tile with the TRUE key index of each accumulator row
        if (kv0 + UV_ATT_KV > p.Lk) {
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = 32 * T + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (kv0 + perm23(i) >= p.Lk) sacc[T][e] = -INFINITY;
                }
        }

        // ---- online softmax (query on the lane)
        float mt = sacc[0][0];
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int e = 0; e < 16; ++e) mt = fmaxf(mt, sacc[T][e]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.scale_log2);
        const float mneg = -m_new * p.scale_log2;
        m_run = m_new;
        float psum = 0.f;
        bf16x8 pf[2][2];
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[T][8 * s + j], p.scale_log2, mneg));
                    psum += pv;
                    pf[T][s][j] = (__bf16)pv;
                }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;

        // ---- O^T += V^T . P^T
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 vf =
                        *(const bf16x8*)(base + d * 32 * 128 + v_row_off + (((4 * T + 2 * s + h) ^ v_key) << 4));
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[T][s], oacc[d], 0, 0, 0);
                }

        if (t + 1 < nt) commit((t + 1) & 1);
        __syncthreads();
    }

    // ---- finish: combine the two half-wave sums, normalise, store bf16 rows
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q = q0w + r;
    if (q < p.Lq) {
        bf16_t* op = p.out + (long)q * p.ldo + hcol + 4 * h;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 o = {pack_bf2(oacc[d][4 * g + 0] * inv, oacc[d][4 * g + 1] * inv),
                           pack_bf2(oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv)};
                *(u32x2*)(op + 32 * d + 8 * g) = o;
            }
    }
}

extern "C" int uv_flash_attn_bf16(const void* q, long ldq, const void* k, long ldk, const void* vt, long ldvt,
                                  void* out, long ldo, int Lq, int Lk, int H, int head_dim,
                                  float softmax_scale, void* stream) {
    UV_CHECK_ARG(q && k && vt && out, "uv_flash_attn_bf16: null pointer");
    UV_CHECK_ARG(head_dim == 128 || head_dim == 64, "uv_flash_attn_bf16: head_dim %d unsupported (64 or 128)", head_dim);
    UV_CHECK_ARG(Lq > 0 && Lk > 0 && H > 0, "uv_flash_attn_bf16: bad shape Lq=%d Lk=%d H=%d", Lq, Lk, H);
    UV_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0,
                 "uv_flash_attn_bf16: leading dimensions must be multiples of 8 elements");
    UV_CHECK_ARG(ldvt >= (long)((Lk + 63) / 64) * 64,
                 "uv_flash_attn_bf16: ldvt=%ld must cover Lk=%d rounded up to 64", ldvt, Lk);
    UV_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt | (uintptr_t)out) & 15) == 0,
                 "uv_flash_attn_bf16: pointers must be 16-byte aligned");
    AttnArgs a;
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.vt = (const bf16_t*)vt; a.out = (bf16_t*)out;
    a.ldq = ldq; a.ldk = ldk; a.ldvt = ldvt; a.ldo = ldo;
    a.Lq = Lq; a.Lk = Lk; a.H = H;
    a.q_blocks = (Lq + UV_ATT_QB - 1) / UV_ATT_QB;
    a.scale_log2 = softmax_scale * 1.4426950408889634f;
    if (head_dim == 128)
        hipLaunchKernelGGL(flash_attn_fwd_kernel<128>, dim3(a.q_blocks * H), dim3(UV_ATT_WAVES * 64), 0,
                           (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(flash_attn_fwd_kernel<64>, dim3(a.q_blocks * H), dim3(UV_ATT_WAVES * 64), 0,
                           (hipStream_t)stream, a);
    UV_CHECK_LAUNCH("uv_flash_attn_bf16");
    return 0;
}
