// Flash attention forward (non-causal, head_dim 128, bf16 in / fp32 softmax+accumulate / bf16 out)
// for gfx950. Serves both the DiT spatiotemporal self-attention (Lq = Lk = L) and the text
// cross-attention (Lk = 512).
//
// Replaces: flash_attention()  models/wan/utils/modules/attention.py:24-130  (FA2/FA3 varlen call,
//           q/k/v cast to bf16 :59-83, result cast back :130), called from
//           WanSelfAttention.forward model.py:145-150 and WanCrossAttention.forward model.py:175.
//
// Three kernels share the operand layouts, LDS images and arithmetic below:
//   flash_attn_fwd12_kernel  head_dim 128, bf16, Lk >= 2048 (self-attention): one 12-wave workgroup per CU, 384 queries per K/V^T stream
//   flash_attn_fwd3_kernel   head_dim 128, bf16, shorter Lk (cross-attention): three 4-wave workgroups per CU (48 KiB LDS, <= 168 registers)
//   flash_attn_fwd_kernel    head_dim 64 / fp16 operands / very large leading dimensions: two 4-wave workgroups per CU (the first design;
//                            also carries the QB = 2 one-wave-per-SIMD experiment and the stamped diagnostic build)
// Common structure (one wave = 32 queries of one head; staged KV tile = 64 keys):
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
//   * LDS-DMA double buffer: tile t+1 is streamed straight into the other LDS buffer (global_load_lds) while tile t
//     is computed; one vmcnt(0) + one barrier per tile.
//   * the softmax reference maximum moves only when a row maximum outgrows it by more than 2^UV_ATT_DEFER.
//   * independent samples are one launch: q/k/out rows and V^T COLUMNS stacked per sample.
// Template parameters: D head_dim (128 / 64), NW waves per workgroup, STAMP in-kernel cycle stamps (diagnostic entry
// uvdbg_flash_attn_stamps), QB = 2 the experimental one-wave-per-SIMD 64-queries-per-wave form (UV_ATTN_QB=2, see DESIGN
// section 9), SGB fragment reads scheduled 6 ahead of their MFMA.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

#define UV_ATT_QW 32     // queries per wave
#define UV_ATT_KV 64
#define UV_ATT_DEFER 8.0f   // log2 of the largest P allowed before the reference maximum is moved

struct AttnArgs {
    const bf16_t* q;   // [Lq, ldq]   head h at column h*128
    const bf16_t* k;   // [Lk, ldk]
    const bf16_t* vt;  // [H*128, ldvt]  (V transposed; ldvt >= roundup(Lk, 64), pad finite)
    bf16_t* out;       // [Lq, ldo]
    long ldq, ldk, ldvt, ldo;
    int Lq, Lk, H, q_blocks, batch;
    float scale_log2;  // softmax_scale * log2(e)
};

typedef __attribute__((address_space(3))) void lds_void_a;

__device__ __forceinline__ int perm23(int i) {  // swap bits 2 and 3
    return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1);
}

// D = head_dim (128 for TI2V-5B; 64 for the reference's CPU-runnable tiny config).
// NW = waves per workgroup: 8 (256 queries, 1 workgroup per CU) or 4 (128 queries, 2 workgroups per CU: the two waves
// that share a SIMD then belong to DIFFERENT workgroups, are not re-aligned by a common barrier every tile, and drift
// into complementary phases - one in its MFMA cluster while the other does softmax VALU work).
// QB = 32-query blocks per wave. QB = 2 with NW = 4 is the one-wave-per-SIMD form: the wave owns the SIMD's whole 512-entry
// register file, every K / V^T fragment read from LDS and every LDS-DMA piece serves 64 queries instead of 32, and the
// softmax VALU work of one block has the other block's MFMAs to hide behind inside the same instruction stream.
template <int D, int NW, bool STAMP = false, int QB = 1, bool SGB = false, bool F16 = false>
__global__ __launch_bounds__(NW * 64, QB == 2 ? 1 : 2) void flash_attn_fwd_kernel(AttnArgs p, unsigned long long* stamps = nullptr) {
    static_assert(!(F16 && QB == 2), "the fp16 operand form is built for the default (QB = 1) structure only");
    constexpr int NT = NW * 64;
    constexpr int KROW = 2 * D;                 // bytes per K row in LDS (256 or 128)
    constexpr int KCH = D / 8;                  // 16-B chunks per K row
    constexpr int NKK = D / 16;                 // MFMA k-steps over the head dim
    constexpr int ND = D / 32;                  // 32-row d tiles of O^T
    constexpr int K_BYTES = UV_ATT_KV * KROW;   // 16 KiB at D=128
    constexpr int V_BYTES = D * 128;            // 16 KiB at D=128
    constexpr int STAGE = K_BYTES + V_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[QB == 2 ? 2 * K_BYTES + 3 * V_BYTES : 2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    // (sample, head)-major block order: consecutive block ids walk the q-blocks of one head of one sample
    const int bh = blockIdx.x / p.q_blocks;
    const int qb = blockIdx.x - bh * p.q_blocks;
    const int head = bh % p.H;
    {   // independent samples are stacked along the token axis: rows of q/k/out, COLUMNS of V^T
        const long b = bh / p.H;
        p.q += b * p.Lq * p.ldq;
        p.k += b * p.Lk * p.ldk;
        p.vt += (long)b * p.Lk;
        p.out += b * p.Lq * p.ldo;
    }
    const int q0w = qb * (NW * QB * UV_ATT_QW) + wave * (QB * UV_ATT_QW);
    const long hcol = (long)head * D;

    // ---- Q fragments (B operand of S^T = K.Q^T): lane (r,h) holds Q[q0w+32b+r][16kk+8h .. +7]
    bf16x8 qf[QB][NKK];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        const int qrow = min(q0w + 32 * b + r, p.Lq - 1);
        const bf16_t* qp = p.q + (long)qrow * p.ldq + hcol + 8 * h;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) qf[b][kk] = *(const bf16x8*)(qp + 16 * kk);
    }

    // ---- staging: K and V^T tiles go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction,
    // no VGPR round trip and no ds_write: the VGPR->LDS store path measured 440-700 cycles per tile when all waves of
    // the CU wrote their 16-byte chunks at once). The LDS image is lane-linear per instruction, so the bank swizzle and
    // the key-row permutation live in the per-lane SOURCE address:
    //   K  : one instruction = KPI rows of KROW bytes; LDS row i <- key perm23(i), physical chunk p <- logical chunk
    //        p ^ key(i)
    //   V^T: one instruction = 8 rows of 128 B; physical chunk p <- logical chunk p ^ ((row>>1)&7)
    constexpr int KPI = 1024 / KROW;                  // K rows per wave-instruction (4 at D=128, 8 at D=64)
    constexpr int K_INSTR = UV_ATT_KV / KPI / NW;     // K wave-instructions per wave per tile
    constexpr int V_INSTR = D / 8 / NW;               // V^T wave-instructions per wave per tile
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bf16_t* ksrc[K_INSTR];
    int krow[K_INSTR];
    const bf16_t* vsrc[V_INSTR];
#pragma unroll
    for (int i = 0; i < K_INSTR; ++i) {
        const int lrow = (i * NW + wave_u) * KPI + lane / KCH;      // LDS row written by this lane
        const int pc = lane % KCH;                                  // physical chunk
        const int c = pc ^ (D == 128 ? (lrow & 15) : ((lrow >> 1) & 7));
        krow[i] = perm23(lrow);                                     // key row (in tile) stored there
        ksrc[i] = p.k + hcol + c * 8;
    }
#pragma unroll
    for (int i = 0; i < V_INSTR; ++i) {
        const int drow = (i * NW + wave_u) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((drow >> 1) & 7);
        vsrc[i] = p.vt + (hcol + drow) * p.ldvt + c * 8;
    }
    auto fetch = [&](int kv0, int buf) {
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < K_INSTR; ++i) {
            const int kr = min(kv0 + krow[i], p.Lk - 1);
            const bf16_t* src = ksrc[i] + (long)kr * p.ldk;
            __builtin_amdgcn_global_load_lds(src, (lds_void_a*)(base + (i * NW + wave_u) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < V_INSTR; ++i) {
            const bf16_t* src = vsrc[i] + kv0;
            __builtin_amdgcn_global_load_lds(src, (lds_void_a*)(base + K_BYTES + (i * NW + wave_u) * 1024), 16, 0, 0);
        }
    };

    // Running source pointers for tiles that lie completely inside [0, Lk): no row clamp, one 64-bit add per LDS-DMA
    // instruction (the clamped form above costs ~7 VALU instructions per piece, ~250 cycles per tile and wave).
    const bf16_t* kptr[K_INSTR];
    const bf16_t* vptr[V_INSTR];
#pragma unroll
    for (int i = 0; i < K_INSTR; ++i) kptr[i] = ksrc[i] + (long)(UV_ATT_KV + krow[i]) * p.ldk;   // tile 1
#pragma unroll
    for (int i = 0; i < V_INSTR; ++i) vptr[i] = vsrc[i] + UV_ATT_KV;
    const long kstep = (long)UV_ATT_KV * p.ldk;
    auto fetch_next_full = [&](int buf) {   // tiles 1, 2, ... in order
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < K_INSTR; ++i) {
            const bf16_t* src = kptr[i];
            __builtin_amdgcn_global_load_lds(src, (lds_void_a*)(base + (i * NW + wave_u) * 1024), 16, 0, 0);
            kptr[i] += kstep;
        }
#pragma unroll
        for (int i = 0; i < V_INSTR; ++i) {
            const bf16_t* src = vptr[i];
            __builtin_amdgcn_global_load_lds(src, (lds_void_a*)(base + K_BYTES + (i * NW + wave_u) * 1024), 16, 0, 0);
            vptr[i] += UV_ATT_KV;
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

    f32x16 oacc[QB][ND];
    float m_run[QB];  // running max of raw scores (both half-waves hold the same value)
    float l_run[QB];  // this half-wave's partial row sum
#pragma unroll
    for (int b = 0; b < QB; ++b) {
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[b][d][e] = 0.f;
        m_run[b] = -INFINITY;
        l_run[b] = 0.f;
    }

    const int nt = (p.Lk + UV_ATT_KV - 1) / UV_ATT_KV;
    const int nt_full = p.Lk / UV_ATT_KV;
    if constexpr (QB == 2) {
        char* kb = smem;
        char* vb = smem + 2 * K_BYTES;
#pragma unroll
        for (int i = 0; i < K_INSTR; ++i) {
            const bf16_t* src = ksrc[i] + (long)min(krow[i], p.Lk - 1) * p.ldk;
            __builtin_amdgcn_global_load_lds(src, (lds_void_a*)(kb + (i * NW + wave_u) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < V_INSTR; ++i) {
            const bf16_t* src = vsrc[i];
            __builtin_amdgcn_global_load_lds(src, (lds_void_a*)(vb + (i * NW + wave_u) * 1024), 16, 0, 0);
        }
    } else {
        fetch(0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // Pin "every prologue load has landed" BEFORE the loop: vmcnt retires in order, so if the compiler has to assume
    // the Q fragment loads may still be in flight at the loop header it guards their first use inside the loop with
    // vmcnt(1)/vmcnt(0) - which in steady state waits for the K/V prefetch issued a few instructions earlier and
    // exposes a full L2/HBM latency in every tile (seen in the ISA; ~1000 cycles of a 4500-cycle tile).
#pragma unroll
    for (int b = 0; b < QB; ++b)
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) asm volatile("" : "+v"(qf[b][kk]));
    __syncthreads();

    // One KV tile. MASKED is a compile-time flag so that the main loop carries no masking code at all (only the ragged
    // last tile is instantiated with it).
    // diagnostic build only (STAMP): per-segment cycle sums [qk, softmax, pv, commit, barrier] per wave
    unsigned long long seg[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tprev = 0;
    auto stamp = [&](int which) {
        if (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long tnow;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tnow)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            if (which >= 0) seg[which] += tnow - tprev;
            tprev = tnow;
        }
    };
    // ---- one-wave-per-SIMD form (QB == 2): the two 32-query blocks A and B of the wave run half a tile apart so that
    // each block's softmax VALU work has the other block's MFMAs to hide behind, inside ONE instruction stream:
    //     QK_A | QK_B + softmax_A | PV_A + softmax_B | PV_B
    // The rescale decision (a rare branch, see below) is taken before the mixed regions so that each of them is one
    // basic block the scheduler can interleave.
    // Row sums by MFMA (QB == 2 path): one more accumulator tile per block whose A operand is all ones, so
    // lacc[b][*] = sum over keys of the bf16 P values (every register of a lane holds the same sum). It replaces 32
    // v_add_f32 per block and tile - the VALU pipe, not the matrix pipe, is the busy one in the mixed regions.
    f32x16 lacc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) lacc[b][e] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
    // Software pipeline of the QB == 2 form. Block B runs one tile behind block A, so that every group of MFMAs has
    // softmax VALU work of the OTHER block to hide (one wave per SIMD: nothing else would):
    //     step 1   MFMA  S_A(t)  = K(t) Q_A             VALU  P_B(t-1) = exp2(S_B(t-1) ...)
    //     step 2   MFMA  O_B    += V(t-1) P_B(t-1)  (1st half)      row max of S_A(t)      -> reference-maximum decision A
    //     step 3   MFMA  O_B (2nd half), S_B(t) = K(t) Q_B          P_A(t) = exp2(S_A(t) ...)
    //     step 4   MFMA  O_A    += V(t) P_A(t)      (1st half)      row max of S_B(t)      -> decision B
    //     step 5   MFMA  O_A (2nd half)
    // V tiles therefore live for two iterations (3-deep V ring, 2-deep K ring). Tile -1 is a dummy: S_B(-1) = -inf
    // gives P = 0, multiplied into V(0)'s (finite) fragments.
    f32x16 sB[2];      // S_B of the previous tile
    float mnegB = 0.f;
    bf16x8 pB[2][2];
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int e = 0; e < 16; ++e) sB[T][e] = -INFINITY;
    const char* const vring = smem + 2 * K_BYTES;
    int vcur = 0, vprev = 0;   // V ring slots of tile t and t-1
    auto rowmax = [&](f32x16 (&s_)[2], int kv0, bool masked) -> float {
        if (masked) {
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = 32 * T + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (kv0 + perm23(i) >= p.Lk) s_[T][e] = -INFINITY;
                }
        }
        float mt = s_[0][0];
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int e = 0; e < 16; ++e) mt = fmaxf(mt, s_[T][e]);
        const unsigned u = __builtin_bit_cast(unsigned, mt);
        const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        return fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
    };
    // The O and l accumulators live in the accumulator half of the register file, which no VALU instruction can touch,
    // so a rescale is three instructions per register; with the deferred reference maximum (UV_ATT_DEFER) it practically
    // happens on the first tile only. Returns -m_ref * scale.
    auto decide = [&](int b, float mt) -> float {
        const float grow = (mt - m_run[b]) * p.scale_log2;      // +inf on the first tile (m_run = -inf)
        if (__any(grow > UV_ATT_DEFER)) {
            const float m_new = fmaxf(m_run[b], mt);
            const float alpha = __builtin_amdgcn_exp2f((m_run[b] - m_new) * p.scale_log2);
            float chain = alpha;   // orders the statements below (asm statements are opaque to the scheduler)
#pragma unroll
            for (int d = 0; d <= ND; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float tmp;
                    // leading nops: the last MFMA that wrote this accumulator must have retired (16-pass XDL ->
                    // accvgpr read); trailing nop: accvgpr write -> MFMA source
                    if (d < ND)
                        asm volatile("s_nop 15\n\ts_nop 3\n\tv_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\t"
                                     "v_accvgpr_write_b32 %0, %1\n\ts_nop 1"
                                     : "+a"(oacc[b][d][e]), "=&v"(tmp), "+v"(chain));
                    else
                        asm volatile("s_nop 15\n\ts_nop 3\n\tv_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\t"
                                     "v_accvgpr_write_b32 %0, %1\n\ts_nop 1"
                                     : "+a"(lacc[b][e]), "=&v"(tmp), "+v"(chain));
                }
            asm volatile("" : "+v"(chain));
            m_run[b] = m_new;
        }
        return -m_run[b] * p.scale_log2;
    };
    auto expP = [&](f32x16 (&s_)[2], float mneg, bf16x8 (&p_)[2][2]) {
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    p_[T][s2][j] = (__bf16)__builtin_amdgcn_exp2f(__builtin_fmaf(s_[T][8 * s2 + j], p.scale_log2, mneg));
    };
    // keeps P from being sunk below the next rescale branch by the IR optimiser (away from the MFMAs it should overlap)
    auto pinP = [&](bf16x8 (&p_)[2][2]) {
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) asm volatile("" : "+v"(p_[T][s2]));
    };
    auto qk2 = [&](const char* kbase, int b, f32x16 (&s_)[2]) {
#pragma unroll
        for (int T = 0; T < 2; ++T) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s_[T][e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const bf16x8 kf = *(const bf16x8*)(kbase + T * 32 * KROW + k_row_off + (((2 * kk + h) ^ k_key) << 4));
                s_[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[b][kk], s_[T], 0, 0, 0);
            }
        }
    };
    // half of O^T += V^T P^T: d tiles {2*half, 2*half+1}; the row-sum tile (A operand = ones) rides with the second half
    auto pv2 = [&](const char* vbase, int b, bf16x8 (&p_)[2][2], int half) {
#pragma unroll
        for (int dd = 0; dd < ND / 2; ++dd) {
            const int d = half * (ND / 2) + dd;
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 vf = *(const bf16x8*)(vbase + d * 32 * 128 + (v_row_off - K_BYTES) + (((4 * T + 2 * s2 + h) ^ v_key) << 4));
                    oacc[b][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, p_[T][s2], oacc[b][d], 0, 0, 0);
                }
        }
        if (half == 1) {
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                    lacc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, p_[T][s2], lacc[b], 0, 0, 0);
        }
    };
    auto fetch2 = [&](int kv0, int kslot, int vslot) {
        char* kb = smem + kslot * K_BYTES;
        char* vb = smem + 2 * K_BYTES + vslot * V_BYTES;
#pragma unroll
        for (int i = 0; i < K_INSTR; ++i) {
            const int kr = min(kv0 + krow[i], p.Lk - 1);
            const bf16_t* src = ksrc[i] + (long)kr * p.ldk;
            __builtin_amdgcn_global_load_lds(src, (lds_void_a*)(kb + (i * NW + wave_u) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < V_INSTR; ++i) {
            const bf16_t* src = vsrc[i] + kv0;
            __builtin_amdgcn_global_load_lds(src, (lds_void_a*)(vb + (i * NW + wave_u) * 1024), 16, 0, 0);
        }
    };
    auto tile2 = [&](int t, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const int kv0 = t * UV_ATT_KV;
        const char* kbase = smem + (t & 1) * K_BYTES;
        const char* vb_cur = vring + vcur * V_BYTES;
        const char* vb_prev = vring + vprev * V_BYTES;
        const int vnext = vcur == 2 ? 0 : vcur + 1;
        if (t + 1 < nt) fetch2(kv0 + UV_ATT_KV, (t + 1) & 1, vnext);
        // Q is only ever an MFMA source: keep it in the accumulator half of the register file (legal for MFMA A/B operands)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) asm volatile("" : "+a"(qf[b][kk]));
        f32x16 sA[2];
        bf16x8 pA[2][2];
        stamp(-1);
        // step 1
        qk2(kbase, 0, sA);
        expP(sB, mnegB, pB);
        pinP(pB);
        __builtin_amdgcn_sched_barrier(0);
        if (STAMP) { asm volatile("" ::"v"(sA[1][15])); }
        stamp(0);
        // step 2
        pv2(vb_prev, 1, pB, 0);
        const float mtA = rowmax(sA, kv0, MASKED);
        __builtin_amdgcn_sched_barrier(0);
        const float mnegA = decide(0, mtA);
        __builtin_amdgcn_sched_barrier(0);
        if (STAMP) { asm volatile("" ::"v"(mnegA)); }
        stamp(1);
        // step 3
        pv2(vb_prev, 1, pB, 1);
        qk2(kbase, 1, sB);
        expP(sA, mnegA, pA);
        pinP(pA);
        __builtin_amdgcn_sched_barrier(0);
        if (STAMP) { asm volatile("" ::"v"(sB[1][15])); }
        stamp(2);
        // step 4
        pv2(vb_cur, 0, pA, 0);
        const float mtB = rowmax(sB, kv0, MASKED);
        __builtin_amdgcn_sched_barrier(0);
        mnegB = decide(1, mtB);
        __builtin_amdgcn_sched_barrier(0);
        if (STAMP) { asm volatile("" ::"v"(mnegB)); }
        stamp(3);
        // step 5
        pv2(vb_cur, 0, pA, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (STAMP) { asm volatile("" ::"a"(oacc[0][ND - 1][15])); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stamp(4);
        vprev = vcur;
        vcur = vnext;
    };
    // drains block B's last tile after the loop
    auto tail2 = [&]() {
        const char* vb_prev = vring + vprev * V_BYTES;
        expP(sB, mnegB, pB);
        pv2(vb_prev, 1, pB, 0);
        pv2(vb_prev, 1, pB, 1);
    };

    // FETCH: 1 = the next tile is a full one (running pointers, no branch: the LDS-DMA instructions then sit in the same
    // scheduling region as the QK MFMAs and are dealt out between them), 0 = decide at run time (last full tile / ragged tile)
    auto tile = [&](int t, auto masked_tag, auto fetch_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        constexpr bool FETCH_FAST = decltype(fetch_tag)::value;
        const int kv0 = t * UV_ATT_KV;
        stamp(-1);
        const char* base = smem + (t & 1) * STAGE;
        // other buffer: last read in tile t-1, fenced by its barrier
        if constexpr (FETCH_FAST) {
            fetch_next_full((t + 1) & 1);
        } else {
            if (t + 1 < nt_full) fetch_next_full((t + 1) & 1);
            else if (t + 1 < nt) fetch(kv0 + UV_ATT_KV, (t + 1) & 1);   // the ragged last tile: clamped rows
        }
        stamp(5);

        // ---- S^T = K . Q^T  (two 32-key tiles) for every 32-query block of the wave
        f32x16 sacc[QB][2];
#pragma unroll
        for (int b = 0; b < QB; ++b)
#pragma unroll
            for (int T = 0; T < 2; ++T) {
#pragma unroll
                for (int e = 0; e < 16; ++e) sacc[b][T][e] = 0.f;
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    const bf16x8 kf =
                        *(const bf16x8*)(base + T * 32 * KROW + k_row_off + (((2 * kk + h) ^ k_key) << 4));
                    sacc[b][T] = mfma_32x32x16<F16>(kf, qf[b][kk], sacc[b][T]);
                }
            }

        if constexpr (SGB && D == 128 && QB == 1) {   // fragment reads 6 ahead of the MFMA that consumes them
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
            for (int i_ = 0; i_ < 10; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (STAMP) { asm volatile("" ::"v"(sacc[0][0][0]), "v"(sacc[QB - 1][1][15])); }
        stamp(0);
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            // ---- mask the ragged last tile with the TRUE key index of each accumulator row
            if (MASKED) {
#pragma unroll
                for (int T = 0; T < 2; ++T)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int i = 32 * T + (e & 3) + 8 * (e >> 2) + 4 * h;
                        if (kv0 + perm23(i) >= p.Lk) sacc[b][T][e] = -INFINITY;
                    }
            }

            // ---- online softmax (query on the lane)
            float mt = sacc[b][0][0];
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int e = 0; e < 16; ++e) mt = fmaxf(mt, sacc[b][T][e]);
            {   // the other half-wave's maximum: v_permlane32_swap (one VALU op) instead of a ds_bpermute round trip
                const unsigned u = __builtin_bit_cast(unsigned, mt);
                const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                mt = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
            }
            float mneg;
            if constexpr (QB == 1) {
                // The reference maximum m_run moves only when some row's maximum outgrows it by more than 2^UV_ATT_DEFER in
                // the exponent domain (then P <= 2^UV_ATT_DEFER instead of <= 1 until the next move; P, l and O share one
                // scale and bf16 / f32 relative precision is scale-invariant). With the exact running maximum about half
                // of all tiles of a long sequence still see a new maximum in SOME row of the wave and pay the O rescale.
                const float grow = (mt - m_run[b]) * p.scale_log2;      // +inf on the first tile (m_run = -inf)
                if (__any(grow > UV_ATT_DEFER)) {
                    const float m_new = fmaxf(m_run[b], mt);
                    const float alpha = __builtin_amdgcn_exp2f((m_run[b] - m_new) * p.scale_log2);
                    l_run[b] *= alpha;
#pragma unroll
                    for (int d = 0; d < ND; ++d)
#pragma unroll
                        for (int e = 0; e < 16; ++e) oacc[b][d][e] *= alpha;
                    m_run[b] = m_new;
                }
                mneg = -m_run[b] * p.scale_log2;
            } else {
                // One wave per SIMD: the O accumulators live in the accumulator half of the register file, which no VALU
                // instruction can touch, so a rescale is three instructions per register. It is therefore deferred: the
                // reference maximum moves only when a row's maximum outgrows it by more than 2^8 in the exponent domain
                // (P <= 256 instead of <= 1; P, l and O all stay at the same scale, bf16/f32 relative precision is
                // scale-invariant), which after the first tile practically never happens.
                const float grow = (mt - m_run[b]) * p.scale_log2;      // +inf on the first tile (m_run = -inf)
                if (__any(grow > UV_ATT_DEFER)) {
                    const float m_new = fmaxf(m_run[b], mt);
                    const float alpha = __builtin_amdgcn_exp2f((m_run[b] - m_new) * p.scale_log2);
                    l_run[b] *= alpha;
                    float chain = alpha;   // orders the statements below (asm statements are opaque to the scheduler)
#pragma unroll
                    for (int d = 0; d < ND; ++d)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            float tmp;
                            // leading nops: the last MFMA that wrote this accumulator must have retired (16-pass XDL ->
                            // accvgpr read); trailing nop: accvgpr write -> MFMA source
                            asm volatile("s_nop 15\n\ts_nop 3\n\tv_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\t"
                                         "v_accvgpr_write_b32 %0, %1\n\ts_nop 1"
                                         : "+a"(oacc[b][d][e]), "=&v"(tmp), "+v"(chain));
                        }
                    asm volatile("" : "+v"(chain));
                    m_run[b] = m_new;
                }
                mneg = -m_run[b] * p.scale_log2;
            }
            float psum = 0.f;
            bf16x8 pf[2][2];
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[b][T][8 * s + j], p.scale_log2, mneg));
                        psum += pv;
                        if constexpr (F16) {   // fp16 bits carried in the bf16x8 container (P <= 2^UV_ATT_DEFER fits fp16 easily)
                            bf16_t bits = out16<true>(pv);
                            pf[T][s][j] = __builtin_bit_cast(__bf16, bits);
                        } else {
                            pf[T][s][j] = (__bf16)pv;
                        }
                    }
            l_run[b] += psum;
            if (STAMP) { asm volatile("" ::"v"(pf[1][1])); }
            if (b == QB - 1) stamp(1);

            // ---- O^T += V^T . P^T
            if constexpr (SGB && D == 128 && QB == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int T = 0; T < 2; ++T)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const bf16x8 vf =
                            *(const bf16x8*)(base + d * 32 * 128 + v_row_off + (((4 * T + 2 * s + h) ^ v_key) << 4));
                        oacc[b][d] = mfma_32x32x16<F16>(vf, pf[T][s], oacc[b][d]);
                    }
        }

        if constexpr (SGB && D == 128 && QB == 1) {
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 1);
#pragma unroll
            for (int i_ = 0; i_ < 10; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (STAMP) { asm volatile("" ::"v"(oacc[QB - 1][ND - 1][15])); }
        stamp(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's share of tile t+1 has landed in LDS
        stamp(3);
        __syncthreads();
        stamp(4);
    };

    if constexpr (QB == 2) {
        for (int t = 0; t < nt_full; ++t) tile2(t, std::false_type{});
        if (nt_full < nt) tile2(nt_full, std::true_type{});
        tail2();
    } else {
        for (int t = 0; t + 1 < nt_full; ++t) tile(t, std::false_type{}, std::true_type{});
        if (nt_full > 0) tile(nt_full - 1, std::false_type{}, std::false_type{});
        if (nt_full < nt) tile(nt_full, std::true_type{}, std::false_type{});
    }

    if (STAMP && stamps && lane == 0) {
        unsigned long long* dst = stamps + ((long)blockIdx.x * NW + wave) * 6;
        for (int i = 0; i < 6; ++i) dst[i] = seg[i];
    }
    // ---- finish: combine the two half-wave sums, normalise, store bf16 rows
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        float l_half = l_run[b];
        if constexpr (QB == 2) l_half = lacc[b][0];   // MFMA row sums: already complete over all keys (no half-wave split)
        const float l_tot = QB == 2 ? l_half : l_half + __shfl_xor(l_half, 32, 64);
        const float inv = 1.0f / l_tot;
        const int q = q0w + 32 * b + r;
        if (q < p.Lq) {
            bf16_t* op = p.out + (long)q * p.ldo + hcol + 4 * h;
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 o = {pack16_2<F16>(oacc[b][d][4 * g + 0] * inv, oacc[b][d][4 * g + 1] * inv),
                               pack16_2<F16>(oacc[b][d][4 * g + 2] * inv, oacc[b][d][4 * g + 3] * inv)};
                    *(u32x2*)(op + 32 * d + 8 * g) = o;
                }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// Three-waves-per-SIMD form (head_dim 128, bf16): same arithmetic, tile order and rounding points as the default kernel
// above - the results are bit-identical - but sized so that THREE 4-wave workgroups fit a CU:
//   * LDS 48 KiB per workgroup: K double-buffered, V^T single-buffered. V^T(t) is requested at the top of tile t (its buffer
//     is free once every wave has left tile t-1) and is needed only after QK(t) + softmax(t); K(t+1) is requested right
//     behind it. Two barriers per tile (V^T landed / tile done); with three independent workgroups per CU a parked wave
//     costs nothing as long as one of the other two has work for the SIMD.
//   * <= 168 registers per wave: the LDS-DMA sources are ONE uniform (SGPR) base pointer per operand, advanced once per
//     tile, plus a constant 32-bit lane offset per piece (the default kernel carries 8 running 64-bit pointers and the
//     clamped-row state of the ragged tile); the ragged tile recomputes its clamped addresses from the lane id.
// A lone wave needs ~3 440 cycles per tile of which 1 024 are MFMA issue; two waves per SIMD overlap almost perfectly
// (3 500 cycles per PAIR of tiles), i.e. the chain is latency- not throughput-bound and a third wave has room.
// ------------------------------------------------------------------------------------------------------------------------
// One LDS-DMA piece with a UNIFORM base pointer (SGPR pair) and a 32-bit lane offset: "global_load_lds_dwordx4 voff, s[base]".
// hipcc selects only the 64-bit-VGPR-address form for the builtin (one v_lshl_add_u64 and a live register pair per piece).
// M0 = wave-uniform LDS byte address of the piece; saved and restored around the statement (the compiler owns M0).
__device__ __forceinline__ void glds16_sbase(const char* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

// AHEAD = fragment reads in flight ahead of their MFMA (2 .. 5 measured: all within 0.5 %); XCD = XCD-aware block order (A/B knob).
template <int AHEAD = 3, bool XCD = true>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void flash_attn_fwd3_kernel(AttnArgs p) {
    constexpr int D = 128, NW = 4, KROW = 256, NKK = 8, ND = 4;
    constexpr int K_BYTES = UV_ATT_KV * KROW, V_BYTES = D * 128;
    __shared__ __attribute__((aligned(16))) char smem[2 * K_BYTES + V_BYTES];
    constexpr int V_OFF = 2 * K_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);

    // Workgroups are dealt round-robin over the 8 XCDs (block b -> XCD b % 8), each with its own L2. In plain (sample, head)-
    // major order the ~96 workgroups resident on an XCD span 8-9 heads, so every L2 streams the K / V^T of 8-9 heads at once and
    // each head's K / V^T is pulled into all 8 L2s. Remapped, XCD x works through the contiguous range [x NB/8, (x+1) NB/8) of
    // (sample, head, q-block) ids: one or two heads at a time per L2, each head in one L2 only.
    int vb = blockIdx.x;
    if constexpr (XCD) {
        const int nb = gridDim.x, per = nb >> 3, rem = nb & 7;
        const int x = vb & 7, j = vb >> 3;
        // XCD x owns per + (x < rem) ids; its range starts after the ranges of XCDs 0 .. x-1
        vb = x * per + min(x, rem) + j;
    }
    const int bh = vb / p.q_blocks;
    const int qb = vb - bh * p.q_blocks;
    const int head = bh % p.H;
    {
        const long b = bh / p.H;
        p.q += b * p.Lq * p.ldq;
        p.k += b * p.Lk * p.ldk;
        p.vt += (long)b * p.Lk;
        p.out += b * p.Lq * p.ldo;
    }
    const int q0w = qb * (NW * UV_ATT_QW) + wave_u * UV_ATT_QW;
    const long hcol = (long)head * D;

    bf16x8 qf[NKK];
    {
        const int qrow = min(q0w + r, p.Lq - 1);
        const bf16_t* qp = p.q + (long)qrow * p.ldq + hcol + 8 * h;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) qf[kk] = *(const bf16x8*)(qp + 16 * kk);
    }

    // LDS-DMA pieces of this wave: 4 of K (4 rows of 256 B each), 4 of V^T (8 rows of 128 B each); see the default kernel for
    // the row permutation and the swizzles. Piece i of a wave covers LDS rows 16 i further on (K) / 32 i (V^T); neither the
    // swizzle key nor the row permutation sees those bits, so ONE lane offset per operand serves all four pieces and the
    // piece stride goes into the uniform base. Offsets in BYTES relative to the tile's first key row / key column.
    unsigned koff, voff;
    {
        const int lrow = wave_u * 4 + (lane >> 4);
        const int c = (lane & 15) ^ (lrow & 15);
        koff = (unsigned)(perm23(lrow) * (int)p.ldk + c * 8) * 2u;
        const int drow = wave_u * 8 + (lane >> 3);
        const int cv = (lane & 7) ^ ((drow >> 1) & 7);
        voff = (unsigned)(drow * (int)p.ldvt + cv * 8) * 2u;
    }
    const char* kbase = (const char*)(p.k + hcol);                       // tile 0; += kstep per tile
    const char* vbase = (const char*)(p.vt + hcol * p.ldvt);             // tile 0; += 128 B per tile
    const long kstep = (long)UV_ATT_KV * p.ldk * 2;
    const long kpiece = 16 * p.ldk * 2, vpiece = 32 * p.ldvt * 2;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_a*)smem + wave_u * 1024;   // this wave's first piece in buffer 0
    auto fetch_k_full = [&](const char* base_t, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16_sbase(base_t + i * kpiece, koff, lds0 + buf * K_BYTES + i * NW * 1024);
    };
    auto fetch_k_clamped = [&](int kv0, int buf) {                       // the ragged last tile (and a lone short tile 0)
        char* dst = smem + buf * K_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int lrow = (i * NW + wave_u) * 4 + (lane >> 4);
            const int c = (lane & 15) ^ (lrow & 15);
            const int kr = min(kv0 + perm23(lrow), p.Lk - 1);
            const bf16_t* src = p.k + hcol + (long)kr * p.ldk + c * 8;
            __builtin_amdgcn_global_load_lds(src, (lds_void_a*)(dst + (i * NW + wave_u) * 1024), 16, 0, 0);
        }
    };
    auto fetch_v = [&](const char* base_t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16_sbase(base_t + i * vpiece, voff, lds0 + V_OFF + i * NW * 1024);
    };

    // fragment read addresses (LDS byte offsets from smem; buffer / key-half / d-tile offsets are instruction immediates)
    typedef const __attribute__((address_space(3))) bf16x8* lds_frag_p;
    const unsigned smem_a = (unsigned)(uintptr_t)(lds_void_a*)smem;
    unsigned kaddr[NKK];
    unsigned vaddr[2][2];
    {
        const int k_key = r & 15, v_key = (r >> 1) & 7;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) kaddr[kk] = smem_a + r * KROW + (((2 * kk + h) ^ k_key) << 4);
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) vaddr[T][s2] = smem_a + V_OFF + r * 128 + (((4 * T + 2 * s2 + h) ^ v_key) << 4);
    }

    f32x16 oacc[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int nt = (p.Lk + UV_ATT_KV - 1) / UV_ATT_KV;
    const int nt_full = p.Lk / UV_ATT_KV;
    if (nt_full > 0) fetch_k_full(kbase, 0);
    else fetch_k_clamped(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) asm volatile("" : "+v"(qf[kk]), "+v"(kaddr[kk]));
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(vaddr[i >> 1][i & 1]));
    asm volatile("" : "+v"(koff), "+v"(voff));
    __syncthreads();

    // One staged tile = 64 keys = two 32-key halves that are computed one after the other (QK, softmax, PV per half): the S
    // accumulator of only one half is live beside O and Q, which is what leaves registers for fragment reads ahead of their
    // MFMAs at 168 registers per wave. The reference maximum is therefore reconsidered per half.
    // NEXT: 1 = tile t+1 is a full one, 0 = decide at run time (ragged or none); PAR = t & 1 (compile-time: the K buffer
    // offsets become instruction immediates)
    auto tile = [&](int t, auto masked_tag, auto next_tag, auto par_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        constexpr bool NEXT_FULL = decltype(next_tag)::value;
        constexpr int PAR = decltype(par_tag)::value;
        const int kv0 = t * UV_ATT_KV;
        fetch_v(vbase);                                  // V^T(t): the buffer was released by the barrier that ended tile t-1
        vbase += 2 * UV_ATT_KV;
        kbase += kstep;
        bool k_pending = true;
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            f32x16 sacc;
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const bf16x8 kf = *(lds_frag_p)(kaddr[kk] + PAR * K_BYTES + T * 32 * KROW);
                sacc = mfma_32x32x16<false>(kf, qf[kk], sacc);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);      // fragment reads AHEAD ahead of their MFMAs
#pragma unroll
            for (int i_ = 0; i_ < 8 - AHEAD; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, AHEAD, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (T == 0) {
                // K(t+1) into the other K buffer (every wave left tile t-1, its last reader, before the barrier that opened tile t)
                if constexpr (NEXT_FULL) {
                    fetch_k_full(kbase, PAR ^ 1);
                } else {
                    if (t + 1 < nt_full) fetch_k_full(kbase, PAR ^ 1);
                    else if (t + 1 < nt) fetch_k_clamped(kv0 + UV_ATT_KV, PAR ^ 1);
                    else k_pending = false;
                }
            }
            if (MASKED) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = 32 * T + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (kv0 + perm23(i) >= p.Lk) sacc[e] = -INFINITY;
                }
            }
            float mt = sacc[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mt = fmaxf(mt, sacc[e]);
            {
                const unsigned u = __builtin_bit_cast(unsigned, mt);
                const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                mt = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
            }
            const float grow = (mt - m_run) * p.scale_log2;
            if (__any(grow > UV_ATT_DEFER)) {
                const float m_new = fmaxf(m_run, mt);
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.scale_log2);
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < ND; ++d)
#pragma unroll
                    for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
                m_run = m_new;
            }
            const float mneg = -m_run * p.scale_log2;
            float psum = 0.f;
            bf16x8 pf[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[8 * s2 + j], p.scale_log2, mneg));
                    psum += pv;
                    pf[s2][j] = (__bf16)pv;
                }
            l_run += psum;
            if (T == 0) {
                // V^T(t) of every wave has landed: the K(t+1) pieces were issued after it and may stay in flight
                if (k_pending) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 vf = *(lds_frag_p)(vaddr[T][s2] + d * 32 * 128);
                    oacc[d] = mfma_32x32x16<false>(vf, pf[s2], oacc[d]);
                }
            __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 1);
#pragma unroll
            for (int i_ = 0; i_ < 8 - AHEAD; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, AHEAD, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // this wave's share of K(t+1); own LDS reads retired
        __builtin_amdgcn_s_barrier();
    };

    using F = std::false_type;
    using T_ = std::true_type;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    int t = 0;
    for (; t + 2 < nt_full; t += 2) {
        tile(t, F{}, T_{}, P0{});
        tile(t + 1, F{}, T_{}, P1{});
    }
    // 0, 1 or 2 full tiles left (t is even)
    if (t + 1 < nt_full) {
        tile(t, F{}, T_{}, P0{});
        tile(t + 1, F{}, F{}, P1{});
    } else if (t < nt_full) {
        tile(t, F{}, F{}, P0{});
    }
    if (nt_full < nt) {
        if (nt_full & 1) tile(nt_full, T_{}, F{}, P1{});
        else tile(nt_full, T_{}, F{}, P0{});
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q = q0w + r;
    if (q < p.Lq) {
        bf16_t* op = p.out + (long)q * p.ldo + hcol + 4 * h;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 o = {pack16_2<false>(oacc[d][4 * g + 0] * inv, oacc[d][4 * g + 1] * inv),
                           pack16_2<false>(oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv)};
                *(u32x2*)(op + 32 * d + 8 * g) = o;
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Long key sequences (default for Lk >= 2048; UV_ATTN_W3=1 keeps the 4-wave form): ONE 12-wave workgroup per CU (384 queries, three
// waves per SIMD) sharing each K / V^T tile: a third of the L2 -> LDS traffic and 2-3 instead of 8 LDS-DMA pieces per wave and tile.
// Bit-identical to flash_attn_fwd3_kernel (same per-wave arithmetic). Removing its barrier (timing probe) gains 0.6 %. Staging as in the two-waves kernel (K and
// V^T double-buffered, tile t+1 requested at the top of tile t, ONE vmcnt(0) + barrier per tile), compute body and register
// diet of flash_attn_fwd3_kernel (32-key halves, SGPR-base DMA, immediates for the buffer parity).
// ------------------------------------------------------------------------------------------------------------------------
template <int AHEAD = 3, bool XCD = true>
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3))) void flash_attn_fwd12_kernel(AttnArgs p) {
    constexpr int D = 128, NW = 12, KROW = 256, NKK = 8, ND = 4;
    constexpr int K_BYTES = UV_ATT_KV * KROW, V_BYTES = D * 128;
    __shared__ __attribute__((aligned(16))) char smem[2 * K_BYTES + 2 * V_BYTES];
    constexpr int V_OFF = 2 * K_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);

    int vb = blockIdx.x;
    if constexpr (XCD) {
        const int nb = gridDim.x, per = nb >> 3, rem = nb & 7;
        const int x = vb & 7, j = vb >> 3;
        vb = x * per + min(x, rem) + j;
    }
    const int bh = vb / p.q_blocks;
    const int qb = vb - bh * p.q_blocks;
    const int head = bh % p.H;
    {
        const long b = bh / p.H;
        p.q += b * p.Lq * p.ldq;
        p.k += b * p.Lk * p.ldk;
        p.vt += (long)b * p.Lk;
        p.out += b * p.Lq * p.ldo;
    }
    const int q0w = qb * (NW * UV_ATT_QW) + wave_u * UV_ATT_QW;
    const long hcol = (long)head * D;

    bf16x8 qf[NKK];
    {
        const int qrow = min(q0w + r, p.Lq - 1);
        const bf16_t* qp = p.q + (long)qrow * p.ldq + hcol + 8 * h;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) qf[kk] = *(const bf16x8*)(qp + 16 * kk);
    }

    // The 32 pieces of a tile (K pieces 0..15 = 4 LDS rows each, V^T pieces 0..15 = 8 rows each) are dealt round-robin: wave w
    // issues K piece w, K piece w + 12 (w < 4), V^T piece w - 4 (w >= 4) and V^T piece w + 8 (w < 8). Pieces 12 apart (K) / of equal
    // parity (V^T) share the swizzle key and the row permutation bits, so one lane offset per operand serves both.
    const int s4 = lane >> 4, s8 = lane >> 3;
    unsigned koff, voff;
    {
        const int p3 = wave_u & 3;
        const int c = (lane & 15) ^ ((4 * p3 + s4) & 15);
        const int swap2 = ((p3 & 1) << 1) | (p3 >> 1);
        koff = (unsigned)((s4 + 4 * swap2) * (int)p.ldk + c * 8) * 2u;
        const int cv = (lane & 7) ^ ((4 * (wave_u & 1) + (s8 >> 1)) & 7);
        voff = (unsigned)(s8 * (int)p.ldvt + cv * 8) * 2u;
    }
    const char* kbase = (const char*)(p.k + hcol);
    const char* vbase = (const char*)(p.vt + hcol * p.ldvt);
    const long kstep = (long)UV_ATT_KV * p.ldk * 2;
    const long k16 = 16 * p.ldk * 2, v8 = 8 * p.ldvt * 2;
    const unsigned smem_a = (unsigned)(uintptr_t)(lds_void_a*)smem;
    const int kp0 = wave_u, kp1 = wave_u + 12;                // K pieces (kp1 only for wave < 4)
    const int vq0 = wave_u - 4, vq1 = wave_u + 8;             // V^T pieces (vq0 for wave >= 4, vq1 for wave < 8)
    auto fetch_full = [&](const char* kb_t, const char* vb_t, int buf) {
        glds16_sbase(kb_t + (kp0 >> 2) * k16, koff, smem_a + buf * K_BYTES + kp0 * 1024);
        if (wave_u < 4) glds16_sbase(kb_t + (kp1 >> 2) * k16, koff, smem_a + buf * K_BYTES + kp1 * 1024);
        if (wave_u >= 4) glds16_sbase(vb_t + vq0 * v8, voff, smem_a + V_OFF + buf * V_BYTES + vq0 * 1024);
        if (wave_u < 8) glds16_sbase(vb_t + vq1 * v8, voff, smem_a + V_OFF + buf * V_BYTES + vq1 * 1024);
    };
    auto fetch_clamped = [&](int kv0, const char* vb_t, int buf) {      // ragged last tile / lone short tile: clamped K rows
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const int pc = w ? kp1 : kp0;
            if (pc < 16) {
                const int lrow = 4 * pc + s4;
                const int c = (lane & 15) ^ (lrow & 15);
                const int kr = min(kv0 + perm23(lrow), p.Lk - 1);
                const bf16_t* src = p.k + hcol + (long)kr * p.ldk + c * 8;
                __builtin_amdgcn_global_load_lds(src, (lds_void_a*)(smem + buf * K_BYTES + pc * 1024), 16, 0, 0);
            }
        }
        if (wave_u >= 4) glds16_sbase(vb_t + vq0 * v8, voff, smem_a + V_OFF + buf * V_BYTES + vq0 * 1024);
        if (wave_u < 8) glds16_sbase(vb_t + vq1 * v8, voff, smem_a + V_OFF + buf * V_BYTES + vq1 * 1024);
    };

    typedef const __attribute__((address_space(3))) bf16x8* lds_frag_p;
    unsigned kaddr[NKK];
    unsigned vaddr[2][2];
    {
        const int k_key = r & 15, v_key = (r >> 1) & 7;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) kaddr[kk] = smem_a + r * KROW + (((2 * kk + h) ^ k_key) << 4);
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) vaddr[T][s2] = smem_a + V_OFF + r * 128 + (((4 * T + 2 * s2 + h) ^ v_key) << 4);
    }

    f32x16 oacc[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int nt = (p.Lk + UV_ATT_KV - 1) / UV_ATT_KV;
    const int nt_full = p.Lk / UV_ATT_KV;
    if (nt_full > 0) fetch_full(kbase, vbase, 0);
    else fetch_clamped(0, vbase, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) asm volatile("" : "+v"(qf[kk]), "+v"(kaddr[kk]));
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(vaddr[i >> 1][i & 1]));
    asm volatile("" : "+v"(koff), "+v"(voff));
    __syncthreads();

    auto tile = [&](int t, auto masked_tag, auto next_tag, auto par_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        constexpr bool NEXT_FULL = decltype(next_tag)::value;
        constexpr int PAR = decltype(par_tag)::value;
        const int kv0 = t * UV_ATT_KV;
        vbase += 2 * UV_ATT_KV;
        kbase += kstep;
        if constexpr (NEXT_FULL) {
            fetch_full(kbase, vbase, PAR ^ 1);
        } else {
            if (t + 1 < nt_full) fetch_full(kbase, vbase, PAR ^ 1);
            else if (t + 1 < nt) fetch_clamped(kv0 + UV_ATT_KV, vbase, PAR ^ 1);
        }
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            f32x16 sacc;
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const bf16x8 kf = *(lds_frag_p)(kaddr[kk] + PAR * K_BYTES + T * 32 * KROW);
                sacc = mfma_32x32x16<false>(kf, qf[kk], sacc);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);
#pragma unroll
            for (int i_ = 0; i_ < 8 - AHEAD; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, AHEAD, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MASKED) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = 32 * T + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (kv0 + perm23(i) >= p.Lk) sacc[e] = -INFINITY;
                }
            }
            float mt = sacc[0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mt = fmaxf(mt, sacc[e]);
            {
                const unsigned u = __builtin_bit_cast(unsigned, mt);
                const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                mt = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
            }
            const float grow = (mt - m_run) * p.scale_log2;
            if (__any(grow > UV_ATT_DEFER)) {
                const float m_new = fmaxf(m_run, mt);
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.scale_log2);
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < ND; ++d)
#pragma unroll
                    for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
                m_run = m_new;
            }
            const float mneg = -m_run * p.scale_log2;
            float psum = 0.f;
            bf16x8 pf[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[8 * s2 + j], p.scale_log2, mneg));
                    psum += pv;
                    pf[s2][j] = (__bf16)pv;
                }
            l_run += psum;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 vf = *(lds_frag_p)(vaddr[T][s2] + PAR * V_BYTES + d * 32 * 128);
                    oacc[d] = mfma_32x32x16<false>(vf, pf[s2], oacc[d]);
                }
            __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 1);
#pragma unroll
            for (int i_ = 0; i_ < 8 - AHEAD; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, AHEAD, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    using F = std::false_type;
    using T_ = std::true_type;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    int t = 0;
    for (; t + 2 < nt_full; t += 2) {
        tile(t, F{}, T_{}, P0{});
        tile(t + 1, F{}, T_{}, P1{});
    }
    if (t + 1 < nt_full) {
        tile(t, F{}, T_{}, P0{});
        tile(t + 1, F{}, F{}, P1{});
    } else if (t < nt_full) {
        tile(t, F{}, F{}, P0{});
    }
    if (nt_full < nt) {
        if (nt_full & 1) tile(nt_full, T_{}, F{}, P1{});
        else tile(nt_full, T_{}, F{}, P0{});
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q = q0w + r;
    if (q < p.Lq) {
        bf16_t* op = p.out + (long)q * p.ldo + hcol + 4 * h;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 o = {pack16_2<false>(oacc[d][4 * g + 0] * inv, oacc[d][4 * g + 1] * inv),
                           pack16_2<false>(oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv)};
                *(u32x2*)(op + 32 * d + 8 * g) = o;
            }
    }
}

extern "C" int uv_flash_attn_bf16(const void* q, long ldq, const void* k, long ldk, const void* vt, long ldvt,
                                  void* out, long ldo, int batch, int Lq, int Lk, int H, int head_dim,
                                  float softmax_scale, void* stream) {
    UV_CHECK_ARG(q && k && vt && out, "uv_flash_attn_bf16: null pointer");
    UV_CHECK_ARG(head_dim == 128 || head_dim == 64, "uv_flash_attn_bf16: head_dim %d unsupported (64 or 128)", head_dim);
    UV_CHECK_ARG(Lq > 0 && Lk > 0 && H > 0 && batch > 0, "uv_flash_attn_bf16: bad shape B=%d Lq=%d Lk=%d H=%d", batch, Lq, Lk, H);
    UV_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0,
                 "uv_flash_attn_bf16: leading dimensions must be multiples of 8 elements");
    UV_CHECK_ARG(ldvt >= (long)(batch - 1) * Lk + (long)((Lk + 63) / 64) * 64,
                 "uv_flash_attn_bf16: ldvt=%ld must cover (batch-1)*Lk + Lk rounded up to 64 (batch=%d Lk=%d)", ldvt, batch, Lk);
    UV_CHECK_ARG(batch == 1 || Lk % 8 == 0, "uv_flash_attn_bf16: batch > 1 needs Lk %% 8 == 0 (Lk=%d)", Lk);
    UV_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt | (uintptr_t)out) & 15) == 0,
                 "uv_flash_attn_bf16: pointers must be 16-byte aligned");
    AttnArgs a;
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.vt = (const bf16_t*)vt; a.out = (bf16_t*)out;
    a.ldq = ldq; a.ldk = ldk; a.ldvt = ldvt; a.ldo = ldo;
    a.Lq = Lq; a.Lk = Lk; a.H = H; a.batch = batch;
    // workgroup shape: 4 waves x 2 workgroups per CU by default; UV_ATTN_WAVES=8 selects the 8-wave workgroup (A/B knob)
    static int nw = 0;
    if (!nw) {
        const char* e = getenv("UV_ATTN_WAVES");
        nw = (e && atoi(e) == 8) ? 8 : 4;
    }
    static int qb2 = -1;
    if (qb2 < 0) {
        const char* e = getenv("UV_ATTN_QB");
        qb2 = e ? (atoi(e) == 2 ? 1 : (atoi(e) == 3 ? 2 : 0)) : 0;
    }
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* nostamps = nullptr;
    a.scale_log2 = softmax_scale * 1.4426950408889634f;
    if (qb2 && head_dim == 128) {   // 4 waves x 64 queries, one wave per SIMD
        a.q_blocks = (Lq + 255) / 256;
        if (qb2 == 2) hipLaunchKernelGGL((flash_attn_fwd_kernel<128, 4, false, 2, true>), dim3(a.q_blocks * H * batch), dim3(256), 0, st, a, nostamps);
        else hipLaunchKernelGGL((flash_attn_fwd_kernel<128, 4, false, 2, false>), dim3(a.q_blocks * H * batch), dim3(256), 0, st, a, nostamps);
        UV_CHECK_LAUNCH("uv_flash_attn_bf16");
        return 0;
    }
    static int w3 = -1;
    // A/B knob: 0 = the two-waves-per-SIMD kernel, 1 = three 4-wave workgroups per CU, 3 (default) = one 12-wave workgroup per CU
    if (w3 < 0) { const char* e = getenv("UV_ATTN_W3"); w3 = e ? atoi(e) : 3; }
    if (w3 && head_dim == 128 && 128 * ldvt < (1L << 30) && 64 * ldk < (1L << 30)) {   // 32-bit lane offsets of the LDS-DMA pieces
        a.q_blocks = (Lq + 127) / 128;
        const dim3 g3(a.q_blocks * H * batch), b3(256);
        if (w3 == 3 && Lk >= 2048) {
            // long key sequences: one 12-wave workgroup per CU shares each K / V^T tile among 384 queries (a third of the L2 -> LDS
            // traffic; -2.8 % on the self-attention launches); short ones (cross-attention, Lk = 512: the prologue and the last
            // round weigh more) keep the 4-wave workgroups (+24 % there otherwise)
            a.q_blocks = (Lq + 383) / 384;
            hipLaunchKernelGGL((flash_attn_fwd12_kernel<3, true>), dim3(a.q_blocks * H * batch), dim3(768), 0, st, a);
        } else if (w3 == 2) hipLaunchKernelGGL((flash_attn_fwd3_kernel<3, false>), g3, b3, 0, st, a);   // A/B: plain block order
        else hipLaunchKernelGGL((flash_attn_fwd3_kernel<3, true>), g3, b3, 0, st, a);
        UV_CHECK_LAUNCH("uv_flash_attn_bf16");
        return 0;
    }
    a.q_blocks = (Lq + nw * UV_ATT_QW - 1) / (nw * UV_ATT_QW);
    const dim3 grid(a.q_blocks * H * batch), block(nw * 64);
    static int sgb1 = -1;
    if (sgb1 < 0) { const char* e = getenv("UV_ATTN_SGB"); sgb1 = (e && atoi(e) == 0) ? 0 : 1; }   // A/B knob, default on
    if (head_dim == 128 && nw == 8) hipLaunchKernelGGL((flash_attn_fwd_kernel<128, 8>), grid, block, 0, st, a, nostamps);
    else if (head_dim == 128 && sgb1) hipLaunchKernelGGL((flash_attn_fwd_kernel<128, 4, false, 1, true>), grid, block, 0, st, a, nostamps);
    else if (head_dim == 128) hipLaunchKernelGGL((flash_attn_fwd_kernel<128, 4>), grid, block, 0, st, a, nostamps);
    else if (nw == 8) hipLaunchKernelGGL((flash_attn_fwd_kernel<64, 8>), grid, block, 0, st, a, nostamps);
    else hipLaunchKernelGGL((flash_attn_fwd_kernel<64, 4>), grid, block, 0, st, a, nostamps);
    UV_CHECK_LAUNCH("uv_flash_attn_bf16");
    return 0;
}

// Developer diagnostic (not part of include/univid_hip.h): the D=128 kernel with per-segment s_memtime stamps.
// stamps: device buffer of q_blocks*H*nw*5 uint64 cycle sums [qk, softmax, pv, commit, barrier] per wave. Its fences
// forbid overlaps the real kernel has: read the SHARES, never its run time.
// The same kernel with IEEE fp16 q / k / V^T / out (fp32 softmax and accumulation): the SigLIP2 ranker's reference dtype.
extern "C" int uv_flash_attn_f16(const void* q, long ldq, const void* k, long ldk, const void* vt, long ldvt,
                                 void* out, long ldo, int batch, int Lq, int Lk, int H, int head_dim,
                                 float softmax_scale, void* stream) {
    UV_CHECK_ARG(q && k && vt && out, "uv_flash_attn_f16: null pointer");
    UV_CHECK_ARG(head_dim == 128 || head_dim == 64, "uv_flash_attn_f16: head_dim %d unsupported (64 or 128)", head_dim);
    UV_CHECK_ARG(Lq > 0 && Lk > 0 && H > 0 && batch > 0, "uv_flash_attn_f16: bad shape B=%d Lq=%d Lk=%d H=%d", batch, Lq, Lk, H);
    UV_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0,
                 "uv_flash_attn_f16: leading dimensions must be multiples of 8 elements");
    UV_CHECK_ARG(ldvt >= (long)(batch - 1) * Lk + (long)((Lk + 63) / 64) * 64,
                 "uv_flash_attn_f16: ldvt=%ld must cover (batch-1)*Lk + Lk rounded up to 64 (batch=%d Lk=%d)", ldvt, batch, Lk);
    UV_CHECK_ARG(batch == 1 || Lk % 8 == 0, "uv_flash_attn_f16: batch > 1 needs Lk %% 8 == 0 (Lk=%d)", Lk);
    UV_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt | (uintptr_t)out) & 15) == 0,
                 "uv_flash_attn_f16: pointers must be 16-byte aligned");
    AttnArgs a;
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.vt = (const bf16_t*)vt; a.out = (bf16_t*)out;
    a.ldq = ldq; a.ldk = ldk; a.ldvt = ldvt; a.ldo = ldo;
    a.Lq = Lq; a.Lk = Lk; a.H = H; a.batch = batch;
    a.scale_log2 = softmax_scale * 1.4426950408889634f;
    a.q_blocks = (Lq + 4 * UV_ATT_QW - 1) / (4 * UV_ATT_QW);
    const dim3 grid(a.q_blocks * H * batch), block(256);
    unsigned long long* nostamps = nullptr;
    if (head_dim == 128) hipLaunchKernelGGL((flash_attn_fwd_kernel<128, 4, false, 1, true, true>), grid, block, 0, (hipStream_t)stream, a, nostamps);
    else hipLaunchKernelGGL((flash_attn_fwd_kernel<64, 4, false, 1, false, true>), grid, block, 0, (hipStream_t)stream, a, nostamps);
    UV_CHECK_LAUNCH("uv_flash_attn_f16");
    return 0;
}

extern "C" int uvdbg_flash_attn_stamps(const void* q, long ldq, const void* k, long ldk, const void* vt, long ldvt, void* out,
                                       long ldo, int Lq, int Lk, int H, float softmax_scale, int nw,
                                       unsigned long long* stamps, int extra_lds, void* stream) {
    AttnArgs a;
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.vt = (const bf16_t*)vt; a.out = (bf16_t*)out;
    a.ldq = ldq; a.ldk = ldk; a.ldvt = ldvt; a.ldo = ldo; a.Lq = Lq; a.Lk = Lk; a.H = H; a.batch = 1;
    a.q_blocks = (Lq + nw * UV_ATT_QW - 1) / (nw * UV_ATT_QW);
    a.scale_log2 = softmax_scale * 1.4426950408889634f;
    const dim3 grid(a.q_blocks * H), block(nw * 64);
    // extra_lds: unused dynamic LDS, only to cap the number of co-resident workgroups per CU in occupancy experiments
    if (extra_lds > 0) {
        hipFuncSetAttribute((const void*)flash_attn_fwd_kernel<128, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, extra_lds);
        hipFuncSetAttribute((const void*)flash_attn_fwd_kernel<128, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, extra_lds);
    }
    if (nw == 42 || nw == 43) {
        a.q_blocks = (Lq + 255) / 256;
        const dim3 g2(a.q_blocks * H), b2(256);
        if (nw == 42) hipLaunchKernelGGL((flash_attn_fwd_kernel<128, 4, true, 2, false>), g2, b2, 0, (hipStream_t)stream, a, stamps);
        else hipLaunchKernelGGL((flash_attn_fwd_kernel<128, 4, true, 2, true>), g2, b2, 0, (hipStream_t)stream, a, stamps);
    } else if (nw == 44) {
        a.q_blocks = (Lq + 127) / 128;
        hipLaunchKernelGGL((flash_attn_fwd_kernel<128, 4, true, 1, true>), dim3(a.q_blocks * H), dim3(256), 0, (hipStream_t)stream, a, stamps);
    } else if (nw == 8) hipLaunchKernelGGL((flash_attn_fwd_kernel<128, 8, true>), grid, block, extra_lds, (hipStream_t)stream, a, stamps);
    else hipLaunchKernelGGL((flash_attn_fwd_kernel<128, 4, true>), grid, block, extra_lds, (hipStream_t)stream, a, stamps);
    UV_CHECK_LAUNCH("uvdbg_flash_attn_stamps");
    return 0;
}
