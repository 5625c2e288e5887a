// Flash attention forward (non-causal, head_dim 128, bf16 in / fp32 softmax+accumulate / bf16 out)
// for gfx950. Serves both the DiT spatiotemporal self-attention (Lq = Lk = L) and the text
// cross-attention (Lk = 512).
//
// Replaces: flash_attention()  models/wan/utils/modules/attention.py:24-130  (FA2/FA3 varlen call,
//           q/k/v cast to bf16 :59-83, result cast back :130), called from
//           WanSelfAttention.forward model.py:145-150 and WanCrossAttention.forward model.py:175.
//
// Three kernels share the operand layouts, LDS images and arithmetic below:
//   flash_attn_fwd12_kernel  head_dim 128, bf16, Lk >= 2048 (self-attention): one 12-wave workgroup per CU, up to 384 queries per K/V^T stream
//   flash_attn_fwd3_kernel   head_dim 128, bf16, shorter Lk (cross-attention): three 4-wave workgroups per CU (48 KiB LDS, <= 168 registers)
//   flash_attn_fwd_kernel    head_dim 64 / fp16 operands / very large leading dimensions: two 4-wave workgroups per CU (the first design)
// attn_select() is the ONE place that decides which of them serves a call (uv_flash_attn_kernel_name reports it).
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
// Template parameters of flash_attn_fwd_kernel: D head_dim (128 / 64), SGB fragment reads scheduled 6 ahead of their MFMA, F16 IEEE fp16 operands.
#include "attn_args.h"
#ifndef UV_ATTN_PROBE
#define UV_ATTN_PROBE 0      // 1 / 2: timing-only LDS-read probes of flash_attn_fwd12_kernel, built only by tools/diag/build_attn_probe.py
#endif
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

// tools/diag/attn_timeline.hip builds this file with -DUV_ATTN_TIMELINE: 100 MHz wall-clock stamps per workgroup at the phase
// boundaries of flash_attn_fwd3_kernel / flash_attn_fwd12_kernel plus the hardware id of the CU it ran on. Compiled out of the library.
#ifdef UV_ATTN_TIMELINE
__device__ unsigned long long* uv_attn_tl;
#define UV_TL(id, slot) do { if (threadIdx.x == 0) uv_attn_tl[(size_t)(id) * 8 + (slot)] = wall_clock64(); } while (0)
#define UV_TL_HW(id) do { if (threadIdx.x == 0) { unsigned hw_, xcc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_)); \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_)); uv_attn_tl[(size_t)(id) * 8 + 7] = ((unsigned long long)xcc_ << 32) | hw_; } } while (0)
#else
#define UV_TL(id, slot) do {} while (0)
#define UV_TL_HW(id) do {} while (0)
#endif

typedef __attribute__((address_space(3))) void lds_void_a;

// D = head_dim (128 for TI2V-5B; 64 for the reference's CPU-runnable tiny config and the SigLIP2 ranker).
// 4 waves x 32 queries per workgroup, 2 workgroups per CU: the two waves that share a SIMD then belong to DIFFERENT workgroups, are
// not re-aligned by a common barrier every tile, and drift into complementary phases - one in its MFMA cluster while the other
// does softmax VALU work.
template <int D, bool SGB = false, bool F16 = false, int QN = 0>
__global__ __launch_bounds__(256, 2) void flash_attn_fwd_kernel(AttnArgs p) {
    static_assert(!F16 || QN == 0, "the fused q-norm prologue is built for bf16 only");
    constexpr int NW = 4;
    constexpr int NT = NW * 64;
    constexpr int KROW = 2 * D;                 // bytes per K row in LDS (256 or 128)
    constexpr int KCH = D / 8;                  // 16-B chunks per K row
    constexpr int NKK = D / 16;                 // MFMA k-steps over the head dim
    constexpr int ND = D / 32;                  // 32-row d tiles of O^T
    constexpr int K_BYTES = UV_ATT_KV * KROW;   // 16 KiB at D=128
    constexpr int V_BYTES = D * 128;            // 16 KiB at D=128
    constexpr int STAGE = K_BYTES + V_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

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
        if (QN) p.q_rs += b * p.Lq;
        p.k += b * p.Lk * p.ldk;
        p.vt += (long)b * p.Lk;
        p.out += b * p.Lq * p.ldo;
    }
    const int q0w = qb * (NW * UV_ATT_QW) + wave * UV_ATT_QW;
    const long hcol = (long)head * D;

    // ---- Q fragments (B operand of S^T = K.Q^T): lane (r,h) holds Q[q0w+r][16kk+8h .. +7]
    bf16x8 qf[NKK];
    AttnQNorm<NKK, QN> qnorm;
    attn_load_q<NKK, QN>(p, qf, qnorm, min(q0w + r, p.Lq - 1), hcol, h);

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

    f32x16 oacc[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
    float m_run = -INFINITY;  // running reference maximum of the raw scores (both half-waves hold the same value)
    float l_run = 0.f;        // this half-wave's partial row sum

    const int nt = (p.Lk + UV_ATT_KV - 1) / UV_ATT_KV;
    const int nt_full = p.Lk / UV_ATT_KV;
    fetch(0, 0);
    attn_apply_qnorm<NKK, QN>(qf, qnorm);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // Pin "every prologue load has landed" BEFORE the loop: vmcnt retires in order, so if the compiler has to assume
    // the Q fragment loads may still be in flight at the loop header it guards their first use inside the loop with
    // vmcnt(1)/vmcnt(0) - which in steady state waits for the K/V prefetch issued a few instructions earlier and
    // exposes a full L2/HBM latency in every tile (seen in the ISA; ~1000 cycles of a 4500-cycle tile).
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) asm volatile("" : "+v"(qf[kk]));
    __syncthreads();

    // One KV tile. MASKED is a compile-time flag so that the main loop carries no masking code at all (only the ragged
    // last tile is instantiated with it).
    // FETCH: 1 = the next tile is a full one (running pointers, no branch: the LDS-DMA instructions then sit in the same
    // scheduling region as the QK MFMAs and are dealt out between them), 0 = decide at run time (last full tile / ragged tile)
    auto tile = [&](int t, auto masked_tag, auto fetch_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        constexpr bool FETCH_FAST = decltype(fetch_tag)::value;
        const int kv0 = t * UV_ATT_KV;
        const char* base = smem + (t & 1) * STAGE;
        // other buffer: last read in tile t-1, fenced by its barrier
        if constexpr (FETCH_FAST) {
            fetch_next_full((t + 1) & 1);
        } else {
            if (t + 1 < nt_full) fetch_next_full((t + 1) & 1);
            else if (t + 1 < nt) fetch(kv0 + UV_ATT_KV, (t + 1) & 1);   // the ragged last tile: clamped rows
        }

        // ---- S^T = K . Q^T  (two 32-key tiles)
        f32x16 sacc[2];
#pragma unroll
        for (int T = 0; T < 2; ++T) {
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[T][e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const bf16x8 kf = *(const bf16x8*)(base + T * 32 * KROW + k_row_off + (((2 * kk + h) ^ k_key) << 4));
                sacc[T] = mfma_32x32x16<F16>(kf, qf[kk], sacc[T]);
            }
        }

        if constexpr (SGB && D == 128) {   // fragment reads 6 ahead of the MFMA that consumes them
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
            for (int i_ = 0; i_ < 10; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        {
            // ---- mask the ragged last tile with the TRUE key index of each accumulator row
            if (MASKED) {
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
            {   // the other half-wave's maximum: v_permlane32_swap (one VALU op) instead of a ds_bpermute round trip
                const unsigned u = __builtin_bit_cast(unsigned, mt);
                const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                mt = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
            }
            float mneg;
            // The reference maximum m_run moves only when some row's maximum outgrows it by more than 2^UV_ATT_DEFER in
            // the exponent domain (then P <= 2^UV_ATT_DEFER instead of <= 1 until the next move; P, l and O share one
            // scale and bf16 / f32 relative precision is scale-invariant). With the exact running maximum about half
            // of all tiles of a long sequence still see a new maximum in SOME row of the wave and pay the O rescale.
            const float grow = (mt - m_run) * p.scale_log2;      // +inf on the first tile (m_run = -inf)
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
            mneg = -m_run * p.scale_log2;
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
                        if constexpr (F16) {   // fp16 bits carried in the bf16x8 container (P <= 2^UV_ATT_DEFER fits fp16 easily)
                            bf16_t bits = out16<true>(pv);
                            pf[T][s][j] = __builtin_bit_cast(__bf16, bits);
                        } else {
                            pf[T][s][j] = (__bf16)pv;
                        }
                    }
            l_run += psum;

            // ---- O^T += V^T . P^T
            if constexpr (SGB && D == 128) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int T = 0; T < 2; ++T)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const bf16x8 vf =
                            *(const bf16x8*)(base + d * 32 * 128 + v_row_off + (((4 * T + 2 * s + h) ^ v_key) << 4));
                        oacc[d] = mfma_32x32x16<F16>(vf, pf[T][s], oacc[d]);
                    }
        }

        if constexpr (SGB && D == 128) {
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 1);
#pragma unroll
            for (int i_ = 0; i_ < 10; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's share of tile t+1 has landed in LDS
        __syncthreads();
    };

    for (int t = 0; t + 1 < nt_full; ++t) tile(t, std::false_type{}, std::true_type{});
    if (nt_full > 0) tile(nt_full - 1, std::false_type{}, std::false_type{});
    if (nt_full < nt) tile(nt_full, std::true_type{}, std::false_type{});

    // ---- finish: combine the two half-wave sums, normalise, store bf16 rows
    {
        float l_half = l_run;
        const float l_tot = l_half + __shfl_xor(l_half, 32, 64);
        const float inv = 1.0f / l_tot;
        const int q = q0w + r;
        if (q < p.Lq) {
            bf16_t* op = p.out + (long)q * p.ldo + hcol + 4 * h;
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 o = {pack16_2<F16>(oacc[d][4 * g + 0] * inv, oacc[d][4 * g + 1] * inv),
                               pack16_2<F16>(oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv)};
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
template <int AHEAD = 3, bool XCD = true, int QN = 0>
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
        if (QN) p.q_rs += b * p.Lq;
        p.k += b * p.Lk * p.ldk;
        p.vt += (long)b * p.Lk;
        p.out += b * p.Lq * p.ldo;
    }
    const int q0w = qb * (NW * UV_ATT_QW) + wave_u * UV_ATT_QW;
    const long hcol = (long)head * D;
    UV_TL(vb, 0);
    UV_TL_HW(vb);

    bf16x8 qf[NKK];
    AttnQNorm<NKK, QN> qnorm;
    attn_load_q<NKK, QN>(p, qf, qnorm, min(q0w + r, p.Lq - 1), hcol, h);

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
    attn_apply_qnorm<NKK, QN>(qf, qnorm);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) asm volatile("" : "+v"(qf[kk]), "+v"(kaddr[kk]));
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(vaddr[i >> 1][i & 1]));
    asm volatile("" : "+v"(koff), "+v"(voff));
    __syncthreads();
    UV_TL(vb, 1);

    // One staged tile = 64 keys = two 32-key halves that are computed one after the other (QK, softmax, PV per half): the S
    // accumulator of only one half is live beside O and Q, which is what leaves registers for fragment reads ahead of their
    // MFMAs at 168 registers per wave. The reference maximum is therefore reconsidered per half.
    // NEXT: 1 = tile t+1 is a full one, 0 = decide at run time (ragged or none); PAR = t & 1 (compile-time: the K buffer
    // offsets become instruction immediates)
    // PAR = -1: the GENERIC form of the tail tiles (parity, "is the next tile full" and "is this the ragged tile" decided at run time).
    // Until round 4 the tail was five compile-time instantiations behind an if / else chain; hipcc allocated registers across that chain
    // and SPILLED 380 of them to scratch (444 bytes per lane; none in the main loop) - and 2 of the cross-attention's 8 tiles ran there.
    // One generic instantiation in a loop of its own keeps the tail at the main loop's register footprint (8 spilled registers, none
    // in a loop): the cross-attention launch 193.9 -> 165.8 us (two builds of the library in one process, tools/attn_so_ab.py), bit-identical.
    auto tile = [&](int t, auto masked_tag, auto next_tag, auto par_tag) {
        constexpr bool DYN = decltype(par_tag)::value < 0;
        constexpr bool MASKED_C = decltype(masked_tag)::value;
        constexpr bool NEXT_FULL = decltype(next_tag)::value && !DYN;
        const int PAR = DYN ? (t & 1) : decltype(par_tag)::value;
        const bool MASKED = DYN ? (t >= nt_full) : MASKED_C;
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
    // the last one or two full tiles and the ragged tile: the generic form, in a loop of its own
    for (; t < nt; ++t) tile(t, F{}, F{}, std::integral_constant<int, -1>{});

    UV_TL(vb, 2);
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q = q0w + r;
    if ((p.ldo & 7) == 0) {
        // O through LDS: in the MFMA's layout a lane owns 16 pieces of 8 bytes along its query's row, and stored directly every
        // store instruction touches 32 rows with 16 bytes each (3.4 us of store issue per workgroup, tools/diag/attn_timeline;
        // cross-attention at the DiT shape 218 -> 208 us in a same-device A/B, bit-identical).
        // Each wave transposes through its own 32 x 272-byte image (the K / V^T buffers are free: the last tile ended on a barrier)
        // and stores 4 whole 256-byte row segments per instruction.
        constexpr int RS = 272;
        char* const img = smem + wave_u * (32 * RS);
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 o = {pack16_2<false>(oacc[d][4 * g + 0] * inv, oacc[d][4 * g + 1] * inv),
                           pack16_2<false>(oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv)};
                *(u32x2*)(img + r * RS + 64 * d + 16 * g + 8 * h) = o;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // wave-private image: no barrier
        const int srow = lane >> 4, chunk = lane & 15;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 4 * i + srow;
            const u32x4 v = *(const u32x4*)(img + row * RS + chunk * 16);
            if (q0w + row < p.Lq) *(u32x4*)(p.out + (long)(q0w + row) * p.ldo + hcol + chunk * 8) = v;
        }
    } else if (q < p.Lq) {
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
    UV_TL(vb, 3);
#ifdef UV_ATTN_TIMELINE_DRAIN
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    UV_TL(vb, 4);
#endif
}

// ------------------------------------------------------------------------------------------------------------------------
// Long key sequences (Lk >= 2048): ONE 12-wave workgroup per CU (up to 384 queries, three waves per SIMD) sharing each K / V^T
// tile: a third of the L2 -> LDS traffic and 2-3 instead of 8 LDS-DMA pieces per wave and tile. Bit-identical to
// flash_attn_fwd3_kernel (same per-wave arithmetic). Staging as in the two-waves kernel (K and V^T double-buffered, tile t+1
// requested at the top of tile t, ONE vmcnt(0) + barrier per tile), compute body and register diet of flash_attn_fwd3_kernel
// (32-key halves, SGPR-base DMA, immediates for the buffer parity).
// QUERY BLOCKS OF 12 AND 8 UNITS: a (sample, head) has NWU = ceil(Lq / 32) wave-units of 32 queries. Cutting it into blocks of 12
// waves only gives, at the DiT's shape (2 x 24 heads x 358 units), 1 440 workgroups = 5.6 rounds of 256 CUs: the sixth round
// runs on 62 % of the chip for the time of a full one. A workgroup's time is set by its busiest SIMD (three waves each at 12
// units), so what shortens the tail is workgroups with TWO waves on every SIMD: the host cuts each head into n12 blocks of 12
// units and n8 blocks of 8 units (attn12_cut: 26 + 6 there) and orders all 12-unit blocks before all 8-unit blocks, so every CU
// ends with short workgroups instead of some CUs ending with a long one. All 12 waves of a workgroup stage K / V^T tiles and
// take every barrier; the waves beyond the block's unit count are LOADER-ONLY (no QK / softmax / PV). Per-query arithmetic does
// not depend on the cut: results are bit-identical for any cut.
// ------------------------------------------------------------------------------------------------------------------------
template <int AHEAD = 3, bool XCD = true, int QN = 0>
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3))) void flash_attn_fwd12_kernel(AttnArgs p) {
    constexpr int D = 128, NW = 12, KROW = 256, NKK = 8, ND = 4;
    constexpr int K_BYTES = UV_ATT_KV * KROW, V_BYTES = D * 128;
    __shared__ __attribute__((aligned(16))) char smem[2 * K_BYTES + 2 * V_BYTES];
    constexpr int V_OFF = 2 * K_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);

    // Workgroup b runs on XCD b % 8 as the (b / 8)-th workgroup dealt to it. Every XCD gets an equal share of the 12-unit blocks
    // FIRST and of the 8-unit blocks AFTER them (so each CU ends on short workgroups), and within each kind a contiguous range of
    // (sample, head, block) ids (so a head's K / V^T streams through one L2). Needs both block totals to divide by 8 (24 heads do);
    // otherwise plain id order, which keeps "12-unit blocks first" but not the L2 locality.
    const int nbh = p.H * p.batch, n8 = p.q_blocks - p.n12;
    const int tot12 = p.n12 * nbh, tot8 = n8 * nbh;
    int id = blockIdx.x;                 // < tot12: a 12-unit block, else 8-unit block number id - tot12
    if (XCD && (tot12 & 7) == 0 && (tot8 & 7) == 0) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3, c12 = tot12 >> 3, c8 = tot8 >> 3;
        id = j < c12 ? x * c12 + j : tot12 + x * c8 + (j - c12);
    }
    int bh, u0, units;
    if (id < tot12) {
        bh = id / p.n12;
        u0 = (id - bh * p.n12) * 12;
        units = 12;
    } else {
        const int v2 = id - tot12;
        bh = v2 / n8;
        u0 = p.n12 * 12 + (v2 - bh * n8) * 8;
        units = 8;
    }
    const int head = bh % p.H;
    {
        const long b = bh / p.H;
        p.q += b * p.Lq * p.ldq;
        if (QN) p.q_rs += b * p.Lq;
        p.k += b * p.Lk * p.ldk;
        p.vt += (long)b * p.Lk;
        p.out += b * p.Lq * p.ldo;
    }
    const int nwu = (p.Lq + UV_ATT_QW - 1) / UV_ATT_QW;
    const bool compute_wave = wave_u < min(units, nwu - u0);      // the head's last block may be ragged
    const int q0w = (u0 + wave_u) * UV_ATT_QW;
    const long hcol = (long)head * D;
    UV_TL(blockIdx.x, 0);
    UV_TL_HW(blockIdx.x);

    bf16x8 qf[NKK];
    AttnQNorm<NKK, QN> qnorm;
    attn_load_q<NKK, QN>(p, qf, qnorm, min(q0w + r, p.Lq - 1), hcol, h);

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
#if UV_ATTN_PROBE
    bf16x8 kf_keep = qf[0], vf_keep = qf[1];      // timing-only probes (see the tile body): never in the product build
#endif

    const int nt = (p.Lk + UV_ATT_KV - 1) / UV_ATT_KV;
    const int nt_full = p.Lk / UV_ATT_KV;
    if (nt_full > 0) fetch_full(kbase, vbase, 0);
    else fetch_clamped(0, vbase, 0);
    attn_apply_qnorm<NKK, QN>(qf, qnorm);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) asm volatile("" : "+v"(qf[kk]), "+v"(kaddr[kk]));
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(vaddr[i >> 1][i & 1]));
    asm volatile("" : "+v"(koff), "+v"(voff));
    __syncthreads();
    UV_TL(blockIdx.x, 1);

    if (!compute_wave) {
        // loader-only wave: its share of every tile's LDS-DMA pieces and every barrier, nothing else
        for (int t = 0; t < nt; ++t) {
            vbase += 2 * UV_ATT_KV;
            kbase += kstep;
            if (t + 1 < nt_full) fetch_full(kbase, vbase, (t & 1) ^ 1);
            else if (t + 1 < nt) fetch_clamped((t + 1) * UV_ATT_KV, vbase, (t & 1) ^ 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // PAR = -1: the generic form of the tail tiles (see flash_attn_fwd3_kernel). Here the if / else chain of five instantiations spilled
    // 315 registers (408 bytes of scratch per lane) in the last two or three tiles of every workgroup: ~150 MB of scratch writes per
    // batch-2 launch at L = 11 440 - HALF of what rocprofv3's WRITE_SIZE reported for this kernel (287 640 KB against 137 280 KB of
    // output; tools/diag/attn_write_probe.*). With the generic tail: 1 spilled register, the launch time unchanged (2.740 against 2.742 ms;
    // three tiles of 179 per workgroup), bit-identical.
    auto tile = [&](int t, auto masked_tag, auto next_tag, auto par_tag) {
        constexpr bool DYN = decltype(par_tag)::value < 0;
        constexpr bool MASKED_C = decltype(masked_tag)::value;
        constexpr bool NEXT_FULL = decltype(next_tag)::value && !DYN;
        const int PAR = DYN ? (t & 1) : decltype(par_tag)::value;
        const bool MASKED = DYN ? (t >= nt_full) : MASKED_C;
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
#if UV_ATTN_PROBE == 1      // timing-only probe (tools/diag/build_attn_probe.py; WRONG results): half of the fragment reads skipped, the skipped fragment = a copy of the previous one
                bf16x8 kf;
                if (kk & 1) kf = kf_keep; else kf = *(lds_frag_p)(kaddr[kk] + PAR * K_BYTES + T * 32 * KROW);
                kf_keep = kf;
#elif UV_ATTN_PROBE == 2    // timing-only probe: every read still issued and waited for, the odd fragments then REPLACED by a copy of the previous one (same operand data as probe 1)
                bf16x8 kf = *(lds_frag_p)(kaddr[kk] + PAR * K_BYTES + T * 32 * KROW);
                if (kk & 1) { asm volatile("" :: "v"(kf)); kf = kf_keep; }
                kf_keep = kf;
#else
                const bf16x8 kf = *(lds_frag_p)(kaddr[kk] + PAR * K_BYTES + T * 32 * KROW);
#endif
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
#if UV_ATTN_PROBE == 1
                    bf16x8 vf;
                    if (s2 & 1) vf = vf_keep; else vf = *(lds_frag_p)(vaddr[T][s2] + PAR * V_BYTES + d * 32 * 128);
                    vf_keep = vf;
#elif UV_ATTN_PROBE == 2
                    bf16x8 vf = *(lds_frag_p)(vaddr[T][s2] + PAR * V_BYTES + d * 32 * 128);
                    if (s2 & 1) { asm volatile("" :: "v"(vf)); vf = vf_keep; }
                    vf_keep = vf;
#else
                    const bf16x8 vf = *(lds_frag_p)(vaddr[T][s2] + PAR * V_BYTES + d * 32 * 128);
#endif
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
    // the last one or two full tiles and the ragged tile: the generic form, in a loop of its own
    for (; t < nt; ++t) tile(t, F{}, F{}, std::integral_constant<int, -1>{});

    UV_TL(blockIdx.x, 2);
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q = q0w + r;
    if ((p.ldo & 7) == 0) {
        // O through LDS (as in flash_attn_fwd3_kernel), in two passes of two d-blocks each: twelve wave-private images of 32 rows x
        // (128 + 16) bytes fit the 64 KiB that held K / V^T (free: the last tile ended on a barrier that the loader-only waves took too);
        // every store instruction writes 8 whole 128-byte lines (self-attention at the DiT shape 2.724 -> 2.711 ms in a same-device A/B,
        // bit-identical).
        constexpr int RS = 144;
        char* const img = smem + wave_u * (32 * RS);
        const int srow = lane >> 3, chunk = lane & 7;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = 2 * half + dd;
                    u32x2 o = {pack16_2<false>(oacc[d][4 * g + 0] * inv, oacc[d][4 * g + 1] * inv),
                               pack16_2<false>(oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv)};
                    *(u32x2*)(img + r * RS + 64 * dd + 16 * g + 8 * h) = o;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private image: no barrier
            u32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = *(const u32x4*)(img + (8 * i + srow) * RS + chunk * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the second pass overwrites the image
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 8 * i + srow;
                if (q0w + row < p.Lq) *(u32x4*)(p.out + (long)(q0w + row) * p.ldo + hcol + 64 * half + chunk * 8) = v[i];
            }
        }
    } else if (q < p.Lq) {
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
    UV_TL(blockIdx.x, 3);
#ifdef UV_ATTN_TIMELINE_DRAIN
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    UV_TL(blockIdx.x, 4);
#endif
}

// ---- kernel selection: the ONE place that decides which kernel serves a call ------------------------------------------------
enum AttnKernel { ATT_FWD12 = 0, ATT_FWD3 = 1, ATT_FWD_D128 = 2, ATT_FWD_D64 = 3 };
static const char* const kAttnKernelName[] = {"flash_attn_fwd12_kernel", "flash_attn_fwd3_kernel", "flash_attn_fwd_kernel<128>",
                                              "flash_attn_fwd_kernel<64>"};

static AttnKernel attn_select(int Lk, int head_dim, long ldk, long ldvt, bool f16) {
    if (head_dim != 128) return ATT_FWD_D64;
    // fwd12 / fwd3 address their LDS-DMA pieces with 32-bit lane offsets from a uniform base
    if (f16 || 128 * ldvt >= (1L << 30) || 64 * ldk >= (1L << 30)) return ATT_FWD_D128;
    // long key sequences: one 12-wave workgroup per CU shares each K / V^T tile among up to 384 queries (a third of the L2 -> LDS
    // traffic; -2.8 % on the self-attention launches); short ones (cross-attention, Lk = 512: prologue and last round weigh
    // more) keep the 4-wave workgroups (the 12-wave form is 24 % slower there)
    if (Lk < 2048) return ATT_FWD3;
    return ATT_FWD12;
}

// Cut of a (sample, head)'s NWU = ceil(Lq / 32) query units into n12 blocks of 12 units followed by n8 blocks of 8 units for
// flash_attn_fwd12_kernel. Model: a 12-unit workgroup takes time 1, an 8-unit one T8 = 0.76 (measured), every XCD's CUs pick
// their workgroups up in id order (12-unit blocks first = longest-first list scheduling); the cut with the smallest simulated
// makespan wins, ties go to fewer workgroups. Measured at the DiT shape (batch 2): 26 + 6 blocks 2.86 ms against 2.90 ms for
// 30 + 0; mixes with more 8-unit blocks lose (24 + 9: 3.03 ms). At batch 1 the model picks 30 + 0 (768 workgroups = 3 rounds).
static void attn12_cut(int Lq, int heads_total, int* n12_out, int* n8_out) {
    const int nwu = (Lq + UV_ATT_QW - 1) / UV_ATT_QW, ncu = uv_num_cus();
    // small per-thread memo (ctypes releases the GIL: two host threads, one per GPU, may be in here with different shapes at once;
    // alternating shapes must not re-run the list-scheduling scan, ~ blocks x CUs x cuts host operations, on every call)
    struct Memo { int nwu, key, n12, n8; };
    static thread_local Memo memo[8];
    static thread_local int memo_next = 0;
    const int mkey = heads_total * 1024 + ncu;
    for (const Memo& e : memo)
        if (e.nwu == nwu && e.key == mkey) { *n12_out = e.n12; *n8_out = e.n8; return; }
    const double T8 = 0.76;   // measured: all-8-unit cut 0.365 ms per round of workgroups, all-12-unit cut 0.48 ms (batch 2, L = 11 440)
    int best12 = (nwu + 11) / 12, best8 = 0;
    double best = 1e30;
    for (int n8 = 0; n8 * 8 < nwu + 8; ++n8) {
        const int rest = nwu - 8 * n8;
        const int n12 = rest > 0 ? (rest + 11) / 12 : 0;
        if (n12 == 0 && n8 * 8 - nwu >= 8) break;
        // list scheduling on ncu identical machines: loads kept in a small array (ncu <= 1024)
        double load[1024];
        const int m = ncu < 1024 ? ncu : 1024;
        for (int i = 0; i < m; ++i) load[i] = 0.0;
        auto place = [&](long count, double t) {
            for (long j = 0; j < count; ++j) {
                int arg = 0;
                for (int i = 1; i < m; ++i) if (load[i] < load[arg]) arg = i;
                load[arg] += t;
            }
        };
        place((long)n12 * heads_total, 1.0);
        place((long)n8 * heads_total, T8);
        double mk = 0.0;
        for (int i = 0; i < m; ++i) mk = load[i] > mk ? load[i] : mk;
        if (mk < best - 1e-9) { best = mk; best12 = n12; best8 = n8; }
    }
    memo[memo_next] = Memo{nwu, mkey, best12, best8};
    memo_next = (memo_next + 1) & 7;
    *n12_out = best12; *n8_out = best8;
}

// Name of the kernel uv_flash_attn_bf16 / _f16 dispatches for this problem (bench.py labels its roofline line with it).
extern "C" int uv_flash_attn_kernel_name(int Lk, int head_dim, long ldk, long ldvt, int f16, char* buf, int len) {
    UV_CHECK_ARG(buf && len > 0, "uv_flash_attn_kernel_name: bad buffer");
    UV_CHECK_ARG(head_dim == 128 || head_dim == 64, "uv_flash_attn_kernel_name: head_dim %d unsupported (64 or 128)", head_dim);
    snprintf(buf, len, "%s", kAttnKernelName[attn_select(Lk, head_dim, ldk, ldvt, f16 != 0)]);
    return 0;
}

template <bool F16>
static int attn_entry(const char* name, const void* q, long ldq, const void* k, long ldk, const void* vt, long ldvt, void* out, long ldo,
                      int batch, int Lq, int Lk, int H, int head_dim, float softmax_scale, void* stream, const float* q_rs = nullptr,
                      const float* q_w = nullptr) {
    UV_CHECK_ARG(q && k && vt && out, "%s: null pointer", name);
    UV_CHECK_ARG(head_dim == 128 || head_dim == 64, "%s: head_dim %d unsupported (64 or 128)", name, head_dim);
    UV_CHECK_ARG(Lq > 0 && Lk > 0 && H > 0 && batch > 0, "%s: bad shape B=%d Lq=%d Lk=%d H=%d", name, batch, Lq, Lk, H);
    UV_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0, "%s: leading dimensions must be multiples of 8 elements", name);
    UV_CHECK_ARG(ldvt >= (long)(batch - 1) * Lk + (long)((Lk + 63) / 64) * 64,
                 "%s: ldvt=%ld must cover (batch-1)*Lk + Lk rounded up to 64 (batch=%d Lk=%d)", name, ldvt, batch, Lk);
    UV_CHECK_ARG(batch == 1 || Lk % 8 == 0, "%s: batch > 1 needs Lk %% 8 == 0 (Lk=%d)", name, Lk);
    UV_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)vt | (uintptr_t)out) & 15) == 0, "%s: pointers must be 16-byte aligned", name);
    AttnArgs a;
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.vt = (const bf16_t*)vt; a.out = (bf16_t*)out;
    a.ldq = ldq; a.ldk = ldk; a.ldvt = ldvt; a.ldo = ldo;
    a.Lq = Lq; a.Lk = Lk; a.H = H; a.batch = batch; a.n12 = 0;
    a.scale_log2 = softmax_scale * 1.4426950408889634f;
    a.q_rs = q_rs; a.q_w = q_w;
    const bool qn = q_rs != nullptr;
    UV_CHECK_ARG(!qn || (q_w && !F16 && (((uintptr_t)q_w | (uintptr_t)q_rs) & 15) == 0), "%s: q_rs needs q_weight (f32, 16-byte aligned), bf16 only", name);
    hipStream_t st = (hipStream_t)stream;
    switch (attn_select(Lk, head_dim, ldk, ldvt, F16)) {
        case ATT_FWD12:
            {
                int n12 = 0, n8 = 0;
                attn12_cut(Lq, H * batch, &n12, &n8);
                if (const int force = uv_option(UV_OPT_ATTN_CUT); force > 0) {       // A/B tools: n8 = force - 1 eight-unit blocks per head
                    const int nwu = (Lq + UV_ATT_QW - 1) / UV_ATT_QW, rest = nwu - 8 * (force - 1);
                    n8 = force - 1;
                    n12 = rest > 0 ? (rest + 11) / 12 : 0;
                }
                a.n12 = n12;
                a.q_blocks = n12 + n8;
            }
            if (qn) hipLaunchKernelGGL((flash_attn_fwd12_kernel<3, true, 1>), dim3(a.q_blocks * H * batch), dim3(768), 0, st, a);
            else hipLaunchKernelGGL((flash_attn_fwd12_kernel<3, true>), dim3(a.q_blocks * H * batch), dim3(768), 0, st, a);
            break;
        case ATT_FWD3:
            a.q_blocks = (Lq + 127) / 128;
            if (qn) hipLaunchKernelGGL((flash_attn_fwd3_kernel<3, true, 1>), dim3(a.q_blocks * H * batch), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((flash_attn_fwd3_kernel<3, true>), dim3(a.q_blocks * H * batch), dim3(256), 0, st, a);
            break;
        case ATT_FWD_D128:
            a.q_blocks = (Lq + 127) / 128;
            if constexpr (!F16) {
                if (qn) { hipLaunchKernelGGL((flash_attn_fwd_kernel<128, true, false, 1>), dim3(a.q_blocks * H * batch), dim3(256), 0, st, a); break; }
            }
            hipLaunchKernelGGL((flash_attn_fwd_kernel<128, true, F16>), dim3(a.q_blocks * H * batch), dim3(256), 0, st, a);
            break;
        case ATT_FWD_D64:
            a.q_blocks = (Lq + 127) / 128;
            if constexpr (!F16) {
                if (qn) { hipLaunchKernelGGL((flash_attn_fwd_kernel<64, false, false, 1>), dim3(a.q_blocks * H * batch), dim3(256), 0, st, a); break; }
            }
            hipLaunchKernelGGL((flash_attn_fwd_kernel<64, false, F16>), dim3(a.q_blocks * H * batch), dim3(256), 0, st, a);
            break;
    }
    UV_CHECK_LAUNCH(name);
    return 0;
}

extern "C" int uv_flash_attn_bf16(const void* q, long ldq, const void* k, long ldk, const void* vt, long ldvt, void* out, long ldo,
                                  int batch, int Lq, int Lk, int H, int head_dim, float softmax_scale, void* stream) {
    return attn_entry<false>("uv_flash_attn_bf16", q, ldq, k, ldk, vt, ldvt, out, ldo, batch, Lq, Lk, H, head_dim, softmax_scale, stream);
}

// uv_flash_attn_bf16 on a RAW q projection: WanRMSNorm of q (norm_q, model.py:138 / 169) applied in the kernels' Q prologue from the per-row scale
// q_rs (uv_rms_scale_from_ssq of the q GEMM's sums of squares) and the norm weight - the separate pass over q is gone. No RoPE here (cross-attention).
extern "C" int uv_flash_attn_bf16_qnorm(const void* q, long ldq, const void* k, long ldk, const void* vt, long ldvt, void* out, long ldo,
                                        int batch, int Lq, int Lk, int H, int head_dim, float softmax_scale, const float* q_rs,
                                        const float* q_weight, void* stream) {
    UV_CHECK_ARG(q_rs && q_weight, "uv_flash_attn_bf16_qnorm: q_rs / q_weight missing");
    return attn_entry<false>("uv_flash_attn_bf16_qnorm", q, ldq, k, ldk, vt, ldvt, out, ldo, batch, Lq, Lk, H, head_dim, softmax_scale, stream,
                             q_rs, q_weight);
}

// The same with IEEE fp16 q / k / V^T / out (fp32 softmax and accumulation): the SigLIP2 ranker's reference dtype.
extern "C" int uv_flash_attn_f16(const void* q, long ldq, const void* k, long ldk, const void* vt, long ldvt, void* out, long ldo,
                                 int batch, int Lq, int Lk, int H, int head_dim, float softmax_scale, void* stream) {
    return attn_entry<true>("uv_flash_attn_f16", q, ldq, k, ldk, vt, ldvt, out, ldo, batch, Lq, Lk, H, head_dim, softmax_scale, stream);
}
