// 256x256x64 tile on FOUR waves, one per SIMD, each owning a 128x128 quadrant of the output (the wave-tile the vendor's
// MT256x256x64 / MIWT8_8 / WG32_8_1 kernel is built on: round-5 calibration).
//
// Against the 8-wave ping-pong kernel (gemm_bf16_8ph_*): per K tile and CU 128 KiB of LDS fragment reads instead of 192, two
// workgroup barriers instead of four, no second wave per SIMD to take turns with - every SIMD's matrix pipe is fed by ONE
// instruction stream in which the LDS reads of the next k-step and the LDS-DMA of the K tile after next sit in the shadows of
// the MFMAs. The accumulators (8 x 8 fragments = 256 registers) live in the AGPR half of the 512-entry file.
//
// Staging: buffer_load_dwordx4 ... lds (LDS-DMA through a buffer descriptor: the per-lane offset - row-in-piece and swizzled
// chunk - is ONE 32-bit register per operand for the whole kernel, everything that moves is a scalar offset: no per-lane address
// arithmetic in the loop). Two 64-KiB LDS buffers; the LDS image (128-byte rows, 16-byte chunks XOR-swizzled by (row >> 1) & 7) is
// the one of the other kernels, so fragment reads are conflict-free ds_read_b128.
//
// Schedule of one K tile t (buffer b), per wave:
//     step A   64 MFMAs of k-step 0            || 16 ds_read_b128: fragments of k-step 1 (buffer b)
//              lgkmcnt(0) + barrier            -> buffer b is consumed by every wave
//     step B1  32 MFMAs of k-step 1            || 16 LDS-DMA: K tile t+2 -> buffer b
//              vmcnt(16) + barrier             -> K tile t+1 (issued one K tile ago) has landed for every wave
//     step B2  32 MFMAs of k-step 1            || 16 ds_read_b128: fragments of k-step 0 of K tile t+1 (buffer b^1)
// Per output element the products are summed in the same order as in every other kernel of this library (K ascending, 32 per
// MFMA 16x16x32): results are bit-identical to gemm_bf16_8ph_kernel / gemm_bf16_nt_kernel.
#pragma once
#include "gemm_bf16_kernels.h"

#define UV4_SCHED(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

template <int EPI, bool F16 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_bf16_4w_kernel(GemmArgs p) {
    constexpr bool TRANS = (EPI == UV_EPI_BF16_T);
    constexpr int BUF = 65536;       // A rows 0..255 | W rows 0..255, 128 B each
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    // ---- tile of this workgroup (XCD-aware walk, as gemm_bf16_8ph_kernel)
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
    const int m0 = (first_m + in_group % gm) * 256, n0 = (in_group / gm) * 256;

    // ---- staging: piece i of an operand = rows 32 i + 8 wave + srow (i = 0..7), 1 KiB per wave-instruction, lane-linear in LDS.
    // The swizzle key (row >> 1) & 7 does not depend on i, so ONE per-lane byte offset serves all pieces; rows past the matrix end
    // are clamped by the descriptor's num_records (reads return 0) - whole tiles only reach this kernel anyway.
    const int srow = lane >> 3, pchunk = lane & 7;
    const int prow = wave * 8 + srow;
    const int pc = pchunk ^ ((prow >> 1) & 7);
    const uint32_t a_voff = (uint32_t)((long)prow * p.lda * 2 + pc * 16);
    const uint32_t w_voff = (uint32_t)((long)prow * p.ldw * 2 + pc * 16);
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (long)m0 * p.lda), 0,
                                                                          (int)min((long)(p.M - m0) * p.lda * 2, 0x7fffffffL), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (long)n0 * p.ldw), 0,
                                                                          (int)min((long)(p.N - n0) * p.ldw * 2, 0x7fffffffL), 0x00020000);
    const uint32_t a_step = (uint32_t)(32 * p.lda * 2), w_step = (uint32_t)(32 * p.ldw * 2);   // bytes between pieces
    char* const my_dst = smem + wave * 1024;
    const int nk = p.K / UV_BK;
#define UV4_STAGE(KT, B)                                                                                                           \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                                                             \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_void*)(my_dst + (B) * BUF + i_ * 4096), 16, a_voff,                     \
                                                 (uint32_t)(KT) * 128u + i_ * a_step, 0, 0);                                        \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_void*)(my_dst + (B) * BUF + 32768 + i_ * 4096), 16, w_voff,             \
                                                 (uint32_t)(KT) * 128u + i_ * w_step, 0, 0);                                        \
    }

    // ---- fragment reads: A row 128 wr + 16 j + frow, W row 128 wc + 16 i + frow; chunk (4 ks + fq) ^ (frow >> 1)
    const int frow = lane & 15, fq = lane >> 4;
    const int lx = (fq ^ (frow >> 1)) << 4;
    const char* const la = smem + (wr * 128 + frow) * 128 + lx;
    const char* const lw = smem + 32768 + (wc * 128 + frow) * 128 + lx;

    f32x4 acc[8][8];  // [i][j]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 af[2][8], wf[2][8];

#define UV4_RD(B, KS)                                                                                        \
    _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) {                                                        \
        af[KS][j_] = *(const bf16x8*)(((KS) ? (const char*)((uintptr_t)la ^ 64) : la) + (B) * BUF + j_ * 2048); \
        wf[KS][j_] = *(const bf16x8*)(((KS) ? (const char*)((uintptr_t)lw ^ 64) : lw) + (B) * BUF + j_ * 2048); \
    }
#define UV4_MFMA(KS, I0, I1)                                                                                  \
    _Pragma("unroll") for (int i_ = (I0); i_ < (I1); ++i_)                                                    \
    _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) {                                                        \
        if (TRANS) acc[i_][j_] = mfma_16x16x32<F16>(af[KS][j_], wf[KS][i_], acc[i_][j_]);                     \
        else acc[i_][j_] = mfma_16x16x32<F16>(wf[KS][i_], af[KS][j_], acc[i_][j_]);                           \
    }
    // one K tile T in buffer B; ST2: K tile T+2 exists (stage it), RD1: K tile T+1 exists (read its first fragments)
#define UV4_KTILE(T, B, ST2, RD1)                                                                             \
    UV4_RD(B, 1)                                                                                              \
    UV4_MFMA(0, 0, 8)                                                                                         \
    _Pragma("unroll") for (int g_ = 0; g_ < 16; ++g_) { UV4_SCHED(0x008, 4); UV4_SCHED(0x100, 1); }           \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                        \
    __builtin_amdgcn_s_barrier();                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    if (ST2) { UV4_STAGE((T) + 2, B) }                                                                        \
    UV4_MFMA(1, 0, 4)                                                                                         \
    if (ST2) { _Pragma("unroll") for (int g_ = 0; g_ < 16; ++g_) { UV4_SCHED(0x008, 2); UV4_SCHED(0x020, 1); } } \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    if (ST2) { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); } else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } \
    __builtin_amdgcn_s_barrier();                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    if (RD1) { UV4_RD((B) ^ 1, 0) }                                                                           \
    UV4_MFMA(1, 4, 8)                                                                                         \
    if (RD1) { _Pragma("unroll") for (int g_ = 0; g_ < 16; ++g_) { UV4_SCHED(0x008, 2); UV4_SCHED(0x100, 1); } } \
    __builtin_amdgcn_sched_barrier(0);

    // prologue: K tiles 0 and 1 in flight; tile 0 landed for everyone; its first fragments read
    UV4_STAGE(0, 0)
    UV4_STAGE(1, 1)
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    UV4_RD(0, 0)

    int t = 0;
    for (; t + 2 < nk; t += 2) {
        UV4_KTILE(t, 0, true, true)
        UV4_KTILE(t + 1, 1, true, true)
    }
    UV4_KTILE(t, 0, false, true)
    UV4_KTILE(t + 1, 1, false, false)

    // ---- epilogue (fragment-wise; the accumulators' layout is the one epi_frag expects)
    if constexpr (EPI == UV_EPI_RESID_F32 || EPI == UV_EPI_GATE_RESID_F32) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int mb[8], nb[8];
            f32x4 av[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                mb[i] = m0 + wr * 128 + j * 16;
                nb[i] = n0 + wc * 128 + i * 16;
                av[i] = acc[i][j];
            }
            epi_rmw_pipe<EPI, 8, 4, F16>(p, mb, nb, av, frow, fq);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                epi_frag<EPI, F16>(p, m0 + wr * 128 + j * 16, n0 + wc * 128 + i * 16, acc[i][j], frow, fq);
    }
#undef UV4_KTILE
#undef UV4_MFMA
#undef UV4_RD
#undef UV4_STAGE
}

template <bool F16 = false>
static int launch_4w(const GemmArgs& a0, int epi, hipStream_t stream) {
    GemmArgs a = a0;
    UV_CHECK_ARG(a.K % 128 == 0 && a.K >= 256, "uv_gemm_bf16_nt: the 4-wave kernel needs K %% 128 == 0 and K >= 256 (K=%d)", a.K);
    UV_CHECK_ARG((long)256 * a.lda * 2 < 0x7fffffffL && (long)256 * a.ldw * 2 < 0x7fffffffL, "uv_gemm_bf16_nt: leading dimension too large for 32-bit tile offsets");
    a.tiles_m = (a.M + 255) / 256;
    a.tiles_n = (a.N + 255) / 256;
    const dim3 grid(a.tiles_m * a.tiles_n), block(256);
    const size_t lds = 128 * 1024;
#define UV_LAUNCH4W(E)                                                                             \
    case E: {                                                                                      \
        auto kern = gemm_bf16_4w_kernel<E, F16>;                                                   \
        UV_ONCE_PER_DEVICE(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));  \
        hipLaunchKernelGGL(kern, grid, block, lds, stream, a);                                     \
        break;                                                                                     \
    }
    switch (epi) {
        UV_LAUNCH4W(UV_EPI_BF16)
        UV_LAUNCH4W(UV_EPI_GELU_BF16)
        UV_LAUNCH4W(UV_EPI_F32_FROM_BF16)
        UV_LAUNCH4W(UV_EPI_RESID_F32)
        UV_LAUNCH4W(UV_EPI_GATE_RESID_F32)
        UV_LAUNCH4W(UV_EPI_BF16_T)
        default:
            uv_set_error("uv_gemm_bf16_nt: unknown epilogue %d", epi);
            return -1;
    }
#undef UV_LAUNCH4W
    UV_CHECK_LAUNCH("uv_gemm_bf16_nt");
    return 0;
}
