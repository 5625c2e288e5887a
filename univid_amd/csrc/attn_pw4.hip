// Flash attention forward for LONG key sequences (head_dim 128, bf16): the 4-wave, one-wave-per-SIMD, 64-queries-per-wave
// structure of the CDNA4 playbook, with the whole 512-entry register file per wave.
//
// Replaces: flash_attention()  models/wan/utils/modules/attention.py:24-130 as called by WanSelfAttention.forward
//           (models/wan/utils/modules/model.py:145-150) - same operand layouts, LDS images, arithmetic, rounding points and
//           per-query operation ORDER as flash_attn_fwd12_kernel / flash_attn_fwd3_kernel in attention.hip: the output is
//           bit-identical to theirs (tested), only the mapping of the work onto the machine differs:
//
//   * a workgroup = 4 waves = 256 queries of one (sample, head); one wave per SIMD; a wave owns TWO 32-query blocks (A, B).
//     Every K / V^T fragment read from LDS feeds two MFMAs (one per block): half the LDS bytes per MFMA of the 32-queries-
//     per-wave kernels.
//   * O^T of both blocks (128 registers) and the Q fragments (64 registers) live in the ACCUMULATOR half of the register
//     file, addressed literally by inline-asm MFMAs (a[0:127] = O, a[128:191] = Q); hipcc allocates only the architectural
//     half (S^T of two tiles, P, fragment queues, softmax state). With builtin MFMAs hipcc moves every accumulator of a
//     512-register kernel into AGPRs and S^T comes back through v_accvgpr_read; the asm MFMAs below write S^T to VGPRs and read
//     Q from AGPRs.
//   * software pipeline over 64-key tiles, two phases of 32 MFMAs per iteration i:
//       phase 1   QK^T of tile i+1 (K fragments from LDS)   beside   exp2 / row sums / bf16 packing of tile i  ("finish")
//       phase 2   P.V of tile i    (V^T fragments from LDS)  beside   row max / deferred-maximum decision / scale-subtract of
//                                                                      tile i+1 ("start")
//     so every MFMA has ~5 vector instructions of the OTHER tile's softmax beside it and a lone wave keeps its SIMD's matrix
//     pipe fed. The order inside a phase is pinned slot by slot (one MFMA + its fillers, __builtin_amdgcn_sched_barrier(0)).
//   * K and V^T tiles arrive by LDS-DMA into two K and two V^T stages (64 KiB); the pieces of tile i+3 (K) / i+2 (V^T) are
//     issued behind the ONE barrier of iteration i (slot 28 of phase 2, after every wave's last LDS read of the stages they
//     overwrite) and are waited for a whole iteration later.
//   * the online-softmax rescale (rare after the first tiles: deferred maximum, UV_ATT_DEFER) is decided branch-free and
//     applied to the AGPR-resident O^T behind a wave-uniform flag at the two points of phase 2 where the reference order has it.
#include "attn_args.h"
#include <type_traits>
#include <utility>

typedef __attribute__((address_space(3))) void lds_void_p;
typedef const __attribute__((address_space(3))) bf16x8* lds_frag_q;

#define UV_ACL_O "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
#define UV_ACL_Q "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191"

// compile-time loop: f(std::integral_constant<int, 0>{}), f(<1>), ... - every index a constant expression (asm "n" operands,
// register-resident arrays)
template <class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
    sfor_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// ---- accumulator-file helpers (register numbers are literal: the compiler never sees these registers). EVERY statement that touches
// the accumulator file lists the whole asm-owned range a[0:191] as clobbered: hipcc parks spilled values in any AGPR it believes free
// between two statements (seen: a128.. overwritten during the P.V phase when only the QK^T statements claimed them).
template <int N> __device__ __forceinline__ void acc_zero_o() { asm volatile("v_accvgpr_write_b32 a%c0, 0" ::"n"(N) : UV_ACL_O, UV_ACL_Q); }
template <int N> __device__ __forceinline__ void acc_write_q(uint32_t v) {
    asm volatile("v_accvgpr_write_b32 a%c1, %0" ::"v"(v), "n"(N) : UV_ACL_O, UV_ACL_Q);
}
template <int N> __device__ __forceinline__ float acc_read(void) {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(v) : "n"(N) : UV_ACL_O, UV_ACL_Q);
    return v;
}
template <int N> __device__ __forceinline__ void acc_write_o(float v) {
    asm volatile("v_accvgpr_write_b32 a%c1, %0" ::"v"(v), "n"(N) : UV_ACL_O, UV_ACL_Q);
}
// single-instruction maxima (as builtins hipcc puts a canonicalising v_max in front of every MFMA output it compares)
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float d;
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// max(m, a, b, c, d) / max(m, a, b, c) as ONE statement (hipcc pads every asm statement with an s_nop)
__device__ __forceinline__ float vmax5(float m, float a, float b, float c, float d) {
    asm volatile("v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %0, %0, %3, %4" : "+v"(m) : "v"(a), "v"(b), "v"(c), "v"(d));
    return m;
}
__device__ __forceinline__ float vmax4(float m, float a, float b, float c) {
    asm volatile("v_max3_f32 %0, %0, %1, %2\n\tv_max_f32 %0, %0, %3" : "+v"(m) : "v"(a), "v"(b), "v"(c));
    return m;
}
__device__ __forceinline__ float vmax(float a, float b) {
    float d;
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// S^T (VGPRs) = / += Kfrag (VGPRs) . Qfrag (AGPRs a[QB:QB+3])
template <int QB, bool FIRST> __device__ __forceinline__ void mfma_qk(f32x16& s, const bf16x8& kf) {
    if constexpr (FIRST)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(s) : "v"(kf), "n"(QB), "n"(QB + 3) : UV_ACL_O, UV_ACL_Q);
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(s) : "v"(kf), "n"(QB), "n"(QB + 3) : UV_ACL_O, UV_ACL_Q);
}
// O^T (AGPRs a[OB:OB+15]) += V^Tfrag (VGPRs) . P^Tfrag (VGPRs)
template <int OB> __device__ __forceinline__ void mfma_pv(const bf16x8& vf, const bf16x8& pf) {
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(vf), "v"(pf), "n"(OB), "n"(OB + 15) : UV_ACL_O, UV_ACL_Q);
}

// One LDS-DMA piece, uniform (SGPR) base + 32-bit lane offset; M0 = LDS byte address of the piece (see attention.hip).
__device__ __forceinline__ void glds16_sb(const char* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

// Empty asm with the value as an in/out operand: everything the value depends on is computed BEFORE this point of the (volatile-
// asm-ordered) instruction stream. hipcc's sinking passes otherwise move the softmax arithmetic of a slot to its first use.
#define UV_PIN(x) asm volatile("" : "+v"(x))

// Diagnostic build only (tools/pw4_diag.hip compiles this file with -DUV_PW4_DIAG; the library never does): in-kernel stamps around
// the main loop and timing-only ablations. Stamp values go to a buffer of their own; no output value is computed from them.
#ifdef UV_PW4_DIAG
__device__ unsigned long long* uv_pw4_dbg;
#ifndef UV_PW4_ABL
#define UV_PW4_ABL 0
#endif
#else
#define UV_PW4_ABL 0
#endif
#define UV_ABL_NO_DMA 1
#define UV_ABL_NO_FIN 2
#define UV_ABL_NO_START 4
#define UV_ABL_NO_LDS 8
#define UV_ABL_NO_BARRIER 16
#define UV_ABL_NO_RESCALE 32

#ifndef UV_PW4_PD
#define UV_PW4_PD 2   // fragment reads issued this many fragments (= 2 MFMAs each) ahead of their first use
#endif

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void flash_attn_pw4_kernel(AttnArgs p) {
    constexpr int KROW = 256, K_BYTES = UV_ATT_KV * KROW, V_BYTES = 128 * 128, V_OFF = 2 * K_BYTES;
    constexpr int PD = UV_PW4_PD;
    __shared__ __attribute__((aligned(16))) char smem[2 * K_BYTES + 2 * V_BYTES];   // K stage 0 | K stage 1 | V^T stage 0 | V^T stage 1

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware block order (as flash_attn_fwd3_kernel): XCD x works through a contiguous range of (sample, head, q-block) ids
    int vb = blockIdx.x;
    {
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
    const int q0w = qb * 256 + wave_u * 64;
    const long hcol = (long)head * 128;
    const float c = p.scale_log2;

    // ---- LDS-DMA pieces of this wave (4 of K = 4 LDS rows each, 4 of V^T = 8 rows each per tile; flash_attn_fwd3_kernel's map)
    unsigned koff, voff;
    {
        const int lrow = wave_u * 4 + (lane >> 4);
        const int cc = (lane & 15) ^ (lrow & 15);
        koff = (unsigned)(perm23(lrow) * (int)p.ldk + cc * 8) * 2u;
        const int drow = wave_u * 8 + (lane >> 3);
        const int cv = (lane & 7) ^ ((drow >> 1) & 7);
        voff = (unsigned)(drow * (int)p.ldvt + cv * 8) * 2u;
    }
    const char* const k0 = (const char*)(p.k + hcol);                 // K tile t: k0 + t * kstep
    const char* const v0 = (const char*)(p.vt + hcol * p.ldvt);       // V^T tile t: v0 + t * 128 bytes
    const long kstep = (long)UV_ATT_KV * p.ldk * 2;
    const long kpiece = 16 * p.ldk * 2, vpiece = 32 * p.ldvt * 2;
    const unsigned smem_a = (unsigned)(uintptr_t)(lds_void_p*)smem;
    const unsigned lds0 = smem_a + wave_u * 1024;
    const int nt = (p.Lk + UV_ATT_KV - 1) / UV_ATT_KV;
    const int nt_full = p.Lk / UV_ATT_KV;
    // piece pi (0..3) of this wave's share of K tile t -> K stage t & 1; FULL: the tile lies inside [0, Lk) for sure
    auto dma_k = [&](int t, int pi, auto full_t) __attribute__((always_inline)) {
        if (decltype(full_t)::value || t < nt_full) {
            glds16_sb(k0 + t * kstep + pi * kpiece, koff, lds0 + (t & 1) * K_BYTES + pi * 4096);
        } else {                                                       // the ragged last tile: clamped key rows
            const int lrow = (pi * 4 + wave_u) * 4 + (lane >> 4);
            const int cc = (lane & 15) ^ (lrow & 15);
            const int kr = min(t * UV_ATT_KV + perm23(lrow), p.Lk - 1);
            const bf16_t* src = p.k + hcol + (long)kr * p.ldk + cc * 8;
            __builtin_amdgcn_global_load_lds(src, (lds_void_p*)(smem + (t & 1) * K_BYTES + (pi * 4 + wave_u) * 1024), 16, 0, 0);
        }
    };
    auto dma_v = [&](int t, int pi) __attribute__((always_inline)) {    // V^T tile t -> V^T stage t & 1
        glds16_sb(v0 + (long)t * (2 * UV_ATT_KV) + pi * vpiece, voff, lds0 + V_OFF + (t & 1) * V_BYTES + pi * 4096);
    };

    // ---- first tiles on their way before anything else: K(0), V^T(0), K(1)
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) dma_k(0, pi, std::true_type{});
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) dma_v(0, pi);
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) dma_k(1, pi, std::true_type{});

    // ---- Q fragments of both blocks -> a[128:191]; O^T = 0 -> a[0:127]
    sfor<2>([&](auto xt) {
        constexpr int X = decltype(xt)::value;
        const int qrow = min(q0w + 32 * X + r, p.Lq - 1);
        const bf16_t* qp = p.q + (long)qrow * p.ldq + hcol + 8 * h;
        u32x4 qv[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) qv[kk] = *(const u32x4*)(qp + 16 * kk);
        sfor<32>([&](auto wt) {
            constexpr int w = decltype(wt)::value;
            acc_write_q<128 + 32 * X + w>(qv[w >> 2][w & 3]);
        });
    });
    sfor<128>([&](auto nt_) { acc_zero_o<decltype(nt_)::value>(); });

    // ---- fragment read addresses (LDS byte addresses; stage / key-half / d-tile offsets are instruction immediates)
    unsigned kaddr[8];
    unsigned vaddr[2][2];
    {
        const int k_key = r & 15, v_key = (r >> 1) & 7;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) kaddr[kk] = smem_a + r * KROW + (((2 * kk + h) ^ k_key) << 4);
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) vaddr[T][s2] = smem_a + V_OFF + r * 128 + (((4 * T + 2 * s2 + h) ^ v_key) << 4);
    }

    // ---- state
    f32x16 S[2][2][2];              // [tile parity][block X][key half T]: S^T, then (in place) S*c - m*c
    u32x4 pf[2][2][2];              // [X][T][s2]: P^T fragments (8 bf16) of the tile in phase 2
    bf16x8 kq[3], vq[3];            // fragment queues
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    float alpha[2][2] = {{1.f, 1.f}, {1.f, 1.f}};      // [X][T] of the tile whose P.V comes next
    bool flag[2][2] = {{false, false}, {false, false}};
    float psum = 0.f;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // O^T[X] *= a (AGPR resident). Rare after the first tiles; generous wait states around the accumulator-file accesses.
    auto rescale = [&](auto xt, float a) __attribute__((always_inline)) {
        constexpr int X = decltype(xt)::value;
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        sfor<64>([&](auto et) {
            constexpr int e = decltype(et)::value;
            const float v = acc_read<64 * X + e>();
            acc_write_o<64 * X + e>(v * a);
        });
        asm volatile("s_nop 7" ::: "memory");
    };

    // One iteration i (PAR = i & 1):  phase 1 = [QK^T(i+1)] beside [finish(i)],  phase 2 = [P.V(i)] beside [start(i+1)].
    // K(i+1) is read from K stage PAR^1, V^T(i) from V^T stage PAR. TAIL: the DMA / prefetch decisions at slot 28 are made at
    // run time (last iterations); MASK: tile i+1 is the ragged last tile.
    auto iter = [&](int i, auto par_t, auto qk_t, auto fin_t, auto pv_t, auto start_t, auto mask_t, auto tail_t) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_t)::value;
        constexpr bool DO_QK = decltype(qk_t)::value, DO_FIN = decltype(fin_t)::value, DO_PV = decltype(pv_t)::value;
        constexpr bool DO_START = decltype(start_t)::value, MASK = decltype(mask_t)::value, TAIL = decltype(tail_t)::value;
        constexpr int CUR = PAR, NXT = PAR ^ 1;
        constexpr int KRS = PAR ^ 1, VRS = PAR;

        // ================= phase 1 =================
        if constexpr (DO_QK && !(DO_FIN || DO_PV)) {
            // the prologue has no preceding slot 28: fragment queue head from scratch
            sfor<PD>([&](auto ft) {
                constexpr int f = decltype(ft)::value;
                kq[f % 3] = *(lds_frag_q)(kaddr[f & 7] + KRS * K_BYTES + (f >> 3) * 32 * KROW);
            });
        }
        sfor<32>([&](auto nt_) {
            constexpr int n = decltype(nt_)::value;
            if constexpr (DO_FIN && !(UV_PW4_ABL & UV_ABL_NO_FIN)) {
                // exp2 / row sum / bf16 of two S values of tile i: block X, half T, elements e0, e0+1 in the reference order
                constexpr int X = (n >> 3) & 1, T = n >> 4, e0 = 2 * (n & 7);
                const float p0 = __builtin_amdgcn_exp2f(S[CUR][X][T][e0]);
                const float p1 = __builtin_amdgcn_exp2f(S[CUR][X][T][e0 + 1]);
                if constexpr ((n & 7) == 0) psum = p0;
                else psum += p0;
                psum += p1;
                uint32_t pw;                     // volatile: pins exp2 / cvt of this slot (hipcc otherwise sinks them to phase 2)
                // (s_nop 0: a v_exp_f32 result needs one wait state before a non-transcendental VALU reads it, and hipcc pads nothing
                // in front of an asm consumer)
                asm volatile("s_nop 0\n\tv_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pw) : "v"(p0), "v"(p1));
                UV_PIN(psum);
                pf[X][T][e0 >> 3][(e0 & 7) >> 1] = pw;
                if constexpr ((n & 7) == 7) {
                    l_run[X] *= alpha[X][T];
                    l_run[X] += psum;
                    UV_PIN(l_run[X]);
                }
            }
            if constexpr (DO_QK) {
                constexpr int X = n & 1, f = n >> 1, kk = f & 7, T = f >> 3;
                if constexpr ((n & 1) == 0 && f + PD < 16 && !(UV_PW4_ABL & UV_ABL_NO_LDS)) {
                    constexpr int g = f + PD;
                    kq[g % 3] = *(lds_frag_q)(kaddr[g & 7] + KRS * K_BYTES + (g >> 3) * 32 * KROW);
                }
                mfma_qk<128 + 32 * X + 4 * kk, kk == 0>(S[NXT][X][T], kq[f % 3]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });

        // ================= phase 2 =================
        float mx[4], mt[4], mneg[4];           // combos cb = X + 2 T
        float alpha_n[2][2];
        bool flag_n[2][2];
        if constexpr (DO_PV) {
            sfor<PD>([&](auto gt) {
                constexpr int g = decltype(gt)::value;
                vq[g % 3] = *(lds_frag_q)(vaddr[g >> 3][(g >> 2) & 1] + VRS * V_BYTES + (g & 3) * 4096);
            });
        }
        sfor<32>([&](auto nt_) {
            constexpr int n = decltype(nt_)::value;
            if constexpr (DO_PV) {
                constexpr int X = n & 1, g = n >> 1, d = g & 3, s2 = (g >> 2) & 1, T = g >> 3;
                if constexpr (n == 0 || n == 16) {
                    // the reference rescales O between the decision of half T and its P.V
                    if constexpr (!(UV_PW4_ABL & UV_ABL_NO_RESCALE)) {
                        if (flag[0][T]) rescale(std::integral_constant<int, 0>{}, alpha[0][T]);
                        if (flag[1][T]) rescale(std::integral_constant<int, 1>{}, alpha[1][T]);
                    }
                }
                if constexpr ((n & 1) == 0 && g + PD < 16 && !(UV_PW4_ABL & UV_ABL_NO_LDS)) {
                    constexpr int g2 = g + PD;
                    vq[g2 % 3] = *(lds_frag_q)(vaddr[g2 >> 3][(g2 >> 2) & 1] + VRS * V_BYTES + (g2 & 3) * 4096);
                }
                mfma_pv<64 * X + 16 * d>(vq[g % 3], __builtin_bit_cast(bf16x8, pf[X][T][s2]));
            }
            if constexpr (DO_START && !(UV_PW4_ABL & UV_ABL_NO_START)) {
                // ---- row maxima of tile i+1: slots 0-3 the two T = 0 chains, 4-7 the two T = 1 chains (their last MFMA is recent)
                if constexpr (n < 8) {
                    constexpr int T = n >> 2, q = n & 3;            // chain ops 2q, 2q+1 of both blocks
                    sfor<2>([&](auto xt) {
                        constexpr int X = decltype(xt)::value, cb = X + 2 * T;
                        f32x16& s = S[NXT][X][T];
                        if constexpr (MASK && q == 0) {
                            const int kv0 = (i + 1) * UV_ATT_KV;
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const int ki = 32 * T + (e & 3) + 8 * (e >> 2) + 4 * h;
                                if (kv0 + perm23(ki) >= p.Lk) s[e] = -INFINITY;
                            }
                        }
                        if constexpr (q == 0) mx[cb] = vmax5(s[0], s[1], s[2], s[3], s[4]);
                        else if constexpr (q < 3) mx[cb] = vmax5(mx[cb], s[4 * q + 1], s[4 * q + 2], s[4 * q + 3], s[4 * q + 4]);
                        else mx[cb] = vmax4(mx[cb], s[13], s[14], s[15]);
                    });
                }
                // ---- the other half-wave's maximum
                if constexpr (n == 8 || n == 9) {
                    sfor<2>([&](auto xt) {
                        constexpr int cb = decltype(xt)::value + 2 * (n - 8);
                        const unsigned u = __builtin_bit_cast(unsigned, mx[cb]);
                        const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                        mt[cb] = vmax(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
                    });
                }
                // ---- deferred-maximum decisions, branch-free: (A,0) (B,0) (A,1) (B,1) in slots 10 .. 13
                if constexpr (n >= 10 && n < 14) {
                    constexpr int cb = n - 10, X = cb & 1, T = cb >> 1;
                    const float grow = (mt[cb] - m_run[X]) * c;
                    const bool cond = __any(grow > UV_ATT_DEFER);
                    const float m_new = vmax(m_run[X], mt[cb]);
                    const float a = __builtin_amdgcn_exp2f((m_run[X] - m_new) * c);
                    alpha_n[X][T] = cond ? a : 1.0f;
                    m_run[X] = cond ? m_new : m_run[X];
                    flag_n[X][T] = cond;
                    mneg[cb] = -m_run[X] * c;
                    UV_PIN(alpha_n[X][T]);
                    UV_PIN(m_run[X]);
                    UV_PIN(mneg[cb]);
                }
                // ---- S*c - m*c in place: (A,0) slots 12-15, (B,0) 16-19, (A,1) 20-23, (B,1) 24-27
                if constexpr (n >= 12 && n < 28) {
                    constexpr int cb = (n - 12) >> 2, X = cb & 1, T = cb >> 1, e0 = 4 * ((n - 12) & 3);
                    f32x16& s = S[NXT][X][T];
#pragma unroll
                    for (int e = e0; e < e0 + 4; ++e) s[e] = __builtin_fmaf(s[e], c, mneg[cb]);
                    UV_PIN(s);
                }
            }
            if constexpr (n == 28 && (DO_QK || DO_START)) {
                // every wave's reads of K stage PAR^1 (phase 1) and V^T stage PAR (issued by slot 26) are complete; its own
                // pieces of K(i+2) / V^T(i+1) (issued an iteration ago) have landed
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                if constexpr (!(UV_PW4_ABL & UV_ABL_NO_BARRIER)) __builtin_amdgcn_s_barrier();
                // head of the next iteration's K fragment queue: K(i+2) from K stage PAR
                if (!TAIL || i + 2 < nt) {
                    sfor<PD>([&](auto ft) {
                        constexpr int f = decltype(ft)::value;
                        kq[f % 3] = *(lds_frag_q)(kaddr[f & 7] + PAR * K_BYTES + (f >> 3) * 32 * KROW);
                    });
                }
            }
            if constexpr (n >= 28 && (DO_QK || DO_START)) {
                // behind the barrier: this wave's pieces of K(i+3) -> K stage PAR^1 (slots 28, 29) and V^T(i+2) -> V^T stage PAR (30, 31)
                constexpr int pi0 = 2 * (n & 1);
                if constexpr (UV_PW4_ABL & UV_ABL_NO_DMA) {
                } else if constexpr (n < 30) {
                    if (!TAIL || i + 3 < nt) {
                        dma_k(i + 3, pi0, std::integral_constant<bool, !TAIL>{});
                        dma_k(i + 3, pi0 + 1, std::integral_constant<bool, !TAIL>{});
                    }
                } else {
                    if (!TAIL || i + 2 < nt) {
                        dma_v(i + 2, pi0);
                        dma_v(i + 2, pi0 + 1);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (DO_START) {
#pragma unroll
            for (int X = 0; X < 2; ++X)
#pragma unroll
                for (int T = 0; T < 2; ++T) {
                    alpha[X][T] = alpha_n[X][T];
                    flag[X][T] = flag_n[X][T];
                }
        }
    };

    using F = std::false_type;
    using T_ = std::true_type;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    //        i   PAR   QK    FIN   PV    START MASK  TAIL
    iter(-1, P1{}, T_{}, F{}, F{}, T_{}, F{}, F{});                   // prologue: QK^T(0), start(0); DMA K(2), V^T(1)
    int i = 0;
#ifdef UV_PW4_DIAG
    const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    for (; i + 4 < nt_full; i += 2) {
        iter(i, P0{}, T_{}, T_{}, T_{}, T_{}, F{}, F{});
        iter(i + 1, P1{}, T_{}, T_{}, T_{}, T_{}, F{}, F{});
    }
#ifdef UV_PW4_DIAG
    if (lane == 0 && uv_pw4_dbg) {
        unsigned long long* d = uv_pw4_dbg + ((long)blockIdx.x * 4 + wave_u) * 4;
        d[0] = __builtin_amdgcn_s_memtime() - st0;
        d[1] = __builtin_amdgcn_s_memrealtime() - sr0;
        d[2] = (unsigned long long)i;
    }
#endif
    for (; i < nt - 1; ++i) {
        const bool masked = (i + 1 == nt - 1) && nt_full < nt;
        if (i & 1) {
            if (masked) iter(i, P1{}, T_{}, T_{}, T_{}, T_{}, T_{}, T_{});
            else iter(i, P1{}, T_{}, T_{}, T_{}, T_{}, F{}, T_{});
        } else {
            if (masked) iter(i, P0{}, T_{}, T_{}, T_{}, T_{}, T_{}, T_{});
            else iter(i, P0{}, T_{}, T_{}, T_{}, T_{}, F{}, T_{});
        }
    }
    if (i & 1) iter(i, P1{}, F{}, T_{}, T_{}, F{}, F{}, T_{});        // i = nt - 1: finish + P.V of the last tile
    else iter(i, P0{}, F{}, T_{}, T_{}, F{}, F{}, T_{});

    // ---- finish: combine the two half-wave sums, normalise, store bf16 rows
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    sfor<2>([&](auto xt) {
        constexpr int X = decltype(xt)::value;
        const float l_tot = l_run[X] + __shfl_xor(l_run[X], 32, 64);
        const float inv = 1.0f / l_tot;
        const int q = q0w + 32 * X + r;
        bf16_t* op = p.out + (long)q * p.ldo + hcol + 4 * h;
        sfor<16>([&](auto gt) {
            constexpr int dg = decltype(gt)::value, d = dg >> 2, g = dg & 3;
            const float o0 = acc_read<64 * X + 16 * d + 4 * g + 0>();
            const float o1 = acc_read<64 * X + 16 * d + 4 * g + 1>();
            const float o2 = acc_read<64 * X + 16 * d + 4 * g + 2>();
            const float o3 = acc_read<64 * X + 16 * d + 4 * g + 3>();
            u32x2 o = {pack16_2<false>(o0 * inv, o1 * inv), pack16_2<false>(o2 * inv, o3 * inv)};
            if (q < p.Lq) *(u32x2*)(op + 32 * d + 8 * g) = o;
        });
    });
}

int uv_launch_attn_pw4(const AttnArgs& a0, hipStream_t st) {
    AttnArgs a = a0;
    a.q_blocks = (a.Lq + 255) / 256;
    a.n12 = 0;
    hipLaunchKernelGGL(flash_attn_pw4_kernel, dim3(a.q_blocks * a.H * a.batch), dim3(256), 0, st, a);
    return 0;
}
