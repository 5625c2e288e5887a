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
// ---- softmax fillers as ONE asm statement per slot. hipcc pads every asm statement whose outputs the next vector instruction touches
// with an s_nop and knows nothing of the hazards inside; written out, a slot's filler has no pad and its only hazard (a v_exp_f32
// result needs one wait state before a non-transcendental VALU reads it) is covered by the instruction order:
//     exp p0 ; exp p1 ; psum (+)= p0 ; pw = cvt_pk(p0, p1) ; psum += p1   [; l = l * alpha ; l = l + psum]
// MODE 0: first pair of a (block, half): psum = p0 (the reference starts from 0 + p0, the same value); 2: last pair, folds the
// row-sum update l = l * alpha + psum (two roundings, as the reference's `l *= alpha; l += psum`).
template <int MODE>
__device__ __forceinline__ uint32_t fin_pair(float s0, float s1, float& psum, float& l, float alpha) {
    uint32_t pw;
    float p0, p1;
    if constexpr (MODE == 0)
        asm volatile("v_exp_f32 %0, %4\n\tv_exp_f32 %1, %5\n\tv_mov_b32 %2, %0\n\tv_cvt_pk_bf16_f32 %3, %0, %1\n\tv_add_f32 %2, %2, %1"
                     : "=&v"(p0), "=&v"(p1), "=&v"(psum), "=&v"(pw) : "v"(s0), "v"(s1));
    else if constexpr (MODE == 1)
        asm volatile("v_exp_f32 %0, %4\n\tv_exp_f32 %1, %5\n\tv_add_f32 %2, %2, %0\n\tv_cvt_pk_bf16_f32 %3, %0, %1\n\tv_add_f32 %2, %2, %1"
                     : "=&v"(p0), "=&v"(p1), "+v"(psum), "=&v"(pw) : "v"(s0), "v"(s1));
    else
        asm volatile("v_exp_f32 %0, %5\n\tv_exp_f32 %1, %6\n\tv_add_f32 %2, %2, %0\n\tv_cvt_pk_bf16_f32 %3, %0, %1\n\tv_add_f32 %2, %2, %1\n\t"
                     "v_mul_f32 %4, %4, %7\n\tv_add_f32 %4, %4, %2"
                     : "=&v"(p0), "=&v"(p1), "+v"(psum), "=&v"(pw), "+v"(l) : "v"(s0), "v"(s1), "v"(alpha));
    return pw;
}
// row-maximum chains of both blocks, two ops each per slot: Q = 0 starts them (5 values), 1 / 2 continue (4 values), 3 ends (3 values)
template <int Q>
__device__ __forceinline__ void max_step(float& ma, float& mb, const f32x16& a, const f32x16& b) {
    if constexpr (Q == 0)
        asm volatile("v_max3_f32 %0, %2, %3, %4\n\tv_max3_f32 %1, %7, %8, %9\n\tv_max3_f32 %0, %0, %5, %6\n\tv_max3_f32 %1, %1, %10, %11"
                     : "=&v"(ma), "=&v"(mb)
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]));
    else if constexpr (Q < 3)
        asm volatile("v_max3_f32 %0, %0, %2, %3\n\tv_max3_f32 %1, %1, %6, %7\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %1, %1, %8, %9"
                     : "+v"(ma), "+v"(mb)
                     : "v"(a[4 * Q + 1]), "v"(a[4 * Q + 2]), "v"(a[4 * Q + 3]), "v"(a[4 * Q + 4]), "v"(b[4 * Q + 1]), "v"(b[4 * Q + 2]),
                       "v"(b[4 * Q + 3]), "v"(b[4 * Q + 4]));
    else
        asm volatile("v_max3_f32 %0, %0, %2, %3\n\tv_max3_f32 %1, %1, %5, %6\n\tv_max_f32 %0, %0, %4\n\tv_max_f32 %1, %1, %7"
                     : "+v"(ma), "+v"(mb) : "v"(a[13]), "v"(a[14]), "v"(a[15]), "v"(b[13]), "v"(b[14]), "v"(b[15]));
}
// the other half-wave's maximum for two chains: copy, v_permlane32_swap (lanes 32-63 of the first operand <-> lanes 0-31 of the
// second: afterwards one register holds the low half-wave's value in both halves, the other the high one's), maximum. The swap reads
// registers a VALU wrote: two wait states (the second v_mov and the s_nop serve the first pair, the s_nop and the first swap the second).
__device__ __forceinline__ void max_xhalf(float& ma, float& mb) {
    float ta, tb;
    asm volatile("v_mov_b32 %2, %0\n\tv_mov_b32 %3, %1\n\ts_nop 0\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\t"
                 "v_max_f32 %0, %0, %2\n\tv_max_f32 %1, %1, %3"
                 : "+v"(ma), "+v"(mb), "=&v"(ta), "=&v"(tb));
}
// s = s * c + mneg on four accumulator registers (in place); a macro: vector elements cannot bind to references
#define UV_FMA4(S_, E0, C_, MNEG)                                                                                                   \
    asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5" \
                 : "+v"(S_[E0]), "+v"(S_[E0 + 1]), "+v"(S_[E0 + 2]), "+v"(S_[E0 + 3]) : "s"(C_), "v"(MNEG))
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


// ---- one asm statement per slot of the steady-state loop: the MFMA first, then its fillers (hipcc puts an s_nop between two adjacent
// asm statements; a slot written as one statement has none). Same instructions and order as the separate helpers above.
#define UV_ACL_ALL UV_ACL_O, UV_ACL_Q
// QK^T MFMA + finish of one pair, SOFTWARE-PIPELINED over the slots: the slot consumes the exponentials (p0c, p1c) the PREVIOUS slot issued
// and issues the next slot's (p0n, p1n). A v_exp_f32 result takes ~30 cycles to arrive; consumed in the slot that issued it, the lone
// wave stood still for that long in every slot (measured: 55 cycles per slot for 36 cycles of issue).
//   MODE 0: first pair of a (block, half): psum = p0;  1: psum += p0;  2: last pair, also l = l * alpha + psum.  LAST: slot 31, no next pair.
template <int MODE, int QB, bool FIRST, bool LAST>
__device__ __forceinline__ uint32_t qk_fin(f32x16& sn, const bf16x8& kf, float& p0c, float& p1c, float s0n, float s1n, float& psum, float& l,
                                           float alpha) {
    uint32_t pw;
    float p0n = 0.f, p1n = 0.f;
#define UV_QKF_BODY(SUM0) "v_exp_f32 %0, %7\n\tv_exp_f32 %1, %8\n\t" SUM0 "\n\tv_cvt_pk_bf16_f32 %3, %5, %6\n\tv_add_f32 %2, %2, %6"
    if constexpr (LAST) {
        static_assert(MODE == 2 && !FIRST, "slot 31 ends a (block, half)");
        asm volatile("v_mfma_f32_32x32x16_bf16 %2, %7, a[%c8:%c9], %2\n\tv_add_f32 %0, %0, %4\n\tv_cvt_pk_bf16_f32 %1, %4, %5\n\tv_add_f32 %0, %0, %5\n\t"
                     "v_mul_f32 %3, %3, %6\n\tv_add_f32 %3, %3, %0"
                     : "+v"(psum), "=&v"(pw), "+v"(sn), "+v"(l) : "v"(p0c), "v"(p1c), "v"(alpha), "v"(kf), "n"(QB), "n"(QB + 3) : UV_ACL_ALL);
    } else if constexpr (MODE == 2) {
        static_assert(!FIRST, "a chain's first MFMA never coincides with the last pair of a (block, half)");
        asm volatile("v_mfma_f32_32x32x16_bf16 %4, %11, a[%c12:%c13], %4\n\tv_exp_f32 %0, %8\n\tv_exp_f32 %1, %9\n\tv_add_f32 %2, %2, %6\n\t"
                     "v_cvt_pk_bf16_f32 %3, %6, %7\n\tv_add_f32 %2, %2, %7\n\tv_mul_f32 %5, %5, %10\n\tv_add_f32 %5, %5, %2"
                     : "=&v"(p0n), "=&v"(p1n), "+v"(psum), "=&v"(pw), "+v"(sn), "+v"(l)
                     : "v"(p0c), "v"(p1c), "v"(s0n), "v"(s1n), "v"(alpha), "v"(kf), "n"(QB), "n"(QB + 3) : UV_ACL_ALL);
    } else if constexpr (FIRST && MODE == 0) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %4, %9, a[%c10:%c11], 0\n\t" UV_QKF_BODY("v_mov_b32 %2, %5")
                     : "=&v"(p0n), "=&v"(p1n), "=&v"(psum), "=&v"(pw), "=&v"(sn)
                     : "v"(p0c), "v"(p1c), "v"(s0n), "v"(s1n), "v"(kf), "n"(QB), "n"(QB + 3) : UV_ACL_ALL);
    } else if constexpr (FIRST) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %4, %9, a[%c10:%c11], 0\n\t" UV_QKF_BODY("v_add_f32 %2, %2, %5")
                     : "=&v"(p0n), "=&v"(p1n), "+v"(psum), "=&v"(pw), "=&v"(sn)
                     : "v"(p0c), "v"(p1c), "v"(s0n), "v"(s1n), "v"(kf), "n"(QB), "n"(QB + 3) : UV_ACL_ALL);
    } else if constexpr (MODE == 0) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %4, %9, a[%c10:%c11], %4\n\t" UV_QKF_BODY("v_mov_b32 %2, %5")
                     : "=&v"(p0n), "=&v"(p1n), "=&v"(psum), "=&v"(pw), "+v"(sn)
                     : "v"(p0c), "v"(p1c), "v"(s0n), "v"(s1n), "v"(kf), "n"(QB), "n"(QB + 3) : UV_ACL_ALL);
    } else {
        asm volatile("v_mfma_f32_32x32x16_bf16 %4, %9, a[%c10:%c11], %4\n\t" UV_QKF_BODY("v_add_f32 %2, %2, %5")
                     : "=&v"(p0n), "=&v"(p1n), "+v"(psum), "=&v"(pw), "+v"(sn)
                     : "v"(p0c), "v"(p1c), "v"(s0n), "v"(s1n), "v"(kf), "n"(QB), "n"(QB + 3) : UV_ACL_ALL);
    }
#undef UV_QKF_BODY
    p0c = p0n;
    p1c = p1n;
    return pw;
}
// the exponentials of the NEXT iteration's first pair, issued with the last P.V MFMA of phase 2 (or alone, in the prologue)
__device__ __forceinline__ void exp_pair(float& p0c, float& p1c, float s0, float s1) {
    asm volatile("v_exp_f32 %0, %2\n\tv_exp_f32 %1, %3" : "=&v"(p0c), "=&v"(p1c) : "v"(s0), "v"(s1));
}
template <int OB>
__device__ __forceinline__ void pv_exp_pair(const bf16x8& vf, const bf16x8& pf, float& p0c, float& p1c, float s0, float s1) {
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c6:%c7], %4, %5, a[%c6:%c7]\n\tv_exp_f32 %0, %2\n\tv_exp_f32 %1, %3"
                 : "=&v"(p0c), "=&v"(p1c) : "v"(s0), "v"(s1), "v"(vf), "v"(pf), "n"(OB), "n"(OB + 15) : UV_ACL_ALL);
}
#define UV_PV_TXT(VF, PF, LO, HI) "v_mfma_f32_32x32x16_bf16 a[%c" #LO ":%c" #HI "], %" #VF ", %" #PF ", a[%c" #LO ":%c" #HI "]\n\t"
template <int OB, int Q>
__device__ __forceinline__ void pv_max(const bf16x8& vf, const bf16x8& pf, float& ma, float& mb, const f32x16& a, const f32x16& b) {
    if constexpr (Q == 0)
        asm volatile(UV_PV_TXT(12, 13, 14, 15)
                     "v_max3_f32 %0, %2, %3, %4\n\tv_max3_f32 %1, %7, %8, %9\n\tv_max3_f32 %0, %0, %5, %6\n\tv_max3_f32 %1, %1, %10, %11"
                     : "=&v"(ma), "=&v"(mb)
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]),
                       "v"(vf), "v"(pf), "n"(OB), "n"(OB + 15) : UV_ACL_ALL);
    else if constexpr (Q < 3)
        asm volatile(UV_PV_TXT(10, 11, 12, 13)
                     "v_max3_f32 %0, %0, %2, %3\n\tv_max3_f32 %1, %1, %6, %7\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %1, %1, %8, %9"
                     : "+v"(ma), "+v"(mb)
                     : "v"(a[4 * Q + 1]), "v"(a[4 * Q + 2]), "v"(a[4 * Q + 3]), "v"(a[4 * Q + 4]), "v"(b[4 * Q + 1]), "v"(b[4 * Q + 2]),
                       "v"(b[4 * Q + 3]), "v"(b[4 * Q + 4]), "v"(vf), "v"(pf), "n"(OB), "n"(OB + 15) : UV_ACL_ALL);
    else
        asm volatile(UV_PV_TXT(8, 9, 10, 11)
                     "v_max3_f32 %0, %0, %2, %3\n\tv_max3_f32 %1, %1, %5, %6\n\tv_max_f32 %0, %0, %4\n\tv_max_f32 %1, %1, %7"
                     : "+v"(ma), "+v"(mb) : "v"(a[13]), "v"(a[14]), "v"(a[15]), "v"(b[13]), "v"(b[14]), "v"(b[15]),
                       "v"(vf), "v"(pf), "n"(OB), "n"(OB + 15) : UV_ACL_ALL);
}
template <int OB>
__device__ __forceinline__ void pv_xhalf(const bf16x8& vf, const bf16x8& pf, float& ma, float& mb) {
    float ta, tb;
    asm volatile(UV_PV_TXT(4, 5, 6, 7)
                 "v_mov_b32 %2, %0\n\tv_mov_b32 %3, %1\n\ts_nop 0\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\t"
                 "v_max_f32 %0, %0, %2\n\tv_max_f32 %1, %1, %3"
                 : "+v"(ma), "+v"(mb), "=&v"(ta), "=&v"(tb) : "v"(vf), "v"(pf), "n"(OB), "n"(OB + 15) : UV_ACL_ALL);
}
#define UV_PV_FMA4(OB, VF, PF, S_, E0, C_, MNEG)                                                                                    \
    asm volatile(UV_PV_TXT(6, 7, 8, 9)                                                                                              \
                 "v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"       \
                 : "+v"(S_[E0]), "+v"(S_[E0 + 1]), "+v"(S_[E0 + 2]), "+v"(S_[E0 + 3])                                                 \
                 : "s"(C_), "v"(MNEG), "v"(VF), "v"(PF), "n"(OB), "n"((OB) + 15) : UV_ACL_ALL)

// One LDS-DMA piece, uniform (SGPR) base + 32-bit lane offset; M0 = LDS byte address of the piece. M0 is declared clobbered instead of
// saved and restored (hipcc warns that it is a reserved register; nothing else in this kernel keeps a value in it): three
// instructions per piece.
__device__ __forceinline__ void glds16_sb(const char* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
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
    unsigned koff[4], voff[4];       // lane offset of piece pi from the tile's base (piece stride folded in: no scalar address add per piece)
    {
        const int lrow = wave_u * 4 + (lane >> 4);
        const int cc = (lane & 15) ^ (lrow & 15);
        const int drow = wave_u * 8 + (lane >> 3);
        const int cv = (lane & 7) ^ ((drow >> 1) & 7);
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) {
            koff[pi] = (unsigned)((perm23(lrow) + 16 * pi) * (int)p.ldk + cc * 8) * 2u;
            voff[pi] = (unsigned)((drow + 32 * pi) * (int)p.ldvt + cv * 8) * 2u;
        }
    }
    const char* const k0 = (const char*)(p.k + hcol);                 // K tile t: k0 + t * kstep
    const char* const v0 = (const char*)(p.vt + hcol * p.ldvt);       // V^T tile t: v0 + t * 128 bytes
    const long kstep = (long)UV_ATT_KV * p.ldk * 2;
    const unsigned smem_a = (unsigned)(uintptr_t)(lds_void_p*)smem;
    const unsigned lds0 = smem_a + wave_u * 1024;
    const int nt = (p.Lk + UV_ATT_KV - 1) / UV_ATT_KV;
    const int nt_full = p.Lk / UV_ATT_KV;
    // piece pi (0..3) of this wave's share of K tile t -> K stage t & 1; FULL: the tile lies inside [0, Lk) for sure
    auto dma_k = [&](int t, int pi, auto full_t) __attribute__((always_inline)) {
        if (decltype(full_t)::value || t < nt_full) {
            glds16_sb(k0 + t * kstep, koff[pi], lds0 + (t & 1) * K_BYTES + pi * 4096);
        } else {                                                       // the ragged last tile: clamped key rows
            const int lrow = (pi * 4 + wave_u) * 4 + (lane >> 4);
            const int cc = (lane & 15) ^ (lrow & 15);
            const int kr = min(t * UV_ATT_KV + perm23(lrow), p.Lk - 1);
            const bf16_t* src = p.k + hcol + (long)kr * p.ldk + cc * 8;
            __builtin_amdgcn_global_load_lds(src, (lds_void_p*)(smem + (t & 1) * K_BYTES + (pi * 4 + wave_u) * 1024), 16, 0, 0);
        }
    };
    auto dma_v = [&](int t, int pi) __attribute__((always_inline)) {    // V^T tile t -> V^T stage t & 1
        glds16_sb(v0 + (long)t * (2 * UV_ATT_KV), voff[pi], lds0 + V_OFF + (t & 1) * V_BYTES + pi * 4096);
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
    bf16x8 kq[3], vq[3];            // fragment queues: ONE ds_read_b128 straight into the register tuple the MFMA reads (assembled from two
                                    // ds_read_b64, hipcc copies the halves together with v_mov right in front of the asm MFMA, which
                                    // then reads stale registers: a VALU write needs wait states before an MFMA reads it)
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    float mneg_run[2] = {0.f, 0.f};                    // -m_run * c (set with the first tile's decision, which always moves m)
    float alpha[2][2] = {{1.f, 1.f}, {1.f, 1.f}};      // [X][T] of the tile whose P.V comes next
    bool flag[2][2] = {{false, false}, {false, false}};
    float psum = 0.f;
    float p0c = 0.f, p1c = 0.f;     // exponentials of the pair the next finish slot consumes (issued one slot earlier)

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

    // K fragment f (key half f >> 3, k-step f & 7) of K stage ks / V^T fragment g (key half, s2, d tile) of V^T stage vs
    auto ld_k = [&](bf16x8& dst, int f, int ks) __attribute__((always_inline)) {
        dst = *(lds_frag_q)(kaddr[f & 7] + ks * K_BYTES + (f >> 3) * 32 * KROW);
    };
    auto ld_v = [&](bf16x8& dst, int g, int vs) __attribute__((always_inline)) {
        dst = *(lds_frag_q)(vaddr[g >> 3][(g >> 2) & 1] + vs * V_BYTES + (g & 3) * 4096);
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
                ld_k(kq[f % 3], f, KRS);
            });
        }
        sfor<32>([&](auto nt_) {
            constexpr int n = decltype(nt_)::value;
            constexpr bool FIN = DO_FIN && !(UV_PW4_ABL & UV_ABL_NO_FIN);
            // finish: exp2 / row sum / bf16 of two S values of tile i: block FX, half FT, elements e0, e0+1 in the reference order
            constexpr int FX = (n >> 3) & 1, FT = n >> 4, e0 = 2 * (n & 7), MODE = (n & 7) == 0 ? 0 : ((n & 7) == 7 ? 2 : 1);
            // QK^T: block X, fragment f = (half T, k-step kk)
            constexpr int X = n & 1, f = n >> 1, kk = f & 7, T = f >> 3;
            if constexpr (DO_QK && (n & 1) == 0 && f + PD < 16 && !(UV_PW4_ABL & UV_ABL_NO_LDS)) ld_k(kq[(f + PD) % 3], f + PD, KRS);
            if constexpr (DO_QK && FIN) {
                // elements of the NEXT slot's pair (its exponentials are issued here)
                constexpr int n1 = n < 31 ? n + 1 : 31, NX = (n1 >> 3) & 1, NT_ = n1 >> 4, ne0 = 2 * (n1 & 7);
                pf[FX][FT][e0 >> 3][(e0 & 7) >> 1] = qk_fin<MODE, 128 + 32 * X + 4 * kk, kk == 0, n == 31>(
                    S[NXT][X][T], kq[f % 3], p0c, p1c, S[CUR][NX][NT_][ne0], S[CUR][NX][NT_][ne0 + 1], psum, l_run[FX], alpha[FX][FT]);
            }
            else if constexpr (DO_QK)
                mfma_qk<128 + 32 * X + 4 * kk, kk == 0>(S[NXT][X][T], kq[f % 3]);
            else if constexpr (FIN)
                pf[FX][FT][e0 >> 3][(e0 & 7) >> 1] = fin_pair<MODE>(S[CUR][FX][FT][e0], S[CUR][FX][FT][e0 + 1], psum, l_run[FX], alpha[FX][FT]);
            __builtin_amdgcn_sched_barrier(0);
        });

        // ================= phase 2 =================
        float mx[4], mneg_t[4];                // combos cb = X + 2 T
        float alpha_n[2][2];
        bool flag_n[2][2];
        if constexpr (DO_PV) {
            sfor<PD>([&](auto gt) {
                constexpr int g = decltype(gt)::value;
                ld_v(vq[g % 3], g, VRS);
            });
        }
        sfor<32>([&](auto nt_) {
            constexpr int n = decltype(nt_)::value;
            constexpr bool START = DO_START && !(UV_PW4_ABL & UV_ABL_NO_START);
            constexpr int X = n & 1, g = n >> 1, d = g & 3, s2 = (g >> 2) & 1, T = g >> 3, OB = 64 * X + 16 * d;     // P.V of this slot
            if constexpr (DO_PV) {
                if constexpr ((n == 0 || n == 16) && !(UV_PW4_ABL & UV_ABL_NO_RESCALE)) {
                    // the reference rescales O between the decision of half T and its P.V
                    if (flag[0][T]) rescale(std::integral_constant<int, 0>{}, alpha[0][T]);
                    if (flag[1][T]) rescale(std::integral_constant<int, 1>{}, alpha[1][T]);
                }
                if constexpr ((n & 1) == 0 && g + PD < 16 && !(UV_PW4_ABL & UV_ABL_NO_LDS)) ld_v(vq[(g + PD) % 3], g + PD, VRS);
            }
            const bf16x8 pfr = __builtin_bit_cast(bf16x8, pf[X][T][s2]), vfr = vq[g % 3];
            // start of tile i+1 beside it. Row maxima: slots 0-3 the two T = 0 chains, 4-7 the two T = 1 chains (their last MFMA is
            // recent); 8, 9: the other half-wave's maximum (combos cb = X + 2 T); 12-27: S*c - m*c in place, (A,0) (B,0) (A,1) (B,1)
            if constexpr (START && n < 8) {
                constexpr int MT = n >> 2, q = n & 3;            // chain ops 2q, 2q+1 of both blocks
                if (q == 0 && MASK && (i + 1) * UV_ATT_KV + UV_ATT_KV > p.Lk) {      // tile i+1 is the ragged last tile (wave-uniform, rare)
                    const int kv0 = (i + 1) * UV_ATT_KV;
                    sfor<2>([&](auto xt) {
                        f32x16& sm = S[NXT][decltype(xt)::value][MT];
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int ki = 32 * MT + (e & 3) + 8 * (e >> 2) + 4 * h;
                            if (kv0 + perm23(ki) >= p.Lk) sm[e] = -INFINITY;
                        }
                    });
                }
                if constexpr (DO_PV) pv_max<OB, q>(vfr, pfr, mx[2 * MT], mx[2 * MT + 1], S[NXT][0][MT], S[NXT][1][MT]);
                else max_step<q>(mx[2 * MT], mx[2 * MT + 1], S[NXT][0][MT], S[NXT][1][MT]);
            } else if constexpr (START && (n == 8 || n == 9)) {
                if constexpr (DO_PV) pv_xhalf<OB>(vfr, pfr, mx[2 * (n - 8)], mx[2 * (n - 8) + 1]);
                else max_xhalf(mx[2 * (n - 8)], mx[2 * (n - 8) + 1]);
            } else if constexpr (START && n >= 14 && n < 28) {
                constexpr int cb = (n - 12) >> 2, FX = cb & 1, FT = cb >> 1, e0 = 4 * ((n - 12) & 3);
                f32x16& sm = S[NXT][FX][FT];
                if constexpr (DO_PV) UV_PV_FMA4(OB, vfr, pfr, sm, e0, c, mneg_t[cb]);
                else UV_FMA4(sm, e0, c, mneg_t[cb]);
            } else if constexpr (START && n == 31) {
                // first pair of the NEXT iteration's finish phase: S[NXT] (A, half 0) was scaled in slots 12-15
                if constexpr (DO_PV) pv_exp_pair<OB>(vfr, pfr, p0c, p1c, S[NXT][0][0][0], S[NXT][0][0][1]);
                else exp_pair(p0c, p1c, S[NXT][0][0][0], S[NXT][0][0][1]);
            } else if constexpr (DO_PV) {
                mfma_pv<OB>(vfr, pfr);
            }
            if constexpr (START && n >= 10 && n < 14) {
                // deferred-maximum decisions (A,0) (B,0) (A,1) (B,1). Common case: nothing moves (alpha = 1, m and -m*c stay); the
                // update runs behind a wave-uniform branch. Slots 12, 13 also carry the first two S*c - m*c groups of (A,0).
                constexpr int cb = n - 10, DX = cb & 1, DT = cb >> 1;
                const float grow = (mx[cb] - m_run[DX]) * c;
                const bool cond = __any(grow > UV_ATT_DEFER);
                alpha_n[DX][DT] = 1.0f;
                flag_n[DX][DT] = cond;
                if (cond) {
                    const float m_new = vmax(m_run[DX], mx[cb]);
                    alpha_n[DX][DT] = __builtin_amdgcn_exp2f((m_run[DX] - m_new) * c);
                    m_run[DX] = m_new;
                    mneg_run[DX] = -m_new * c;
                }
                mneg_t[cb] = mneg_run[DX];
                UV_PIN(mneg_t[cb]);
                if constexpr (n >= 12) {
                    constexpr int e0 = 4 * (n - 12);
                    f32x16& sm = S[NXT][0][0];
                    UV_FMA4(sm, e0, c, mneg_t[0]);
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
                        ld_k(kq[f % 3], f, PAR);
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
    // steady state: ONE instantiation per tile parity serves every iteration (the DMA / prefetch / mask decisions of the last
    // iterations are wave-uniform run-time branches; separate tail instantiations made hipcc spill into AGPRs around them)
    int i = 0;
#ifdef UV_PW4_DIAG
    const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    for (;;) {
        if (i >= nt - 1) break;
        iter(i, P0{}, T_{}, T_{}, T_{}, T_{}, T_{}, T_{});
        ++i;
        if (i >= nt - 1) break;
        iter(i, P1{}, T_{}, T_{}, T_{}, T_{}, T_{}, T_{});
        ++i;
    }
#ifdef UV_PW4_DIAG
    if (lane == 0 && uv_pw4_dbg) {
        unsigned long long* d = uv_pw4_dbg + ((long)blockIdx.x * 4 + wave_u) * 4;
        d[0] = __builtin_amdgcn_s_memtime() - st0;
        d[1] = __builtin_amdgcn_s_memrealtime() - sr0;
        d[2] = (unsigned long long)i;
    }
#endif
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
