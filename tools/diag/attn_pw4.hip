// DIAGNOSTIC KERNEL - not part of libunivid_hip.so since round 4 (it is bit-identical to and 10 % slower than the product's
// flash_attn_fwd12_kernel; it stays in tools/diag/ as a second, independently scheduled implementation of the same arithmetic:
// tools/diag/build_diag.py builds tools/diag/libuv_diag.so with the C entry uv_diag_flash_attn_pw4, the GPU test
// test_flash_attention_pw4_kernel_is_bit_identical_to_fwd12 and tools/attn_ab.py call it).
//
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
//   * the ACCUMULATOR half of the register file is owned by the asm statements below, addressed literally:
//         a[0:127] O^T of both blocks   a[128:191] Q fragments   a[192:203] K fragment ring (3)   a[204:215] V^T fragment ring (3)
//     hipcc allocates only the architectural half (S^T of two tiles, P, softmax state). With builtin MFMAs hipcc moves every
//     accumulator of a 512-register kernel into AGPRs and S^T comes back through v_accvgpr_read; the asm MFMAs write S^T to
//     VGPRs and read Q / K / V^T from AGPRs. Fragment reads are asm too (ds_read_b128 with an AGPR destination) behind
//     hand-counted s_waitcnt lgkmcnt(N): left to hipcc they sat in 24 more VGPRs (P spilled to AGPRs through v_accvgpr copies)
//     and were waited for with lgkmcnt(0) right behind the newest read.
//   * software pipeline over 64-key tiles, two phases of 32 slots (1 MFMA + fillers, ONE asm statement) per iteration i:
//       phase 1   QK^T of tile i+1   beside   exp2 / row sums / bf16 packing of tile i ("finish"), itself pipelined over the
//                                             slots: a slot consumes the exponentials the previous slot issued
//       phase 2   P.V of tile i      beside   row max / deferred-maximum decision / scale-subtract of tile i+1 ("start")
//     Every even slot issues exactly one fragment read, two fragments ahead of its first use: K2 .. K15, V0, V1 in phase 1,
//     V2 .. V15, K0', K1' in phase 2 - so the wait in front of a fragment's first MFMA is always lgkmcnt(2).
//   * K and V^T tiles arrive by LDS-DMA into two K and two V^T stages (64 KiB); the pieces of tile i+3 (K) / i+2 (V^T) are
//     issued behind the ONE barrier of iteration i (slot 28 of phase 2, after every wave's last read of the stages they overwrite)
//     and are waited for a whole iteration later.
//   * the online-softmax rescale (rare after the first tiles: deferred maximum, UV_ATT_DEFER) runs behind wave-uniform branches:
//     the decision's update path in the start phase, O^T *= alpha at the two points of phase 2 where the reference order has it.
// tools/diag/pw4_audit.py checks the compiled ISA (no compiler access to the asm-owned AGPRs, no VALU write in front of an MFMA
// that reads it); tests/test_host_logic.py runs it.
#include "attn_args.h"      // univid_amd/csrc (-I): argument block and constants shared with attention.hip
#include <type_traits>
#include <utility>

typedef __attribute__((address_space(3))) void lds_void_p;

// EVERY statement that touches the accumulator file lists the whole asm-owned range as clobbered: hipcc parks spilled values in
// any AGPR it believes free between two statements (seen: a128.. overwritten during the P.V phase when only the QK^T statements
// claimed them).
#define UV_ACL_ALL "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215"
#define UV_KRING 192
#define UV_VRING 204

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

// ---- accumulator-file helpers ----------------------------------------------------------------------------------------------------
template <int N> __device__ __forceinline__ void acc_zero() { asm volatile("v_accvgpr_write_b32 a%c0, 0" ::"n"(N) : UV_ACL_ALL); }
template <int N> __device__ __forceinline__ void acc_write(uint32_t v) { asm volatile("v_accvgpr_write_b32 a%c1, %0" ::"v"(v), "n"(N) : UV_ACL_ALL); }
template <int N> __device__ __forceinline__ void acc_write(float v) { asm volatile("v_accvgpr_write_b32 a%c1, %0" ::"v"(v), "n"(N) : UV_ACL_ALL); }
template <int N> __device__ __forceinline__ float acc_read(void) {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(v) : "n"(N) : UV_ACL_ALL);
    return v;
}
// fragment read: 16 bytes per lane from LDS byte address `addr` + OFF into a[RA:RA+3]
template <int RA, int OFF> __device__ __forceinline__ void lds_rd(unsigned addr) {
    asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%c3" ::"v"(addr), "n"(RA), "n"(RA + 3), "n"(OFF) : UV_ACL_ALL);
}

// ---- asm text pieces. A slot statement = [fragment read] [counted wait] MFMA fillers. Operand numbers are fixed per family; the
// read's address / destination / offset operands and the wait count sit at the END of the input list so that a statement with
// and without them shares one numbering.
#define UV_RD(ADDR, RA, RA3, OFF) "ds_read_b128 a[%c" #RA ":%c" #RA3 "], %" #ADDR " offset:%c" #OFF "\n\t"
#define UV_WT(W) "s_waitcnt lgkmcnt(%c" #W ")\n\t"

// Phase-1 slot: S^T(next tile) (+)= Kfrag a[KA:] . Qfrag a[QB:] and the finish of one pair of the current tile:
//     exp p0n, p1n (next slot's pair) ; psum (+)= p0c ; pw = cvt_pk(p0c, p1c) ; psum += p1c   [; l = l * alpha ; l += psum]
// p0c / p1c were issued by the previous slot: a v_exp_f32 result takes ~30 cycles to arrive, and consumed in the slot that issued it
// the lone wave stood still that long in every slot. MODE 0: first pair of a (block, half): psum = p0 (the reference starts from
// 0 + p0, the same value); 1: middle; 2: last pair, with the row-sum update (two roundings, as the reference's `l *= alpha; l += psum`).
// FIN = false: the MFMA alone (prologue). LAST: slot 31, no next pair. RD: fragment read + lgkmcnt(WN) wait in front (even slots).
template <int MODE, bool FIRST, bool LAST, bool FIN, bool RD, int WN, int QB, int KA, int RA, int ROFF>
__device__ __forceinline__ uint32_t p1_slot(f32x16& sn, unsigned raddr, float& p0c, float& p1c, float s0n, float s1n, float& psum, float& l,
                                           float alpha) {
    uint32_t pw = 0;
    if constexpr (!FIN) {
        // operands: 0 sn | 1 QB 2 QB3 3 KA 4 KA3 | 5 raddr 6 RA 7 RA3 8 ROFF 9 WN
#define UV_P1_OPS(SN) SN(sn) : "n"(QB), "n"(QB + 3), "n"(KA), "n"(KA + 3), "v"(raddr), "n"(RA), "n"(RA + 3), "n"(ROFF), "n"(WN) : UV_ACL_ALL
#define UV_P1_MF(C) "v_mfma_f32_32x32x16_bf16 %0, a[%c3:%c4], a[%c1:%c2], " C
        if constexpr (FIRST && RD) asm volatile(UV_RD(5, 6, 7, 8) UV_WT(9) UV_P1_MF("0") : UV_P1_OPS("=&v"));
        else if constexpr (FIRST) asm volatile(UV_P1_MF("0") : UV_P1_OPS("=&v"));
        else if constexpr (RD) asm volatile(UV_RD(5, 6, 7, 8) UV_WT(9) UV_P1_MF("%0") : UV_P1_OPS("+v"));
        else if constexpr (WN >= 0) asm volatile(UV_WT(9) UV_P1_MF("%0") : UV_P1_OPS("+v"));
        else asm volatile(UV_P1_MF("%0") : UV_P1_OPS("+v"));
#undef UV_P1_OPS
#undef UV_P1_MF
    } else if constexpr (LAST) {
        static_assert(MODE == 2 && !FIRST && !RD, "slot 31 ends a (block, half)");
        asm volatile("v_mfma_f32_32x32x16_bf16 %2, a[%c9:%c10], a[%c7:%c8], %2\n\tv_add_f32 %0, %0, %4\n\tv_cvt_pk_bf16_f32 %1, %4, %5\n\t"
                     "v_add_f32 %0, %0, %5\n\tv_mul_f32 %3, %3, %6\n\tv_add_f32 %3, %3, %0"
                     : "+v"(psum), "=&v"(pw), "+v"(sn), "+v"(l)
                     : "v"(p0c), "v"(p1c), "v"(alpha), "n"(QB), "n"(QB + 3), "n"(KA), "n"(KA + 3) : UV_ACL_ALL);
    } else if constexpr (MODE == 2) {
        static_assert(!FIRST && !RD, "the last pair of a (block, half) sits in an odd slot");
        float p0n, p1n;
        asm volatile("v_mfma_f32_32x32x16_bf16 %4, a[%c13:%c14], a[%c11:%c12], %4\n\tv_exp_f32 %0, %8\n\tv_exp_f32 %1, %9\n\t"
                     "v_add_f32 %2, %2, %6\n\tv_cvt_pk_bf16_f32 %3, %6, %7\n\tv_add_f32 %2, %2, %7\n\tv_mul_f32 %5, %5, %10\n\tv_add_f32 %5, %5, %2"
                     : "=&v"(p0n), "=&v"(p1n), "+v"(psum), "=&v"(pw), "+v"(sn), "+v"(l)
                     : "v"(p0c), "v"(p1c), "v"(s0n), "v"(s1n), "v"(alpha), "n"(QB), "n"(QB + 3), "n"(KA), "n"(KA + 3) : UV_ACL_ALL);
        p0c = p0n;
        p1c = p1n;
    } else {
        // operands: 0 p0n 1 p1n 2 psum 3 pw 4 sn | 5 p0c 6 p1c 7 s0n 8 s1n | 9 QB 10 QB3 11 KA 12 KA3 | 13 raddr 14 RA 15 RA3 16 ROFF 17 WN
        float p0n, p1n;
#define UV_P1_OPS(PS, SN)                                                                                                          \
    "=&v"(p0n), "=&v"(p1n), PS(psum), "=&v"(pw), SN(sn)                                                                            \
        : "v"(p0c), "v"(p1c), "v"(s0n), "v"(s1n), "n"(QB), "n"(QB + 3), "n"(KA), "n"(KA + 3), "v"(raddr), "n"(RA), "n"(RA + 3), "n"(ROFF), \
          "n"(WN) : UV_ACL_ALL
#define UV_P1_BODY(C, SUM0) "v_mfma_f32_32x32x16_bf16 %4, a[%c11:%c12], a[%c9:%c10], " C "\n\tv_exp_f32 %0, %7\n\tv_exp_f32 %1, %8\n\t" SUM0 \
    "\n\tv_cvt_pk_bf16_f32 %3, %5, %6\n\tv_add_f32 %2, %2, %6"
#define UV_P1_PFX UV_RD(13, 14, 15, 16) UV_WT(17)
        if constexpr (MODE == 0 && FIRST && RD) asm volatile(UV_P1_PFX UV_P1_BODY("0", "v_mov_b32 %2, %5") : UV_P1_OPS("=&v", "=&v"));
        else if constexpr (MODE == 0 && FIRST) asm volatile(UV_P1_BODY("0", "v_mov_b32 %2, %5") : UV_P1_OPS("=&v", "=&v"));
        else if constexpr (MODE == 0 && RD) asm volatile(UV_P1_PFX UV_P1_BODY("%4", "v_mov_b32 %2, %5") : UV_P1_OPS("=&v", "+v"));
        else if constexpr (MODE == 0) asm volatile(UV_P1_BODY("%4", "v_mov_b32 %2, %5") : UV_P1_OPS("=&v", "+v"));
        else if constexpr (FIRST && RD) asm volatile(UV_P1_PFX UV_P1_BODY("0", "v_add_f32 %2, %2, %5") : UV_P1_OPS("+v", "=&v"));
        else if constexpr (FIRST) asm volatile(UV_P1_BODY("0", "v_add_f32 %2, %2, %5") : UV_P1_OPS("+v", "=&v"));
        else if constexpr (RD) asm volatile(UV_P1_PFX UV_P1_BODY("%4", "v_add_f32 %2, %2, %5") : UV_P1_OPS("+v", "+v"));
        else asm volatile(UV_P1_BODY("%4", "v_add_f32 %2, %2, %5") : UV_P1_OPS("+v", "+v"));
#undef UV_P1_OPS
#undef UV_P1_BODY
#undef UV_P1_PFX
        p0c = p0n;
        p1c = p1n;
    }
    return pw;
}

// finish of one pair WITHOUT an MFMA beside it and without the slot-to-slot pipelining (last tile, after the loop)
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

// ---- start-phase fillers, text only (operand numbers relative to a base: the P.V prefix operands come after them) ----------------
// row-maximum chains of both blocks, two ops each per slot: Q = 0 starts them (5 values each), 1 / 2 continue (4), 3 ends (3)
#define UV_MAX_Q0 "v_max3_f32 %0, %2, %3, %4\n\tv_max3_f32 %1, %7, %8, %9\n\tv_max3_f32 %0, %0, %5, %6\n\tv_max3_f32 %1, %1, %10, %11"
#define UV_MAX_Q12 "v_max3_f32 %0, %0, %2, %3\n\tv_max3_f32 %1, %1, %6, %7\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %1, %1, %8, %9"
#define UV_MAX_Q3 "v_max3_f32 %0, %0, %2, %3\n\tv_max3_f32 %1, %1, %5, %6\n\tv_max_f32 %0, %0, %4\n\tv_max_f32 %1, %1, %7"
// the other half-wave's maximum for two chains: copy, v_permlane32_swap (lanes 32-63 of the first operand <-> lanes 0-31 of the
// second: afterwards one register holds the low half-wave's value in both halves, the other the high one's), maximum. The swap reads
// registers a VALU wrote: two wait states (the second v_mov and the s_nop serve the first pair, the s_nop and the first swap the second).
#define UV_XHALF "v_mov_b32 %2, %0\n\tv_mov_b32 %3, %1\n\ts_nop 0\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\t" \
    "v_max_f32 %0, %0, %2\n\tv_max_f32 %1, %1, %3"
#define UV_FMA4 "v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"

// P.V prefix of a phase-2 slot: [read] [wait] O^T a[OB:] += V^Tfrag a[VA:] . P^Tfrag (VGPRs). B = number of the first prefix operand:
// B pf | B+1 OB  B+2 OB15  B+3 VA  B+4 VA3 | B+5 raddr  B+6 RA  B+7 RA3  B+8 ROFF  B+9 WN
#define UV_PV_MF(PF, OB, OB15, VA, VA3) "v_mfma_f32_32x32x16_bf16 a[%c" #OB ":%c" #OB15 "], a[%c" #VA ":%c" #VA3 "], %" #PF ", a[%c" #OB ":%c" #OB15 "]"
#define UV_PV_INS_(PFV) "v"(PFV), "n"(OB), "n"(OB + 15), "n"(VA), "n"(VA + 3), "v"(raddr), "n"(RA), "n"(RA + 3), "n"(ROFF), "n"(WN)
#define UV_PV_INS UV_PV_INS_(pf)
// PFX: 0 none, 1 wait only, 2 read + wait
#define UV_PV_ASM(B0, B1, B2, B3, B4, B5, B6, B7, B8, B9, TAIL_TXT, ...)                                                              \
    do {                                                                                                                             \
        if constexpr (PFX == 2) asm volatile(UV_RD(B5, B6, B7, B8) UV_WT(B9) UV_PV_MF(B0, B1, B2, B3, B4) TAIL_TXT : __VA_ARGS__);    \
        else if constexpr (PFX == 1) asm volatile(UV_WT(B9) UV_PV_MF(B0, B1, B2, B3, B4) TAIL_TXT : __VA_ARGS__);                      \
        else asm volatile(UV_PV_MF(B0, B1, B2, B3, B4) TAIL_TXT : __VA_ARGS__);                                                        \
    } while (0)

template <int PFX, int WN, int OB, int VA, int RA, int ROFF>
__device__ __forceinline__ void p2_plain(const bf16x8& pf, unsigned raddr) {
    UV_PV_ASM(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, "", : UV_PV_INS : UV_ACL_ALL);
}
template <int Q, int PFX, int WN, int OB, int VA, int RA, int ROFF>
__device__ __forceinline__ void p2_max(const bf16x8& pf, unsigned raddr, float& ma, float& mb, const f32x16& a, const f32x16& b) {
    if constexpr (Q == 0)
        UV_PV_ASM(12, 13, 14, 15, 16, 17, 18, 19, 20, 21, "\n\t" UV_MAX_Q0, "=&v"(ma), "=&v"(mb)
                  : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), UV_PV_INS : UV_ACL_ALL);
    else if constexpr (Q < 3)
        UV_PV_ASM(10, 11, 12, 13, 14, 15, 16, 17, 18, 19, "\n\t" UV_MAX_Q12, "+v"(ma), "+v"(mb)
                  : "v"(a[4 * Q + 1]), "v"(a[4 * Q + 2]), "v"(a[4 * Q + 3]), "v"(a[4 * Q + 4]), "v"(b[4 * Q + 1]), "v"(b[4 * Q + 2]),
                    "v"(b[4 * Q + 3]), "v"(b[4 * Q + 4]), UV_PV_INS : UV_ACL_ALL);
    else
        UV_PV_ASM(8, 9, 10, 11, 12, 13, 14, 15, 16, 17, "\n\t" UV_MAX_Q3, "+v"(ma), "+v"(mb)
                  : "v"(a[13]), "v"(a[14]), "v"(a[15]), "v"(b[13]), "v"(b[14]), "v"(b[15]), UV_PV_INS : UV_ACL_ALL);
}
template <int PFX, int WN, int OB, int VA, int RA, int ROFF>
__device__ __forceinline__ void p2_xhalf(const bf16x8& pf, unsigned raddr, float& ma, float& mb) {
    float ta, tb;
    UV_PV_ASM(4, 5, 6, 7, 8, 9, 10, 11, 12, 13, "\n\t" UV_XHALF, "+v"(ma), "+v"(mb), "=&v"(ta), "=&v"(tb) : UV_PV_INS : UV_ACL_ALL);
}
// s = s * c + mneg on four accumulator registers (in place) beside the P.V; a macro: vector elements cannot bind to references
#define UV_P2_FMA4(PFV, S_, E0, C_, MNEG)                                                                                          \
    UV_PV_ASM(6, 7, 8, 9, 10, 11, 12, 13, 14, 15, "\n\t" UV_FMA4, "+v"(S_[E0]), "+v"(S_[E0 + 1]), "+v"(S_[E0 + 2]), "+v"(S_[E0 + 3])  \
              : "s"(C_), "v"(MNEG), UV_PV_INS_(PFV) : UV_ACL_ALL)
// the exponentials of the NEXT iteration's first pair beside the last P.V of phase 2
template <int PFX, int WN, int OB, int VA, int RA, int ROFF>
__device__ __forceinline__ void p2_exp_pair(const bf16x8& pf, unsigned raddr, float& p0c, float& p1c, float s0, float s1) {
    UV_PV_ASM(4, 5, 6, 7, 8, 9, 10, 11, 12, 13, "\n\tv_exp_f32 %0, %2\n\tv_exp_f32 %1, %3", "=&v"(p0c), "=&v"(p1c) : "v"(s0), "v"(s1), UV_PV_INS : UV_ACL_ALL);
}
// the same fillers without a P.V beside them (prologue: start of tile 0)
template <int Q> __device__ __forceinline__ void max_step(float& ma, float& mb, const f32x16& a, const f32x16& b) {
    if constexpr (Q == 0)
        asm volatile(UV_MAX_Q0 : "=&v"(ma), "=&v"(mb)
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]));
    else if constexpr (Q < 3)
        asm volatile(UV_MAX_Q12 : "+v"(ma), "+v"(mb)
                     : "v"(a[4 * Q + 1]), "v"(a[4 * Q + 2]), "v"(a[4 * Q + 3]), "v"(a[4 * Q + 4]), "v"(b[4 * Q + 1]), "v"(b[4 * Q + 2]),
                       "v"(b[4 * Q + 3]), "v"(b[4 * Q + 4]));
    else
        asm volatile(UV_MAX_Q3 : "+v"(ma), "+v"(mb) : "v"(a[13]), "v"(a[14]), "v"(a[15]), "v"(b[13]), "v"(b[14]), "v"(b[15]));
}
__device__ __forceinline__ void max_xhalf(float& ma, float& mb) {
    float ta, tb;
    asm volatile(UV_XHALF : "+v"(ma), "+v"(mb), "=&v"(ta), "=&v"(tb));
}
#define UV_FMA4_ONLY(S_, E0, C_, MNEG) \
    asm volatile(UV_FMA4 : "+v"(S_[E0]), "+v"(S_[E0 + 1]), "+v"(S_[E0 + 2]), "+v"(S_[E0 + 3]) : "s"(C_), "v"(MNEG))
__device__ __forceinline__ void exp_pair(float& p0c, float& p1c, float s0, float s1) {
    asm volatile("v_exp_f32 %0, %2\n\tv_exp_f32 %1, %3" : "=&v"(p0c), "=&v"(p1c) : "v"(s0), "v"(s1));
}
__device__ __forceinline__ float vmax(float a, float b) {
    float d;
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// One LDS-DMA piece, uniform (SGPR) base + 32-bit lane offset; M0 = LDS byte address of the piece. M0 is declared clobbered instead of
// saved and restored (hipcc warns that it is a reserved register; nothing else in this kernel keeps a value in it).
__device__ __forceinline__ void glds16_sb(const char* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}

// Empty asm with the value as an in/out operand: everything the value depends on is computed BEFORE this point of the (volatile-
// asm-ordered) instruction stream. hipcc's sinking passes otherwise move arithmetic to its first use.
#define UV_PIN(x) asm volatile("" : "+v"(x))

// Diagnostic build only (tools/diag/pw4_diag.hip compiles this file with -DUV_PW4_DIAG; the library never does): in-kernel stamps
// around the main loop. Stamp values go to a buffer of their own; no output value is computed from them.
#ifdef UV_PW4_DIAG
__device__ unsigned long long* uv_pw4_dbg;
#endif
#ifndef UV_PW4_ABL
#define UV_PW4_ABL 0
#endif
#define UV_ABL_NO_DMA 1
#define UV_ABL_NO_BARRIER 16
#define UV_ABL_P1_ONLY 64      // steady state runs QK^T + finish only
#define UV_ABL_P2_ONLY 128     // steady state runs P.V + start only

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void flash_attn_pw4_kernel(AttnArgs p) {
    constexpr int KROW = 256, K_BYTES = UV_ATT_KV * KROW, V_BYTES = 128 * 128, V_OFF = 2 * K_BYTES;
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
    // piece pi (0..3) of this wave's share of K tile t -> K stage t & 1
    auto dma_k = [&](int t, int pi) __attribute__((always_inline)) {
        if (t < nt_full) {
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
    for (int pi = 0; pi < 4; ++pi) dma_k(0, pi);
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) dma_v(0, pi);
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) dma_k(1, pi);

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
            acc_write<128 + 32 * X + w>(qv[w >> 2][w & 3]);
        });
    });
    sfor<128>([&](auto nt_) { acc_zero<decltype(nt_)::value>(); });

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
            acc_write<64 * X + e>(v * a);
        });
        asm volatile("s_nop 7" ::: "memory");
    };

    // Fragment read stream. K fragment f (key half f >> 3, k-step f & 7) of K stage ks -> K ring slot f % 3; V^T fragment g (key half
    // g >> 3, s2 = (g >> 2) & 1, d tile g & 3) of V^T stage vs -> V ring slot g % 3. Address VGPR + immediate offset:
#define UV_KADDR(f) kaddr[(f) & 7]
#define UV_KOFF(f, ks) ((ks) * K_BYTES + ((f) >> 3) * 32 * KROW)
#define UV_VADDR(g) vaddr[(g) >> 3][((g) >> 2) & 1]
#define UV_VOFF(g, vs) ((vs) * V_BYTES + ((g) & 3) * 4096)

    // One iteration i (PAR = i & 1):  phase 1 = [QK^T(i+1)] beside [finish(i)],  phase 2 = [P.V(i)] beside [start(i+1)].
    // K(i+1) is read from K stage PAR^1, V^T(i) from V^T stage PAR. Variants: prologue (QK + START), steady state (all four; the
    // DMA / mask decisions of the last iterations are wave-uniform run-time branches), last tile (FIN + PV).
    auto iter = [&](int i, auto par_t, auto qk_t, auto fin_t, auto pv_t, auto start_t) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_t)::value;
        constexpr bool DO_QK = decltype(qk_t)::value, DO_FIN = decltype(fin_t)::value, DO_PV = decltype(pv_t)::value;
        constexpr bool DO_START = decltype(start_t)::value;
        constexpr int CUR = PAR, NXT = PAR ^ 1;
        constexpr int KRS = PAR ^ 1, VRS = PAR;

        // ================= phase 1 =================
        if constexpr (DO_QK && !DO_FIN) {
            // the prologue has no preceding slot 28 / 30: head of the K fragment stream from scratch
            lds_rd<UV_KRING + 0, UV_KOFF(0, KRS)>(UV_KADDR(0));
            lds_rd<UV_KRING + 4, UV_KOFF(1, KRS)>(UV_KADDR(1));
        }
        sfor<32>([&](auto nt_) {
            constexpr int n = decltype(nt_)::value;
            // finish: exp2 / row sum / bf16 of two S values of tile i: block FX, half FT, elements e0, e0+1 in the reference order
            constexpr int FX = (n >> 3) & 1, FT = n >> 4, e0 = 2 * (n & 7), MODE = (n & 7) == 0 ? 0 : ((n & 7) == 7 ? 2 : 1);
            // QK^T: block X, fragment f = (half T, k-step kk)
            constexpr int X = n & 1, f = n >> 1, kk = f & 7, T = f >> 3;
            // this slot's fragment read: K fragment f + 2, or (slots 28, 30, when a P.V phase follows) V^T fragment 0 / 1
            constexpr bool EVEN = (n & 1) == 0, RDK = EVEN && f + 2 < 16, RDV = EVEN && f + 2 >= 16 && DO_PV;
            constexpr int RA = RDK ? UV_KRING + 4 * ((f + 2) % 3) : UV_VRING + 4 * ((f + 2 - 16) % 3);
            constexpr int ROFF = RDK ? UV_KOFF(f + 2, KRS) : UV_VOFF(f + 2 - 16, VRS);
            // reads issued after fragment f's own, up to this slot's: 2 in the steady stream, fewer at the end of a QK-only phase
            constexpr int WN = !EVEN ? -1 : ((RDK || RDV) ? 2 : (f == 14 ? 1 : 0));
            const unsigned raddr = RDK ? UV_KADDR(f + 2) : UV_VADDR((f + 2) & 15);
            if constexpr (DO_QK) {
                constexpr int n1 = n < 31 ? n + 1 : 31, NX = (n1 >> 3) & 1, NT_ = n1 >> 4, ne0 = 2 * (n1 & 7);   // the NEXT slot's pair
                const uint32_t pw = p1_slot<MODE, kk == 0, n == 31, DO_FIN, RDK || RDV, WN, 128 + 32 * X + 4 * kk, UV_KRING + 4 * (f % 3), RA, ROFF>(
                    S[NXT][X][T], raddr, p0c, p1c, S[CUR][NX][NT_][ne0], S[CUR][NX][NT_][ne0 + 1], psum, l_run[FX], alpha[FX][FT]);
                if constexpr (DO_FIN) pf[FX][FT][e0 >> 3][(e0 & 7) >> 1] = pw;
            } else if constexpr (DO_FIN) {
                if constexpr (RDV) lds_rd<RA, ROFF>(raddr);
                pf[FX][FT][e0 >> 3][(e0 & 7) >> 1] = fin_pair<MODE>(S[CUR][FX][FT][e0], S[CUR][FX][FT][e0 + 1], psum, l_run[FX], alpha[FX][FT]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });

        // ================= phase 2 =================
        float mx[4], mneg_t[4];                // combos cb = X + 2 T
        float alpha_n[2][2];
        bool flag_n[2][2];
        sfor<32>([&](auto nt_) {
            constexpr int n = decltype(nt_)::value;
            constexpr int X = n & 1, g = n >> 1, d = g & 3, s2 = (g >> 2) & 1, T = g >> 3, OB = 64 * X + 16 * d;     // P.V of this slot
            constexpr int VA = UV_VRING + 4 * (g % 3);
            // this slot's fragment read: V^T fragment g + 2 in front of the MFMA (even slots up to 26); the K head of the next tile
            // follows the barrier (slot 28) / precedes the MFMA as a statement of its own (slot 30)
            constexpr bool EVEN = (n & 1) == 0, RDV = EVEN && g + 2 < 16;
            constexpr int RA = UV_VRING + 4 * ((g + 2) % 3), ROFF = UV_VOFF(g + 2, VRS);
            constexpr int PFX = !EVEN ? 0 : (RDV ? 2 : 1);
            // slot 28: V15 is the only newer read; slot 30: K0', K1' (steady state) or nothing (last tile) are newer than V15
            constexpr int WN = !EVEN ? -1 : (RDV ? 2 : (n == 28 ? 1 : (DO_QK || DO_START ? 2 : 0)));
            const unsigned raddr = UV_VADDR(RDV ? g + 2 : 0);
            if constexpr (DO_PV && (n == 0 || n == 16)) {
                // the reference rescales O between the decision of half T and its P.V
                if (flag[0][T]) rescale(std::integral_constant<int, 0>{}, alpha[0][T]);
                if (flag[1][T]) rescale(std::integral_constant<int, 1>{}, alpha[1][T]);
            }
            if constexpr (n == 30 && (DO_QK || DO_START)) lds_rd<UV_KRING + 4, UV_KOFF(1, PAR)>(UV_KADDR(1));   // K1' of K stage PAR
            const bf16x8 pfr = __builtin_bit_cast(bf16x8, pf[X][T][s2]);
            // start of tile i+1 beside the P.V. Row maxima: slots 0-3 the two T = 0 chains, 4-7 the two T = 1 chains (their last MFMA
            // is recent); 8, 9: the other half-wave's maximum (combos cb = X + 2 T); 12-27: S*c - m*c in place, (A,0) (B,0) (A,1) (B,1)
            if constexpr (DO_START && n < 8) {
                constexpr int MT = n >> 2, q = n & 3;            // chain ops 2q, 2q+1 of both blocks
                if (q == 0 && (i + 1) * UV_ATT_KV + UV_ATT_KV > p.Lk) {      // tile i+1 is the ragged last tile (wave-uniform, rare)
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
                if constexpr (DO_PV) p2_max<q, PFX, WN, OB, VA, RA, ROFF>(pfr, raddr, mx[2 * MT], mx[2 * MT + 1], S[NXT][0][MT], S[NXT][1][MT]);
                else max_step<q>(mx[2 * MT], mx[2 * MT + 1], S[NXT][0][MT], S[NXT][1][MT]);
            } else if constexpr (DO_START && (n == 8 || n == 9)) {
                if constexpr (DO_PV) p2_xhalf<PFX, WN, OB, VA, RA, ROFF>(pfr, raddr, mx[2 * (n - 8)], mx[2 * (n - 8) + 1]);
                else max_xhalf(mx[2 * (n - 8)], mx[2 * (n - 8) + 1]);
            } else if constexpr (DO_START && n >= 14 && n < 28) {
                constexpr int cb = (n - 12) >> 2, FX = cb & 1, FT = cb >> 1, e0 = 4 * ((n - 12) & 3);
                f32x16& sm = S[NXT][FX][FT];
                if constexpr (DO_PV) UV_P2_FMA4(pfr, sm, e0, c, mneg_t[cb]);
                else UV_FMA4_ONLY(sm, e0, c, mneg_t[cb]);
            } else if constexpr (DO_START && n == 31) {
                // first pair of the NEXT iteration's finish phase: S[NXT] (A, half 0) was scaled in slots 12-15
                if constexpr (DO_PV) p2_exp_pair<PFX, WN, OB, VA, RA, ROFF>(pfr, raddr, p0c, p1c, S[NXT][0][0][0], S[NXT][0][0][1]);
                else exp_pair(p0c, p1c, S[NXT][0][0][0], S[NXT][0][0][1]);
            } else if constexpr (DO_PV) {
                p2_plain<PFX, WN, OB, VA, RA, ROFF>(pfr, raddr);
            }
            if constexpr (DO_START && n >= 10 && n < 14) {
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
                    UV_FMA4_ONLY(sm, e0, c, mneg_t[0]);
                }
            }
            if constexpr (n == 28 && (DO_QK || DO_START)) {
                // every wave's reads of K stage PAR^1 (phase 1) and V^T stage PAR (last one issued in slot 26) are complete; its own
                // pieces of K(i+2) / V^T(i+1) (issued an iteration ago) have landed
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                if constexpr (!(UV_PW4_ABL & UV_ABL_NO_BARRIER)) __builtin_amdgcn_s_barrier();
                // head of the next iteration's K fragment stream: K(i+2) from K stage PAR (always issued: keeps the read stream and its
                // wait counts uniform; behind the last tile it reads stale LDS that nothing uses)
                lds_rd<UV_KRING + 0, UV_KOFF(0, PAR)>(UV_KADDR(0));
            }
            if constexpr (n >= 28 && (DO_QK || DO_START) && !(UV_PW4_ABL & UV_ABL_NO_DMA)) {
                // behind the barrier: this wave's pieces of K(i+3) -> K stage PAR^1 (slots 28, 29) and V^T(i+2) -> V^T stage PAR (30, 31)
                constexpr int pi0 = 2 * (n & 1);
                if constexpr (n < 30) {
                    if (i + 3 < nt) {
                        dma_k(i + 3, pi0);
                        dma_k(i + 3, pi0 + 1);
                    }
                } else {
                    if (i + 2 < nt) {
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
    //        i   PAR   QK    FIN   PV    START
    iter(-1, P1{}, T_{}, F{}, F{}, T_{});                             // prologue: QK^T(0), start(0); DMA K(2), V^T(1)
    int i = 0;
#ifdef UV_PW4_DIAG
    const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    using AQ = std::integral_constant<bool, !(UV_PW4_ABL & UV_ABL_P2_ONLY)>;      // timing-only ablations of the diagnostic build
    using AP = std::integral_constant<bool, !(UV_PW4_ABL & UV_ABL_P1_ONLY)>;
    for (;;) {
        if (i >= nt - 1) break;
        iter(i, P0{}, AQ{}, AQ{}, AP{}, AP{});
        ++i;
        if (i >= nt - 1) break;
        iter(i, P1{}, AQ{}, AQ{}, AP{}, AP{});
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
    if (i & 1) iter(i, P1{}, F{}, T_{}, T_{}, F{});                   // i = nt - 1: finish + P.V of the last tile
    else iter(i, P0{}, F{}, T_{}, T_{}, F{});

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

static int uv_launch_attn_pw4(const AttnArgs& a0, hipStream_t st) {
    AttnArgs a = a0;
    a.q_blocks = (a.Lq + 255) / 256;
    a.n12 = 0;
    hipLaunchKernelGGL(flash_attn_pw4_kernel, dim3(a.q_blocks * a.H * a.batch), dim3(256), 0, st, a);
    return 0;
}

#ifndef UV_PW4_DIAG
// C entry of tools/diag/libuv_diag.so: the argument list of uv_flash_attn_bf16 (include/univid_hip.h). Returns 0, or -1 for a shape this
// kernel does not serve (head_dim 128, bf16, 32-bit lane offsets).
extern "C" int uv_diag_flash_attn_pw4(const void* q, long ldq, const void* k, long ldk, const void* vt, long ldvt, void* out, long ldo,
                                      int batch, int Lq, int Lk, int H, int head_dim, float softmax_scale, void* stream) {
    if (!q || !k || !vt || !out || head_dim != 128 || Lq <= 0 || Lk <= 0 || H <= 0 || batch <= 0) return -1;
    if (ldq % 8 || ldk % 8 || ldvt % 8 || ldo % 4 || 128 * ldvt >= (1L << 30) || 64 * ldk >= (1L << 30)) return -1;
    if (ldvt < (long)(batch - 1) * Lk + (long)((Lk + 63) / 64) * 64 || (batch > 1 && Lk % 8)) return -1;
    AttnArgs a;
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.vt = (const bf16_t*)vt; a.out = (bf16_t*)out;
    a.ldq = ldq; a.ldk = ldk; a.ldvt = ldvt; a.ldo = ldo;
    a.Lq = Lq; a.Lk = Lk; a.H = H; a.batch = batch; a.n12 = 0; a.q_blocks = 0;
    a.scale_log2 = softmax_scale * 1.4426950408889634f;
    uv_launch_attn_pw4(a, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
void uv_set_error(const char*, ...) {}
int uv_option(int) { return 0; }
#endif
