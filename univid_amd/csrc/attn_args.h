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
};

__device__ __forceinline__ int perm23(int i) {  // swap bits 2 and 3
    return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1);
}
