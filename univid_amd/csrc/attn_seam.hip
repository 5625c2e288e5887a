// Data-movement kernels of the `flash_attention(q, k, v, ...)` operator seam (reference models/wan/utils/modules/attention.py:24-130):
// the reference function takes q/k/v as [B, L, N, C] in any float dtype, casts them to a half dtype (:59-83), and returns the
// result in q's dtype (:130). uv_flash_attn_* consumes 16-bit rows and V TRANSPOSED ([N*C, keys], the layout that makes the
// P.V operand a plain 16-byte row read), so the seam needs: f32 -> 16-bit casts, a 16-bit transpose with zero-filled key
// padding, and the 16-bit -> f32 cast of the output. All HBM-bound, a few MB per call; none is on the fused DiT path (there the
// V projection GEMM writes V^T directly, UV_EPI_BF16_T).
#include "common.h"

// out[c][l] = in[l][c] for l < L; columns L .. Lpad-1 are written as ZERO (the attention kernel's key padding must be finite).
// 64 x 64 tiles through LDS: 128-byte row reads, 128-byte row writes.
__global__ __launch_bounds__(256) void transpose16_kernel(const uint16_t* __restrict__ in, long ldi, uint16_t* __restrict__ out,
                                                          long ldo, int L, int C, int Lpad) {
    __shared__ uint16_t tile[64][66];
    const int l0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
#pragma unroll
    for (int r = ty; r < 64; r += 4) {
        const int l = l0 + r, c = c0 + tx;
        tile[r][tx] = (l < L && c < C) ? in[(long)l * ldi + c] : (uint16_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r, l = l0 + tx;
        if (c < C && l < Lpad) out[(long)c * ldo + l] = tile[tx][r];
    }
}

extern "C" int uv_transpose_16(const void* in, long ldi, void* out, long ldo, int L, int C, int Lpad, void* stream) {
    UV_CHECK_ARG(in && out, "uv_transpose_16: null pointer");
    UV_CHECK_ARG(L > 0 && C > 0 && Lpad >= L && ldi >= C && ldo >= Lpad, "uv_transpose_16: bad shape L=%d C=%d Lpad=%d ldi=%ld ldo=%ld",
                 L, C, Lpad, ldi, ldo);
    const dim3 grid((Lpad + 63) / 64, (C + 63) / 64);
    hipLaunchKernelGGL(transpose16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)in, ldi, (uint16_t*)out, ldo,
                       L, C, Lpad);
    UV_CHECK_LAUNCH("uv_transpose_16");
    return 0;
}

template <bool F16>
__global__ void cast_f32_to16_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, long n) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const f32x4 v = *(const f32x4*)(in + i);
        u32x2 o;
        o[0] = pack16_2<F16>(v[0], v[1]);
        o[1] = pack16_2<F16>(v[2], v[3]);
        *(u32x2*)(out + i) = o;
    } else {
        for (long j = i; j < n; ++j) out[j] = out16<F16>(in[j]);
    }
}

// Contiguous f32 -> bf16 (f16 = 0) or IEEE fp16 (f16 = 1), round-to-nearest-even: attention.py:59-60 `half(x)`.
extern "C" int uv_cast_f32_to16(const float* in, void* out, long n, int f16, void* stream) {
    UV_CHECK_ARG(in && out && n > 0, "uv_cast_f32_to16: bad arguments");
    UV_CHECK_ARG((((uintptr_t)in & 15) | ((uintptr_t)out & 7)) == 0, "uv_cast_f32_to16: misaligned pointers");
    const long blocks = (n / 4 + 256) / 256;
    if (f16) hipLaunchKernelGGL(cast_f32_to16_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, (uint16_t*)out, n);
    else hipLaunchKernelGGL(cast_f32_to16_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, (uint16_t*)out, n);
    UV_CHECK_LAUNCH("uv_cast_f32_to16");
    return 0;
}

template <bool F16>
__global__ void cast_16_to_f32_kernel(const uint16_t* __restrict__ in, float* __restrict__ out, long n) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const u32x2 v = *(const u32x2*)(in + i);
        f32x4 o;
        o[0] = in16<F16>((bf16_t)(v[0] & 0xffff)); o[1] = in16<F16>((bf16_t)(v[0] >> 16));
        o[2] = in16<F16>((bf16_t)(v[1] & 0xffff)); o[3] = in16<F16>((bf16_t)(v[1] >> 16));
        *(f32x4*)(out + i) = o;
    } else {
        for (long j = i; j < n; ++j) out[j] = in16<F16>(in[j]);
    }
}

// Contiguous bf16 / fp16 -> f32 (exact): attention.py:130 `x.type(out_dtype)` for an fp32 caller.
extern "C" int uv_cast_16_to_f32(const void* in, float* out, long n, int f16, void* stream) {
    UV_CHECK_ARG(in && out && n > 0, "uv_cast_16_to_f32: bad arguments");
    UV_CHECK_ARG((((uintptr_t)out & 15) | ((uintptr_t)in & 7)) == 0, "uv_cast_16_to_f32: misaligned pointers");
    const long blocks = (n / 4 + 256) / 256;
    if (f16) hipLaunchKernelGGL(cast_16_to_f32_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)in, out, n);
    else hipLaunchKernelGGL(cast_16_to_f32_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)in, out, n);
    UV_CHECK_LAUNCH("uv_cast_16_to_f32");
    return 0;
}
