// Shared device helpers for the gfx950 (CDNA4 / MI355X) kernels of univid_amd.
// Wave = 64 lanes everywhere; nothing here is portable to 32-wide hardware.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>

typedef uint16_t bf16_t;  // raw bf16 bits in HBM

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define UV_WAVE 64

// Host-side per-device state (function attributes, CU count, zero page) is indexed by the CURRENT HIP device: the Python layer
// makes the tensors' device current around every entry point, so one process can drive several GPUs (reference: manual model
// placement, models/model_pipeline.py `wan_gpu`).
#define UV_MAX_DEV 16
static inline int uv_cur_dev() {
    int d = 0;
    (void)hipGetDevice(&d);
    return (unsigned)d < UV_MAX_DEV ? d : 0;
}
// One-time, per-device host setup (hipFuncSetAttribute of a kernel's dynamic LDS size): one std::once_flag per device at the call
// site, so two host threads (one per GPU; ctypes releases the GIL) never race on a plain flag.
#define UV_ONCE_PER_DEVICE(...)                                                      \
    do {                                                                             \
        static std::once_flag uv_once_[UV_MAX_DEV];                                  \
        std::call_once(uv_once_[uv_cur_dev()], [&] { __VA_ARGS__; });                \
    } while (0)
// Compute units of the current device (256 on MI355X); launch-geometry decisions (whole rounds of workgroups) use it.
static inline int uv_num_cus() {
    static int cus[UV_MAX_DEV];
    static std::once_flag once[UV_MAX_DEV];
    const int dev = uv_cur_dev();
    std::call_once(once[dev], [&] {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    });
    return cus[dev];
}

// f32 -> bf16, round-to-nearest-even (same rounding torch's .to(bfloat16) uses).
// A plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN.
__device__ __forceinline__ bf16_t f2bf(float x) {
    __bf16 b = (__bf16)x;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float bf2f(bf16_t b) {
    return __builtin_bit_cast(float, (uint32_t)b << 16);
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
// round an f32 through bf16 and back (the value a bf16 tensor would hold)
__device__ __forceinline__ float round_bf(float x) { return bf2f(f2bf(x)); }

// ---- 16-bit operand type of the MFMA kernels as a compile-time switch: F16 = false -> bf16 (the DiT), true -> IEEE fp16 (the
// SigLIP2 ranker's reference dtype). The bits travel as bf16_t / bf16x8 either way; only conversions and the MFMA opcode differ.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
template <bool F16> __device__ __forceinline__ float in16(bf16_t b) {
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, b);
    else return bf2f(b);
}
template <bool F16> __device__ __forceinline__ bf16_t out16(float x) {
    if constexpr (F16) return __builtin_bit_cast(bf16_t, (_Float16)x);
    else return f2bf(x);
}
template <bool F16> __device__ __forceinline__ float round16(float x) { return in16<F16>(out16<F16>(x)); }
template <bool F16> __device__ __forceinline__ uint32_t pack16_2(float lo, float hi) {
    return (uint32_t)out16<F16>(lo) | ((uint32_t)out16<F16>(hi) << 16);
}
template <bool F16> __device__ __forceinline__ f32x4 mfma_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <bool F16> __device__ __forceinline__ f32x16 mfma_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// GELU(approximate='tanh') of an f32 value (torch upcasts a bf16 tensor, evaluates in f32, rounds once):
// 0.5*x*(1+tanh(u)), u = sqrt(2/pi)*(x+0.044715*x^3). Since 1+tanh(u) = 2/(1+exp(-2u)) this is x / (1 + exp(-2u)):
// one v_exp_f32 and one v_rcp_f32 (1 ulp each) instead of libm's tanhf (~40 instructions, the dominant cost of the
// ffn.0 epilogue), and without the cancellation of 1+tanh(u) for x < -2. The exponent -2u*log2(e) is evaluated as
// x*(A + B*x^2) with the constants folded (A = -2*log2(e)*sqrt(2/pi), B = A*0.044715): 3 VALU operations instead of 6 in
// front of the exp; the two forms agree to ~1e-7 relative, far inside the bf16 rounding of the result.
// Limits: u -> -inf gives x*0 = -0, u -> +inf gives x.
__device__ __forceinline__ float gelu_tanh_f32(float x) {
    const float kA = -2.3022081986f;    // -2 * log2(e) * sqrt(2/pi)
    const float kB = -0.1029432396f;    // kA * 0.044715
    const float q = __builtin_fmaf(x * x, kB, kA);
    const float e = __builtin_amdgcn_exp2f(x * q);          // exp(-2u)
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// The same function on two values at once with PACKED f32 arithmetic (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two lanes' worth of work per
// issue slot; the two exponentials and reciprocals stay scalar instructions). Each component is computed by exactly the operations of
// gelu_tanh_f32 (a packed instruction is two independent IEEE operations): bit-identical results. For epilogues, where no MFMA is in
// flight beside them (packed f32 VALU next to MFMAs costs issue cycles: MI355X_MICROARCH.md).
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_tanh_f32x2(f32x2 x) {
    const f32x2 kA = {-2.3022081986f, -2.3022081986f}, kB = {-0.1029432396f, -0.1029432396f}, one = {1.0f, 1.0f};
    const f32x2 q = __builtin_elementwise_fma(x * x, kB, kA);
    const f32x2 t = x * q;
    const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    const f32x2 d = one + e;
    const f32x2 c = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    return x * c;
}

__device__ __forceinline__ float silu_f32(float x) { return x / (1.0f + expf(-x)); }

// Developer options (include/univid_hip.h: uv_set_option). Process-wide relaxed atomics: set explicitly through the C ABI, never
// read from the environment (getenv racing with a putenv on another host thread is undefined behaviour, and a stray variable would
// silently switch a summation order).
#define UV_OPT_CONV_HALO 0     // -1 automatic (default), 0 never the LDS-halo convolution kernel, 1 whenever the geometry fits
#define UV_OPT_GEMM_GM 1       // 0 automatic (default), > 0: tile-walk group height of the persistent GEMM
#define UV_OPT_ATTN_CUT 2      // 0 automatic (default); v > 0: flash_attn_fwd12_kernel cuts every head with n8 = v - 1 eight-unit blocks (A/B tools)
#define UV_OPT_COUNT 3
int uv_option(int key);

// error plumbing shared by the extern "C" entry points
extern "C" const char* uv_last_error(void);
void uv_set_error(const char* fmt, ...);

#define UV_CHECK_ARG(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            uv_set_error(__VA_ARGS__);     \
            return -1;                     \
        }                                  \
    } while (0)

#define UV_CHECK_LAUNCH(name)                                                     \
    do {                                                                          \
        hipError_t _e = hipGetLastError();                                        \
        if (_e != hipSuccess) {                                                   \
            uv_set_error("%s: launch failed: %s", name, hipGetErrorString(_e));   \
            return (int)_e;                                                       \
        }                                                                         \
    } while (0)
