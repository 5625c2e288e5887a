// fp32 "NT" GEMM on the f32-input MFMA (v_mfma_f32_16x16x4_f32; exact f32 products, f32 accumulate):
//   C[M,N] = A[M,K] . W[N,K]^T + bias (+ residual)
// Used where the reference computes in fp32 on purpose:
//   Head.head Linear(3072 -> 192) inside the fp32 autocast island   models/wan/utils/modules/model.py:286-290
//   VAE 1x1(x1) convolutions / attention projections (vae2_2.py:211, 249-250, 766-767), channels-last
// Same tile machinery as gemm_bf16.hip: 128-byte LDS rows (32 floats), global_load_lds staging with the
// (row>>1)&7 chunk swizzle on the source address, ds_read_b128 fragments. One b128 fragment feeds FOUR
// 16x16x4 MFMAs: element e of lane (row, kq) is used as k-slot kq of step e on both operands, which only
// permutes the order in which the K products are summed.
#include "common.h"

typedef __attribute__((address_space(3))) void lds_void_f;

struct GemmF32Args {
    const float* A; const float* W; const float* bias; const float* resid; float* out;
    long lda, ldw, ldo, ldr;
    int M, N, K, tiles_m, tiles_n;
    const float* zeros;  // >= 128 B of zeros for the K tail
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void gemm_f32_nt_kernel(GemmF32Args p) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
    constexpr int A_INSTR = BM / 8 / NW, W_INSTR = BN / 8 / NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int tile_m = blockIdx.x % p.tiles_m, tile_n = blockIdx.x / p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int srow = lane >> 3, pchunk = lane & 7;
    const float* a_src[A_INSTR];
    const float* w_src[W_INSTR];
    int a_c[A_INSTR], w_c[W_INSTR];
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int row = (i * NW + wave) * 8 + srow;
        a_c[i] = (pchunk ^ ((row >> 1) & 7)) * 4;
        a_src[i] = p.A + (long)min(m0 + row, p.M - 1) * p.lda + a_c[i];
    }
#pragma unroll
    for (int i = 0; i < W_INSTR; ++i) {
        const int row = (i * NW + wave) * 8 + srow;
        w_c[i] = (pchunk ^ ((row >> 1) & 7)) * 4;
        w_src[i] = p.W + (long)min(n0 + row, p.N - 1) * p.ldw + w_c[i];
    }
    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * STAGE;
        const int koff = kt * 32;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            const float* s = (koff + a_c[i] < p.K) ? a_src[i] + koff : p.zeros;
            __builtin_amdgcn_global_load_lds(s, (lds_void_f*)(base + (i * NW + wave) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < W_INSTR; ++i) {
            const float* s = (koff + w_c[i] < p.K) ? w_src[i] + koff : p.zeros;
            __builtin_amdgcn_global_load_lds(s, (lds_void_f*)(base + A_BYTES + (i * NW + wave) * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    int a_off[TM], a_key[TM], w_off[TN], w_key[TN];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int row = wm * (BM / WM) + j * 16 + frow;
        a_off[j] = row * 128; a_key[j] = (row >> 1) & 7;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int row = wn * (BN / WN) + i * 16 + frow;
        w_off[i] = A_BYTES + row * 128; w_key[i] = (row >> 1) & 7;
    }

    const int nk = (p.K + 31) / 32;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
        const char* base = smem + buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f32x4 af[TM], wf[TN];
            const int c = ks * 4 + fq;
#pragma unroll
            for (int j = 0; j < TM; ++j) af[j] = *(const f32x4*)(base + a_off[j] + ((c ^ a_key[j]) << 4));
#pragma unroll
            for (int i = 0; i < TN; ++i) wf[i] = *(const f32x4*)(base + w_off[i] + ((c ^ w_key[i]) << 4));
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][e], af[j][e], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + wm * (BM / WM) + j * 16 + frow;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int n = n0 + wn * (BN / WN) + i * 16 + 4 * fq;
            if (n >= p.N) continue;
            f32x4 v = acc[i][j];
            if (p.bias) {
                const f32x4 b = *(const f32x4*)(p.bias + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += b[e];
            }
            if (p.resid) {
                const f32x4 rr = *(const f32x4*)(p.resid + (long)m * p.ldr + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += rr[e];
            }
            *(f32x4*)(p.out + (long)m * p.ldo + n) = v;
        }
    }
}

static float* g_zero_page[UV_MAX_DEV];
const float* uv_zero_page() {   // one per device; allocated by uv_init() (outside any stream capture)
    float*& z = g_zero_page[uv_cur_dev()];
    if (!z) {
        if (hipMalloc(&z, 4096) != hipSuccess) return nullptr;
        hipMemset(z, 0, 4096);
    }
    return z;
}

extern "C" int uv_gemm_f32_nt(const float* A, long lda, const float* W, long ldw, const float* bias, int M, int N,
                              int K, float* out, long ldo, const float* resid, long ldr, void* stream) {
    UV_CHECK_ARG(A && W && out, "uv_gemm_f32_nt: null pointer");
    UV_CHECK_ARG(M > 0 && N > 0 && K > 0, "uv_gemm_f32_nt: bad shape");
    UV_CHECK_ARG(K % 4 == 0 && N % 4 == 0, "uv_gemm_f32_nt: K and N must be multiples of 4 (K=%d N=%d)", K, N);
    UV_CHECK_ARG(lda % 4 == 0 && ldw % 4 == 0 && ldo % 4 == 0 && ldr % 4 == 0, "uv_gemm_f32_nt: ld* must be multiples of 4");
    UV_CHECK_ARG((((uintptr_t)A | (uintptr_t)W | (uintptr_t)out | (uintptr_t)bias | (uintptr_t)resid) & 15) == 0,
                 "uv_gemm_f32_nt: pointers must be 16-byte aligned");
    GemmF32Args a;
    a.A = A; a.W = W; a.bias = bias; a.resid = resid; a.out = out;
    a.lda = lda; a.ldw = ldw; a.ldo = ldo; a.ldr = ldr; a.M = M; a.N = N; a.K = K;
    a.zeros = uv_zero_page();
    UV_CHECK_ARG(a.zeros, "uv_gemm_f32_nt: zero page allocation failed");
    constexpr int BM = 128, BN = 128;
    a.tiles_m = (M + BM - 1) / BM; a.tiles_n = (N + BN - 1) / BN;
    auto kern = gemm_f32_nt_kernel<BM, BN, 2, 2>;
    const size_t lds = 2 * (BM + BN) * 128;
    UV_ONCE_PER_DEVICE(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(256), lds, (hipStream_t)stream, a);
    UV_CHECK_LAUNCH("uv_gemm_f32_nt");
    return 0;
}
