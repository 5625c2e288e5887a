// bf16 "NT" GEMM on gfx950 MFMA:  C[M,N] = A[M,K] . W[N,K]^T (+ bias), fp32 accumulate,
// with the fused epilogues the Wan DiT block needs.
//
// Replaces (reference, PyTorch ops under autocast-bf16):
//   nn.Linear q/k/v/o      models/wan/utils/modules/model.py:119-122,138-140,154,170-172,179
//   ffn Linear/GELU/Linear models/wan/utils/modules/model.py:212-214,252-253
//   gated fp32 residuals   models/wan/utils/modules/model.py:247,251,255
//   patch / text embedding models/wan/utils/modules/model.py:378-382,448,473
//
// Layout: both operands are K-contiguous (activations [M,K], nn.Linear weight [N,K]), so both
// MFMA fragments are 16-byte row reads. Tiles go HBM -> LDS with global_load_lds_dwordx4
// (1 KiB per wave-instruction = 8 rows x 128 B), LDS image is lane-linear and the XOR swizzle
// (chunk ^= (row>>1)&7) is applied on the SOURCE address and again on the ds_read_b128, which makes
// the 16x16x32 fragment reads bank-conflict-free on 128-byte rows.
// The MFMA is issued as D = Wfrag x Afrag so that a lane ends up with 4 CONSECUTIVE n for one m:
// epilogue stores are 8 B (bf16) / 16 B (fp32) per lane instead of 2 B scatters.
#include "common.h"

#define UV_BK 64  // k elements per LDS tile (128-byte rows)

enum {
    UV_EPI_BF16 = 0,           // out_bf16 = bf16(acc + bias)
    UV_EPI_GELU_BF16 = 1,      // out_bf16 = bf16(gelu_tanh(bf16(acc + bias)))
    UV_EPI_F32_FROM_BF16 = 2,  // out_f32  = float(bf16(acc + bias))
    UV_EPI_RESID_F32 = 3,      // x_f32   += float(bf16(acc + bias))
    UV_EPI_GATE_RESID_F32 = 4, // x_f32    = x + float(bf16(acc + bias)) * gate[tid[m]][n]
    UV_EPI_BF16_T = 5,         // outT_bf16[n][m] = bf16(acc + bias)   (V^T for attention)
};

struct GemmArgs {
    const bf16_t* A;
    const bf16_t* W;
    const bf16_t* bias;  // [N] bf16 or nullptr
    void* out;
    const float* gate;      // [n_t, gate_stride]  (EPI 4)
    const int32_t* gate_tid;  // [M] row -> gate row (EPI 4), may be nullptr => row 0
    long lda, ldw, ldo, gate_stride;
    int M, N, K;
    int tiles_m, tiles_n;
};

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void glds16(const void* g, lds_void* l) {
    __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
}

template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_nt_kernel(GemmArgs p) {
    constexpr int NW = WM * WN;
    constexpr int NT = NW * 64;
    constexpr int TM = BM / WM / 16;  // 16-row m tiles per wave
    constexpr int TN = BN / WN / 16;
    constexpr int A_BYTES = BM * 128;
    constexpr int W_BYTES = BN * 128;
    constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
    constexpr int A_INSTR = (BM / 8 + NW - 1) / NW;  // glds wave-instructions per wave for the A tile
    constexpr int W_INSTR = (BN / 8 + NW - 1) / NW;  // (the last pass may cover only part of the waves: guarded below)
    static_assert(BM % 8 == 0 && BN % 8 == 0 && (BM / WM) % 16 == 0 && (BN / WN) % 16 == 0, "tile/wave mismatch");
    constexpr bool TRANS = (EPI == UV_EPI_BF16_T);

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- XCD-aware tile mapping: blocks b, b+8, ... share an XCD's L2; give each XCD a
    // contiguous range of logical tiles, and walk logical tiles in GM-tall column groups so the
    // ~32 blocks resident on one XCD share few A / W panels.
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
    const int tile_m = first_m + in_group % gm;
    const int tile_n = in_group / gm;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-lane staging source pointers (row clamp keeps every load in bounds)
    const int srow = lane >> 3;  // row inside the 8-row glds piece
    const int pchunk = lane & 7; // physical 16-B chunk inside the 128-B row
    const bf16_t* a_src[A_INSTR];
    const bf16_t* w_src[W_INSTR];
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int row = (i * NW + wave) * 8 + srow;
        const int c = pchunk ^ ((row >> 1) & 7);
        const int gr = min(m0 + row, p.M - 1);
        a_src[i] = p.A + (long)gr * p.lda + c * 8;
    }
#pragma unroll
    for (int i = 0; i < W_INSTR; ++i) {
        const int row = (i * NW + wave) * 8 + srow;
        const int c = pchunk ^ ((row >> 1) & 7);
        const int gr = min(n0 + row, p.N - 1);
        w_src[i] = p.W + (long)gr * p.ldw + c * 8;
    }

    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * STAGE_BYTES;
        const int koff = kt * UV_BK;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i)
            if ((i * NW + wave) * 8 < BM) glds16(a_src[i] + koff, (lds_void*)(base + (i * NW + wave) * 1024));
#pragma unroll
        for (int i = 0; i < W_INSTR; ++i)
            if ((i * NW + wave) * 8 < BN) glds16(w_src[i] + koff, (lds_void*)(base + A_BYTES + (i * NW + wave) * 1024));
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: row = base + (lane&15), logical chunk = ks*4 + (lane>>4)
    const int frow = lane & 15;
    const int fq = lane >> 4;
    int a_off[TM], w_off[TN];  // byte offset of the row start; swizzle key per row
    int a_key[TM], w_key[TN];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int row = wm * (BM / WM) + j * 16 + frow;
        a_off[j] = row * 128;
        a_key[j] = (row >> 1) & 7;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int row = wn * (BN / WN) + i * 16 + frow;
        w_off[i] = A_BYTES + row * 128;
        w_key[i] = (row >> 1) & 7;
    }

    const int nk = p.K / UV_BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
        const char* base = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[TM], wf[TN];
            const int c = ks * 4 + fq;
#pragma unroll
            for (int j = 0; j < TM; ++j)
                af[j] = *(const bf16x8*)(base + a_off[j] + ((c ^ a_key[j]) << 4));
#pragma unroll
            for (int i = 0; i < TN; ++i)
                wf[i] = *(const bf16x8*)(base + w_off[i] + ((c ^ w_key[i]) << 4));
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    if (TRANS)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j], wf[i], acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue
    if (!TRANS) {
        // lane holds n = nb + 4*fq + {0..3} for m = mb + frow
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = m0 + wm * (BM / WM) + j * 16 + frow;
            if (m >= p.M) continue;
            const float* grow = nullptr;
            if (EPI == UV_EPI_GATE_RESID_F32) {
                const int t = p.gate_tid ? p.gate_tid[m] : 0;
                grow = p.gate + (long)t * p.gate_stride;
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int n = n0 + wn * (BN / WN) + i * 16 + 4 * fq;
                if (n >= p.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e];
                if (p.bias) {
                    const u32x2 bb = *(const u32x2*)(p.bias + n);
                    v[0] += bf2f((bf16_t)(bb[0] & 0xffff));
                    v[1] += bf2f((bf16_t)(bb[0] >> 16));
                    v[2] += bf2f((bf16_t)(bb[1] & 0xffff));
                    v[3] += bf2f((bf16_t)(bb[1] >> 16));
                }
                if (EPI == UV_EPI_BF16) {
                    u32x2 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
                    *(u32x2*)((bf16_t*)p.out + (long)m * p.ldo + n) = o;
                } else if (EPI == UV_EPI_GELU_BF16) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f32(round_bf(v[e]));
                    u32x2 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
                    *(u32x2*)((bf16_t*)p.out + (long)m * p.ldo + n) = o;
                } else if (EPI == UV_EPI_F32_FROM_BF16) {
                    f32x4 o = {round_bf(v[0]), round_bf(v[1]), round_bf(v[2]), round_bf(v[3])};
                    *(f32x4*)((float*)p.out + (long)m * p.ldo + n) = o;
                } else if (EPI == UV_EPI_RESID_F32) {
                    float* xp = (float*)p.out + (long)m * p.ldo + n;
                    f32x4 x = *(const f32x4*)xp;
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = __fadd_rn(x[e], round_bf(v[e]));
                    *(f32x4*)xp = x;
                } else if (EPI == UV_EPI_GATE_RESID_F32) {
                    float* xp = (float*)p.out + (long)m * p.ldo + n;
                    f32x4 x = *(const f32x4*)xp;
                    const f32x4 g = *(const f32x4*)(grow + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = __fadd_rn(x[e], __fmul_rn(round_bf(v[e]), g[e]));
                    *(f32x4*)xp = x;
                }
            }
        }
    } else {
        // transposed product: lane holds m = mb + 4*fq + {0..3} for n = nb + frow
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int n = n0 + wn * (BN / WN) + i * 16 + frow;
            if (n >= p.N) continue;
            const float b = p.bias ? bf2f(p.bias[n]) : 0.f;
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int m = m0 + wm * (BM / WM) + j * 16 + 4 * fq;
                if (m >= p.M) continue;  // M is padded to a multiple of 4 by the caller's ldo
                bf16_t* op = (bf16_t*)p.out + (long)n * p.ldo + m;
                if (m + 3 < p.M) {
                    u32x2 o = {pack_bf2(acc[i][j][0] + b, acc[i][j][1] + b),
                               pack_bf2(acc[i][j][2] + b, acc[i][j][3] + b)};
                    *(u32x2*)op = o;
                } else {
                    for (int e = 0; e < 4 && m + e < p.M; ++e) op[e] = f2bf(acc[i][j][e] + b);
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg(const GemmArgs& a0, int epi, hipStream_t stream) {
    GemmArgs a = a0;
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.N + BN - 1) / BN;
    const dim3 grid(a.tiles_m * a.tiles_n), block(WM * WN * 64);
    const size_t lds = 2 * (BM + BN) * 128;
#define UV_LAUNCH(E)                                                                               \
    case E: {                                                                                      \
        auto kern = gemm_bf16_nt_kernel<BM, BN, WM, WN, E>;                                        \
        static bool attr_set = false;                                                              \
        if (!attr_set) {                                                                           \
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                (int)lds);                                                         \
            attr_set = true;                                                                       \
        }                                                                                          \
        hipLaunchKernelGGL(kern, grid, block, lds, stream, a);                                     \
        break;                                                                                     \
    }
    switch (epi) {
        UV_LAUNCH(UV_EPI_BF16)
        UV_LAUNCH(UV_EPI_GELU_BF16)
        UV_LAUNCH(UV_EPI_F32_FROM_BF16)
        UV_LAUNCH(UV_EPI_RESID_F32)
        UV_LAUNCH(UV_EPI_GATE_RESID_F32)
        UV_LAUNCH(UV_EPI_BF16_T)
        default:
            uv_set_error("uv_gemm_bf16_nt: unknown epilogue %d", epi);
            return -1;
    }
#undef UV_LAUNCH
    UV_CHECK_LAUNCH("uv_gemm_bf16_nt");
    return 0;
}

extern "C" int uv_gemm_bf16_nt(const void* A, long lda, const void* W, long ldw, const void* bias_bf16,
                               int M, int N, int K, int epilogue, void* out, long ldo,
                               const float* gate, const int32_t* gate_tid, long gate_stride,
                               int tile_cfg, void* stream) {
    UV_CHECK_ARG(A && W && out, "uv_gemm_bf16_nt: null pointer");
    UV_CHECK_ARG(M > 0 && N > 0 && K > 0, "uv_gemm_bf16_nt: bad shape M=%d N=%d K=%d", M, N, K);
    UV_CHECK_ARG(K % UV_BK == 0, "uv_gemm_bf16_nt: K=%d must be a multiple of %d", K, UV_BK);
    UV_CHECK_ARG(N % 16 == 0, "uv_gemm_bf16_nt: N=%d must be a multiple of 16", N);
    UV_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0, "uv_gemm_bf16_nt: lda/ldw must be multiples of 8 elements");
    UV_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)out & 15) == 0,
                 "uv_gemm_bf16_nt: pointers must be 16-byte aligned");
    UV_CHECK_ARG(ldo % 4 == 0, "uv_gemm_bf16_nt: ldo must be a multiple of 4 elements");
    if (epilogue == UV_EPI_GATE_RESID_F32)
        UV_CHECK_ARG(gate && gate_stride % 4 == 0, "uv_gemm_bf16_nt: gate table required (stride %% 4 == 0)");
    GemmArgs a;
    a.A = (const bf16_t*)A; a.W = (const bf16_t*)W; a.bias = (const bf16_t*)bias_bf16;
    a.out = out; a.gate = gate; a.gate_tid = gate_tid;
    a.lda = lda; a.ldw = ldw; a.ldo = ldo; a.gate_stride = gate_stride;
    a.M = M; a.N = N; a.K = K; a.tiles_m = a.tiles_n = 0;
    hipStream_t s = (hipStream_t)stream;
    switch (tile_cfg) {
        case 0: {  // default: the big tile whose grid quantises best onto the 256 CUs (1 block per CU)
            if (M < 2048 || N < 1024) return launch_cfg<128, 128, 2, 2>(a, epilogue, s);
            const long tm = (M + 255) / 256;
            const long t256 = tm * ((N + 255) / 256), t192 = tm * ((N + 191) / 192);
            // cost ~ rounds x tile area; 256x192 tiles carry 3/4 of the work of 256x256 at slightly lower efficiency
            const double c256 = (double)((t256 + 255) / 256) * 1.00, c192 = (double)((t192 + 255) / 256) * 0.78;
            // 16 waves per workgroup (64x64 / 64x48 per wave, 4 waves per SIMD): more waves to hide the LDS and DMA
            // latency than the 8-wave layout (measured +6 % on the ffn.0 shape)
            // long-K shapes (ffn.2, K = 14336) stream A once per n-tile: the wider tile wins there despite worse quantisation
            if (N % 192 == 0 && c192 < c256 && K <= 4096) return launch_cfg<256, 192, 4, 4>(a, epilogue, s);
            return launch_cfg<256, 256, 4, 4>(a, epilogue, s);
        }
        case 1: return launch_cfg<128, 128, 2, 2>(a, epilogue, s);
        case 2: return launch_cfg<256, 256, 2, 4>(a, epilogue, s);
        case 3: return launch_cfg<256, 128, 4, 2>(a, epilogue, s);
        case 4: return launch_cfg<256, 192, 2, 4>(a, epilogue, s);
        case 5: return launch_cfg<256, 256, 4, 4>(a, epilogue, s);
        case 6: return launch_cfg<256, 192, 4, 4>(a, epilogue, s);
        default:
            uv_set_error("uv_gemm_bf16_nt: unknown tile_cfg %d", tile_cfg);
            return -1;
    }
}
