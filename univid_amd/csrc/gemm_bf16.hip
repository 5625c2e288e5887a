// bf16 / fp16 "NT" GEMM entry points: C[M,N] = A[M,K] . W[N,K]^T (+ bias) with the fused epilogues the Wan DiT block needs.
// Kernels and launch helpers: gemm_bf16_kernels.h (the reference lines each epilogue replaces are cited there). This file holds the
// shape-based choice the product path uses (tile_cfg 0), the leftover-row split, and the numbered configurations the choice is made
// of; configurations that only tests and developer tools select live in gemm_bf16_diag.hip.
#include "gemm_bf16_kernels.h"

// gemm_bf16_diag.hip: tile_cfg 2, 3, 4, 10, 11, 13, 14 (A/B and race-screen references; bf16 only)
int uv_gemm_diag_launch(const GemmArgs& a, int epilogue, int tile_cfg, hipStream_t s);

static int num_cus() { return uv_num_cus(); }

template <bool F16>
static int launch_by_cfg(const GemmArgs& a, int epilogue, int tile_cfg, hipStream_t s);

// The caller's split-K workspace (uv_gemm_bf16_nt_ws), or nothing.
struct GemmWs {
    void* p = nullptr;
    long bytes = 0;
};

// Does the leftover strip `at` run as ONE round of 256x256 tiles x split-K 4? Only where that measured faster than the 128x128 ring
// (tools/gemm_bench.py, round 6): the long-K strips (ffn.2: K = 14 336, 144 -> 121 us in isolation); at K = 3 072 a slice is 12 K tiles
// and the publish + combine (~40 us: 63 MB of partial tiles out and back in) costs more than the whole ring launch (63 against 35 us). Needs the residual / bf16 epilogues, whole slices, at most one round of workgroups, and a
// caller-provided workspace.
template <bool F16>
static bool strip_takes_splitk(const GemmArgs& at, int epilogue, const GemmWs& ws) {
    if (F16 || !ws.p) return false;
    if (epilogue != UV_EPI_BF16 && epilogue != UV_EPI_RESID_F32 && epilogue != UV_EPI_GATE_RESID_F32) return false;
    if (at.K < 8192 || at.K % 512 != 0 || at.N % 256 != 0) return false;
    const long tiles = (long)((at.M + 255) / 256) * (at.N / 256);
    return tiles * 4 <= num_cus() && ws.bytes >= splitk_ws_bytes(at.M, at.N, 4) && ((uintptr_t)ws.p & 255) == 0;
}

// The rows of an [M, N] projection (N % 256 == 0) that the persistent kernel takes under tile_cfg 0: every whole 256-row tile - or, when the
// tile count is just above a whole number of rounds (the partial round under 45 % full: *short_round), only the row tiles of the whole rounds.
// ONE place: gemm_entry cuts its launches here and uv_gemm_splitk_ws_bytes sizes the strip's workspace from the same cut.
static long persistent_rows(int M, int N, bool* short_round) {
    const long tn = N / 256, tm_full = M / 256;
    const long tiles = (long)((M + 255) / 256) * tn;
    const long rounds = tiles / num_cus(), rest = tiles - rounds * num_cus();
    *short_round = rounds >= 1 && rest > 0 && rest * 100 < 45 * num_cus();
    return *short_round ? rounds * num_cus() / tn * 256 : tm_full * 256;
}

// Rows that fill whole rounds of 256x256 tiles go to the big-tile kernel; the leftover rows (which would otherwise cost a
// whole extra round on a fraction of the CUs) run as 128x128 tiles. Same arithmetic per element either way.
// m_main > 0 names the split point explicitly (a multiple of 256).
template <bool F16>
static int launch_m_split(const GemmArgs& a, int epilogue, int main_cfg, hipStream_t s, long m_main = 0, const GemmWs& ws = GemmWs()) {
    const long tiles_n = (a.N + 255) / 256, tiles_m = (a.M + 255) / 256;
    if (m_main <= 0) {
        const long rounds = tiles_m * tiles_n / num_cus();
        m_main = rounds * num_cus() / tiles_n * 256;
        if (rounds == 0) m_main = 0;
    }
    if (m_main <= 0 || m_main >= a.M) return launch_by_cfg<F16>(a, epilogue, main_cfg, s);
    GemmArgs am = a, at = a;
    am.M = (int)m_main;
    at.M = a.M - (int)m_main;
    at.A = a.A + m_main * a.lda;
    if (a.gate_tid) at.gate_tid = a.gate_tid + m_main;
    if (epilogue == UV_EPI_BF16_T) at.out = (bf16_t*)a.out + m_main;
    else if (epilogue == UV_EPI_BF16 || epilogue == UV_EPI_GELU_BF16 || epilogue == UV_EPI_BF16_SSQ) at.out = (bf16_t*)a.out + m_main * a.ldo;
    else at.out = (float*)a.out + m_main * a.ldo;
    if (a.ssq) at.ssq = a.ssq + m_main * a.ld_ssq;
    const int rc = launch_by_cfg<F16>(am, epilogue, main_cfg, s);
    if (rc) return rc;
    if constexpr (!F16)
        if (strip_takes_splitk<F16>(at, epilogue, ws)) return launch_8ph_splitk<4, false>(at, epilogue, s, ws.p, ws.bytes);
    return launch_by_cfg<F16>(at, epilogue, 12, s);
}

template <bool F16>
static int gemm_entry(const void* A, long lda, const void* W, long ldw, const void* bias_bf16, int M, int N, int K, int epilogue,
                      void* out, long ldo, const float* gate, const int32_t* gate_tid, long gate_stride, int tile_cfg, void* stream,
                      float* ssq = nullptr, long ld_ssq = 0, const GemmWs& ws = GemmWs()) {
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
    a.ssq = ssq; a.ld_ssq = ld_ssq;
    a.ws_slab = nullptr; a.ws_cnt = nullptr; a.gm = 0;
    if (epilogue == UV_EPI_BF16_SSQ)
        UV_CHECK_ARG(ssq && N % 32 == 0 && ld_ssq >= N / 32 && ldo % 8 == 0, "uv_gemm_bf16_nt_ssq: needs N %% 32 == 0, ld_ssq >= N / 32, ldo %% 8 == 0 (N=%d ld_ssq=%ld ldo=%ld)", N, ld_ssq, ldo);
    a.zeros = uv_zero_page();
    UV_CHECK_ARG(a.zeros, "uv_gemm_bf16_nt: zero page missing (call uv_init)");
    hipStream_t s = (hipStream_t)stream;
    if (tile_cfg == 0 && M >= 2048 && N >= 1024 && N % 256 == 0 && K % 128 == 0 && K >= 384 &&
        (ldo % 8 == 0 || epilogue == UV_EPI_BF16_T || (epilogue >= UV_EPI_F32_FROM_BF16 && epilogue != UV_EPI_BF16_SSQ))) {
        // Large projections: 256x256 tiles on the PERSISTENT 8-wave ping-pong kernel (one workgroup per CU walking its tile
        // list). It takes whole tiles only; rows beyond the last multiple of 256 - and, when the tile count is just above a
        // whole number of rounds, the rows of that partial round - run as 128x128 tiles on the small-tile kernel.
        const long tn = N / 256;
        bool short_round = false;
        const long m_main = persistent_rows(M, N, &short_round);
        // (also when the transposed output's leading dimension does not allow the persistent kernel's 16-byte stores)
        if (m_main / 256 * tn < 2L * num_cus() || (epilogue == UV_EPI_BF16_T && ldo % 8 != 0)) {      // under two rounds of work: the one-tile-per-workgroup launch
            if (short_round) return launch_m_split<F16>(a, epilogue, 7, s);
            return launch_by_cfg<F16>(a, epilogue, 7, s);
        }
        return launch_m_split<F16>(a, epilogue, 17, s, m_main, ws);
    }
    if constexpr (!F16) {      // tests / tools: the WHOLE problem as split-K 4 (19) or 2 (20) on the one-tile-per-workgroup ping-pong kernel
        if (tile_cfg == 19) return launch_8ph_splitk<4, false>(a, epilogue, s, ws.p, ws.bytes);
        if (tile_cfg == 20) return launch_8ph_splitk<2, false>(a, epilogue, s, ws.p, ws.bytes);
        if (tile_cfg == 21) { a.gm = 1; return launch_8ph_splitk<4, false>(a, epilogue, s, ws.p, ws.bytes); }      // A/B: one K range per XCD
    }
    if (tile_cfg == 8) return launch_m_split<F16>(a, epilogue, 7, s);
    if (tile_cfg == 9) return launch_m_split<F16>(a, epilogue, 5, s);
    if (tile_cfg == 18) return launch_m_split<F16>(a, epilogue, 17, s, (long)(M / 256) * 256 < M ? (long)(M / 256) * 256 : 0);
    return launch_by_cfg<F16>(a, epilogue, tile_cfg, s);
}

extern "C" int uv_gemm_bf16_nt(const void* A, long lda, const void* W, long ldw, const void* bias_bf16,
                               int M, int N, int K, int epilogue, void* out, long ldo,
                               const float* gate, const int32_t* gate_tid, long gate_stride,
                               int tile_cfg, void* stream) {
    return gemm_entry<false>(A, lda, W, ldw, bias_bf16, M, N, K, epilogue, out, ldo, gate, gate_tid, gate_stride, tile_cfg, stream);
}

// uv_gemm_bf16_nt with a caller-owned WORKSPACE (see include/univid_hip.h): lets the leftover-row strip of a long-K projection run as one
// round of split-K workgroups. Without a (large enough) workspace it IS uv_gemm_bf16_nt.
extern "C" int uv_gemm_bf16_nt_ws(const void* A, long lda, const void* W, long ldw, const void* bias_bf16,
                                  int M, int N, int K, int epilogue, void* out, long ldo,
                                  const float* gate, const int32_t* gate_tid, long gate_stride,
                                  int tile_cfg, void* workspace, long workspace_bytes, void* stream) {
    UV_CHECK_ARG(workspace_bytes >= 0 && (workspace || workspace_bytes == 0), "uv_gemm_bf16_nt_ws: bad workspace");
    GemmWs ws;
    ws.p = workspace; ws.bytes = workspace_bytes;
    return gemm_entry<false>(A, lda, W, ldw, bias_bf16, M, N, K, epilogue, out, ldo, gate, gate_tid, gate_stride, tile_cfg, stream, nullptr, 0, ws);
}

// Bytes of workspace with which uv_gemm_bf16_nt_ws(M, N, K, tile_cfg 0) takes its split-K strip (0: this shape has none): the
// largest strip tile_cfg 0 can cut from M rows is one round of 256x256 tiles less one row of tiles.
extern "C" long uv_gemm_splitk_ws_bytes(int M, int N, int K) {
    if (M < 2048 || N < 1024 || N % 256 != 0 || K < 8192 || K % 512 != 0) return 0;
    const long tn = N / 256;
    bool short_round = false;
    const long m_main = persistent_rows(M, N, &short_round);
    if (m_main / 256 * tn < 2L * num_cus() || m_main >= M) return 0;
    const long strip_tiles = ((M - m_main) + 255) / 256 * tn;
    return strip_tiles * 4 <= num_cus() ? splitk_ws_bytes((int)(M - m_main), N, 4) : 0;
}

// UV_EPI_BF16 plus the output's sums of squares per aligned 32-column group (see include/univid_hip.h): the q projection whose RMSNorm is applied
// by the attention kernel's prologue (uv_flash_attn_bf16_qnorm) instead of a pass of its own over q.
extern "C" int uv_gemm_bf16_nt_ssq(const void* A, long lda, const void* W, long ldw, const void* bias_bf16, int M, int N, int K,
                                   void* out, long ldo, float* ssq, long ld_ssq, int tile_cfg, void* stream) {
    return gemm_entry<false>(A, lda, W, ldw, bias_bf16, M, N, K, UV_EPI_BF16_SSQ, out, ldo, nullptr, nullptr, 0, tile_cfg, stream, ssq, ld_ssq);
}

// The same GEMM with IEEE fp16 operands / bias / 16-bit outputs (fp32 accumulate): the SigLIP2 ranker's reference dtype
// (models/BAGEL/eval_understanding.py:172). Only the automatic tile choice (tile_cfg 0) is built for fp16.
extern "C" int uv_gemm_f16_nt(const void* A, long lda, const void* W, long ldw, const void* bias_f16,
                              int M, int N, int K, int epilogue, void* out, long ldo,
                              const float* gate, const int32_t* gate_tid, long gate_stride,
                              int tile_cfg, void* stream) {
    UV_CHECK_ARG(tile_cfg == 0, "uv_gemm_f16_nt: only tile_cfg 0 (automatic) is built for fp16 operands");
    return gemm_entry<true>(A, lda, W, ldw, bias_f16, M, N, K, epilogue, out, ldo, gate, gate_tid, gate_stride, tile_cfg, stream);
}

template <bool F16>
static int launch_by_cfg(const GemmArgs& a, int epilogue, int tile_cfg, hipStream_t s) {
    const int M = a.M, N = a.N, K = a.K;
    switch (tile_cfg) {
        case 0: {  // default (only reached for shapes the split/ping-pong path in uv_gemm_bf16_nt does not take)
            if (M < 2048 || N < 1024) {
                // tall and narrow (the SigLIP2 towers: 16 384 x 768): 256x256 tiles on the ping-pong kernel beat 128x128 tiles even
                // at 3/4 of a round of workgroups (q / k / v / o 33.7 -> 30.1 us, fc2 with K = 3072 89.7 -> 71.9 us)
                if (M >= 4096 && N >= 512 && N % 256 == 0 && K % 128 == 0 && K >= 256 && a.ldo % 8 == 0 && a.M % 256 == 0 &&
                    2L * (M / 256) * (N / 256) >= num_cus())
                    return launch_8ph<5, F16>(a, epilogue, s);
                // few tiles (at most ~2 per CU): 8 waves on a 4-stage ring hide the DMA/LDS latency that one 4-wave
                // workgroup per CU leaves exposed; many tiles: 4-wave workgroups, 2-3 resident per CU
                const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
                if (t128 <= 2 * num_cus()) return launch_cfg<128, 128, 4, 2, 4, F16>(a, epilogue, s);
                return launch_cfg<128, 128, 2, 2, 2, F16>(a, epilogue, s);
            }
            const long tm = (M + 255) / 256;
            const long t256 = tm * ((N + 255) / 256), t192 = tm * ((N + 191) / 192);
            // cost ~ rounds x tile area; 256x192 tiles carry 3/4 of the work of 256x256 at slightly lower efficiency
            const double c256 = (double)((t256 + 255) / 256) * 1.00, c192 = (double)((t192 + 255) / 256) * 0.78;
            if (!F16 && N % 192 == 0 && c192 < c256 && K <= 4096 && epilogue != UV_EPI_BF16_SSQ) return launch_cfg<256, 192, 4, 4>(a, epilogue, s);
            return launch_cfg<256, 256, 4, 4, 2, F16>(a, epilogue, s);
        }
        case 1: if constexpr (!F16) return launch_cfg<128, 128, 2, 2>(a, epilogue, s); else break;
        case 5: if constexpr (!F16) return launch_cfg<256, 256, 4, 4>(a, epilogue, s); else break;
        case 6: if constexpr (!F16) return launch_cfg<256, 192, 4, 4>(a, epilogue, s); else break;
        case 12: return launch_cfg<128, 128, 4, 2, 4, F16>(a, epilogue, s);
        case 7:
            UV_CHECK_ARG(K % 128 == 0 && K >= 256, "uv_gemm_bf16_nt: tile_cfg 7 needs K %% 128 == 0 and K >= 256 (K=%d)", K);
            return launch_8ph<5, F16>(a, epilogue, s);
        case 17:
            UV_CHECK_ARG(K % 128 == 0 && K >= 384, "uv_gemm_bf16_nt: tile_cfg 17 needs K %% 128 == 0 and K >= 384 (K=%d)", K);
            return launch_8ph_persist<F16>(a, epilogue, s);
        default:
            if constexpr (!F16) return uv_gemm_diag_launch(a, epilogue, tile_cfg, s);     // test / tool configurations (gemm_bf16_diag.hip)
            else break;
    }
    uv_set_error("uv_gemm_f16_nt: tile_cfg %d is not built for fp16 operands", tile_cfg);
    return -1;
}
