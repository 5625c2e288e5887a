/* univid_hip.h - C ABI of libunivid_hip.so (MI355X / gfx950 only).
 *
 * This is the drop-in boundary for UniVid's diffusion-decoder hot path. The reference has no native code
 * and no FFI: its seam is a set of PyTorch op sequences inside
 * models/wan/utils/modules/{model,attention,vae2_2}.py and models/wan/utils/fm_solvers_unipc.py. Each entry
 * point below replaces one such sequence (cited per function, paths relative to the reference root) and is what
 * a ctypes / pybind11 / cffi stub on the reference side would bind (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless stated; no torch types.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream). Calls are asynchronous on it and never
 *     synchronise, allocate or free (uv_init() does the one-time allocations), so they can be graph-captured.
 *   - bf16 tensors are raw uint16 bit patterns; leading dimensions (ld*) are in ELEMENTS.
 *   - return 0 on success; non-zero on a rejected argument or failed launch, with uv_last_error() describing it.
 *     Nothing is written on a rejected call. There is no CPU fallback anywhere in this library.
 */
#ifndef UNIVID_HIP_H
#define UNIVID_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- library ---------------------------------------------------------------------------------------------- */
int uv_version(void);                      /* 200 = 0.2.0 */
const char* uv_build_id(void);             /* digest of the kernel sources the library was built from (loader checks it) */
int uv_init(void);                         /* one-time allocations for the CURRENT device; call once per device before use */
const char* uv_last_error(void);           /* thread-local text of the last failure */
int uv_device_arch(char* buf, int len);    /* e.g. "gfx950:sramecc+:xnack-" */
/* Host scheduling policy of the CURRENT device: on = hipDeviceScheduleBlockingSync (a waiting host thread sleeps instead of spinning),
 * 0 = the runtime's default. Never set implicitly (process-wide policy); one-rank-per-GPU launchers call it once per device. */
int uv_host_blocking_sync(int on);

/* Developer options: process-wide switches for A/B tools and tests, set EXPLICITLY through these calls - the library never reads the
 * environment. Defaults are what production runs; none of them changes results beyond the f32 summation order noted per key.
 *   UV_OPT_CONV_HALO  -1 automatic (default) | 0 never | 1 whenever the geometry fits: which 3x3 stride-1 convolutions of uv_conv3d_* take
 *                     the LDS-halo kernel instead of the gather kernel (the two sum their k-tiles in different orders: ~1e-5 apart)
 *   UV_OPT_GEMM_GM    0 automatic (default) | 1..64: tile-walk group height of the persistent GEMM (tile order only, results unchanged)
 *   UV_OPT_ATTN_CUT   0 automatic (default) | v: flash_attn_fwd12_kernel cuts every head with v - 1 eight-unit blocks (results unchanged)
 * uv_set_option rejects unknown keys / out-of-range values; uv_reset_options restores every default. Host-only, thread-safe. */
#define UV_OPT_CONV_HALO 0
#define UV_OPT_GEMM_GM 1
#define UV_OPT_ATTN_CUT 2
int uv_set_option(int key, int value);
int uv_get_option(int key, int* value);
int uv_reset_options(void);

/* ---- DiT: GEMMs -------------------------------------------------------------------------------------------- */
/* epilogue selectors of uv_gemm_bf16_nt */
#define UV_EPI_BF16 0            /* out_bf16 = bf16(acc + bias)                               nn.Linear under autocast */
#define UV_EPI_GELU_BF16 1       /* out_bf16 = bf16(gelu_tanh(bf16(acc + bias)))              ffn.0 + GELU, model.py:212-213 */
#define UV_EPI_F32_FROM_BF16 2   /* out_f32  = float(bf16(acc + bias))                        patch_embedding, model.py:448 */
#define UV_EPI_RESID_F32 3       /* x_f32   += float(bf16(acc + bias))                        x + cross_attn(...), model.py:251 */
#define UV_EPI_GATE_RESID_F32 4  /* x_f32    = x + float(bf16(acc + bias)) * gate[tid[m]][n]  x + y*e, model.py:247,255 */
#define UV_EPI_BF16_T 5          /* outT_bf16[n][m] = bf16(acc + bias)                        V^T for uv_flash_attn_bf16 */
/* (6 = UV_EPI_BF16 + per-group sums of squares: reached through uv_gemm_bf16_nt_ssq, which carries the extra output) */

/* C[M,N] = A[M,K] . W[N,K]^T + bias, bf16 operands, fp32 accumulate (MFMA 16x16x32).
 * Replaces nn.Linear q/k/v/o, ffn.0/ffn.2, text_embedding, patch_embedding (model.py:119-122, 212-214, 378-382).
 * K % 64 == 0, N % 16 == 0; M arbitrary. bias: bf16 [N] or NULL. gate/gate_tid only for UV_EPI_GATE_RESID_F32
 * (gate: f32 [n_t, gate_stride]; gate_tid: int32 [M] or NULL = row 0). tile_cfg 0 = auto. */
int uv_gemm_bf16_nt(const void* A, long lda, const void* W, long ldw, const void* bias_bf16, int M, int N, int K,
                    int epilogue, void* out, long ldo, const float* gate, const int32_t* gate_tid, long gate_stride,
                    int tile_cfg, void* stream);

/* uv_gemm_bf16_nt with a caller-owned workspace: the rows a large projection leaves after whole rounds of 256x256 tiles (1 120 of
 * the 22 880 rows of the CFG pair at 49 x 704 x 1280) run, for K >= 8192 (ffn.2, model.py:214,255), as ONE round of 256x256 tiles x
 * split-K 4 instead of 128x128 tiles: four f32 partial tiles per output tile, published through the workspace and summed IN SLICE ORDER
 * by the last-arriving workgroup, which then applies the ordinary epilogue. Deterministic (same bits run to run); the strip's rows differ
 * from uv_gemm_bf16_nt's by the f32 summation order only (4 partial sums instead of one chain; <= 1 ulp of the epilogue's 16-bit rounding).
 * workspace: device memory, 256-byte aligned, workspace_bytes >= uv_gemm_splitk_ws_bytes(M, N, K); contents need no initialisation and
 * are scratch afterwards; launches that share a workspace must be ordered (same stream). NULL / too small: exactly uv_gemm_bf16_nt.
 * tile_cfg 19 / 20 (tests, tools): the whole problem as split-K 4 / 2 (epilogues 0, 3, 4; workspace 4096 + tiles x split x 256 KiB);
 * 21 = 19 with one K range per XCD instead of one tile per XCD (placement A/B, same results). */
int uv_gemm_bf16_nt_ws(const void* A, long lda, const void* W, long ldw, const void* bias_bf16, int M, int N, int K,
                       int epilogue, void* out, long ldo, const float* gate, const int32_t* gate_tid, long gate_stride,
                       int tile_cfg, void* workspace, long workspace_bytes, void* stream);
long uv_gemm_splitk_ws_bytes(int M, int N, int K);   /* 0: tile_cfg 0 has no split-K strip for this shape */

/* uv_gemm_bf16_nt with IEEE fp16 operands, bias and 16-bit outputs (fp32 accumulate): the dtype the reference runs the SigLIP2
 * ranker in (models/BAGEL/eval_understanding.py:172,181,191: fp16 autocast). Same epilogues; tile_cfg must be 0. */
int uv_gemm_f16_nt(const void* A, long lda, const void* W, long ldw, const void* bias_f16, int M, int N, int K, int epilogue,
                   void* out, long ldo, const float* gate, const int32_t* gate_tid, long gate_stride, int tile_cfg, void* stream);

/* The q projection whose RMSNorm the attention kernel applies itself (WanSelfAttention / WanCrossAttention: q = norm_q(self.q(x)),
 * model.py:138, 169; WanRMSNorm over ALL heads' columns, :82-85): out = bf16(A W^T + bias) exactly as UV_EPI_BF16, and in the same pass
 * ssq[m][g] = sum over the 32 columns n in [32 g, 32 g + 32) of float(out[m][n])^2, f32 [M, ld_ssq], N % 32 == 0, ld_ssq >= N / 32.
 * Every group is summed in ONE fixed order whatever kernel and tile shape computes it (quads of 4 consecutive columns left to right, then
 * (s_k + s_{k+4}) pairs, then a balanced tree), so the values do not depend on the schedule. uv_rms_scale_from_ssq turns them into the
 * per-row scale uv_flash_attn_bf16_qnorm consumes: one pass over q (read + write of [M, N] bf16) is gone. */
int uv_gemm_bf16_nt_ssq(const void* A, long lda, const void* W, long ldw, const void* bias_bf16, int M, int N, int K, void* out, long ldo,
                        float* ssq, long ld_ssq, int tile_cfg, void* stream);
/* rs[m] = 1 / sqrt(sum_g ssq[m][g] / C + eps), groups added in a fixed order (four ascending quarter sums, then (p0 + p1) + (p2 + p3)):
 * WanRMSNorm's x.pow(2).mean(-1) + eps, rsqrt (model.py:85). */
int uv_rms_scale_from_ssq(const float* ssq, long ld_ssq, int M, int groups, int C, float eps, float* rs, void* stream);

/* C[M,N] = A[M,K] . W[N,K]^T + bias (+ resid), all fp32, exact-f32 MFMA (16x16x4).
 * Replaces Head.head (fp32 island, model.py:286-290) and the VAE's 1x1 convolutions (vae2_2.py:211,249-250,766-767).
 * K % 4 == 0, N % 4 == 0. */
int uv_gemm_f32_nt(const float* A, long lda, const float* W, long ldw, const float* bias, int M, int N, int K, float* out,
                   long ldo, const float* resid, long ldr, void* stream);

/* ---- DiT: attention ---------------------------------------------------------------------------------------- */
/* out[Lq, H*D] = softmax(q k^T * softmax_scale) v per head, non-causal, all Lk keys attended, for `batch` independent
 * samples. Replaces flash_attention() (attention.py:24-130) as called at model.py:145-150 and :175.
 * q [batch*Lq, ldq], k [batch*Lk, ldk] bf16 (samples stacked along the token axis) with head h at columns
 * [h*D, (h+1)*D); vt = V TRANSPOSED, bf16 [H*D, ldvt]: sample b's keys are columns [b*Lk, (b+1)*Lk) (so one
 * UV_EPI_BF16_T GEMM over the stacked rows produces it), ldvt >= (batch-1)*Lk + roundup(Lk, 64), every column up to
 * that bound finite, Lk % 8 == 0 when batch > 1; out [batch*Lq, ldo]; head_dim D in {64, 128}. */
int uv_flash_attn_bf16(const void* q, long ldq, const void* k, long ldk, const void* vt, long ldvt, void* out, long ldo,
                       int batch, int Lq, int Lk, int H, int head_dim, float softmax_scale, void* stream);

/* uv_flash_attn_bf16 on the RAW q projection: the Q prologue of every kernel applies norm_q - q'[m][c] = bf16( bf16(q[m][c] * q_rs[m]) * q_weight[c] ),
 * the roundings of uv_rmsnorm_rope - before the first MFMA. q_rs f32 [batch * Lq] (uv_rms_scale_from_ssq), q_weight f32 [H * D] (norm_q.weight).
 * No rotary embedding: the cross-attention form (model.py:160-180). k / vt as for uv_flash_attn_bf16 (k already normalised). */
int uv_flash_attn_bf16_qnorm(const void* q, long ldq, const void* k, long ldk, const void* vt, long ldvt, void* out, long ldo,
                             int batch, int Lq, int Lk, int H, int head_dim, float softmax_scale, const float* q_rs, const float* q_weight,
                             void* stream);

/* uv_flash_attn_bf16 with IEEE fp16 q / k / vt / out (SigLIP2 ranker: transformers' attention under fp16 autocast). */
int uv_flash_attn_f16(const void* q, long ldq, const void* k, long ldk, const void* vt, long ldvt, void* out, long ldo,
                      int batch, int Lq, int Lk, int H, int head_dim, float softmax_scale, void* stream);

/* Name of the kernel uv_flash_attn_bf16 (f16 = 0) / uv_flash_attn_f16 (f16 = 1) dispatches for a problem of this geometry
 * (selection depends on Lk, head_dim and the leading dimensions only); for benchmark / profile labels. Host-only. */
int uv_flash_attn_kernel_name(int Lk, int head_dim, long ldk, long ldvt, int f16, char* buf, int len);

/* Operator-seam helpers: what flash_attention(q, k, v, ...) (attention.py:24-130) does around the core when it is handed
 * [B, L, N, C] tensors of any float dtype: `half(x)` casts (:59-83), and `.type(out_dtype)` of the result (:130); the
 * transpose produces the V^T operand of uv_flash_attn_* from token-major V rows.
 * uv_transpose_16: out[c][l] = in[l][c] (16-bit elements, l < L, c < C); columns L .. Lpad-1 are written as zero. */
int uv_transpose_16(const void* in, long ldi, void* out, long ldo, int L, int C, int Lpad, void* stream);
/* contiguous f32 -> bf16 (f16 = 0) / IEEE fp16 (f16 = 1), round to nearest even; and the exact widening back */
int uv_cast_f32_to16(const float* in, void* out, long n, int f16, void* stream);
int uv_cast_16_to_f32(const void* in, float* out, long n, int f16, void* stream);

/* umT5 text-encoder attention (models/wan/utils/modules/t5.py:93-120), one prompt of n <= 1024 tokens, head_dim 64, bf16:
 * scores = bf16(bf16(q k^T) + rel_bias[h][key - query]) (no scaling), P = bf16(softmax_fp32(scores)), out = bf16(P v) - the
 * reference's rounding points. q, k, v, out [n, H*64]; rel_bias fp32 [H, 2*span - 1] (bf16 values of T5RelativeEmbedding for
 * relative positions -(span-1) .. span-1), span >= n. */
int uv_t5_attention_bf16(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* out, long ldo, int n, int H,
                         const float* rel_bias, int span, void* stream);

/* ---- DiT: fused HBM-bound glue ------------------------------------------------------------------------------ */
/* LayerNorm(no affine) over C then mode 0: y | 1: y*(1+scale[t])+shift[t] (t = tid[row]) | 2: y*w+b.
 * Replaces WanLayerNorm + AdaLN modulation (model.py:93-98, 239-245, 253, 287-290). round_ln=1 rounds y to bf16 first
 * (block 0, whose residual stream is still bf16). out is f32 (out_bf16=0), bf16 (1) or IEEE fp16 (2). C % 256 == 0, C <= 8192. */
int uv_layernorm_mod(const float* x, long ldx, void* out, long ldo, int L, int C, float eps, int mode, const float* tab,
                     long tab_stride, int shift_off, int scale_off, const int32_t* tid, const float* w, const float* b,
                     int round_ln, int out_bf16, void* stream);

/* out = bf16( RoPE( float(bf16(x * rsqrt(mean(x^2)+eps))) * weight ) ); RoPE (complex128 table `freqs`
 * [1024, head_dim/2] as (re,im) doubles) is skipped when freqs == NULL and for tokens >= F*Hh*Ww. Row i is token
 * row0 + i of the sequence (row0 > 0 for a sequence-parallel shard, distributed/sequence_parallel.py:47-56).
 * Replaces WanRMSNorm (model.py:82-85) + rope_apply (model.py:38-66) + the bf16 cast of attention.py:59-83. */
int uv_rmsnorm_rope(const void* x, long ldx, void* out, long ldo, const float* weight, int L, int C, int head_dim,
                    float eps, const double* freqs, int F, int Hh, int Ww, int row0, void* stream);
/* uv_rmsnorm_rope for q AND k of a self-attention in one launch (model.py:138-139, 146-147): same geometry, own norm weights;
 * the L rows hold L / Ls stacked samples whose RoPE positions restart every Ls rows. */
int uv_rmsnorm_rope_qk(const void* q, void* q_out, const float* q_weight, const void* k, void* k_out, const float* k_weight,
                       long ldx, long ldo, int L, int Ls, int C, int head_dim, float eps, const double* freqs, int F, int Hh,
                       int Ww, int row0, void* stream);

/* latent [Cin,F,H,W] f32 -> im2col rows [L, Kpad] bf16, column order (c,kt,kh,kw) = Conv3d.weight.flatten(1)
 * (model.py:378-379, 448-451). */
int uv_patchify_bf16(const float* x, void* out, long ldo, int Cin, int F, int H, int W, int pt, int ph, int pw, int Kpad,
                     void* stream);
/* head rows [L, pt*ph*pw*Cout] f32 -> [Cout, Fp*pt, Hp*ph, Wp*pw] f32 (WanModel.unpatchify, model.py:499-522). */
int uv_unpatchify_f32(const float* in, long ldi, float* out, int Cout, int Fp, int Hp, int Wp, int pt, int ph, int pw,
                      void* stream);
/* sinusoidal_embedding_1d (model.py:14-24): out[n, dim] f32 = cos||sin(t * 10000^(-i/half)) evaluated in fp64. */
int uv_sinusoid_f32(const float* t, float* out, int n, int dim, void* stream);
/* out[r][n] = act_in(x[r]) . W[n] + b[n], fp32, for the few distinct timestep rows (time_embedding / time_projection,
 * model.py:384-386, 465-468). act_in: 0 none, 1 SiLU. */
int uv_linear_rows_f32(const float* x, long ldx, const float* W, const float* b, float* out, long ldo, int R, int N, int K,
                       int act_in, void* stream);
/* out[r][i] = mod[i] + e0[r][i]  (modulation + e, model.py:239, 287) */
int uv_add_rows_f32(const float* mod, const float* e0, float* out, int R, long n, void* stream);
int uv_cast_f32_bf16(const float* in, void* out, long n, void* stream);
/* x_f32 += float(y_bf16): un-fused residual used when WanCrossAttention.forward has been re-assigned (UniVid's hook). */
int uv_add_bf16_resid(float* x, long ldx, const void* y, long ldy, int L, int C, void* stream);
/* UniVid's dynamic text weight on the embedded context of ONE sample (Wan22ContextWrapper's per-layer hook,
 * models/model_pipeline.py:1779-1797: `context * weight_mask`, weight_mask = ones_like(context) with rows [0, text_len) `*= w`):
 * out[r] = bf16(in[r] * bf16(w)) for r < n_scaled, out[r] = in[r] for n_scaled <= r < R. in / out bf16 [R, C], C % 8 == 0; in == out allowed.
 * The result is what the K / V projections of the hooked layers read (WanModel.set_text_weight). */
int uv_text_weight_rows_bf16(const void* in, long ldi, void* out, long ldo, int R, int n_scaled, int C, float w, void* stream);

/* out[r] = x[r] / max(||x[r]||_2, eps): torch.nn.functional.normalize(dim=-1) of the SigLIP2 ranker's embeddings
 * (models/BAGEL/eval_understanding.py:185,195). */
int uv_l2_normalize_rows_f32(const float* x, long ldx, float* out, long ldo, int R, int C, float eps, void* stream);

/* ContextProjector glue (models/model_pipeline.py:1516-1523, 1532-1549; the module is bf16): exact (erf) GELU of a bf16 tensor,
 * and F.interpolate(mode='linear', align_corners=False) along the token axis of [Lin, C] bf16 rows. */
int uv_gelu_erf_bf16(const void* in, void* out, long n, void* stream);
int uv_interp_linear_rows_bf16(const void* in, long ldi, void* out, long ldo, int Lin, int Lout, int C, void* stream);

/* umT5 encoder glue (t5.py, bf16 module): bf16 residual add (:175-176); fc1 * GELU(gate) with the reference's op-by-op bf16 GELU
 * (:46-50, :138). */
int uv_add_bf16(const void* x, const void* y, void* out, long n, void* stream);
int uv_t5_gated_gelu_bf16(const void* gate, const void* fc1, void* out, long n, void* stream);

/* ---- sampler (CFG + flow UniPC order 2 / bh2 / predict-x0) -------------------------------------------------- */
/* noise_pred = uncond + gs*(cond - uncond) (textimage2video.py:385); x0 = sample - sigma*noise_pred
 * (fm_solvers_unipc.py:323). noise_pred may be NULL. */
int uv_cfg_convert(const float* cond, const float* uncond, const float* sample, float guide_scale, float sigma,
                   float* noise_pred, float* x0, long n, void* stream);
/* multistep_uni_c_bh_update (fm_solvers_unipc.py:545-628) with host-computed fp32 coefficients. */
int uv_unipc_corrector(const float* x_last, const float* m0, const float* m_prev, const float* model_t, float* out, float r,
                       float c1, float c2, float rho0, float rho_last, float rk, int order, long n, void* stream);
/* multistep_uni_p_bh_update (fm_solvers_unipc.py:397-486). */
int uv_unipc_predictor(const float* x, const float* m0, const float* m_prev, float* out, float r, float c1, float c2,
                       float rk, int order, long n, void* stream);
/* DPM-Solver++ update of sample_solver='dpm++' (textimage2video.py:343-351): dpm_solver_first_order_update (fm_solvers.py:417-485,
 * order 1) and multistep_dpm_solver_second_order_update, midpoint (fm_solvers.py:488-595, order 2; m1 = the older x0), with the
 * host-computed fp32 scalars r = sigma_t/sigma_s0, c = alpha_t*(exp(-h)-1), inv_r0 = 1/r0. */
int uv_dpmpp_update(const float* x, const float* m0, const float* m1, float* out, float r, float c, float inv_r0, int order,
                    long n, void* stream);

/* ---- VAE (fp32, channels-last [T, H, W, C]) -------------------------------------------------------------------- */
/* Causal 3D / 2D convolution as implicit GEMM on the exact-f32 MFMA. Replaces CausalConv3d (vae2_2.py:17-42) and the
 * Resample convolutions (vae2_2.py:86-108, 153-155). in: input ring [Tin, Hin, Win, ld_in]; w: [Cout, kt*kh*kw*Cin]
 * (tap-major, channel-minor); out rows = output pixels (t,h,w) row-major, [M, ldo]. Input frame of tap dt for output
 * frame t is t*st + dt + t_off; rows/cols are h*sh + dh - ph, w*sw + dw - pw, out-of-range taps contribute zero.
 * up=1: the input is read through a nearest-exact 2x spatial upsample (Hin/Win are the PRE-upsample sizes).
 * up=2+2a+b (a, b in {0,1}): ONE OUTPUT PHASE of such an upsampling 3x3 convolution. On the 2x nearest-upsampled image the three taps
 * of a row collapse onto two source rows (phase a=0: {dy=0} -> y-1, {1,2} -> y; a=1: {0,1} -> y, {2} -> y+1; columns alike), so output
 * pixels (2y+a, 2x+b) are a 2x2 convolution of the SOURCE image with the pre-summed weights of the collapsed taps: 16 instead of 36
 * multiply-adds per four output pixels. The launch takes that 2x2 kernel (kh = kw = 2, ph = 1-a, pw = 1-b), Hout/Wout = the SOURCE
 * resolution, and stores pixel (t, y, x) at (t, 2y+a, 2x+b) of the [Tout, 2 Hout, 2 Wout, ldo] output; four launches make the layer.
 * interleave=1: output channel halves become consecutive frames (Resample.time_conv, vae2_2.py:143-151).
 * Cin % 32 == 0 (zero-pad channels), Cout % 4 == 0. */
int uv_conv3d_f32(const float* in, long ld_in, int Tin, int Hin, int Win, const float* w, const float* bias, float* out,
                  long ldo, int Tout, int Hout, int Wout, int Cin, int Cout, int kt, int kh, int kw, int st, int sh, int sw,
                  int t_off, int ph, int pw, int up, int interleave, const float* resid, long ldr, void* stream);
/* uv_conv3d_f32 with the products computed on the bf16 matrix pipe by exact three-way splitting of both f32 operands (x = x0 + x1 + x2,
 * round-to-nearest bf16 planes; the six terms with i + j <= 2; f32 accumulate): per-product error below 2^-26 (under f32 rounding).
 * Activations / bias / residual / output as in uv_conv3d_f32 (split in registers); w_split6 = uv_split_weights_bf16x6 of the f32
 * weights. Replaces the same reference lines as uv_conv3d_f32. */
int uv_conv3d_bf16x6(const float* in, long ld_in, int Tin, int Hin, int Win, const void* w_split6, const float* bias, float* out, long ldo,
                     int Tout, int Hout, int Wout, int Cin, int Cout, int kt, int kh, int kw, int st, int sh, int sw, int t_off, int ph,
                     int pw, int up, int interleave, const float* resid, long ldr, void* stream);
/* w [rows][K] f32 (K % 32 == 0) -> [rows][K/32][32 p0 | 32 p1 | 32 p2] bf16, w = p0 + p1 + p2 exactly (n = rows * K elements in,
 * 3 n bf16 out). */
int uv_split_weights_bf16x6(const float* w, void* out, long n, void* stream);
/* The same convolution in split-bf16 ("bf16x3") arithmetic: x*w ~ xh*wh + xh*wl + xl*wh on the bf16 MFMA with f32 accumulate
 * (relative error per product ~1e-5; up to 16/3 x the f32-MFMA rate). w_split comes from uv_split_weights_bf16x3.
 * in_split=1: the input ring already holds split activations ([C/32][32 hi | 32 lo] bf16 per pixel, as written by
 * uv_vae_rms_silu(split_out=1)); in_split=0: f32 activations, split in registers. */
int uv_conv3d_bf16x3(const float* in, long ld_in, int Tin, int Hin, int Win, const void* w_split, const float* bias, float* out,
                     long ldo, int Tout, int Hout, int Wout, int Cin, int Cout, int kt, int kh, int kw, int st, int sh, int sw,
                     int t_off, int ph, int pw, int up, int interleave, const float* resid, long ldr, int in_split, void* stream);
/* w [n] f32 (rows of K, K % 32 == 0) -> [n/32][32 hi | 32 lo] bf16 with hi = bf16(w), lo = bf16(w - hi) */
int uv_split_weights_bf16x3(const float* w, void* out, long n, void* stream);
/* The same convolution, f32-GRADE, in THREE fp16 MFMA passes ("f16x3"): both operands pre-split into two IEEE fp16 pieces (hi = fp16(x),
 * lo = fp16(x - hi): x to 2^-22 relative), x*w ~ xh*wh + xh*wl + xl*wh on v_mfma_f32_16x16x32_f16, f32 accumulate. Against an fp64
 * convolution it is as close as uv_conv3d_f32: at K = 27 C both are dominated by the f32 accumulation error (tests: test_conv3d_f16x3_is_f32_grade).
 * `in` holds PRE-SPLIT activations ([C/32][32 hi | 32 lo] fp16 per pixel = the bytes of an f32 pixel; written by uv_vae_rms_silu(split_out=2);
 * |x| < 65 504: an RMS-normalised row is bounded by sqrt(C) max|gamma|, which the caller checks); w_split = uv_split_weights_f16x3(w, w_scale)
 * with w_scale a power of two (max|w| * w_scale in [2^13, 2^14) keeps the lo pieces normal); out = acc / w_scale + bias (+ resid), f32.
 * act_scale: NULL, or a DEVICE scalar the result is multiplied by too - the 1 / s of activations split by uv_vae_split_f16 under a
 * per-tensor power-of-two scale (convolutions whose input is not an RMS_norm output: Resample, vae2_2.py:86-108).
 * Replaces the same reference lines as uv_conv3d_f32 for the convolutions that follow an RMS_norm (ResidualBlock, heads: vae2_2.py:193-235). */
int uv_conv3d_f16x3(const float* in, long ld_in, int Tin, int Hin, int Win, const void* w_split, const float* bias, float* out, long ldo,
                    int Tout, int Hout, int Wout, int Cin, int Cout, int kt, int kh, int kw, int st, int sh, int sw, int t_off, int ph,
                    int pw, int up, int interleave, const float* resid, long ldr, float w_scale, const float* act_scale, void* stream);
/* x [P, C] f32 rows (any feature map; C % 32 == 0) -> `out`: the same bytes per pixel holding [C/32][32 hi | 32 lo] IEEE fp16 pieces of
 * x * s, s a per-tensor power of two found on the device: 1 while max|x| < 2^15 (then this equals uv_vae_rms_silu's split_out=2 format of
 * the same values), else the scale that puts max|x| into [2^14, 2^15). scale: two device floats; scale[0] <- 1 / s for uv_conv3d_f16x3's
 * act_scale, scale[1] = work space. Three stream-ordered operations (memset; split under s = 1 with the maximum found in the same pass; a second
 * launch that writes 1 / s and re-splits only if s != 1); no host round trip. `out` must not alias `x`. */
int uv_vae_split_f16(const float* x, long ld, float* out, long ld_out, long P, int C, float* scale, void* stream);
/* w [n] f32 (rows of K, K % 32 == 0) -> [n/32][32 hi | 32 lo] IEEE fp16 with hi = fp16(w * scale), lo = fp16(w * scale - hi); scale = 2^s */
int uv_split_weights_f16x3(const float* w, void* out, long n, float scale, void* stream);
/* y = x / max(||x||,1e-12) * sqrt(C) * gamma [-> SiLU] per pixel (RMS_norm + SiLU, vae2_2.py:45-59, 201-206).
 * split_out: 0 = f32 rows; 1 = [C/32][32 hi | 32 lo] bf16 pieces (uv_conv3d_bf16x3 in_split); 2 = the same layout in IEEE fp16 (uv_conv3d_f16x3). */
int uv_vae_rms_silu(const float* in, long ld_in, const float* gamma, float* out, long ld_out, long P, int C, int do_silu,
                    int split_out, void* stream);
/* in-place row softmax of x*scale (AttentionBlock's SDPA, vae2_2.py:267-271) */
int uv_softmax_rows_f32(float* x, long ld, int R, int n, float scale, void* stream);
/* out += DupUp3D(x) (vae2_2.py:390-412); out += AvgDown3D(x) (vae2_2.py:335-367) */
int uv_vae_dupup_add(const float* x, float* out, int T, int H, int W, int Cin, int Cout, int ft, int drop, void* stream);
int uv_vae_avgdown_add(const float* x, float* out, int T, int H, int W, int Cin, int Cout, int ft, int fs, void* stream);
/* latent / video layout conversions at the VAE boundary (vae2_2.py:280-313, 803-808, 814-818, 1045) */
int uv_vae_latent_in(const float* z, const float* mean, const float* inv_std, float* out, long ld, int Z, long P, void* stream);
int uv_vae_latent_out(const float* mu, long ld, const float* mean, const float* inv_std, float* out, int Z, long P, void* stream);
int uv_vae_video_in(const float* vid, float* out, long ld, int F, int H, int W, int f0, int T, void* stream);
int uv_vae_video_out(const float* y, long ld, float* vid, int F, int Hp, int Wp, int f0, int T, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UNIVID_HIP_H */
