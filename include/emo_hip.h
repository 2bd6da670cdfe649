/* emo_hip.h - C ABI of the MI355X (gfx950) kernels behind the Emote-hack diffusion hot path.
 *
 * The reference is Python calling Python: it has NO FFI for this path.
 * The drop-in boundary is therefore the Python surface (UNet3DConditionModel /
 * ReferenceAttentionControl / EMOAnimationPipeline, mirrored in emote_hack_amd/), and this
 * header is the C ABI underneath it.  Each entry point names the reference ATen/xformers op
 * (file:line under /root/reference) it replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - plain pointers + sizes; no C++/torch types; every pointer is a DEVICE pointer unless
 *     the name ends in _host.  Buffers are caller-owned and must stay alive until `stream`
 *     has passed the call.
 *   - activations are "rows x channels" row-major (NHWC: row = ((b*F+f)*H + y)*W + x); `ld*`
 *     are leading dimensions in ELEMENTS.  Weights are [N][K] row-major (torch Linear layout;
 *     conv weights re-laid once at load to [Cout][ky][kx][Cin]).
 *   - dtype: EMO_F32 (validation mode: f32 MFMA, exact-f32 accumulate), EMO_BF16 or EMO_F16 (IEEE half: the reference's
 *     `weight_dtype=torch.float16` configurations; same kernels, v_mfma_f32_32x32x16_f16)
 *     (production: bf16 MFMA, f32 accumulate).  Statistics / softmax / latents are always f32.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), never allocates,
 *     never synchronises, keeps no global mutable state (re-entrant per stream).
 *   - returns EMO_OK (0) or a negative emo_status; emo_last_error_string() describes the last
 *     failure on the calling thread.
 */
#ifndef EMO_HIP_H
#define EMO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { EMO_F32 = 0, EMO_BF16 = 1, EMO_F16 = 2 } emo_dtype;

typedef enum {
  EMO_OK = 0,
  EMO_ERR_BAD_SHAPE = -1,
  EMO_ERR_BAD_DTYPE = -2,
  EMO_ERR_UNSUPPORTED = -3,
  EMO_ERR_HIP = -4,
  EMO_ERR_NULL = -5
} emo_status;

int emo_version(void);
const char* emo_last_error_string(void);

/* ---- layout / elementwise ------------------------------------------------------------------ */

/* (B,C,F,H,W) f32 -> rows ((b f) h w, ldo>=C) in `dtype`; channels [C, Cpad) are written as 0.
 * Replaces einops "b c f h w -> (b f) c h w" + channels-last (resnet.py:33, attention.py:115). */
int emo_ncfhw_to_rows(const float* x, void* y, int B, int C, int F, int H, int W, int Cpad, int ldo,
                      int dtype, void* stream);
/* rows -> (B,C,F,H,W) f32 (the UNet output boundary, unet_controlnet.py:478-483). */
int emo_rows_to_ncfhw(const void* x, float* y, int B, int C, int F, int H, int W, int ldi, int dtype,
                      void* stream);
/* y[m, coff:coff+C] = x[m, 0:C]   (torch.cat along channels, unet_3d_blocks.py:629,731). */
int emo_copy_cols(const void* x, int ldx, void* y, int ldy, int coff, int64_t M, int C, int dtype, void* stream);
/* y = a + alpha*b elementwise over M x C (ControlNet residual adds, unet_controlnet.py:430-447). */
int emo_add(const void* a, int lda, const void* b, int ldb, float alpha, void* y, int ldy, int64_t M, int C,
            int dtype, void* stream);
/* dtype conversion of a contiguous buffer (src f32 <-> dst dtype); fp16_round!=0 rounds through IEEE
 * half first (bank hand-off, mutual_self_attention.py:588 `.to(float16)`). */
int emo_convert(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, int fp16_round, void* stream);
/* y = silu(x) over n contiguous elements (F.silu(temb), resnet.py:186). */
int emo_silu(const void* x, void* y, int64_t n, int dtype, void* stream);

/* sinusoidal timestep embedding (embeddings.py:28-68 get_timestep_embedding): out[b] =
 * [sin | cos](t_b * freqs) (flipped to [cos | sin] if flip_sin_to_cos).  `freqs` f32 [dim/2] is the
 * constant table exp(-ln(1e4)*arange(half)/(half - freq_shift)) built once by the host; the integer
 * timestep is consumed bit-exactly. */
int emo_timestep_embedding(const int64_t* timesteps, const float* freqs, void* out, int B, int dim,
                           int flip_sin_to_cos, int dtype, void* stream);

/* ---- normalisation ---------------------------------------------------------------------------
 * GroupNorm over NHWC rows.  An "instance" is a contiguous run of S rows normalised together:
 *   5-D joint statistics (resnet.py:180,191; unet_controlnet.py:476): N=B,   S=F*H*W
 *   per-frame            (attention.py:124; motion_module.py:147)   : N=B*F, S=H*W
 * Two launches: emo_groupnorm_stats writes row-chunk partial (sum, sumsq) per group into `partials` (caller
 * workspace of emo_groupnorm_workspace_bytes(N, S, C, G) bytes); emo_groupnorm_apply combines them in f64 in a
 * fixed order (deterministic; every block recomputes the same (mean, rstd)) and normalises (+SiLU).
 * gamma / beta are f32, 16-byte aligned. */
size_t emo_groupnorm_workspace_bytes(int N, int64_t S, int C, int G);
int emo_groupnorm_stats(const void* x, int ldx, void* partials, int N, int64_t S, int C, int G, int dtype,
                        void* stream);
int emo_groupnorm_apply(const void* x, int ldx, const void* partials, const float* gamma, const float* beta,
                        void* y, int ldy, int N, int64_t S, int C, int G, float eps, int silu, int dtype,
                        void* stream);

/* The coefficient half of emo_groupnorm_apply for consumers that normalise on the fly (emo_gemm_params.gn_coef):
 * scale_c = rstd_{n,g(c)} * gamma_c, shift_c = beta_c - mean_{n,g(c)} * scale_c from the `partials` of emo_groupnorm_stats with the
 * same fixed-order f64 combine - bit-identical to the factors emo_groupnorm_apply uses.  coef is f32 [N][2 * C], channel PAIRS
 * interleaved: coef[n][4 * j ..] = (scale_2j, scale_2j+1, shift_2j, shift_2j+1) - one 16-byte read per packed pair.  C even. */
int emo_groupnorm_coeffs(const void* partials, const float* gamma, const float* beta, float* coef, int N, int64_t S,
                         int C, int G, float eps, int dtype, void* stream);

/* The same GroupNorm in ONE launch, for instances small enough that one workgroup holds (instance, slab of whole groups)
 * in registers (<= 32 K elements per workgroup: the 8x8 level and the per-frame 16x16 norms at the bench size): one read of x, statistics in the same fixed
 * order (f32 partials, f64 mean / variance), one write.  emo_groupnorm_one_launch_ok returns 1 when the geometry fits;
 * emo_groupnorm returns EMO_ERR_UNSUPPORTED otherwise (the caller then uses the two launches above). */
int emo_groupnorm_one_launch_ok(int N, int64_t S, int C, int G, int dtype);
int emo_groupnorm(const void* x, int ldx, const float* gamma, const float* beta, void* y, int ldy, int N, int64_t S,
                  int C, int G, float eps, int silu, int dtype, void* stream);

/* GroupNorm folded into the Linear / 1x1 conv that consumes it (attention.py:124,135-146 `norm` -> `proj_in`;
 * motion_module.py:147-151): GN(x) W^T + b over an instance n = x W'_n^T + b'_n with
 *   W'_n[o, c] = W[o, c] * gamma_c * rstd_{n, g(c)}      (rounded to the compute dtype, [N][Cout][C] -> `w_out`)
 *   b'_n[o]    = b[o] + sum_c W[o, c] * beta_c - sum_c mean_{n, g(c)} * W'_n[o, c]   (f32, [N][Cout] -> `rowbias_out`)
 * from the `partials` of emo_groupnorm_stats (same fixed-order f64 combine as emo_groupnorm_apply): the normalised tensor is
 * never written or re-read; emo_gemm then runs with W = w_out, bias = rowbias_out, w_slab_rows = S, w_slab_stride = Cout * C.
 * W is [Cout][C] in `dtype`, bias f32 [Cout] or NULL. */
int emo_groupnorm_fold_linear(const void* partials, const float* gamma, const float* beta, const void* W, const float* bias,
                              void* w_out, float* rowbias_out, int N, int64_t S, int C, int G, int Cout, float eps, int dtype,
                              void* stream);

/* LayerNorm over the last dim (attention.py:279-316, motion_module.py:216-224), eps 1e-5 default.
 * Optional fused temporal positional-encoding add (motion_module.py:246-248,282-283):
 * y[row] += pe[frame(row)] with frame(row) = (row / rows_per_frame) % frames, pe f32 [max_len][C]. */
int emo_layernorm(const void* x, int ldx, const float* gamma, const float* beta, void* y, int ldy, int64_t M,
                  int C, float eps, const float* pe, int rows_per_frame, int frames, int dtype, void* stream);
/* the statistics half of it: stats[m] = (mean, 1 / sqrt(var + eps)) of row m (two-pass, f32) for the LayerNorm fold of
 * emo_gemm (emo_gemm_params.ln_stats) - a read-only pass over x. */
int emo_layernorm_stats(const void* x, int ldx, float* stats, int64_t M, int C, float eps, int dtype, void* stream);

/* ---- GEMM / convolution (MFMA) ----------------------------------------------------------------
 * C[M,N] = epilogue( A[M,K] . W[N,K]^T ).  Replaces F.linear / 1x1 conv (orig_attention.py:566-575,
 * 776,817; attention.py:82,110; resnet.py:175) and, with conv geometry, the per-frame 3x3 conv
 * (resnet.py:30-38 InflatedConv3d).  Epilogue, in order:
 *   + bias[N]  (f32, may be NULL)
 *   + rowbias[(m / rows_per_batch)][N] (f32; the `+ temb[:, :, None, None, None]` of resnet.py:188)
 *   GEGLU: W rows are interleaved (32 value rows, 32 gate rows) per 32 outputs -> out[m,j] = v*gelu_erf(g),
 *          N_out = N/2 (orig_attention.py:825-827)
 *   + residual[M, N_out] (resnet.py:205, attention.py:292-317)        * out_scale
 *   store: row-major (ldc) or TRANSPOSED per batch: Ct[(m / t_rows)][n][m % t_rows] with ld t_ld
 *          (the V^T layout the attention kernels consume). */
typedef struct {
  const void* A; int64_t lda;
  const void* W;              /* [N][K] (K = taps*Cin for conv) */
  const float* bias;
  const float* rowbias; int rows_per_batch; int ld_rowbias;
  const void* residual; int64_t ldr;
  void* C; int64_t ldc;
  int64_t M; int N; int K;
  int geglu;
  float out_scale;
  int transpose_out; int t_rows; int64_t t_ld; int64_t t_batch_stride;
  /* conv geometry (conv_taps==9 -> implicit 3x3 GEMM over NHWC A; 0 -> dense) */
  int conv_taps; int H; int W_; int Cin; int stride; int upsample2x; int Ho; int Wo;
  int dtype;
  /* split-K for problems with too few output tiles to fill 256 CUs (small M at the 8x8 / 16x16 levels):
   * split_k > 1 slices K over grid.y, f32 partial tiles go to `workspace` ([split_k][M][N] floats, at least
   * emo_gemm_workspace_bytes(p) bytes) and a second kernel reduces them in a fixed order (deterministic)
   * and applies the epilogue.  split_k <= 1: single pass, workspace unused. */
  int split_k; void* workspace;
  /* LayerNorm folded into the GEMM (attention.py:279-316, motion_module.py:216-224: every LayerNorm of the transformer
   * blocks feeds a Linear).  With ln_colsum / ln_stats != NULL the kernel computes
   *     C = epilogue( ((A - mean_m) * rstd_m) . W^T + bias ),   ln_stats[m] = (mean_m, rstd_m) from emo_layernorm_stats
   * as rstd_m * (bias[n] / rstd_m - mean_m * ln_colsum[n] + A.W^T) with ln_colsum[n] = sum_k W[n][k]: the correction enters
   * through the accumulator init, the epilogue pays one multiply, and no normalised copy of A ever exists in HBM.  The caller
   * folds the LayerNorm affine into the operands once at load: W <- W * gamma[k], bias <- bias + W . beta.  Dense, split_k <= 1. */
  const float* ln_colsum; const float* ln_stats;
  int tile;   /* 0 = planned from the shape; 1..6 pin a tile (64x64, 128x128, 128x160, 256x256, 256x160, 256x320) - tuning hook
                 in the spirit of a BLAS algorithm id; combinations a tile cannot serve fall back to the nearest one that can */
  int conv_asym;   /* conv only: 1 = padding (0, 1, 0, 1) instead of 1 all round - the `F.pad(x, (0,1,0,1))` + stride-2
                      conv of the VAE encoder's Downsample2D (diffusers AutoencoderKL; Ho = (H + 1 - 3) / stride + 1) */
  int up_h; int up_w;  /* conv only: nearest upsampling to an EXPLICIT size folded into the loader (F.interpolate(size=...),
                      resnet.py:74-82 with `output_size`: inputs that are not a multiple of 2^num_upsamplers,
                      unet_controlnet.py:357-365,456-459); source pixel = floor(dst * H / up_h).  0 = off (upsample2x covers x2) */
  int w_slab_rows; int64_t w_slab_stride;  /* dense only: per-instance weights - rows [i * w_slab_rows, (i+1) * w_slab_rows) of A
                      multiply the weight slab W + i * w_slab_stride (elements); w_slab_rows must be a multiple of 256 (a tile never
                      straddles two slabs).  `bias` is then per instance as well: f32 [M / w_slab_rows][N].  0 = one W (and one bias)
                      for every row.  Made by emo_groupnorm_fold_linear (GroupNorm folded into proj_in); row-major output with
                      N % 4 == 0 only, not with the LayerNorm fold, split-K or the conv loader. */
  /* 3x3 conv only: GroupNorm (+ SiLU) of the conv's INPUT applied inside the conv (resnet.py:180-183,191-196: norm -> nonlinearity ->
   * conv).  A holds the RAW producer output; gn_coef is the f32 [instances][2 * Cin] table of emo_groupnorm_coeffs
   * (per channel pair: scale, scale, shift, shift with scale = rstd * gamma, shift = beta - mean * scale), image i of A belongs to
   * instance i / gn_imgs_per_inst (F for the joint 5-D statistics of the resnets, 1 per frame).  The kernel normalises, activates and rounds each halo chunk in LDS right
   * after its direct-to-LDS load lands - the same arithmetic, in the same order, as emo_groupnorm_apply followed by the plain conv
   * (the padding stays zero) - so the normalised tensor is never written to or re-read from HBM.  Served by the halo-reuse kernel
   * only: emo_conv3x3_gn_fusable(p) says whether a given conv qualifies; emo_gemm returns EMO_ERR_UNSUPPORTED otherwise. */
  const float* gn_coef; int gn_imgs_per_inst; int gn_silu;
  /* dense, single pass, row-major: the columns [vt_col0, N) are stored TRANSPOSED into `vt` instead - V^T [(m / t_rows)][n - vt_col0][m % t_rows]
   * with t_ld / t_batch_stride as for transpose_out - while the columns [0, vt_col0) go to C as usual (ldc >= vt_col0): ONE launch for the
   * q | k | v projection of a self-attention whose V the attention kernel wants key-contiguous (orig_attention.py:598-600 to_q / to_k / to_v on
   * the same rows; attention.py:279-293) - the rows are read once.  The accumulator layout of the row-major kernels already has consecutive rows m in
   * consecutive lanes: a register is 32 contiguous elements of a V^T row.  vt_col0 must be a multiple of the planned tile's wave width
   * (emo_gemm_vt_ok), N - vt_col0 a multiple of 8, M a multiple of 32; with the LayerNorm fold (ln_colsum / ln_stats) only; no GEGLU / residual / row bias /
   * out_scale.  NULL = off. */
  void* vt; int vt_col0;
} emo_gemm_params;
int emo_gemm(const emo_gemm_params* p, void* stream);
/* 1 when the conv described by p (gn_* fields ignored) runs on the halo-reuse kernel, i.e. may carry gn_coef */
int emo_conv3x3_gn_fusable(const emo_gemm_params* p);
/* 1 when p (with vt / vt_col0 set) is a GEMM the split row-major | transposed store serves on the tile the planner picks for it */
int emo_gemm_vt_ok(const emo_gemm_params* p);
/* heuristic split factor for (M, N, K) and the workspace it needs */
int emo_gemm_suggest_split_k(int64_t M, int N, int K, int dtype, int geglu, int transpose_out);
size_t emo_gemm_workspace_bytes(int64_t M, int N, int split_k);

/* ---- attention -------------------------------------------------------------------------------
 * Flash-style softmax(q k^T * scale) v, scores never leave the chip.  Replaces
 * CrossAttention._attention (orig_attention.py:655-684) and xformers.memory_efficient_attention
 * (models/motionmodule.py:300, models/videonet.py:62,117).
 *   q  : [B*Lq][ldq] rows, head h at columns h*d
 *   k0 : [B*Lk0][ldk0] rows (self / context keys of batch b)
 *   v0t: V^T [B][heads*d][ldv0t] (keys contiguous, ldv0t >= Lk0, multiple of 8)
 *   k1/v1t: optional second KV segment shared by `seg1_div` consecutive batches (the ReferenceNet
 *           bank repeated over frames, mutual_self_attention.py:238-241): batch b reads bank row
 *           b / seg1_div - seg1_skip; batches b < seg1_first_batch skip it (uc rows, :243-256).
 *           seg1_skip = leading bank rows that were never materialised: under classifier-free guidance the
 *           bank row of the uncond batch is overwritten by the uc path (:243-256) and need not exist. */
typedef struct {
  const void* q; int64_t ldq;
  const void* k0; int64_t ldk0; const void* v0t; int64_t ldv0t; int Lk0;
  const void* k1; int64_t ldk1; const void* v1t; int64_t ldv1t; int Lk1;
  int seg0_div;   /* batch b reads k0/v0t row-block b / seg0_div (1 = per-batch keys; F = a text context
                     shared by the F frames of a clip, attention.py:118-119 without the repeat) */
  int seg1_div; int seg1_first_batch; int seg1_skip;
  void* out; int64_t ldo;
  int B; int Lq; int heads; int d;
  float scale;
  int dtype;
  const int32_t* seg1_row;   /* optional DEVICE int32: when non-NULL every batch b >= seg1_first_batch reads bank row
                                seg1_row[0] (seg1_div / seg1_skip unused).  A sampling loop keeps the projected banks of a
                                whole group of timesteps resident and selects the current one by writing this word - the
                                launch parameters (hence a captured hipGraph) stay the same from step to step. */
} emo_attention_params;
int emo_attention(const emo_attention_params* p, void* stream);

/* Temporal self-attention of the AnimateDiff motion module (motion_module.py:275-334): tokens
 * "(b f) d c -> (b d) f c"; qkv rows are [(b*F+f)*HW + pix][3*C] (q|k|v), output [(b*F+f)*HW+pix][C].
 * F <= 32.  The transposes are folded into the indexing. */
int emo_temporal_attention(const void* qkv, int64_t ldqkv, void* out, int64_t ldo, int B, int F, int HW,
                           int heads, int d, float scale, int dtype, void* stream);

/* ---- sampler -------------------------------------------------------------------------------
 * Fused window-average + classifier-free guidance + scheduler step
 * (EMOAnimationPipeline.py:812-817):  eps = uc + s*(c - uc) on noise_pred/counter;
 * x <- c_x*x + c_eps*eps + c_noise*z,  z = counter-based N(0,1) keyed by (seed, step, element).
 * noise_pred f32 [2][n] (uc, c), counter f32 per frame [F] broadcast over (C, H*W): element
 * (c, f, p) has index (c*F + f)*HW + p.  latents f32 [n], updated in place; eps_out optional.
 * guidance_scale <= 1 is the reference's "no classifier-free guidance" (:622 `do_classifier_free_guidance =
 * guidance_scale > 1.0`): noise_pred is then [1][n] and eps = noise_pred / counter. */
int emo_cfg_step(const float* noise_pred, const float* counter, float* latents, float* eps_out, int C, int F,
                 int HW, float guidance_scale, float c_x, float c_eps, float c_noise, uint32_t seed, uint32_t step,
                 void* stream);
/* noise_pred[branch, :, frames[j]] += pred rows; counter[frames[j]] += 1 (EMOAnimationPipeline.py:790-794).
 * pred: rows ((j) h w, ld) in dtype for ONE branch of ONE window; frames: device int32 [nf].  The frames of a
 * window must be distinct; a position with frames[j] < 0 is skipped (a wrapped window of the uniform scheduler with
 * context_stride > 1 can list a frame twice: `noise_pred[:, :, c] = noise_pred[:, :, c] + pred` then keeps ONE
 * occurrence - the host marks the others negative). */
int emo_accumulate_window(const void* pred, int ld, float* noise_pred_branch, float* counter, const int32_t* frames,
                          int nf, int C, int F, int HW, int add_counter, int dtype, void* stream);

/* ---- EMO conditioning ----------------------------------------------
 * emo_act: y = act(x), kind 0 SiLU | 1 ReLU | 2 tanh | 3 erf-GELU over n contiguous elements (the ReLU / tanh of
 *   Net.py:214-218,246 and train_stage_3_speedlayers.py:36-40,66-73; GELU: the activation of the wav2vec2 encoder
 *   Net.py:611-612 loads - transformers Wav2Vec2FeedForward / conv layers).
 * emo_speed_encode: SpeedEncoder.encode_speed (Net.py:231-247): out[b,i] = tanh((v[b]-centers[i])/radii[i]*3).
 * emo_speed_bucket: SpeedController.map_speed_to_bucket (train_stage_3_speedlayers.py:42-47), INT bit-exact:
 *   idx[b] = argmin_i |v[b] - centers[i]| (first minimum on ties).
 * emo_gather_rows: nn.Embedding lookup out[b,:] = table[idx[b],:].
 * emo_add_rowbias: y[m,:] = x[m,:] + rb[m / rows_per_batch,:] (EMOStage3.forward combine, :268-269). */
int emo_act(const void* x, void* y, int64_t n, int kind, int dtype, void* stream);
int emo_speed_encode(const float* v, const float* centers, const float* radii, void* out, int B, int nb, int dtype, void* stream);
int emo_speed_bucket(const float* v, const float* centers, int32_t* idx, int B, int nb, void* stream);
int emo_gather_rows(const void* table, const int32_t* idx, void* out, int B, int D, int rows, int dtype, void* stream);
int emo_add_rowbias(const void* x, int ldx, const void* rb, int ldr, void* y, int ldy, int64_t M, int C, int rows_per_batch,
                    int dtype, void* stream);

/* ---- either side of the loop ----------------------------------------------------
 * emo_softmax_rows: y[m, :] = softmax(scale * x[m, :]) over N columns - the VAE mid-block attention (one head of 512
 *   channels; diffusers AutoencoderKL behind EMOAnimationPipeline.py:291-307,402-414) as Q.K^T GEMM -> this -> P.V GEMM.
 * emo_audio_windows: Wav2VecFeatureExtractor.extract_features_from_wav (Net.py:649-667): out[t][j][:] = feats[t-m+j][:],
 *   zero where t-m+j falls outside [0, T); out is [T][m+n+1][D].  Bit-exact (a copy).
 * emo_rows_to_video: rows ((b f) h w, ld) -> (B, C, F, H, W) f32 with y = clamp(x*mul + add, lo, hi): the
 *   `(video / 2 + 0.5).clamp(0, 1)` of decode_latents (EMOAnimationPipeline.py:303-306). */
int emo_softmax_rows(const void* x, int64_t ldx, void* y, int64_t ldy, int64_t M, int N, float scale, int dtype, void* stream);
int emo_audio_windows(const void* feats, void* out, int T, int D, int m, int n, int dtype, void* stream);
int emo_rows_to_video(const void* x, int64_t ld, float* y, int B, int C, int F, int HW, float mul, float add, float lo, float hi,
                      int dtype, void* stream);
/* emo_channelnorm: per-channel normalisation over the S rows of a sequence, y = act((x - mean_c) * rstd_c * gamma_c + beta_c), act 0 none |
 * 1 erf-GELU: nn.GroupNorm(C, C) on (1, C, T) - the first conv layer of the wav2vec2 feature extractor behind
 * Wav2VecFeatureExtractor (Net.py:607-648; transformers Wav2Vec2GroupNormConvLayer).  workspace: emo_channelnorm_workspace_bytes(S, C). */
size_t emo_channelnorm_workspace_bytes(int64_t S, int C);
int emo_channelnorm(const void* x, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy, int64_t S, int C, float eps,
                    int act, void* workspace, int dtype, void* stream);
/* FaceLocator (Net.py:819-855): nn.MaxPool2d(2, 2) over NHWC rows (H, W -> H/2, W/2), and
 * F.interpolate(logits, size=(Ho, Wo), mode='bilinear', align_corners=False) of rows ((n) h w, ld) into (n, C, Ho, Wo) f32. */
int emo_maxpool2x2(const void* x, int64_t ldx, void* y, int64_t ldy, int n_img, int H, int W, int C, int dtype, void* stream);
int emo_bilinear_to_nchw(const void* x, int64_t ld, float* y, int n_img, int C, int h, int w, int Ho, int Wo, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EMO_HIP_H */
