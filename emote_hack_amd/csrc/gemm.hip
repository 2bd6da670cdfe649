// gemm.hip - C ABI entry points of the GEMM / 3x3 conv family: argument checks, tile / split-K planning, dtype dispatch.
// The kernels live in gemm_impl.h and are instantiated per element type in gemm_f32.hip / gemm_bf16.hip / gemm_f16.hip.
#include "gemm_api.h"

extern template int gemm_run<float>(const emo_gemm_params&, const GemmPlan&, int, hipStream_t);
extern template int gemm_run<bf16_t>(const emo_gemm_params&, const GemmPlan&, int, hipStream_t);
extern template int gemm_run<f16_t>(const emo_gemm_params&, const GemmPlan&, int, hipStream_t);
extern template int gemm_run_halo<float>(const emo_gemm_params&, int, int, int64_t, hipStream_t);
extern template int gemm_run_halo<bf16_t>(const emo_gemm_params&, int, int, int64_t, hipStream_t);
extern template int gemm_run_halo<f16_t>(const emo_gemm_params&, int, int, int64_t, hipStream_t);

extern "C" int emo_gemm_suggest_split_k(int64_t M, int N, int K, int dtype, int geglu, int transpose_out) {
  return plan_gemm(M, N, K, dtype, geglu, transpose_out).split_k;
}
extern "C" size_t emo_gemm_workspace_bytes(int64_t M, int N, int split_k) {
  return split_k > 1 ? (size_t)split_k * (size_t)M * (size_t)N * sizeof(float) : 0;
}

// The same row-major GEMM / conv restricted to the output columns [n0, n0 + n): EVERY per-column operand moves with them (weights rows,
// output, bias, residual, row bias) - one place to extend when a per-column field is added to emo_gemm_params.
static emo_gemm_params output_columns(const emo_gemm_params& p, int n0, int n) {
  const int esz = p.dtype == EMO_F32 ? 4 : 2;
  emo_gemm_params q = p;
  q.N = n;
  q.W = (const char*)p.W + (int64_t)n0 * p.K * esz;
  q.C = (char*)p.C + (int64_t)n0 * esz;
  if (p.bias) q.bias = p.bias + n0;
  if (p.residual) q.residual = (const char*)p.residual + (int64_t)n0 * esz;
  if (p.rowbias) q.rowbias = p.rowbias + n0;
  return q;
}

// the stride-1 3x3 convs the halo-reuse kernel serves (conv_halo_impl.h); everything else takes the im2col loader of the GEMM
static bool halo_conv_ok(const emo_gemm_params& p) {
  const int S = p.split_k > 1 ? p.split_k : 1;
  const int bk = KBYTES / (p.dtype == EMO_F32 ? 4 : 2);
  // (the nearest x2 upsampling of resnet.py:74-82 rides along: the patch grid lives on the upsampled frame)
  const int He = p.upsample2x ? 2 * p.H : p.H, We = p.upsample2x ? 2 * p.W_ : p.W_;
  // Frames that are not a whole number of 8 x 16 patches (24 x 24: BASELINE configs[4]) run with an overlapped last patch row / column
  // whose pixels two blocks store (identical bits): not with the GroupNorm fold, and not when the output aliases the residual
  const bool whole = He % 8 == 0 && We % HaloGeom::PW == 0;
  const int esz_ = p.dtype == EMO_F32 ? 4 : 2;
  const char* c0 = (const char*)p.C; const char* r0 = (const char*)p.residual;
  const bool alias = r0 && c0 && !(r0 + ((p.M - 1) * p.ldr + p.N) * esz_ <= c0 || c0 + ((p.M - 1) * p.ldc + p.N) * esz_ <= r0);
  const bool ragged_ok = He >= 8 && We >= HaloGeom::PW && !p.gn_coef && !alias;
  return p.conv_taps == 9 && p.stride == 1 && !p.conv_asym && !p.up_h && !p.transpose_out && !p.geglu && S == 1 && p.Cin > 0 && p.Cin % bk == 0 &&
         (whole || ragged_ok) && (p.N & 3) == 0 &&
         // (its loaders address one frame / one weight tile through 32-bit buffer offsets)
         (int64_t)p.H * p.W_ * p.lda * (p.dtype == EMO_F32 ? 4 : 2) < (1ll << 31) && (int64_t)HaloGeom::BN * p.K * (p.dtype == EMO_F32 ? 4 : 2) < (1ll << 31) &&
         (!p.rowbias || (p.rows_per_batch % (He * We) == 0 && (p.ld_rowbias & 3) == 0));
}
extern "C" int emo_conv3x3_gn_fusable(const emo_gemm_params* pp) {
  return pp && emo_dtype_ok(pp->dtype) && pp->H > 0 && pp->W_ > 0 && pp->H % 8 == 0 && pp->W_ % HaloGeom::PW == 0 && halo_conv_ok(*pp) && !pp->upsample2x ? 1 : 0;
}

// columns a wave of the planned tile covers: the split store of emo_gemm_params.vt switches per wave
static int tile_wave_cols(int tile) {
  switch (tile) {
    case EMO_TILE_64x64: return 32;
    case EMO_TILE_128x160: case EMO_TILE_256x160: case EMO_TILE_256x320: return 160;
    default: return 64;   // 128x128, 256x256 (+ ping-pong / phase loop)
  }
}
static bool gemm_vt_ok(const emo_gemm_params& p) {
  if (!p.vt || !p.ln_colsum || p.conv_taps || p.transpose_out || p.geglu || p.residual || p.rowbias || p.split_k > 1 || p.w_slab_rows || p.out_scale != 1.0f) return false;
  if (p.vt_col0 <= 0 || p.vt_col0 >= p.N || (p.N - p.vt_col0) % 8 || p.M % 32 || p.t_rows <= 0 || p.t_ld < p.t_rows || p.ldc < p.vt_col0) return false;
  const GemmPlan pl = plan_gemm(p.M, p.N, p.K, p.dtype, 0, 0, p.tile & 15, p.ln_colsum != nullptr);
  if (pl.split_k > 1) return false;
  int bm, bn;
  tile_dims(pl.tile, bm, bn);
  // a wave's rows (32 .. 128 of the tile's) must lie inside one batch of t_rows rows, its columns on one side of vt_col0
  return p.t_rows % bm == 0 && p.vt_col0 % tile_wave_cols(pl.tile) == 0 && ((uintptr_t)p.vt % 16) == 0;
}
extern "C" int emo_gemm_vt_ok(const emo_gemm_params* pp) { return pp && emo_dtype_ok(pp->dtype) && gemm_vt_ok(*pp) ? 1 : 0; }

extern "C" int emo_gemm(const emo_gemm_params* pp, void* stream) {
  EMO_CHECK(pp, EMO_ERR_NULL, "emo_gemm: null params");
  const emo_gemm_params& p = *pp;
  EMO_CHECK(p.A && p.W && p.C, EMO_ERR_NULL, "emo_gemm: null pointer");
  EMO_CHECK(emo_dtype_ok(p.dtype), EMO_ERR_BAD_DTYPE, "emo_gemm: dtype %d", p.dtype);
  const int V = emo_dtype_vec(p.dtype);
  EMO_CHECK(p.M > 0 && p.N > 0 && p.K > 0, EMO_ERR_BAD_SHAPE, "emo_gemm: M=%lld N=%d K=%d", (long long)p.M, p.N, p.K);
  EMO_CHECK(p.K % V == 0 && p.lda % V == 0, EMO_ERR_BAD_SHAPE, "emo_gemm: K=%d lda=%lld must be multiples of %d", p.K,
            (long long)p.lda, V);
  EMO_CHECK(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.W % 16) == 0, EMO_ERR_BAD_SHAPE, "emo_gemm: A/W must be 16-byte aligned");
  if (p.geglu) EMO_CHECK(p.N % 64 == 0, EMO_ERR_BAD_SHAPE, "emo_gemm: GEGLU needs N %% 64 == 0 (N=%d)", p.N);
  if (p.rowbias) EMO_CHECK(p.rows_per_batch > 0 && p.ld_rowbias >= p.N, EMO_ERR_BAD_SHAPE, "emo_gemm: rowbias geometry");
  if (p.transpose_out) EMO_CHECK(p.t_rows > 0 && p.t_ld >= p.t_rows && !p.geglu && !p.residual, EMO_ERR_BAD_SHAPE, "emo_gemm: transpose geometry");
  const bool conv = p.conv_taps != 0;
  if (conv) {
    EMO_CHECK(p.conv_taps == 9, EMO_ERR_UNSUPPORTED, "emo_gemm: conv_taps=%d (only 3x3)", p.conv_taps);
    EMO_CHECK(p.Cin > 0 && p.Cin % V == 0 && p.K == 9 * p.Cin && p.lda >= p.Cin, EMO_ERR_BAD_SHAPE,
              "emo_gemm: conv needs Cin %% %d == 0 and K == 9*Cin (Cin=%d K=%d)", V, p.Cin, p.K);
    EMO_CHECK(p.H > 0 && p.W_ > 0 && p.Ho > 0 && p.Wo > 0 && (p.stride == 1 || p.stride == 2), EMO_ERR_BAD_SHAPE, "emo_gemm: conv geometry");
    EMO_CHECK(p.M % ((int64_t)p.Ho * p.Wo) == 0, EMO_ERR_BAD_SHAPE, "emo_gemm: conv M not a multiple of Ho*Wo");
    EMO_CHECK((p.up_h == 0) == (p.up_w == 0) && p.up_h >= 0 && !(p.up_h && p.upsample2x), EMO_ERR_BAD_SHAPE, "emo_gemm: up_h / up_w");
    const int He = p.up_h ? p.up_h : (p.upsample2x ? 2 * p.H : p.H), We = p.up_h ? p.up_w : (p.upsample2x ? 2 * p.W_ : p.W_);
    const int pad_tot = p.conv_asym ? 1 : 2;
    EMO_CHECK(p.Ho == (He + pad_tot - 3) / p.stride + 1 && p.Wo == (We + pad_tot - 3) / p.stride + 1, EMO_ERR_BAD_SHAPE, "emo_gemm: conv output size");
  }
  if (!p.transpose_out) {
    EMO_CHECK(((uintptr_t)p.C % 16) == 0 && (!p.residual || ((uintptr_t)p.residual % 8) == 0), EMO_ERR_BAD_SHAPE, "emo_gemm: C/residual alignment");
  }
  const int S = p.split_k > 1 ? p.split_k : 1;
  if (p.w_slab_rows) {
    EMO_CHECK(p.w_slab_rows > 0 && p.w_slab_rows % 256 == 0 && p.w_slab_stride >= (int64_t)p.N * p.K && p.M % p.w_slab_rows == 0, EMO_ERR_BAD_SHAPE,
              "emo_gemm: w_slab_rows=%d must be a positive multiple of 256 dividing M, w_slab_stride >= N*K", p.w_slab_rows);
    EMO_CHECK(!conv && S == 1 && !p.ln_colsum && !p.transpose_out && p.N % 4 == 0, EMO_ERR_UNSUPPORTED,
              "emo_gemm: per-instance weights need a dense, single-pass, row-major GEMM with N %% 4 == 0 and without the LayerNorm fold");
  }
  if (p.ln_colsum || p.ln_stats) {
    EMO_CHECK(p.ln_colsum && p.ln_stats, EMO_ERR_NULL, "emo_gemm: the LayerNorm fold needs both ln_colsum and ln_stats");
    EMO_CHECK(!conv && S == 1, EMO_ERR_UNSUPPORTED, "emo_gemm: the LayerNorm fold needs a dense, single-pass GEMM (conv=%d split_k=%d)", (int)conv, S);
    EMO_CHECK(p.N % 4 == 0 && ((uintptr_t)p.ln_colsum % 16) == 0 && ((uintptr_t)p.ln_stats % 8) == 0 && (!p.bias || ((uintptr_t)p.bias % 16) == 0),
              EMO_ERR_BAD_SHAPE, "emo_gemm: the LayerNorm fold needs N %% 4 == 0 and aligned colsum / stats / bias");
  }
  if (p.vt) EMO_CHECK(gemm_vt_ok(p), EMO_ERR_UNSUPPORTED, "emo_gemm: vt / vt_col0=%d is not served for this GEMM (ask emo_gemm_vt_ok)", p.vt_col0);
  if (p.gn_coef) {
    EMO_CHECK(conv && halo_conv_ok(p) && !p.upsample2x, EMO_ERR_UNSUPPORTED,
              "emo_gemm: gn_coef is served by the halo-reuse 3x3 conv only (ask emo_conv3x3_gn_fusable)");
    EMO_CHECK(p.gn_imgs_per_inst > 0 && (p.M / ((int64_t)p.H * p.W_)) % p.gn_imgs_per_inst == 0 && ((uintptr_t)p.gn_coef % 16) == 0,
              EMO_ERR_BAD_SHAPE, "emo_gemm: gn_imgs_per_inst=%d must divide the image count; gn_coef 16-byte aligned", p.gn_imgs_per_inst);
  }
  {
    const int He = p.upsample2x ? 2 * p.H : p.H;
    if (conv && halo_conv_ok(p)) {
      const int64_t nt = (p.N + HaloGeom::BN - 1) / HaloGeom::BN;
      // 16-row patches (8 waves, one block per CU) when they still give (nearly) every CU a block; p.tile 1 / 2 pins 8 / 16
      const int We = p.upsample2x ? 2 * p.W_ : p.W_;
      const int64_t n_img = p.M / ((int64_t)He * We), tpx = (We + HaloGeom::PW - 1) / HaloGeom::PW;
      const int64_t tiles16 = n_img * tpx * ((He + 15) / 16) * nt;
      const bool ph16 = He % 16 == 0 && ((p.tile & 3) == 2 || ((p.tile & 3) != 1 && tiles16 >= 200));
      // A width that is an odd multiple of 64 (N = 320 = 128 + 128 + 64): the last 128-column tile would multiply 64 columns of
      // zeros - 17 % of the launch at N = 320.  The 64 remainder columns get their own launch of 64-channel blocks instead
      // (same patches, half the weight tile, half the MFMAs per stage); p.tile bit 2 (4) keeps the single launch (tools/bench A/B).
      const int n_rem = (p.N > HaloGeom::BN && p.N % HaloGeom::BN == 64 && !(p.tile & 4)) ? 64 : 0;
      // ... and with 16-row patches the LAST 192 columns may go to 192-channel blocks (N = 320 = 128 + 192: two launches of full-width
      // blocks instead of 128 + 128 + a half-width 64) - when a launch of one block per patch still fits the chip in ONE round: at
      // M = 49152 (192 patches: the shared-prefix half batch) +15 %, N = 192 +23 %; at M = 98304 (384 patches) each of the two
      // launches would run a second, half-empty round: 0.4-4.5 % SLOWER than 768 + 384 blocks (profiles/r06k_conv_bn192.txt).
      // p.tile bit 3 (8) keeps the 64-column remainder launch, bit 4 (16) forces the 192-column blocks (tools/bench A/B)
      const bool wide = n_rem && ph16 && !p.gn_coef && !(p.tile & 8) && (n_img * tpx * (He / 16) <= 256 || p.N == 192 || (p.tile & 16));   // (N = 192: ONE launch instead of two)
      auto launch = [&](const emo_gemm_params& q, int bn) {
        const int64_t ntq = (q.N + bn - 1) / bn;
        const int64_t tiles = n_img * tpx * (ph16 ? He / 16 : (He + 7) / 8) * ntq, slots = ph16 ? 256 : 512;
        const int64_t gx = tiles > slots ? slots : tiles;
        int rc_h = EMO_OK;
        EMO_DISPATCH(q.dtype, "emo_gemm", rc_h = gemm_run_halo<T>(q, ph16 ? 16 : 8, bn, gx, as_stream(stream)));
        return rc_h;
      };
      if (!n_rem) return launch(p, p.N == 64 ? 64 : ((p.N == 192 && ph16 && !p.gn_coef && !(p.tile & 8)) ? 192 : HaloGeom::BN));
      emo_gemm_params a = p;
      a.N = p.N - (wide ? 192 : n_rem);
      const int rc_a = a.N > 0 ? launch(a, HaloGeom::BN) : EMO_OK;
      return rc_a ? rc_a : launch(output_columns(p, a.N, p.N - a.N), wide ? 192 : 64);
    }
  }
  GemmPlan pl = plan_gemm(p.M, p.N, p.K, p.dtype, p.geglu, p.transpose_out, p.tile & 15, p.ln_colsum != nullptr);   // (tile >> 4: tile-order override, gemm_impl.h)
  if (S > 1) {
    EMO_CHECK(p.workspace != nullptr && S <= 65535, EMO_ERR_NULL, "emo_gemm: split_k=%d needs a workspace", S);
    EMO_CHECK(p.N % 4 == 0, EMO_ERR_BAD_SHAPE, "emo_gemm: split-K needs N %% 4 == 0");
  }
  hipStream_t st = as_stream(stream);
  int rc = EMO_OK;
  EMO_DISPATCH(p.dtype, "emo_gemm", rc = gemm_run<T>(p, pl, S, st));
  return rc;
}
