// gemm_api.h - host-side interface between gemm.hip (C ABI entry points, planning) and the per-dtype translation units
// gemm_f32.hip / gemm_bf16.hip / gemm_f16.hip, which instantiate the kernels of gemm_impl.h for ONE element type each
// (the three compile in parallel; one TU with every instantiation took ~5 minutes).
#pragma once
#include <stdlib.h>
#include "common.h"

#ifndef EMO_GEMM_KBYTES
#define EMO_GEMM_KBYTES 128
#endif
static constexpr int KBYTES = EMO_GEMM_KBYTES;   // bytes of K per ring stage and per LDS row (128 = one full L2 line per row)

// geometry of the halo-reuse 3x3 conv that the host needs for its eligibility test (kernel: gemm_impl.h)
struct HaloGeom { static constexpr int PH = 8, PW = 16, BN = 128; };

struct GemmPlan { int nt5, big, small, split_k; };

static int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }

static GemmPlan plan_gemm(int64_t M, int N, int K, int dtype, int geglu, int transpose_out) {
  GemmPlan pl;
  // 256x256 / 8 waves for the big compute-bound shapes (N a multiple of 256, or wide enough that the ragged last
  // tile is small), else 128x160 when N is a multiple of 160 (every SD-1.5 width), else 128x128
  const int64_t tiles256 = ((M + 255) / 256) * ((N + 255) / 256);
  const int bk = KBYTES / (dtype == EMO_F32 ? 4 : 2);
  const int nk = (K + bk - 1) / bk;
  static const int big_min_nk = env_int("EMO_GEMM_BIG_MINNK", 0);
  pl.big = (dtype != EMO_F32 && tiles256 >= 224 && (N % 256 == 0 || N >= 1792) && nk >= big_min_nk) ? 1 : 0;
  pl.nt5 = (!pl.big && !geglu && N % 160 == 0) ? 1 : 0;
  pl.small = 0;
  if (pl.nt5 && N % 128 == 0) {
    // both 128x160 and 128x128 tile N exactly: take the one that fills the 2-blocks-per-CU slots better (the 2x2 wave
    // layout also reads 1.0 instead of 1.2 LDS fragments per MFMA, so it wins ties)
    const int64_t mt = (M + 127) / 128, slots = 512;
    const int64_t b5 = mt * (N / 160), b4 = mt * (N / 128);
    const double f5 = (double)b5 / (double)(((b5 + slots - 1) / slots) * slots);
    const double f4 = (double)b4 / (double)(((b4 + slots - 1) / slots) * slots);
    static const int force = env_int("EMO_GEMM_TILE", 0);
    if (force == 4 || (force == 0 && f4 * 1.05 >= f5)) pl.nt5 = 0;
  }
  // fewer 128-row blocks than CUs and a short K (the 8x8 / 16x16 levels, the ReferenceNet pass): splitting K pays an f32
  // round trip + a second launch and a block is mostly prologue + epilogue -> 64x64 tiles (4 waves of 32x32, 32 KB of
  // LDS: ~4 co-resident blocks per CU overlap each other's prologue/epilogue).  They read 2 LDS fragments per MFMA, so
  // long-K shapes (every conv) stay on 128-row tiles + split-K.  V^T outputs also take them: the column-per-lane
  // store of the transposed epilogue is cheaper from 32x32 wave tiles (measured 91 -> 59 us at M=98304 N=K=320).
  static const int small_mode = env_int("EMO_GEMM_SMALL", 1), small_slots = env_int("EMO_GEMM_SMALL_SLOTS", 512);
  static const int small_below = env_int("EMO_GEMM_SMALL_BELOW", 256), small_nk = env_int("EMO_GEMM_SMALL_NK", 24);
  const int64_t blocks128 = ((M + 127) / 128) * ((N + (pl.nt5 ? 159 : 127)) / (pl.nt5 ? 160 : 128));
  if (small_mode && !pl.big && !geglu && nk <= small_nk && (blocks128 < small_below || transpose_out)) { pl.small = 1; pl.nt5 = 0; }
  const int bm = pl.small ? 64 : 128, bn = pl.small ? 64 : (pl.nt5 ? 160 : 128);
  const int64_t tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
  const int64_t slots = pl.small ? small_slots : 512;     // co-resident blocks to aim at
  int s = 1;
  if (!pl.big && tiles * 2 <= slots && nk >= 8 && N % 4 == 0) {
    s = (int)(slots / tiles);                     // whole blocks only: one block more than the slots costs a second round
    const int max_by_k = nk / 4;                  // keep >= 4 stages per slice
    if (s > max_by_k) s = max_by_k;
    if (s > 32) s = 32;
    if (s < 2) s = 1;
  }
  pl.split_k = s;
  return pl;
}


// per-dtype entry points (defined in gemm_impl.h, explicitly instantiated in gemm_<dtype>.hip)
template <typename T> int gemm_run(const emo_gemm_params& p, const GemmPlan& pl, int S, hipStream_t st);        // tiles (+ split-K reduce)
template <typename T> int gemm_run_halo(const emo_gemm_params& p, int64_t gx, hipStream_t st);                   // conv3x3_halo_kernel
