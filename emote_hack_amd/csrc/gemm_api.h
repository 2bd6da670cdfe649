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
struct HaloGeom { static constexpr int PW = 16, BN = 128; };   // patch height: 8 or 16 rows (gemm_impl.h HaloT)

enum { EMO_TILE_AUTO = 0, EMO_TILE_64x64 = 1, EMO_TILE_128x128 = 2, EMO_TILE_128x160 = 3, EMO_TILE_256x256 = 4,
       EMO_TILE_256x160 = 5, EMO_TILE_256x320 = 6, EMO_TILE_256x256_PP = 7 };   // 7: the ping-pong main loop on the 256x256 tile (gemm_impl.h)
// (measured and dropped: 4 waves of 128x128 with 512 registers - 880-900 TFLOP/s at 8192^3 against 1100 and 2-3x slower on short
// K, hipcc spills ~220 VGPRs around the epilogue; a register-staged loader - tools/bench/patches/gemm_staged_loader.patch)
struct GemmPlan { int tile, split_k; };

static inline void tile_dims(int tile, int& bm, int& bn) {
  switch (tile) {
    case EMO_TILE_64x64: bm = 64; bn = 64; break;
    case EMO_TILE_128x160: bm = 128; bn = 160; break;
    case EMO_TILE_256x256: case EMO_TILE_256x256_PP: bm = 256; bn = 256; break;
    case EMO_TILE_256x160: bm = 256; bn = 160; break;
    case EMO_TILE_256x320: bm = 256; bn = 320; break;
    default: bm = 128; bn = 128; break;
  }
}

// Tile / split-K plan of one GEMM.  `hint` (emo_gemm_params.tile) pins the tile shape; 0 = the measured heuristics below.
static GemmPlan plan_gemm(int64_t M, int N, int K, int dtype, int geglu, int transpose_out, int hint = 0, int ln = 0) {
  GemmPlan pl;
  // 256x256 / 8 waves for the big compute-bound shapes (N a multiple of 256, or wide enough that the ragged last
  // tile is small), else 128x160 when N is a multiple of 160 (every SD-1.5 width), else 128x128
  const int64_t tiles256 = ((M + 255) / 256) * ((N + 255) / 256);
  const int bk = KBYTES / (dtype == EMO_F32 ? 4 : 2);
  const int nk = (K + bk - 1) / bk;
  // (LayerNorm-folded projections: the 256-row tile also wins at N = 640 / 960 - 105 vs 128 us at M=98304 N=960 K=320, 84 vs 92
  // at N=640 - although the last 256-column tile is ragged)
  // (round 5, profiles/r05l_tile_sweep.txt: with a ragged last column tile the 256-row tile pays only from ~512 tiles on - M=24576 N=640 K=640
  // ln, the 32x32 level's attn2 to_q: 40.1 us on 96 x 3 tiles of 256x256 against 31.1 on 128x128)
  const bool big = dtype != EMO_F32 && tiles256 >= 224 && (N % 256 == 0 || N >= 1792 || (ln && !transpose_out && N >= 640 && tiles256 >= 512 && N % 256 >= 128));
  bool nt5 = !big && !geglu && N % 160 == 0;
  if (nt5 && N % 128 == 0) {
    // both 128x160 and 128x128 tile N exactly: take the one that fills the 2-blocks-per-CU slots better (the 2x2 wave
    // layout also reads 1.0 instead of 1.2 LDS fragments per MFMA, so it wins ties)
    const int64_t mt = (M + 127) / 128, slots = 512;
    const int64_t b5 = mt * (N / 160), b4 = mt * (N / 128);
    const double f5 = (double)b5 / (double)(((b5 + slots - 1) / slots) * slots);
    const double f4 = (double)b4 / (double)(((b4 + slots - 1) / slots) * slots);
    if (f4 * 1.05 >= f5) nt5 = false;
  }
  // fewer 128-row blocks than CUs and a short K (the 8x8 / 16x16 levels, the ReferenceNet pass): splitting K pays an f32
  // round trip + a second launch and a block is mostly prologue + epilogue -> 64x64 tiles (4 waves of 32x32, 32 KB of
  // LDS: ~4 co-resident blocks per CU overlap each other's prologue/epilogue).  They read 2 LDS fragments per MFMA, so
  // long-K shapes (every conv) stay on 128-row tiles + split-K.  V^T outputs also take them: the column-per-lane
  // store of the transposed epilogue is cheaper from 32x32 wave tiles (measured 91 -> 59 us at M=98304 N=K=320).
  const int64_t blocks128 = ((M + 127) / 128) * ((N + (nt5 ? 159 : 127)) / (nt5 ? 160 : 128));
  // (nk <= 40: M=1536 N=1280 K=2560 27 us on 64x64 tiles against 32 split 4 ways on 128x128, tools/bench/conv_split_sweep.py)
  const bool small = !big && !geglu && (nk <= 24 || (nk <= 40 && blocks128 < 256)) && (blocks128 < 256 || transpose_out);
  // few tiles and a LONG K (the 8x8-level 3x3 convs through the im2col loader: M = 1536, K = 11520 / 23040): 256x256 tiles
  // split 256 / tiles ways - the same block count as 128x128 tiles split 4 ways at half the LDS-DMA traffic per MFMA
  // (85 vs 105 us at Cin 1280, 132 vs 172 us at Cin 2560; M = 640, the ReferenceNet group: 55 vs 75)
  const bool deep = !big && !small && dtype != EMO_F32 && !ln && !geglu && !transpose_out && nk >= 128 && tiles256 * 4 <= 256 && N % 4 == 0;
  pl.tile = big || deep ? EMO_TILE_256x256 : (small ? EMO_TILE_64x64 : (nt5 ? EMO_TILE_128x160 : EMO_TILE_128x128));
  // the ping-pong main loop (gemm_impl.h PP) for the single-pass 2-byte dense shapes: +1 % at 8192^3, +2.6 % at M=98304 N=2560
  // K=320 GEGLU, +4.6 % at N=960 K=320, never behind (tools/bench/pp_bench.py); the conv loader / V^T / split-K dispatch of this
  // id fall back to the lockstep loop
  if (big && !transpose_out && K % bk == 0) pl.tile = EMO_TILE_256x256_PP;
  // (256x160 / 256x320 tiles exist behind the hint: on the UNet's shapes they measured SLOWER than the 128-row tiles - 342 /
  // 250 vs 351 TFLOP/s at M=98304 N=K=320, 576 / 732 vs 696-725 at M=24576 N=640 K=2560 - one block per CU leaves the
  // epilogue uncovered, which costs more than the lighter LDS-DMA stream saves; the planner does not pick them)
  if (hint > 0) {
    pl.tile = hint;
    if ((hint == EMO_TILE_256x160 || hint == EMO_TILE_256x320) && (transpose_out || (hint == EMO_TILE_256x320 && dtype == EMO_F32))) pl.tile = EMO_TILE_128x160;
    // GEGLU pairs a value tile with its gate tile inside one wave: only the even-WTN shapes (128x128, 256x256) serve it
    if (hint > EMO_TILE_256x256_PP) pl.tile = EMO_TILE_128x128;
    if (hint == EMO_TILE_256x256_PP && (transpose_out || dtype == EMO_F32 || K % bk != 0)) pl.tile = EMO_TILE_256x256;
    if (geglu && pl.tile != EMO_TILE_128x128 && pl.tile != EMO_TILE_256x256 && pl.tile != EMO_TILE_256x256_PP) pl.tile = EMO_TILE_128x128;
  }
  int bm, bn;
  tile_dims(pl.tile, bm, bn);
  const int64_t tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
  const int64_t slots = 512;                      // co-resident blocks to aim at
  int s = 1;
  if (deep && hint <= 0) {
    s = (int)(256 / tiles);                       // one block per CU
    if (s > nk / 4) s = nk / 4;
    if (s > 32) s = 32;
  } else if (!ln && bm <= 128 && tiles * 2 <= slots && nk >= 8 && N % 4 == 0) {
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
template <typename T> int gemm_run_halo(const emo_gemm_params& p, int ph, int bn, int64_t gx, hipStream_t st);                   // conv3x3_halo_kernel
