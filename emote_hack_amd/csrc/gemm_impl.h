// gemm.hip - MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950 (wave64, 32x32 MFMA tiles).
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T )        A: NHWC activation rows, W: torch Linear layout
//
// One kernel template serves every GEMM-shaped op on the hot path: Linear q/k/v/out/FF
// (orig_attention.py:566-575,776,817), 1x1 conv proj_in/out + shortcuts (attention.py:82,110;
// resnet.py:175) and - with the conv A-loader - the per-frame 3x3 convs (resnet.py:30-38), stride 2
// (resnet.py:98) and nearest-x2-upsample-folded (resnet.py:74-82: the interpolate is folded into the
// load indexing, the upsampled tensor never exists).
//
// Structure (v6):
//   * PERSISTENT blocks: a launch has as many blocks as the chip holds at once (by LDS, <= 4 per CU; a multiple of 8),
//     block b walks tiles b, b+G, ... of an XCD-aware order (block b runs on XCD b%8; each XCD owns a contiguous run of
//     tiles, n fastest, so the A rows it re-reads stay in that XCD's L2).
//   * a wave owns WTM x WTN MFMA 32x32 tiles: 2x4 = 8 waves of 128x64 (256x256, the big compute-bound shapes), 2x2 waves
//     of 64x64 (128x128), 4x1 waves of 32x160 (128x160: every channel count of the SD-1.5 UNet is a multiple of 160),
//     2x2 waves of 32x32 (64x64: few-block short-K shapes and V^T outputs, 4 co-resident blocks per CU).
//   * K is streamed in 128-BYTE stages (bf16: 64, f32: 32 k-values; one full L2 line per row per stage) through a
//     2-deep LDS ring filled by ASYNCHRONOUS direct-to-LDS loads (global_load_lds_dwordx4, no VGPR round trip).  The
//     loader state is a stream of (tile, stage) pairs that runs one stage AHEAD of the MFMA loop and does not stop at tile
//     boundaries: a tile's epilogue runs with the next tile's first stage in flight.  The whole next stage is requested
//     right behind the per-stage s_barrier (it then has the full MFMA time of this stage to land).
//   * software-pipelined stage: after the barrier only the first k-step's fragment read is exposed; the reads of
//     step kk+1 are issued one at a time BETWEEN the MFMAs of step kk (compile-time `static_for` so every register-array
//     index is a constant: a runtime index sends the arrays to scratch and breaks the asm-read/wait protocol).
//   * the LDS image of a stage is lane-linear (a glds writes wave-base + lane*16), so the 16-byte chunk position is
//     XOR-swizzled with (row / rows-per-bank-row) on the SOURCE address and on the fragment read: the ds_read_b128 of
//     a 16-lane group then hits 16 distinct 16-byte slots (conflict-free, SQ_LDS_BANK_CONFLICT = 0 measured).  One
//     swizzle key per lane serves every glds; the k-step fragment addresses are the step-0 address XOR (kk << 5).
//   * fragment reads are inline-asm ds_read_b128 with hand-counted lgkmcnt: hipcc drains vmcnt(0) in front of
//     every C++-level LDS read while a glds is in flight, which would serialise the ring.
//   * dense operands: rows past M / N are clamped to the last valid row (never stored), so every source pointer steps
//     by one constant per stage; K tails and the conv loader's zero padding source a 16-byte zero page.
//   * the MFMA is issued with the operands SWAPPED (acc = W_frag x A_frag), so a lane ends up with 4 consecutive
//     output COLUMNS of one row per register quad.  V^T store (TRANS): operands unswapped, a lane holds 4 consecutive
//     ROWS of one column.
//   * EPILOGUE (what bounds the short-K shapes): accumulators start at the bias; bf16 row-major outputs are staged
//     through the ring slot the main loop has just released and written back as full 128-byte lines, 16 bytes per lane,
//     with equally coalesced residual loads issued ahead of the staging phase (epilogue_lds); everything else (f32,
//     ragged N, split-K partials, V^T) uses the row-per-lane / column-per-lane paths.
//   bf16: v_mfma_f32_32x32x16_bf16, f32 accumulate.   f32: v_mfma_f32_32x32x2_f32 (exact f32 fma chain).
// Epilogue fused: bias, per-batch row bias (temb), GEGLU, residual, scale, row-major or V^T store.
// Split-K (small M): f32 partial slabs + a fixed-order reduce/epilogue kernel (deterministic).
// conv3x3_halo_kernel (below): the stride-1 3x3 convs with input-halo reuse instead of the im2col loader.
// LayerNorm fold (template LN): the LayerNorms of the transformer blocks all feed a Linear; with ln_colsum / ln_stats set the
// kernel multiplies the RAW rows: out = rstd_m * (acc0 + x . W'^T), acc0 = b'[n] / rstd_m - mean_m * colsum[n] - the whole
// correction rides in the accumulator init and the epilogue pays one multiply.  (mean, rstd) come from a read-only statistics
// pass (emo_layernorm_stats: one read of x instead of LayerNorm's read + write + the GEMM's re-read).  A first version summed
// the statistics from the A fragments inside the MFMA loop and applied colsum / bias in the epilogue: no extra pass, but two
// float4 loads per output quad with nothing to hide them - 20-30 % slower GEMMs at N >= 960, a net loss on the GEGLU shapes.
// emo_gemm_params.tile pins a tile shape (0 = planned); the planner's thresholds are the measured ones of DESIGN.md 7.
#pragma once
#include "gemm_api.h"

static constexpr int CPR = KBYTES / 16;          // 16-byte chunks per LDS row
static constexpr int KSTEPS = KBYTES / 32;       // mma16 steps per stage (a step consumes 2 chunks: one per lane half)
static constexpr int RPB = 256 / KBYTES;         // LDS rows per 256-byte bank row
// XOR swizzle of the chunk position: rows that share a 256-B bank row get the same key, consecutive bank rows
// different keys -> the 16 lanes of a ds_read_b128 group (rows {0-3,12-15,20-27} / {4-11,16-19,28-31}) hit 16
// distinct 16-byte slots
__device__ __forceinline__ int swz(int row) { return (row / RPB) & (CPR - 1); }

static __device__ __attribute__((aligned(16))) unsigned int g_zero_page[4] = {0, 0, 0, 0};   // one copy per dtype TU

template <int WTM, int WTN, int WVM, int WVN, int NS_> struct GemmTile {
  static constexpr int NS = NS_;   // LDS ring depth
  static constexpr int NW = WVM * WVN, THREADS = 64 * NW;
  static constexpr int BM = 32 * WTM * WVM, BN = 32 * WTN * WVN;
  static constexpr int RPI = 64 * NW / CPR;   // LDS rows filled by one glds "round" of the whole block
  static constexpr int A_ROWS = (BM + RPI - 1) / RPI * RPI, B_ROWS = (BN + RPI - 1) / RPI * RPI;
  static constexpr int A_BYTES = A_ROWS * KBYTES, B_BYTES = B_ROWS * KBYTES;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int LDS_BYTES = NS * STAGE_BYTES;
  static constexpr int LA = A_ROWS / RPI, LB = B_ROWS / RPI, LPS = LA + LB;   // glds per wave per stage
  // co-resident blocks per CU the launch is sized for (by LDS, <= 4) and the waves per SIMD that needs: the register
  // budget handed to the compiler (the persistent loop keeps the loader state live across the epilogue)
  static constexpr int BPC = (160 * 1024 / LDS_BYTES) > 4 ? 4 : ((160 * 1024 / LDS_BYTES) < 1 ? 1 : (160 * 1024 / LDS_BYTES));
  static constexpr int WPE = (BPC * NW + 3) / 4;
};

#define EMO_GLDS16(gptr, lptr) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// the same 16-byte direct-to-LDS load through a buffer resource: lane offset (range-checked: out of range reads 0) + scalar offset
// (a __device__ function: with the builtin in a kernel template's own body the host pass drops the kernel's stub)
__device__ __forceinline__ void glds16_buffer(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_dst, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) unsigned char*)lds_dst, 16, voff, soff, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ uint4 lds_read16(unsigned addr) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// Row-major fused epilogue of one 32-row MFMA tile row (lane <-> output row m; register quad g of tile j <-> columns
// j*32 + 8*g + 4*half + {0..3}): bias, per-batch row bias, GEGLU, residual, scale, 8-byte (bf16) / 16-byte (f32) stores.
// ln: LayerNorm folded into the GEMM - the accumulator started at bias / rstd_m - mean_m * colsum[n] and holds that plus A.W^T
// of the RAW rows; one multiply by rstd_m finishes it (see emo_gemm_params.ln_colsum).
template <typename T, int WTN>
__device__ __forceinline__ void epilogue_row(const f32x16 (&acc)[WTN], const emo_gemm_params& p, int64_t m, bool m_ok, int wn0, int half,
                                             T* __restrict__ C, const T* __restrict__ R, bool ln = false, float ln_rstd = 1.f) {
  const float* rbias = (p.rowbias && m_ok) ? p.rowbias + (m / p.rows_per_batch) * p.ld_rowbias : nullptr;
  const int n_out = p.geglu ? p.N / 2 : p.N;
  const bool vec_ok = (p.N & 3) == 0 && ((p.ldc | (R ? p.ldr : 0)) & 3) == 0 && (!p.rowbias || (p.ld_rowbias & 3) == 0);
#pragma unroll
  for (int j = 0; j < WTN; j++) {
    if (p.geglu && (j & 1)) continue;   // gate tile is consumed with its value tile
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int nw0 = wn0 + j * 32 + 8 * g + 4 * half;              // column in W-row space
      const int no0 = p.geglu ? ((wn0 + j * 32) >> 1) + 8 * g + 4 * half : nw0;   // output column
      if (nw0 >= p.N) continue;
      float o[4] = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
      if (vec_ok) {   // the whole quad is in range (N % 4 == 0)
        if (ln) { o[0] *= ln_rstd; o[1] *= ln_rstd; o[2] *= ln_rstd; o[3] *= ln_rstd; }
        if (p.bias) { const float4 b4 = *(const float4*)(p.bias + nw0); o[0] += b4.x; o[1] += b4.y; o[2] += b4.z; o[3] += b4.w; }
        if (rbias) { const float4 b4 = *(const float4*)(rbias + nw0); o[0] += b4.x; o[1] += b4.y; o[2] += b4.z; o[3] += b4.w; }
        if (p.geglu) {
          float gt[4] = {acc[(j + 1) % WTN][4 * g], acc[(j + 1) % WTN][4 * g + 1], acc[(j + 1) % WTN][4 * g + 2],
                         acc[(j + 1) % WTN][4 * g + 3]};
          if (ln) { gt[0] *= ln_rstd; gt[1] *= ln_rstd; gt[2] *= ln_rstd; gt[3] *= ln_rstd; }
          if (p.bias) { const float4 b4 = *(const float4*)(p.bias + nw0 + 32); gt[0] += b4.x; gt[1] += b4.y; gt[2] += b4.z; gt[3] += b4.w; }
          if constexpr (sizeof(T) == 2) {
            geglu_poly2(o[0], o[1], gt[0], gt[1], o[0], o[1]);
            geglu_poly2(o[2], o[3], gt[2], gt[3], o[2], o[3]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] *= gelu_for<T>(gt[e]);
          }
        }
        if (!m_ok) continue;
        if (R) {
          if constexpr (sizeof(T) == 2) {
            float r4[4];
            unpack4<T>(*(const uint2*)(R + m * p.ldr + no0), r4);
            o[0] += r4[0]; o[1] += r4[1]; o[2] += r4[2]; o[3] += r4[3];
          } else {
            const float4 rv = *(const float4*)(R + m * p.ldr + no0);
            o[0] += rv.x; o[1] += rv.y; o[2] += rv.z; o[3] += rv.w;
          }
        }
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] *= p.out_scale;
        if constexpr (sizeof(T) == 2) *(uint2*)(C + m * p.ldc + no0) = make_uint2(pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3]));
        else *(float4*)(C + m * p.ldc + no0) = make_float4(o[0], o[1], o[2], o[3]);
      } else {   // ragged N / unaligned leading dims: scalar path
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int nw = nw0 + e;
          if (nw >= p.N || !m_ok) continue;
          float v = o[e];
          if (ln) v *= ln_rstd;
          if (p.bias) v += p.bias[nw];
          if (rbias) v += rbias[nw];
          if (p.geglu) {
            float gt = acc[(j + 1) % WTN][4 * g + e];
            if (ln) gt *= ln_rstd;
            if (p.bias) gt += p.bias[nw + 32];
            v *= gelu_for<T>(gt);
          }
          if (no0 + e < n_out) {
            if (R) v += TT<T>::ld(R + m * p.ldr + no0 + e);
            TT<T>::st(C + m * p.ldc + no0 + e, v * p.out_scale);
          }
        }
      }
    }
  }
}

// bf16 epilogue through LDS: the row-per-lane stores of epilogue_row touch 64 different 128-byte lines with 8 bytes each per
// wave instruction (and so do its residual loads) - on the memory-bound short-K shapes the epilogue then runs at the L2
// REQUEST rate, ~3 TB/s.  Here a wave stages its 32-row tile rows (bias / temb / GEGLU applied, rounded to bf16 like the
// reference's Linear output) in a private region of the ring slot the main loop has just released, reads them back
// row-major - 16 bytes per lane, consecutive lanes along the row - adds the residual from an equally coalesced 16-byte
// load and stores full lines.  LDS ops are inline asm with counted lgkmcnt (a C++ LDS access would make hipcc drain the
// next tile's prefetched stage first).
__device__ __forceinline__ void lds_write8(unsigned addr, unsigned lo, unsigned hi) {
  const unsigned long long v = ((unsigned long long)hi << 32) | lo;
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// V^T store of a wave's tile from the ROW-MAJOR accumulator layout (lane <-> output row m, register <-> column n): register r of tile
// (i, j) holds, over the 32 lanes of a half-wave, 32 CONSECUTIVE rows m of column n = wn0 + j*32 + 8*(r>>2) + 4*half + (r&3) - 64 (bf16)
// contiguous bytes of V^T row n.  One 2-byte (4-byte: f32) buffer store per register: the lane part of the address (m, and the half-wave's
// 4 columns) is one constant VGPR, the (tile, register) part a scalar offset; rows past M carry an out-of-range lane offset.
// emo_gemm_params.vt: the V columns of a merged q | k | v projection.  (bias / LayerNorm-folded bias are in the accumulators already.)
template <typename T, int WTM, int WTN, bool LN>
__device__ __forceinline__ void epilogue_vt(const f32x16 (&acc)[WTM][WTN], const emo_gemm_params& p, int64_t wm0, int wn0, int lane,
                                            const float (&ln_rstd)[WTM]) {
  const int half = lane >> 5, l31 = lane & 31;
  const int64_t b = wm0 / p.t_rows, ml0 = wm0 % p.t_rows;             // (a wave's 32 * WTM rows lie inside one batch: t_rows % (32 * WTM) == 0 is checked by the host)
  T* base = (T*)p.vt + b * p.t_batch_stride + (int64_t)(wn0 - p.vt_col0) * p.t_ld + ml0;
  const int n_left = p.N - wn0;                                        // columns of this wave that exist (a multiple of 8)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)((((int64_t)(n_left > 0 ? n_left : 1) - 1) * p.t_ld + 32 * WTM) * (int64_t)sizeof(T)),
                                                                      0x00020000);
  const unsigned t_ld_b = (unsigned)p.t_ld * (unsigned)sizeof(T);
#pragma unroll
  for (int i = 0; i < WTM; i++) {
    const bool m_ok = wm0 + i * 32 + l31 < p.M;
    const unsigned voff = m_ok ? (unsigned)(i * 32 + l31) * (unsigned)sizeof(T) + (unsigned)(4 * half) * t_ld_b : 0x80000000u;
    const float rs_m = LN ? ln_rstd[i] : 1.0f;
#pragma unroll
    for (int j = 0; j < WTN; j++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        if (j * 32 + 8 * g >= n_left) continue;                        // wave-uniform (N - vt_col0 is a multiple of 8)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float v = acc[i][j][4 * g + e] * rs_m;
          const unsigned soff = (unsigned)(j * 32 + 8 * g + e) * t_ld_b;
          if constexpr (sizeof(T) == 2) {
            const unsigned short h = (unsigned short)(pack2<T>(v, 0.f) & 0xffffu);
            __builtin_amdgcn_raw_buffer_store_b16(h, rs, voff, soff, 2);   // (aux 2 = non-temporal, like the staged epilogue: EPI_STORE_AUX)
          } else {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, voff, soff, 2);
          }
        }
      }
  }
}

// Addressing and order follow one observation: on gfx950 stores and loads share the in-order vmcnt, so a load issued after a
// pass's stores (the next pass's residual chunks, a row-bias quad, a spilled 64-bit row address coming back from scratch) cannot
// be waited for without waiting for those stores to be acknowledged by L2.  The first version (64-bit row addresses per chunk,
// run-time branches for bias / row bias / residual) paid that round trip in every pass: hipcc had a vmcnt(0) in front of nearly
// every LDS write.  In the instrumented build (tools/bench/gemm_timing.py) the epilogue of a 256x256 tile took 14-18 k cycles
// whether 20 or 256 CUs were storing; now 6 k (plain) / 13 k (GEGLU).  Here
//   * output and residual go through buffer resources anchored at the wave's first element, with ONE per-lane byte offset per
//     operand and pass; what moves (tile row, chunk round) is a wave-uniform addend - no 64-bit row addresses to keep alive or
//     spill; rows past M and masked columns fall outside the resource's range (stores dropped, loads zero): no branches.
//     The whole offset goes into the VECTOR offset field (the hardware's range check does not see the scalar offset), and
//     keeping the scalar field the literal 0 matters for a second reason, see the store below;
//   * the residual chunks of pass s+1 are requested BEFORE the stores of pass s (two register sets);
//   * LN / residual-or-scale / row bias are TEMPLATE flags: with them as run-time branches hipcc's waitcnt insertion fell back
//     to vmcnt(0) at every join (and spilled row addresses around them) - the very round trips this function removes.
//   * ROWW = 0: row r of the wave's tile is output row wm0 + r.  ROWW = 16 (halo conv): the tile is a stack of 16-pixel patch
//     rows, row r is output row wm0 + (r / 16) * row_pitch + r % 16 (a chunk round never straddles a patch row).
// Output tiles are STREAMED: a tile's 16-128 KB are written once and read back by another kernel (on another XCD, i.e. through the
// fabric anyway) - stored non-temporally (aux bit 1 = nt) they do not displace the operand panels the XCD's L2 holds for the
// blocks still in their main loops: -8 % at M=98304 N=2560 K=320 GEGLU, -11 % at N=960, -18 % at M=24576 N=K=640 + residual,
// neutral on the long-K shapes (profiles/r03h_epilogue_store_policy.txt; sc1 / sc0+sc1 write-through: neutral; nt on the
// residual LOADS as well gives the gain back).
static constexpr int EPI_STORE_AUX = 2;
template <typename T, int WTM, int WTN, int NW, int XBYTES, bool GEGLU, bool LN, bool RES, bool ROWB, int ROWW = 0>
__device__ __forceinline__ void epilogue_lds(const f32x16 (&acc)[WTM][WTN], const emo_gemm_params& p, int64_t wm0, int wn0, int wave, int lane,
                                                    unsigned xbase, T* __restrict__ C, const T* __restrict__ R, const float (&ln_rstd)[WTM],
                                                    int row_pitch = 0) {
  auto roff = [&](int r0) -> unsigned { return ROWW == 0 ? (unsigned)r0 : (unsigned)((r0 / (ROWW ? ROWW : 1)) * row_pitch + r0 % (ROWW ? ROWW : 1)); };
  static_assert(sizeof(T) == 2, "the staged epilogue is for the 2-byte element types");
  constexpr int OTW = GEGLU ? WTN / 2 : WTN;                  // 32-column output tiles per wave row
  constexpr int JMAX = (XBYTES / (NW * 32) - 16) / 64;        // tiles per pass that fit this wave's share of the slot
  constexpr int JFIT = JMAX < OTW ? JMAX : OTW;
  constexpr int JG = JFIT >= 4 ? 4 : (JFIT >= 2 ? 2 : 1);     // power of two: a chunk round then covers whole rows
  static_assert(JFIT >= 1, "transpose slot too small");
  constexpr int NPO = (OTW + JG - 1) / JG, NPASS = WTM * NPO; // passes per tile row, passes in all
  constexpr int PITCH = JG * 64 + 16;                          // bytes per staged row (+16: rows on different banks)
  constexpr int NIT = 2 * JG;                                  // 16-byte chunks per lane per pass: 32 rows * 4*nt chunks / 64 lanes
  const int half = lane >> 5, l31 = lane & 31;
  const unsigned xw = xbase + wave * (32 * PITCH);
  const int n_out = GEGLU ? p.N / 2 : p.N;
  const int oc0 = GEGLU ? (wn0 >> 1) : wn0;                    // first output column of this wave
  const int64_t rows_left = p.M - wm0;
  auto span = [&](int64_t ld) -> int {                         // bytes from the wave's first element to the end of its last valid row
    const int64_t b = (rows_left > 0 && n_out > oc0) ? ((rows_left - 1) * ld + (n_out - oc0)) * 2 : 0;
    return (int)(b > 0x7fffffff ? 0x7fffffff : b);
  };
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)(C + wm0 * p.ldc + oc0), 0, span(p.ldc), 0x00020000);
  const bool has_r = RES && R != nullptr;
  const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void*)(has_r ? R + wm0 * p.ldr + oc0 : C), 0, has_r ? span(p.ldr) : 0, 0x00020000);
  const unsigned ldc2 = (unsigned)p.ldc * 2u, ldr2 = (unsigned)p.ldr * 2u;
  // row bias (f32, one row per batch of rows_per_batch output rows): the whole table as one resource, rows past M clamp to the last
  const int64_t rb_rows = ROWB ? (p.M + p.rows_per_batch - 1) / p.rows_per_batch : 0;
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)(ROWB ? (const void*)p.rowbias : (const void*)C), 0,
                                                                           ROWB ? (int)(((rb_rows - 1) * p.ld_rowbias + p.N) * 4) : 0, 0x00020000);
  u32x4_t rv[2][RES ? NIT : 1];
  // pass s -> (tile row i, first output tile ot0, tiles nt); chunk idx = it*64 + lane -> row = it*(64/cw) + lane/cw, c = lane%cw
  auto lane_off = [&](int cw, int ot0, unsigned ld2) -> unsigned {
    const int lrow = lane / cw, lc = lane - lrow * cw;
    return (oc0 + ot0 * 32 + lc * 8 < n_out) ? (unsigned)lrow * ld2 + (unsigned)lc * 16u : 0x80000000u;   // masked column: out of range (num_records < 2^31)
  };
  auto load_res = [&](auto S) {
    constexpr int s_ = decltype(S)::value, i = s_ / NPO, ot0 = (s_ % NPO) * JG, nt = (OTW - ot0) < JG ? (OTW - ot0) : JG, cw = nt * 4;
    static_assert((cw & (cw - 1)) == 0, "chunks per staged row must be a power of two");
    if constexpr (RES) {
      const unsigned vo = lane_off(cw, ot0, ldr2);
#pragma unroll
      for (int it = 0; it < 2 * nt; it++)
        rv[s_ & 1][it] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, vo + roff(i * 32 + it * (64 / cw)) * ldr2 + (unsigned)(ot0 * 64), 0, 0);   // (the range check sees the vector offset only)
    }
  };
  load_res(std::integral_constant<int, 0>{});
  static_for<NPASS>([&](auto S) {
    constexpr int s_ = decltype(S)::value, i = s_ / NPO, ot0 = (s_ % NPO) * JG, nt = (OTW - ot0) < JG ? (OTW - ot0) : JG, cw = nt * 4;
    // (the bias is in the accumulators - init_acc_bias / the LayerNorm fold; the per-batch row bias is not)
    unsigned rb_off = 0;
    if constexpr (ROWB) {
      int64_t m_lane = ROWW ? wm0 : wm0 + i * 32 + l31;   // (a patch lies inside one frame: one bias row)
      if (m_lane >= p.M) m_lane = p.M - 1;
      rb_off = (unsigned)((m_lane / p.rows_per_batch) * p.ld_rowbias) * 4u;
    }
    // ---- phase 1: lane <-> row, quads of 4 columns -> LDS (columns past N carry junk: they are masked at the store)
#pragma unroll
    for (int t = 0; t < nt; t++) {
      const int jv = GEGLU ? 2 * (ot0 + t) : (ot0 + t);
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int nw0 = wn0 + jv * 32 + 8 * g + 4 * half;    // column in W-row space
        float o[4] = {acc[i][jv][4 * g], acc[i][jv][4 * g + 1], acc[i][jv][4 * g + 2], acc[i][jv][4 * g + 3]};
        if constexpr (LN) { const float rs = ln_rstd[i]; o[0] *= rs; o[1] *= rs; o[2] *= rs; o[3] *= rs; }   // LayerNorm fold: finish with rstd_m
        if constexpr (ROWB) {
          const u32x4_t bq = __builtin_amdgcn_raw_buffer_load_b128(rs_b, rb_off + (unsigned)nw0 * 4u, 0, 0);
          o[0] += __uint_as_float(bq.x); o[1] += __uint_as_float(bq.y); o[2] += __uint_as_float(bq.z); o[3] += __uint_as_float(bq.w);
        }
        if constexpr (GEGLU) {
          float gt[4] = {acc[i][jv + 1][4 * g], acc[i][jv + 1][4 * g + 1], acc[i][jv + 1][4 * g + 2], acc[i][jv + 1][4 * g + 3]};
          if constexpr (LN) { const float rs = ln_rstd[i]; gt[0] *= rs; gt[1] *= rs; gt[2] *= rs; gt[3] *= rs; }
          geglu_poly2(o[0], o[1], gt[0], gt[1], o[0], o[1]);
          geglu_poly2(o[2], o[3], gt[2], gt[3], o[2], o[3]);
        }
        lds_write8(xw + l31 * PITCH + t * 64 + (8 * g + 4 * half) * 2, pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3]));
      }
    }
    wait_lgkmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 2: read back row-major; request the NEXT pass's residual chunks; add this pass's residual; store full lines
    uint4 xv[NIT];
    const int lrow = lane / cw, lc = lane - lrow * cw;
#pragma unroll
    for (int it = 0; it < 2 * nt; it++) xv[it] = lds_read16(xw + (it * (64 / cw) + lrow) * PITCH + lc * 16);
    if constexpr (s_ + 1 < NPASS) load_res(std::integral_constant<int, s_ + 1>{});
    wait_lgkmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
    const unsigned vo = lane_off(cw, ot0, ldc2);
#pragma unroll
    for (int it = 0; it < 2 * nt; it++) {
      static_assert(ROWW == 0 || (64 / cw <= ROWW && ROWW % (64 / cw) == 0), "a chunk round must stay inside a patch row");
      const unsigned so = roff(i * 32 + it * (64 / cw)) * ldc2 + (unsigned)(ot0 * 64);
      u32x4_t out = {xv[it].x, xv[it].y, xv[it].z, xv[it].w};
      if constexpr (RES) {
        float x[8], r[8];
        unpack16<T>(xv[it], x);
        unpack16<T>(make_uint4(rv[s_ & 1][it].x, rv[s_ & 1][it].y, rv[s_ & 1][it].z, rv[s_ & 1][it].w), r);   // (no residual: out of range, zeros)
#pragma unroll
        for (int e = 0; e < 8; e++) x[e] = (x[e] + r[e]) * p.out_scale;
        const uint4 pk = pack16<T>(x);
        out = u32x4_t{pk.x, pk.y, pk.z, pk.w};
      }
      // (the moving part of the address goes into the VECTOR offset, the scalar offset stays the literal 0: with an SGPR
      // offset hipcc lets the next chunk's unpack overwrite the store's data registers in the very next instruction, and the
      // store then writes that instead for some lanes - seen as `x << 16` patterns in the last tile of a pass once a block
      // walks several tiles.  Without an SGPR offset the compiler applies the > 8-byte store-data hazard rule.)
      __builtin_amdgcn_raw_buffer_store_b128(out, rs_c, vo + so, 0, EPI_STORE_AUX);
    }
    __builtin_amdgcn_sched_barrier(0);
  });
}

// Accumulators start at the bias (row-major layout: register quad g of tile j <-> columns j*32 + 8*g + 4*half + {0..3}):
// the epilogue then has no bias loads (an L2 round trip per pass with only 2-4 waves per SIMD to hide it) or adds.
template <int WTM, int WTN>
__device__ __forceinline__ void init_acc_bias(f32x16 (&acc)[WTM][WTN], const float* __restrict__ bias, int wn0, int half, int N) {
#pragma unroll
  for (int j = 0; j < WTN; j++)
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int n0 = wn0 + j * 32 + 8 * g + 4 * half;
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bias && n0 < N) b4 = *(const float4*)(bias + n0);   // N % 4 == 0 checked by the caller
#pragma unroll
      for (int i = 0; i < WTM; i++) { acc[i][j][4 * g] = b4.x; acc[i][j][4 * g + 1] = b4.y; acc[i][j][4 * g + 2] = b4.z; acc[i][j][4 * g + 3] = b4.w; }
    }
}

// LayerNorm fold: out = rstd_m * (x . W'^T - mean_m * colsum[n]) + b'[n] = rstd_m * (acc0 + x . W'^T) with
// acc0 = b'[n] / rstd_m - mean_m * colsum[n]: the whole correction enters through the accumulator init (its loads hide behind
// the tile's first stage like the bias init) and the epilogue is one multiply per element.
template <int WTM, int WTN>
__device__ __forceinline__ void init_acc_ln(f32x16 (&acc)[WTM][WTN], const float* __restrict__ bias, const float* __restrict__ colsum,
                                            const float (&mean)[WTM], const float (&rstd)[WTM], int wn0, int half, int N) {
  float stdv[WTM];
#pragma unroll
  for (int i = 0; i < WTM; i++) stdv[i] = 1.0f / rstd[i];
#pragma unroll
  for (int j = 0; j < WTN; j++)
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int n0 = wn0 + j * 32 + 8 * g + 4 * half;
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), s4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n0 < N) {   // N % 4 == 0 checked by the caller
        s4 = *(const float4*)(colsum + n0);
        if (bias) b4 = *(const float4*)(bias + n0);
      }
#pragma unroll
      for (int i = 0; i < WTM; i++) {
        acc[i][j][4 * g] = fmaf(-mean[i], s4.x, b4.x * stdv[i]); acc[i][j][4 * g + 1] = fmaf(-mean[i], s4.y, b4.y * stdv[i]);
        acc[i][j][4 * g + 2] = fmaf(-mean[i], s4.z, b4.z * stdv[i]); acc[i][j][4 * g + 3] = fmaf(-mean[i], s4.w, b4.w * stdv[i]);
      }
    }
}
// V^T layout: lane <-> column n = wn0 + j*32 + l31, register 4g + e <-> row 8g + 4half + e (statistics of row r live in lane r)
template <int WTM, int WTN>
__device__ __forceinline__ void init_acc_ln_trans(f32x16 (&acc)[WTM][WTN], const float* __restrict__ bias, const float* __restrict__ colsum,
                                                  const float (&mean)[WTM], const float (&rstd)[WTM], int wn0, int l31, int half, int N) {
  float sn[WTN], bnv[WTN];
#pragma unroll
  for (int j = 0; j < WTN; j++) {
    const int n = wn0 + j * 32 + l31;
    sn[j] = n < N ? colsum[n] : 0.f;
    bnv[j] = (bias && n < N) ? bias[n] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < WTM; i++) {
    const float stdv = 1.0f / rstd[i];
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float rm = __shfl(mean[i], 8 * g + 4 * half + e, 64), rs = __shfl(stdv, 8 * g + 4 * half + e, 64);
#pragma unroll
        for (int j = 0; j < WTN; j++) acc[i][j][4 * g + e] = fmaf(-rm, sn[j], bnv[j] * rs);
      }
  }
}

struct ConvRow { int img, iy0, ix0; };

// PP ("ping-pong", 256x256 dense tiles only): the two wave rows of the block (waves 0-3 = group 0, 4-7 = group 1: one wave of
// each per SIMD) run HALF A STAGE APART - while one group is in its 32-MFMA phase the other issues its LDS-DMA and waits, so
// the matrix pipe of a SIMD is fed by one wave at a time and the DMA issue (~1000 cycles per stage and wave in the lockstep
// loop, during which BOTH waves of a SIMD are off the pipe) hides behind the partner's MFMAs.  Two barriers per stage delimit
// "epochs"; group 0 issues in even epochs and multiplies in odd ones, group 1 the other way round, one stage behind:
//   epoch 2s   : G0 requests A0(s+1), then waits for A0(s)      | G1 multiplies stage s-1, then waits for B(s), A1(s)
//   epoch 2s+1 : G0 multiplies stage s                          | G1 requests B(s+1), A1(s+1)
// The A half-panels are private to a group (rows 0-127 / 128-255); the W panel is read by both, so it is loaded by the LAGGING
// group only: B(s+1) overwrites B(s-1), which group 1 reads until the end of epoch 2s - nobody may write it before epoch 2s+1,
// and that is group 1's request epoch.  A plain 2-deep ring (128 KB) therefore suffices.  Every wait is the issuing wave's
// own vmcnt followed by a barrier the readers pass (LDS-DMA data is ordered by nothing else).  The offset collapses at a
// tile's end (joint LDS-staged epilogue) - one half-idle epoch per group and tile.
#ifndef EMO_GEMM_PH
#define EMO_GEMM_PH 1      // 1: the ping-pong tile runs the PHASE main loop (below); 0: the stage-granular ping-pong of round 3 (A/B builds)
#endif
template <typename T, bool CONV, bool TRANS, int WTM, int WTN, int WVM, int WVN, int NS, bool LN = false, bool PP = false>
__global__ __launch_bounds__(64 * WVM * WVN, (GemmTile<WTM, WTN, WVM, WVN, NS>::WPE)) void gemm_kernel(const emo_gemm_params p) {
  static_assert(!(LN && CONV), "the LayerNorm fold is for the dense loader");
  static_assert(!PP || (!CONV && !TRANS && WTM == 4 && WTN == 2 && WVM == 2 && WVN == 4 && NS == 2), "ping-pong: dense 256x256 tiles, 2-deep ring");
  using Tile = GemmTile<WTM, WTN, WVM, WVN, NS>;
  constexpr int NW = Tile::NW;
  constexpr int V = TT<T>::VEC;          // elements per 16 B
  constexpr int BK = KBYTES / (int)sizeof(T);
  constexpr int BM = Tile::BM, BN = Tile::BN, LA = Tile::LA, LB = Tile::LB, LPS = Tile::LPS;
  extern __shared__ __attribute__((aligned(128))) unsigned char lds[];   // 128: the XOR k-step addressing below

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wvm = wave / WVN, wvn = wave % WVN;
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (int)((p.M + BM - 1) / BM);
  const int tiles_all = tiles_m * tiles_n;
  const int G = gridDim.x;                     // persistent: block b works tiles b, b+G, b+2G, ... of the XCD-aware order
  // XCD-aware tile order (bijective): stream index i runs on XCD i % 8 (G is a multiple of 8 whenever a block has more
  // than one tile); XCD x owns a contiguous run of tiles (n fastest), so the A rows it re-reads stay in that XCD's L2
  auto tile_of = [&](int i) {
    const int qn = tiles_all >> 3, rn = tiles_all & 7, x = i & 7, idx = i >> 3;
    return (x < rn ? x * (qn + 1) : rn * (qn + 1) + (x - rn) * qn) + idx;
  };
  // GROUPED order inside an XCD's run: the ~32 * BPC blocks of an XCD work on CONSECUTIVE positions of the order at any time.
  // With n fastest that is one tile row - one A panel shared by all, but tiles_n DIFFERENT W panels, each used by one CU:
  // at N = 8192 the XCD's L2 hit rate was 48 % (A hits, W misses), VMEM latency 1160 cycles, 37 % of the wave cycles parked
  // in s_waitcnt (hipBLASLt on the same shape: 80 %, 450 cycles, 5 %).  Bands of gm tile rows with m fastest inside a band
  // make the concurrent set a gm x (32 * BPC / gm) block: 4 + 8 panels instead of 1 + 32.  Narrow outputs (tiles_n < 5) keep
  // n fastest - their concurrent set already spans several rows.  (p.tile >> 4 overrides gm: tools/bench.)
  const int gm_hint = (p.tile >> 4) & 15;
  // (bands of 4 already from 5 column tiles on: -3 % at M=98304 N=2560 K=320 GEGLU, -2..3 % at N=640 K=2560 / N=1280 K=5120,
  // neutral on the square projections - tools/bench/gemm_gm_sweep.py)
  const int gm = gm_hint > 0 ? gm_hint : (tiles_n > 12 ? (Tile::BPC >= 2 ? 8 : 4) : (tiles_n >= 5 ? 4 : 1));
  auto tile_mn = [&](int t, int& tm, int& tn) {
    if (gm <= 1) { tm = t / tiles_n; tn = t - tm * tiles_n; return; }
    const int per = gm * tiles_n, g = t / per, r = t - g * per, m0 = g * gm;
    const int gsz = tiles_m - m0 < gm ? tiles_m - m0 : gm;
    tn = r / gsz; tm = m0 + (r - tn * gsz);
  };
  const int nsplit = p.split_k > 1 ? p.split_k : 1;

  const T* __restrict__ A = (const T*)p.A;
  const T* __restrict__ W = (const T*)p.W;
  const T* zero = (const T*)g_zero_page;

  const int nk_all = (p.K + BK - 1) / BK;
  const int nk_per = (nk_all + nsplit - 1) / nsplit;
  const int kt0 = blockIdx.y * nk_per;
  const int nk = (kt0 + nk_per <= nk_all ? nk_per : nk_all - kt0);   // may be <= 0 for a trailing empty slice

  // ---- loader state.  The loader streams (tile, k-stage) pairs NS-1 stages AHEAD of the MFMA loop and does not stop at
  // tile boundaries: while a tile's epilogue runs, the first stages of the block's next tile are already in flight.
  // glds #i of this wave fills LDS rows (i*NW + wave)*(64/CPR) .. of the operand; lane l writes physical chunk l%CPR of
  // row l/CPR, which holds LOGICAL chunk (l%CPR) ^ swz(row)
  const int lrow = lane / CPR, lchunk = lane % CPR;
  // the swizzle key of row (i*NW + wave)*(64/CPR) + lrow does not depend on the round i (NW*(64/CPR)/RPB is a multiple of
  // CPR): one logical chunk index per lane serves every glds of A and B
  static_assert((NW * (64 / CPR) / RPB) % CPR == 0, "swizzle key must be round-independent");
  const int klog = lchunk ^ swz(wave * (64 / CPR) + lrow);
  ConvRow a_cr[LA];
  bool a_ok[LA];          // conv loader only
  // Dense operands go through BUFFER descriptors anchored at the tile's first A row / W row (rebuilt per tile: scalar work): one
  // 32-bit lane offset per request that never changes (row in the tile x leading dimension + the lane's swizzled chunk), the K
  // stage in the scalar offset.  Rows past M / N lie outside the descriptor's range and read 0 (their products land in
  // accumulator rows / columns that are never stored): no 64-bit pointers to keep and advance, no clamping, no zero page
  // (round 5: the halo conv's loaders went this way first, -3.5 % on the conv family).
  unsigned a_off[LA], b_off[LB];
#pragma unroll
  for (int i = 0; i < LA; i++) a_off[i] = (unsigned)((i * NW + wave) * (64 / CPR) + lrow) * (unsigned)p.lda * (unsigned)sizeof(T) + (unsigned)(klog * 16);
#pragma unroll
  for (int i = 0; i < LB; i++) b_off[i] = (unsigned)((i * NW + wave) * (64 / CPR) + lrow) * (unsigned)p.K * (unsigned)sizeof(T) + (unsigned)(klog * 16);
  __amdgpu_buffer_rsrc_t a_rs, b_rs;
  int l_iter = blockIdx.x, l_kt = 0;   // the loader's tile (stream index) and next stage within the slice
  auto setup_loader = [&](int iter) {
    int ltm, ltn;
    tile_mn(tile_of(iter), ltm, ltn);
    const int64_t lbm = (int64_t)ltm * BM;
    const int lbn = ltn * BN;
#pragma unroll
    for (int i = 0; i < LA; i++) {
      const int row = (i * NW + wave) * (64 / CPR) + lrow;
      const int64_t m = lbm + row;
      a_ok[i] = row < BM && m < p.M;
      if (CONV) {
        const int hw = p.Ho * p.Wo;
        const int64_t mm = a_ok[i] ? m : 0;
        const int img = (int)(mm / hw), rem = (int)(mm % hw);
        const int oy = rem / p.Wo, ox = rem % p.Wo;
        const int pad_tl = p.conv_asym ? 0 : 1;   // (0, 1, 0, 1) padding: nothing above / left of the image
        a_cr[i].img = img; a_cr[i].iy0 = oy * p.stride - pad_tl; a_cr[i].ix0 = ox * p.stride - pad_tl;
      }
    }
    if constexpr (!CONV) {
      const int64_t arows = p.M - lbm < BM ? p.M - lbm : BM;
      a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(A + lbm * p.lda), 0, (int)(((arows - 1) * p.lda + p.K) * (int64_t)sizeof(T)), 0x00020000);
    }
    // per-instance weights (emo_gemm_params.w_slab_rows: GroupNorm folded into proj_in): the tile's rows pick the slab
    const T* Wt = W;
    if constexpr (!CONV) { if (p.w_slab_rows > 0) Wt += (lbm / p.w_slab_rows) * p.w_slab_stride; }
    const int64_t brows = p.N - lbn < BN ? p.N - lbn : BN;
    b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(Wt + (int64_t)lbn * p.K), 0, (int)(brows * (int64_t)p.K * (int64_t)sizeof(T)), 0x00020000);
  };
  const bool k_ragged = (p.K % BK) != 0;            // only the very last stage can run past K
  const bool cin_aligned = CONV && (p.Cin % BK) == 0; // a stage then lies inside one 3x3 tap (tap is wave-uniform)

  // one glds of operand A (index i < LA) / B (i < LB) of k-stage kt (absolute) into ring slot `slot`
  auto issue_a = [&](int kt, int slot, auto I) {
    constexpr int i = decltype(I)::value;
    unsigned char* sa = lds + slot * Tile::STAGE_BYTES;
    const bool tail = k_ragged && (kt + 1) * BK > p.K;
    if constexpr (!CONV) {
      // (a chunk past K in the last, ragged stage: the range check sees the lane offset only - send it out of range)
      const unsigned voff = (tail && kt * BK + klog * V >= p.K) ? 0x80000000u : a_off[i];
      glds16_buffer(a_rs, sa + (i * NW + wave) * 1024, voff, (unsigned)kt * (unsigned)KBYTES);
    } else {
      const int Hin = p.up_h ? p.up_h : (p.upsample2x ? 2 * p.H : p.H), Win = p.up_h ? p.up_w : (p.upsample2x ? 2 * p.W_ : p.W_);
      const int k0 = kt * BK + klog * V;
      int tap, ci;
      if (cin_aligned) { tap = (kt * BK) / p.Cin; ci = kt * BK - tap * p.Cin + klog * V; }   // tap is wave-uniform (SALU)
      else { tap = k0 / p.Cin; ci = k0 - tap * p.Cin; }
      const int ky = tap / 3, kx = tap - ky * 3;
      int iy = a_cr[i].iy0 + ky, ix = a_cr[i].ix0 + kx;
      const T* src = zero;
      if (a_ok[i] && k0 < p.K && iy >= 0 && iy < Hin && ix >= 0 && ix < Win) {
        if (p.upsample2x) { iy >>= 1; ix >>= 1; }
        else if (p.up_h) { iy = iy * p.H / p.up_h; ix = ix * p.W_ / p.up_w; }   // nearest to an explicit size (rare path)
        src = A + (((int64_t)a_cr[i].img * p.H + iy) * p.W_ + ix) * p.lda + ci;
      }
      EMO_GLDS16(src, sa + (i * NW + wave) * 1024);
    }
  };
  auto issue_b = [&](int kt, int slot, auto I) {
    constexpr int i = decltype(I)::value;
    unsigned char* sb = lds + slot * Tile::STAGE_BYTES + Tile::A_BYTES;
    const bool tail = k_ragged && (kt + 1) * BK > p.K;
    const unsigned voff = (tail && kt * BK + klog * V >= p.K) ? 0x80000000u : b_off[i];
    glds16_buffer(b_rs, sb + (i * NW + wave) * 1024, voff, (unsigned)kt * (unsigned)KBYTES);
  };
  // after the last glds of a stage: step the loader to the next stage of the stream (next tile when this one is done)
  auto advance_loader = [&]() {
    if (++l_kt >= nk) {
      l_kt = 0;
      l_iter += G;
      if (l_iter < tiles_all) setup_loader(l_iter);
    }
  };

  // fragment read addresses (LDS byte offsets, stage-relative), swizzled
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  // k-step kk reads chunk (kk*2 + half) ^ swz(r) = ((half ^ swz(r)) ^ (kk << 1)): with 128-byte aligned stages the address
  // of step kk is the step-0 address XOR (kk << 5) - one register per fragment row instead of KSTEPS
  static_assert(KSTEPS * 32 <= KBYTES && (Tile::STAGE_BYTES % 128) == 0 && (Tile::A_BYTES % 128) == 0, "XOR k-step addressing");
  unsigned fa0[WTM], fb0[WTN];
#pragma unroll
  for (int i = 0; i < WTM; i++) {
    const int r = wvm * 32 * WTM + i * 32 + l31;
    fa0[i] = lds_base + r * KBYTES + ((half ^ swz(r)) * 16);
  }
#pragma unroll
  for (int j = 0; j < WTN; j++) {
    const int r = wvn * 32 * WTN + j * 32 + l31;
    fb0[j] = lds_base + Tile::A_BYTES + r * KBYTES + ((half ^ swz(r)) * 16);
  }

  T* __restrict__ C = (T*)p.C;
  const T* __restrict__ R = (const T*)p.residual;
  // row-major single-pass outputs start their accumulators at the bias; the epilogues then see bias == nullptr
  const bool bias_in_acc = !LN && !TRANS && nsplit == 1 && p.bias != nullptr && (p.N & 3) == 0;
  emo_gemm_params pe = p;
  if (bias_in_acc || LN) pe.bias = nullptr;   // (LN: the folded bias enters the accumulator init scaled by 1 / rstd)
  // the per-batch row bias (temb of a resnet, the positional-encoding term of a temporal q|k|v projection) is uniform over a tile
  // whose rows lie inside one batch of rows_per_batch rows: it then starts the accumulators as well (under the LayerNorm fold
  // divided by rstd_m like the bias) and the epilogue has no row-bias loads (per staged pass: one L2 round trip each)
#ifndef EMO_GEMM_RB_IN_ACC
#define EMO_GEMM_RB_IN_ACC 1   // (0: tools/bench A/B build)
#endif
  const bool rb_in_acc = EMO_GEMM_RB_IN_ACC && !CONV && !TRANS && nsplit == 1 && p.rowbias != nullptr && (p.N & 3) == 0 && (p.ld_rowbias & 3) == 0 &&
                         p.rows_per_batch > 0 && (p.rows_per_batch % BM) == 0;
  if (rb_in_acc) pe.rowbias = nullptr;
  // coalesced LDS-staged epilogue (bf16, row-major, single pass): needs whole 16-byte chunks everywhere
  const int n_out_all = p.geglu ? p.N / 2 : p.N;
  const bool use_lds_epi = !TRANS && sizeof(T) == 2 && nsplit == 1 && nk > 0 && (n_out_all & 7) == 0 && (p.N & 3) == 0 &&
                           (p.ldc & 7) == 0 && (!R || (p.ldr & 7) == 0) && (!p.rowbias || (p.ld_rowbias & 3) == 0) &&
                           (!p.geglu || (WTN % 2 == 0 && !p.rowbias)) && (!p.rowbias || p.rows_per_batch > 0);

  // ---- ping-pong loader (PP): group g's waves load A rows [g*128, g*128+128) of the tile, group 1's also the whole W panel;
  // glds round i of wave wg covers LDS rows (i*4 + wg)*8 .. +8 of the (half-)panel.  Source pointers are formed per request
  // from two per-lane row numbers (rows past M / N clamp to the last valid one, as in the lockstep loader).
  const int pp_grp = wvm, pp_wg = wave & 3;
  const int pp_klog = lchunk ^ swz(pp_wg * (64 / CPR) + lrow);
  int64_t pp_arow = 0; int pp_brow = 0, pp_koff = 0;
  const T* pp_w = W;      // the tile's weight slab (per-instance weights, emo_gemm_params.w_slab_rows - as in setup_loader)
  auto pp_setup = [&](int iter) {
    int ltm, ltn;
    tile_mn(tile_of(iter), ltm, ltn);
    pp_arow = (int64_t)ltm * BM + pp_grp * 128 + pp_wg * (64 / CPR) + lrow;
    pp_brow = ltn * BN + pp_wg * (64 / CPR) + lrow;
    pp_koff = pp_klog * V;
    pp_w = p.w_slab_rows > 0 ? W + (((int64_t)ltm * BM) / p.w_slab_rows) * p.w_slab_stride : W;
  };
  auto pp_issue = [&](int slot) {     // this wave's share of the loader's current stage into ring slot `slot`
    unsigned char* sa = lds + slot * Tile::STAGE_BYTES;
    static_for<4>([&](auto I) {
      constexpr int i = decltype(I)::value;
      int64_t m = pp_arow + i * 32;
      if (m >= p.M) m = p.M - 1;
      EMO_GLDS16(A + m * p.lda + pp_koff, sa + (pp_grp * 16 + i * 4 + pp_wg) * 1024);
    });
    if (pp_grp == 1) {
      static_for<8>([&](auto I) {
        constexpr int i = decltype(I)::value;
        int n = pp_brow + i * 32;
        if (n >= p.N) n = p.N - 1;
        EMO_GLDS16(pp_w + (int64_t)n * p.K + pp_koff, sa + Tile::A_BYTES + (i * 4 + pp_wg) * 1024);
      });
    }
  };
  auto pp_advance = [&]() {
    pp_koff += BK;
    if (++l_kt >= nk) {
      l_kt = 0;
      l_iter += G;
      if (l_iter < tiles_all) pp_setup(l_iter);
    }
  };

  // ---- phase loader (PP with EMO_GEMM_PH): a K-tile's stage is requested in QUARTERS of 128 rows that follow the phases'
  // last reads - A0 / A1 = the first / second 64 rows of each wave group's 128, B0 / B1 = the first / second 32 rows of each wave
  // column's 64 (W rows = output columns).  A quarter is 16 pieces of 8 rows; wave w requests pieces 2w, 2w + 1.  Pointers are
  // formed per request from (tile base, row, K-tile): the stream crosses tile boundaries without a loader state machine.
  // Requests go through buffer resources anchored at the tile's first A row / W row: one 32-bit lane offset per piece, the
  // K-tile in the scalar offset, the LDS address in M0 - 3-4 instructions per request; rows past M / N fall outside the
  // resource's range and read zeros (their products land in accumulator rows / columns that are never stored).
  int ph_abase[2], ph_bbase[2];     // [piece] first row (wave-uniform) of this wave's piece of quarter A0 / B0 (A1: + 64, B1: + 32)
  unsigned ph_avoff[2], ph_bvoff[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int pr0 = (2 * wave + j) * (64 / CPR);
    ph_abase[j] = (pr0 >> 6) * 128 + (pr0 & 63);
    ph_bbase[j] = (pr0 >> 5) * 64 + (pr0 & 31);
    const int ra = ph_abase[j] + lrow, rb = ph_bbase[j] + lrow;     // (swz(r + 64) == swz(r), swz(r + 32) == swz(r): one key per piece)
    ph_avoff[j] = (unsigned)ra * (unsigned)p.lda * (unsigned)sizeof(T) + (unsigned)((lchunk ^ swz(ra)) * 16);
    ph_bvoff[j] = (unsigned)rb * (unsigned)p.K * (unsigned)sizeof(T) + (unsigned)((lchunk ^ swz(rb)) * 16);
  }
  struct PhTile { __amdgpu_buffer_rsrc_t ra, rb; };
  auto ph_tile = [&](int iter) -> PhTile {
    int ltm, ltn;
    tile_mn(tile_of(iter), ltm, ltn);
    const int64_t tbm = (int64_t)ltm * BM;
    const int tbn = ltn * BN;
    const T* tw = p.w_slab_rows > 0 ? W + (tbm / p.w_slab_rows) * p.w_slab_stride : W;
    const int64_t arows = p.M - tbm < BM ? p.M - tbm : BM, brows = p.N - tbn < BN ? p.N - tbn : BN;
    PhTile t;
    t.ra = __builtin_amdgcn_make_buffer_rsrc((void*)(A + tbm * p.lda), 0, (int)(((arows - 1) * p.lda + p.K) * (int64_t)sizeof(T)), 0x00020000);
    t.rb = __builtin_amdgcn_make_buffer_rsrc((void*)(tw + (int64_t)tbn * p.K), 0, (int)(brows * (int64_t)p.K * (int64_t)sizeof(T)), 0x00020000);
    return t;
  };
  auto ph_req_a = [&](int hf, const PhTile& t, int kt, int slot) {
#pragma unroll
    for (int j = 0; j < 2; j++)
      glds16_buffer(t.ra, lds + slot * Tile::STAGE_BYTES + (ph_abase[j] + hf * 64) * KBYTES, ph_avoff[j] + (unsigned)(hf * 64) * (unsigned)p.lda * (unsigned)sizeof(T),
                    (unsigned)kt * (unsigned)KBYTES);
  };
  auto ph_req_b = [&](int hf, const PhTile& t, int kt, int slot) {
#pragma unroll
    for (int j = 0; j < 2; j++)
      glds16_buffer(t.rb, lds + slot * Tile::STAGE_BYTES + Tile::A_BYTES + (ph_bbase[j] + hf * 32) * KBYTES,
                    ph_bvoff[j] + (unsigned)(hf * 32) * (unsigned)p.K * (unsigned)sizeof(T), (unsigned)kt * (unsigned)KBYTES);
  };

  // stream prologue: NS-1 stages in flight
  int gs = 0;   // stream stage counter of the MFMA loop (ring slot = gs % NS)
  if constexpr (PP && EMO_GEMM_PH) {
    if (nk > 0 && (int)blockIdx.x < tiles_all) {   // all four quarters of the first tile's K-tile 0
      const PhTile ft = ph_tile(blockIdx.x);
      ph_req_a(0, ft, 0, 0); ph_req_b(0, ft, 0, 0); ph_req_b(1, ft, 0, 0); ph_req_a(1, ft, 0, 0);
    }
  } else
  if constexpr (PP) {
    if (nk > 0 && l_iter < tiles_all) {
      pp_setup(l_iter);
      pp_issue(0);
      pp_advance();
    }
  } else
  if (nk > 0) {
    setup_loader(l_iter);
#pragma unroll
    for (int s = 0; s < NS - 1; s++)
      if (l_iter < tiles_all) {
        static_for<LA>([&](auto I) { issue_a(kt0 + l_kt, s, I); });
        static_for<LB>([&](auto I) { issue_b(kt0 + l_kt, s, I); });
        advance_loader();
      }
  }

  for (int c_iter = blockIdx.x; c_iter < tiles_all; c_iter += G) {
  int ctm, ctn;
  tile_mn(tile_of(c_iter), ctm, ctn);
  const int64_t bm = (int64_t)ctm * BM;
  const int bn = ctn * BN;
  const int tiles_left = (tiles_all - 1 - c_iter) / G;   // tiles of this block after this one

  f32x16 acc[WTM][WTN];
  // LayerNorm fold: (mean, rstd) of this lane's A rows (lane <-> row l31 of MFMA tile row i) from the statistics pass
  float ln_mean[WTM], ln_rstd[WTM];
#pragma unroll
  for (int i = 0; i < WTM; i++) { ln_mean[i] = 0.f; ln_rstd[i] = 1.f; }
  if constexpr (LN) {
#pragma unroll
    for (int i = 0; i < WTM; i++) {
      int64_t m = bm + wvm * 32 * WTM + i * 32 + l31;
      if (m >= p.M) m = p.M - 1;
      const float2 st2 = *(const float2*)(p.ln_stats + 2 * m);
      ln_mean[i] = st2.x; ln_rstd[i] = st2.y;
    }
    if constexpr (!TRANS) init_acc_ln<WTM, WTN>(acc, p.bias, p.ln_colsum, ln_mean, ln_rstd, bn + wvn * 32 * WTN, half, p.N);
    else init_acc_ln_trans<WTM, WTN>(acc, p.bias, p.ln_colsum, ln_mean, ln_rstd, bn + wvn * 32 * WTN, l31, half, p.N);
  } else if (bias_in_acc) {
    // (per-instance weights carry a per-instance bias [slab][N]: uniform over the tile's rows, so it rides here too)
    const float* tbias = p.bias;
    if constexpr (!CONV) { if (p.w_slab_rows > 0) tbias += (bm / p.w_slab_rows) * (int64_t)p.N; }
    init_acc_bias<WTM, WTN>(acc, tbias, bn + wvn * 32 * WTN, half, p.N);
  } else {
#pragma unroll
    for (int i = 0; i < WTM; i++)
#pragma unroll
      for (int j = 0; j < WTN; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  }

  if (rb_in_acc) {
    const float* trb = p.rowbias + (bm / p.rows_per_batch) * (int64_t)p.ld_rowbias;
    const int wnb = bn + wvn * 32 * WTN;
#pragma unroll
    for (int j = 0; j < WTN; j++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int n0 = wnb + j * 32 + 8 * g + 4 * half;
        if (n0 < p.N) {
          const float4 b4 = *(const float4*)(trb + n0);
#pragma unroll
          for (int i = 0; i < WTM; i++) {
            const float sc = LN ? 1.0f / ln_rstd[i] : 1.0f;
            acc[i][j][4 * g] = fmaf(b4.x, sc, acc[i][j][4 * g]); acc[i][j][4 * g + 1] = fmaf(b4.y, sc, acc[i][j][4 * g + 1]);
            acc[i][j][4 * g + 2] = fmaf(b4.z, sc, acc[i][j][4 * g + 2]); acc[i][j][4 * g + 3] = fmaf(b4.w, sc, acc[i][j][4 * g + 3]);
          }
        }
      }
  }

  if constexpr (PP && EMO_GEMM_PH) {
    // PHASE main loop (the structure of the guide's 256^2 8-phase template on this kernel's 32x32x16 tiles; measured standalone in
    // tools/bench/micro/gemm8p.hip, profiles/r04y_gemm8p_micro.txt).  A K-tile is 4 phases = the wave's 4 quadrants of 64 x 32:
    //   P1 (A0, B0)   P2 (A0, B1)   P3 (A1, B1)   P4 (A1, B0 from registers)
    // each  [fragment reads of the quadrant + one quarter's LDS-DMA requests]  s_barrier  lgkmcnt(0)  [8 MFMAs, nothing else]  s_barrier,
    // and wave group 1 runs ONE barrier behind group 0: one wave per SIMD multiplies from registers while the other reads and
    // requests.  Quarters are restaged behind their last read (both groups are past it two barriers later):
    //   P1: B1 of K-tile kt+1 (last read: P2 of kt-1)      P2: A1 of kt+1 (P3 of kt-1)
    //   P3: A0 of K-tile kt+2 (last read: P1 of kt)        P4: B0 of kt+2 (P1 of kt), then vmcnt(4): all of kt+1 has landed
    // The K-tile stream continues into the block's next tile; only the two requests that would land in the slot the epilogue
    // stages through (A0 / B0 of the next tile's K-tile 1) wait for the next tile's start.
    wait_vmcnt<0>();                  // K-tile 0 of this tile (requested a tile ago) and the previous epilogue's stores
    __builtin_amdgcn_s_barrier();     // ... visible to all; everyone is out of the previous epilogue's staging slot
    const bool has_next = c_iter + G < tiles_all;
    const PhTile ct = ph_tile(c_iter), nt = ph_tile(has_next ? c_iter + G : c_iter);
    ph_req_a(0, ct, 1, (gs + 1) & 1); ph_req_b(0, ct, 1, (gs + 1) & 1);      // (nk >= 2: dispatch_tile)
    if (pp_grp == 1) __builtin_amdgcn_s_barrier();
    // (Measured and not kept: B0 of K-tile kt+1 read during P4 of kt into the registers B1 has left - 8 / 4 / 8 / 4 reads per load
    // section instead of 12 / 4 / 8 / 0, the loop unrolled over two K-tiles: hipcc spills into the loop, and scratch accesses sit
    // in the same vmcnt the requests are counted on - wrong tiles and 2x slower; the same balance without unrolling - B0 read in P4 into
    // B1's registers, 16 v_mov at the next P1, a counted vmcnt in P3 - is correct and 5-9 % slower: profiles/r04ph_phase_main_loop.txt.)
    uint4 ra[2][KSTEPS], rb0[KSTEPS], rb1[KSTEPS];
    auto rd_a = [&](unsigned st, auto H) {
      constexpr int h = decltype(H)::value;
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int kk = 0; kk < KSTEPS; kk++) ra[i][kk] = lds_read16((st + fa0[h * 2 + i]) ^ (kk << 5));
    };
    auto rd_b = [&](unsigned st, auto J, uint4 (&rb)[KSTEPS]) {
      constexpr int j = decltype(J)::value;
#pragma unroll
      for (int kk = 0; kk < KSTEPS; kk++) rb[kk] = lds_read16((st + fb0[j]) ^ (kk << 5));
    };
    auto quad = [&](auto H, auto J, const uint4 (&rb)[KSTEPS]) {
      constexpr int h = decltype(H)::value, j = decltype(J)::value;
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < KSTEPS; kk++)
#pragma unroll
        for (int i = 0; i < 2; i++) acc[h * 2 + i][j] = mma16<T>(rb[kk], ra[i][kk], acc[h * 2 + i][j]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    for (int kt = 0; kt < nk; kt++, gs++) {
      const unsigned st = (gs & 1) * Tile::STAGE_BYTES;
      const int slot_c = gs & 1, slot_o = slot_c ^ 1;
      const bool last = kt == nk - 1;
      const bool t1 = !last || has_next;                          // K-tile kt+1 of the stream exists
      const bool t1_here = !last;
      const bool t2_here = kt + 2 < nk;
      const bool t2 = !last && (t2_here || has_next);             // K-tile kt+2, unless it belongs behind the epilogue
      // ---- P1
      rd_b(st, I0{}, rb0);
      __builtin_amdgcn_sched_barrier(0);
      rd_a(st, I0{});
      if (t1) ph_req_b(1, t1_here ? ct : nt, t1_here ? kt + 1 : 0, slot_o);
      __builtin_amdgcn_s_barrier();
      wait_lgkmcnt<0>();
      quad(I0{}, I0{}, rb0);
      __builtin_amdgcn_s_barrier();
      // ---- P2
      rd_b(st, I1{}, rb1);
      if (t1) ph_req_a(1, t1_here ? ct : nt, t1_here ? kt + 1 : 0, slot_o);
      __builtin_amdgcn_s_barrier();
      wait_lgkmcnt<0>();
      quad(I0{}, I1{}, rb1);
      __builtin_amdgcn_s_barrier();
      // ---- P3
      rd_a(st, I1{});
      if (t2) ph_req_a(0, t2_here ? ct : nt, t2_here ? kt + 2 : 0, slot_c);
      __builtin_amdgcn_s_barrier();
      wait_lgkmcnt<0>();
      quad(I1{}, I1{}, rb1);
      __builtin_amdgcn_s_barrier();
      // ---- P4
      if (t2) { ph_req_b(0, t2_here ? ct : nt, t2_here ? kt + 2 : 0, slot_c); wait_vmcnt<4>(); }
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      quad(I1{}, I0{}, rb0);
      __builtin_amdgcn_s_barrier();
    }
    if (pp_grp == 0) __builtin_amdgcn_s_barrier();   // realign: group 1's last barrier
  } else
  if constexpr (PP) {
    // the 32 MFMAs of one stage (4 k-steps x 8), next k-step's fragment reads between them - the lockstep loop's cluster
    auto pp_mma = [&](unsigned st) {
      uint4 fa[2][WTM], fb[2][WTN];
#pragma unroll
      for (int i = 0; i < WTM; i++) fa[0][i] = lds_read16(st + fa0[i]);
#pragma unroll
      for (int j = 0; j < WTN; j++) fb[0][j] = lds_read16(st + fb0[j]);
      constexpr int NMMA = WTM * WTN, NRD = WTM + WTN;
      __builtin_amdgcn_s_setprio(1);
      static_for<KSTEPS>([&](auto KK) {
        constexpr int kk = decltype(KK)::value, cur = kk & 1, nxt = cur ^ 1;
        wait_lgkmcnt<0>();
        __builtin_amdgcn_sched_barrier(0);
        constexpr int n_side = (kk + 1 < KSTEPS) ? NRD : 0;
        static_for<NMMA>([&](auto Q) {
          constexpr int q = decltype(Q)::value, i = q / WTN, j = q % WTN;
          acc[i][j] = mma16<T>(fb[cur][j], fa[cur][i], acc[i][j]);
          __builtin_amdgcn_sched_barrier(0);
          static_for<n_side>([&](auto O) {
            constexpr int o = decltype(O)::value;
            if constexpr ((o * NMMA) / n_side == q) {
              if constexpr (o < WTM) fa[nxt][o] = lds_read16((st + fa0[o]) ^ (((kk + 1) % KSTEPS) << 5));
              else fb[nxt][o - WTM] = lds_read16((st + fb0[o - WTM]) ^ (((kk + 1) % KSTEPS) << 5));
            }
          });
          __builtin_amdgcn_sched_barrier(0);
        });
      });
      __builtin_amdgcn_s_setprio(0);
    };
    // Both groups run the SAME loop body - [request, wait] barrier [multiply] [wait] barrier; group 1 takes one extra barrier
    // in front of it (and group 0 one behind), which puts it exactly one epoch behind:
    //   barrier #2j   : G0 leaves pre(j)    | G1 leaves mma(j-1) + its wait for stage j's B / A1 (-> published)
    //   barrier #2j+1 : G0 leaves mma(j)    | G1 leaves pre(j) = request of B(j+1), A1(j+1)
    wait_vmcnt<0>();                  // stage 0 of this tile (requested a tile ago) and the previous epilogue's stores
    __builtin_amdgcn_s_barrier();     // ... visible to all; everyone is out of the previous epilogue's staging slot
    if (pp_grp == 1) __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; kt++, gs++) {
      const unsigned st = (gs & 1) * Tile::STAGE_BYTES;
      const bool more = l_iter < tiles_all;
      // G0: A0(kt+1) over A0(kt-1) (its own last read: epoch 2kt-1).  G1: B(kt+1), A1(kt+1) over stage kt-1 (last reads: G0 in
      // epoch 2kt-1, G1 in epoch 2kt; this is epoch 2kt+1)
      if (more) { pp_issue((gs + 1) & 1); pp_advance(); }
      if (pp_grp == 0 && kt > 0) { if (more) wait_vmcnt<4>(); else wait_vmcnt<0>(); }   // A0(kt) landed (kt = 0: the tile's first wait)
      __builtin_amdgcn_s_barrier();
      pp_mma(st);
      if (pp_grp == 1) wait_vmcnt<0>();   // B(kt+1), A1(kt+1) landed: the next barrier publishes them to G0's mma(kt+1)
      __builtin_amdgcn_s_barrier();
    }
    if (pp_grp == 0) __builtin_amdgcn_s_barrier();   // realign: G1's last barrier
  } else
  for (int kt = 0; kt < nk; kt++, gs++) {
    // stage gs must have landed; up to NS-2 younger stages may still be in flight - fewer at the end of the stream, and none
    // are counted on at a tile's first stage: the previous tile's epilogue stores sit in the same counter
    const int rem = nk - 1 - kt + tiles_left * nk;   // younger stages of this block's stream still to come
    if (rem >= NS - 2 && (kt > 0 || c_iter == (int)blockIdx.x)) wait_vmcnt<(NS - 2) * LPS>();
    else if (NS > 3 && rem >= 1 && kt > 0) {   // draining: rem younger stages in flight
      if (rem == 1) wait_vmcnt<LPS>();
      else if (NS > 4 && rem == 2) wait_vmcnt<2 * LPS>();
      else if (NS > 5 && rem == 3) wait_vmcnt<3 * LPS>();
      else if (NS > 6 && rem == 4) wait_vmcnt<4 * LPS>();
      else if (NS > 7 && rem == 5) wait_vmcnt<5 * LPS>();
      else wait_vmcnt<0>();
    }
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // everyone's part of stage gs landed; everyone finished reading slot (gs-1)%NS
    const unsigned st = (gs % NS) * Tile::STAGE_BYTES;   // stage offset (fa0 / fb0 carry the LDS base)
    const bool more = l_iter < tiles_all;
    const int kt_next = kt0 + l_kt, slot_next = (gs + NS - 1) % NS;
    // Software-pipelined stage: the only exposed latency is the first k-step's fragment read.  The fragment reads
    // of step kk+1 and the next ring stage's glds (with their address arithmetic) are issued one at a time BETWEEN the
    // MFMAs of step kk, so their issue cost and latency hide under the matrix pipe.
    uint4 fa[2][WTM], fb[2][WTN];
#pragma unroll
    for (int i = 0; i < WTM; i++) fa[0][i] = lds_read16(st + fa0[i]);
#pragma unroll
    for (int j = 0; j < WTN; j++) fb[0][j] = lds_read16(st + fb0[j]);
    constexpr int NMMA = WTM * WTN, NRD = WTM + WTN;
    // the whole next ring stage is requested right behind the barrier: it then has this stage's full MFMA time to land
    // (spreading the glds between the MFMAs of all four k-steps left the last ones ~no time: 8192^3 980 -> 1130 TF/s clustered;
    // one glds per MFMA over the first one or two k-steps only: equal to clustered within the run-to-run noise, r02k)
    if (more) {
      static_for<LA>([&](auto I) { issue_a(kt_next, slot_next, I); });
      static_for<LB>([&](auto I) { issue_b(kt_next, slot_next, I); });
    }
    static_for<KSTEPS>([&](auto KK) {
      constexpr int kk = decltype(KK)::value, cur = kk & 1, nxt = cur ^ 1;
      wait_lgkmcnt<0>();                       // fragments of step kk
      __builtin_amdgcn_sched_barrier(0);
      // side ops of this cluster: the next step's fragment reads, one at a time between the MFMAs.
      // Measured and dropped (tools/bench/patches): (a) fragment reads ordered by first use + s_waitcnt lgkmcnt(n)
      // per MFMA instead of lgkmcnt(0) per step, reads front-loaded two per MFMA - 8192^3 949 -> 989 us; (b) a register-staged
      // loader (buffer_load_dwordx4 two stages ahead, ds_write_b128 into the ring) instead of LDS-DMA: the glds issue cost
      // (~1150 cycles per stage in front of the first MFMA) goes away, the k-steps grow by as much - equal within 2 %.
      constexpr int n_rd = (kk + 1 < KSTEPS) ? NRD : 0;
      constexpr int n_side = n_rd;
      static_for<NMMA>([&](auto Q) {
        constexpr int q = decltype(Q)::value, i = q / WTN, j = q % WTN;
        if constexpr (TRANS) acc[i][j] = mma16<T>(fa[cur][i], fb[cur][j], acc[i][j]);   // rows = m, lane = n
        else acc[i][j] = mma16<T>(fb[cur][j], fa[cur][i], acc[i][j]);                   // rows = n, lane = m
        __builtin_amdgcn_sched_barrier(0);
        static_for<n_side>([&](auto O) {
          constexpr int o = decltype(O)::value;
          if constexpr ((o * NMMA) / n_side == q) {
            if constexpr (o < WTM) fa[nxt][o] = lds_read16((st + fa0[o]) ^ (((kk + 1) % KSTEPS) << 5));
            else fb[nxt][o - WTM] = lds_read16((st + fb0[o - WTM]) ^ (((kk + 1) % KSTEPS) << 5));
          }
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    if (more) advance_loader();
  }

  const int64_t wm0 = bm + wvm * 32 * WTM;
  const int wn0 = bn + wvn * 32 * WTN;

  const bool ln_on = LN;
  bool lds_epilogue = false;
  // merged q | k | v projection (emo_gemm_params.vt): the waves whose columns are V columns store transposed, the others as usual
  bool vt_wave = false;
  if constexpr (!TRANS && !CONV && LN) vt_wave = p.vt != nullptr && wn0 >= p.vt_col0;
  if constexpr (!TRANS && sizeof(T) == 2 && Tile::STAGE_BYTES >= 80 * 32 * NW) {   // (a ring slot must hold a wave's 32 x 32 staging tile)
    lds_epilogue = use_lds_epi;
    if (lds_epilogue) {
      // the slot the last stage was read from is free once every wave has finished that stage (the other slot holds the
      // next tile's prefetched first stage); it is rewritten by the loader only behind the next stage's barrier
      __builtin_amdgcn_s_barrier();
      const unsigned xbase = lds_base + ((gs + NS - 1) % NS) * Tile::STAGE_BYTES;
      const bool res = R != nullptr || p.out_scale != 1.0f, rowb = pe.rowbias != nullptr;
      auto run = [&](auto GG, auto RS, auto RB) {
        epilogue_lds<T, WTM, WTN, NW, Tile::STAGE_BYTES, decltype(GG)::value, LN, decltype(RS)::value, decltype(RB)::value>(
            acc, pe, wm0, wn0, wave, lane, xbase, C, R, ln_rstd);
      };
      using Tr = std::true_type; using Fa = std::false_type;
      if (vt_wave) {
        if constexpr (!TRANS && !CONV && LN) epilogue_vt<T, WTM, WTN, LN>(acc, pe, wm0, wn0, lane, ln_rstd);
      } else
      if (p.geglu) {   // (GEGLU with a row bias is not staged: use_lds_epi)
        if constexpr (WTN % 2 == 0) { if (res) run(Tr{}, Tr{}, Fa{}); else run(Tr{}, Fa{}, Fa{}); }
      } else if (rowb) {
        if (res) run(Fa{}, Tr{}, Tr{}); else run(Fa{}, Fa{}, Tr{});
      } else {
        if (res) run(Fa{}, Tr{}, Fa{}); else run(Fa{}, Fa{}, Fa{});
      }
    }
  }
  if (lds_epilogue) {
  } else if (vt_wave) {
    if constexpr (!TRANS && !CONV && LN) epilogue_vt<T, WTM, WTN, LN>(acc, pe, wm0, wn0, lane, ln_rstd);
  } else if constexpr (!TRANS) {
    // lane <-> output row m; register quad g of tile j <-> columns j*32 + 8*g + 4*half + {0..3}
#pragma unroll
    for (int i = 0; i < WTM; i++) {
      const int64_t m = wm0 + i * 32 + l31;
      const bool m_ok = m < p.M;
      if (nsplit > 1) {
        float* __restrict__ ws = (float*)p.workspace + ((int64_t)blockIdx.y * p.M + (m_ok ? m : 0)) * p.N;
#pragma unroll
        for (int j = 0; j < WTN; j++)
#pragma unroll
          for (int g = 0; g < 4; g++) {
            const int n0 = wn0 + j * 32 + 8 * g + 4 * half;
            if (m_ok && n0 < p.N)   // N % 4 == 0 is enforced for split-K
              *(float4*)(ws + n0) = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
          }
        continue;
      }
      epilogue_row<T, WTN>(acc[i], pe, m, m_ok, wn0, half, C, R, ln_on, ln_rstd[i]);
    }
  } else {
    // TRANS: lane <-> column n; register quad g <-> rows 8*g + 4*half + {0..3} (4 CONSECUTIVE rows), which are 4
    // contiguous elements of V^T: Ct[m / t_rows][n][m % t_rows] -> one 8-byte (bf16) / 16-byte (f32) store per quad
    const bool quad_ok = (p.t_rows & 3) == 0 && (p.t_ld & 3) == 0 && (p.t_batch_stride & 3) == 0;
#pragma unroll
    for (int i = 0; i < WTM; i++) {
      // LayerNorm fold: the statistics of row 8g + 4half + e live in lane (row) of this wave - fetched with EVERY lane active,
      // before the column mask below (a lane masked off for n >= N is the SOURCE of other lanes' shuffles: in a ragged last N
      // tile the valid columns read rstd from inactive lanes - wrong V^T columns whenever N is not a multiple of the tile)
      float rr_all[16];
#pragma unroll
      for (int r_ = 0; r_ < 16; r_++) {
        if constexpr (LN) rr_all[r_] = __shfl(ln_rstd[i], 8 * (r_ >> 2) + 4 * half + (r_ & 3), 64);
        else rr_all[r_] = 1.f;
      }
#pragma unroll
      for (int j = 0; j < WTN; j++) {
        const int n = wn0 + j * 32 + l31;
        if (n >= p.N) continue;
        const float bias_v = (p.bias && !LN) ? p.bias[n] : 0.f;   // (LN: folded into the accumulator init)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int64_t m0 = wm0 + i * 32 + 8 * g + 4 * half;
          float rrstd[4];
#pragma unroll
          for (int e = 0; e < 4; e++) rrstd[e] = rr_all[4 * g + e];
          if (m0 >= p.M) continue;
          if (nsplit > 1) {
#pragma unroll
            for (int e = 0; e < 4; e++)
              if (m0 + e < p.M) ((float*)p.workspace)[((int64_t)blockIdx.y * p.M + m0 + e) * p.N + n] = acc[i][j][4 * g + e];
            continue;
          }
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; e++) {
            float v = acc[i][j][4 * g + e];
            if constexpr (LN) v *= rrstd[e];
            v += bias_v;
            if (p.rowbias && m0 + e < p.M) v += p.rowbias[((m0 + e) / p.rows_per_batch) * p.ld_rowbias + n];
            o[e] = v * p.out_scale;
          }
          const int64_t b = m0 / p.t_rows, ml = m0 % p.t_rows;
          T* dst = C + b * p.t_batch_stride + (int64_t)n * p.t_ld + ml;
          if (quad_ok && m0 + 3 < p.M) {   // t_rows % 4 == 0 => the quad never straddles a batch
            if constexpr (sizeof(T) == 2) *(uint2*)dst = make_uint2(pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3]));
            else *(float4*)dst = make_float4(o[0], o[1], o[2], o[3]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; e++) {
              const int64_t m = m0 + e;
              if (m < p.M) TT<T>::st(C + (m / p.t_rows) * p.t_batch_stride + (int64_t)n * p.t_ld + (m % p.t_rows), o[e]);
            }
          }
        }
      }
    }
  }
  }   // tiles of this block
}

// split-K second pass: fixed-order reduction of the f32 partials + the same fused epilogue; a thread owns 4 consecutive
// output columns (N % 4 == 0 is enforced for split-K): 16-byte partial loads, 8/16-byte stores
template <typename T>
__global__ __launch_bounds__(256) void gemm_splitk_epilogue_kernel(const emo_gemm_params p) {
  const int n_out = p.geglu ? p.N / 2 : p.N;
  const int nq = n_out >> 2;
  const int64_t total = p.M * nq;
  const float* __restrict__ ws = (const float*)p.workspace;
  const int64_t slab = p.M * (int64_t)p.N;
  T* __restrict__ C = (T*)p.C;
  const T* __restrict__ R = (const T*)p.residual;
  const bool vec_io = ((p.ldc | (R ? p.ldr : 0)) & 3) == 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / nq;
    const int no = (int)(i % nq) * 4;
    const int nw = p.geglu ? (no / 32) * 64 + (no % 32) : no;   // column in W-row space
    float v[4] = {0.f, 0.f, 0.f, 0.f}, g[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.split_k; s++) {
      const float4 a = *(const float4*)(ws + s * slab + m * p.N + nw);
      v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
      if (p.geglu) {
        const float4 b = *(const float4*)(ws + s * slab + m * p.N + nw + 32);
        g[0] += b.x; g[1] += b.y; g[2] += b.z; g[3] += b.w;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      if (p.bias) { v[e] += p.bias[nw + e]; if (p.geglu) g[e] += p.bias[nw + 32 + e]; }
      if (p.rowbias) v[e] += p.rowbias[(m / p.rows_per_batch) * p.ld_rowbias + nw + e];
      if (p.geglu) v[e] = v[e] * gelu_erf_f(g[e]);
    }
    if (!p.transpose_out) {
      if (vec_io) {
        if (R) {
          if constexpr (sizeof(T) == 2) {
            float r4[4];
            unpack4<T>(*(const uint2*)(R + m * p.ldr + no), r4);
            v[0] += r4[0]; v[1] += r4[1]; v[2] += r4[2]; v[3] += r4[3];
          } else {
            const float4 rv = *(const float4*)(R + m * p.ldr + no);
            v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
          }
        }
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] *= p.out_scale;
        if constexpr (sizeof(T) == 2) *(uint2*)(C + m * p.ldc + no) = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
        else *(float4*)(C + m * p.ldc + no) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          float o = v[e];
          if (R) o += TT<T>::ld(R + m * p.ldr + no + e);
          TT<T>::st(C + m * p.ldc + no + e, o * p.out_scale);
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; e++)
        TT<T>::st(C + (m / p.t_rows) * p.t_batch_stride + (int64_t)(no + e) * p.t_ld + (m % p.t_rows), v[e] * p.out_scale);
    }
  }
}

// ------------------------------------------------------------------------------------------ host side
template <typename T, bool CONV, bool TRANS, int WTM, int WTN, int WVM, int WVN, int NS, bool LN, bool PP = false>
static int launch_gemm(const emo_gemm_params& p, int S, hipStream_t st) {
  using Tile = GemmTile<WTM, WTN, WVM, WVN, NS>;
  auto kern = gemm_kernel<T, CONV, TRANS, WTM, WTN, WVM, WVN, NS, LN, PP>;
  if (Tile::LDS_BYTES > 64 * 1024) {
    static bool once = false;
    if (!once) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Tile::LDS_BYTES);
      if (e != hipSuccess) return emo_fail(EMO_ERR_HIP, "emo_gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
      once = true;
    }
  }
  const int64_t tiles = ((p.M + Tile::BM - 1) / Tile::BM) * ((p.N + Tile::BN - 1) / Tile::BN);
  if (tiles >= (1ll << 31)) return emo_fail(EMO_ERR_BAD_SHAPE, "emo_gemm: too many tiles");
  // the split store of emo_gemm_params.vt switches per WAVE (epilogue_vt): whatever tile the dispatch (or one of its fallbacks) ended
  // up with, a wave's 32 * WTN columns must lie on one side of vt_col0 and its 32 * WTM rows inside one batch of t_rows rows -
  // checked against the geometry of THIS instantiation, not against gemm.hip's table of planned tiles
  if (p.vt && (p.vt_col0 % (32 * WTN) != 0 || p.t_rows % (32 * WTM) != 0))
    return emo_fail(EMO_ERR_UNSUPPORTED, "emo_gemm: vt_col0=%d / t_rows=%d do not fall on the wave boundaries (%d columns, %d rows) of the launched tile",
                    p.vt_col0, p.t_rows, 32 * WTN, 32 * WTM);
  // persistent launch: as many blocks as the chip holds at once (by LDS, <= 4 per CU), each walking its tiles; a
  // multiple of 8 so that a block's tiles all map to its own XCD
  int64_t slots = (256 * Tile::BPC / S) & ~7;
  if (slots < 8) slots = 8;
  const int64_t gx = tiles > slots ? slots : tiles;
  dim3 grid((unsigned)gx, (unsigned)S);
  kern<<<grid, Tile::THREADS, Tile::LDS_BYTES, st>>>(p);
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

// tile shapes (GemmPlan.tile / emo_gemm_params.tile):
//   EMO_TILE_64x64    2x2 waves of 32x32   few-block short-K shapes, V^T outputs (4 co-resident blocks per CU)
//   EMO_TILE_128x128  2x2 waves of 64x64
//   EMO_TILE_128x160  4x1 waves of 32x160  (every SD-1.5 width is a multiple of 160)
//   EMO_TILE_256x256  2x4 waves of 128x64  the big compute-bound shapes: 7.8 KB of LDS-DMA traffic per MFLOP instead of 15.6
//   EMO_TILE_256x160  8x1 waves of 32x160  10.2 KB / MFLOP for N = 320 / 640 / 960 / 1920 ... with M large
//   EMO_TILE_256x320  4x2 waves of 64x160   7.0 KB / MFLOP, A panel read once for N = 320
// (4-wave 128x256 / 256x128, 192x128, 128x192, 128x320 tiles were measured 15-35 % slower than these at equal LDS traffic per MFMA)
#ifndef EMO_GEMM_NS
#define EMO_GEMM_NS 2   // LDS ring depth (tools/bench/build_variant.sh builds deeper-ring / shorter-stage variants for A/B runs)
#endif
template <typename T, bool CONV, bool TRANS, bool LN>
static int dispatch_tile(const emo_gemm_params& p, const GemmPlan& pl, int S, hipStream_t st) {
  constexpr int NS = EMO_GEMM_NS;
  constexpr int NS_BIG = (NS * 512 * KBYTES <= 160 * 1024) ? NS : 2;   // a 256x256 stage holds 512 rows of KBYTES
  switch (pl.tile) {
    case EMO_TILE_256x256_PP:
      if constexpr (!CONV && !TRANS && sizeof(T) == 2) {
        // (the phase loader: >= 2 K-tiles, and a tile's 256 rows of A / W inside the 31-bit range of its buffer resources)
        const bool ph_ok = p.K >= 2 * (KBYTES / (int)sizeof(T)) && 256 * p.lda * (int64_t)sizeof(T) < (1ll << 31) && 256 * (int64_t)p.K * (int64_t)sizeof(T) < (1ll << 31);
        if (S == 1 && p.K % (KBYTES / (int)sizeof(T)) == 0 && ph_ok && p.split_k <= 1) return launch_gemm<T, CONV, TRANS, 4, 2, 2, 4, 2, LN, true>(p, S, st);
      }
      return launch_gemm<T, CONV, TRANS, 4, 2, 2, 4, NS_BIG, LN>(p, S, st);
    case EMO_TILE_256x256: return launch_gemm<T, CONV, TRANS, 4, 2, 2, 4, NS_BIG, LN>(p, S, st);
    case EMO_TILE_64x64: return launch_gemm<T, CONV, TRANS, 1, 1, 2, 2, NS, LN>(p, S, st);
    case EMO_TILE_128x160: return launch_gemm<T, CONV, TRANS, 1, 5, 4, 1, NS, LN>(p, S, st);
    case EMO_TILE_256x160:
      if constexpr (!CONV && !TRANS) return launch_gemm<T, CONV, TRANS, 1, 5, 8, 1, 2, LN>(p, S, st);
      break;
    case EMO_TILE_256x320:
      if constexpr (!CONV && !TRANS && sizeof(T) == 2) return launch_gemm<T, CONV, TRANS, 2, 5, 4, 2, 2, LN>(p, S, st);
      break;
    default: break;
  }
  return launch_gemm<T, CONV, TRANS, 2, 2, 2, 2, NS, LN>(p, S, st);
}

template <typename T>
static int dispatch_gemm(const emo_gemm_params& p, const GemmPlan& pl, int S, hipStream_t st) {
  const bool conv = p.conv_taps != 0;
  if (p.ln_colsum) return p.transpose_out ? dispatch_tile<T, false, true, true>(p, pl, S, st) : dispatch_tile<T, false, false, true>(p, pl, S, st);
  if (p.transpose_out) return conv ? dispatch_tile<T, true, true, false>(p, pl, S, st) : dispatch_tile<T, false, true, false>(p, pl, S, st);
  return conv ? dispatch_tile<T, true, false, false>(p, pl, S, st) : dispatch_tile<T, false, false, false>(p, pl, S, st);
}


template <typename T> int gemm_run(const emo_gemm_params& p, const GemmPlan& pl, int S, hipStream_t st) {
  int rc = dispatch_gemm<T>(p, pl, S, st);
  if (rc) return rc;
  if (S > 1) {
    const int64_t total = p.M * (p.geglu ? p.N / 2 : p.N) / 4;
    int64_t g = (total + 255) / 256; if (g > 4096) g = 4096;
    gemm_splitk_epilogue_kernel<T><<<(int)g, 256, 0, st>>>(p);
    EMO_LAUNCH_CHECK();
  }
  return EMO_OK;
}
