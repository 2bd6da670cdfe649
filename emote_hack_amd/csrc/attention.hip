// attention.hip - flash-style spatial / cross attention for gfx950: softmax(q k^T * scale) v with the
// score matrix kept on chip (8.6 G score elements per forward if materialised).
// Replaces CrossAttention._attention (orig_attention.py:655-684: baddbmm -> softmax -> bmm) and
// xformers.ops.memory_efficient_attention (models/motionmodule.py:300, models/videonet.py:62,117).
//
// Design (wave64, 32x32 MFMA), per workgroup = 4 waves x 32 query rows, KV tiles of 64 keys:
//   S^T = K . Q^T    A = K tile rows from LDS, B = Q fragments held in registers for the whole kernel.
//                    Output lane <-> query column, so the online-softmax state (m, l) and the rescale
//                    are lane-local.
//   O^T = V^T . P^T  A = V^T tile from LDS (keys contiguous: V arrives pre-transposed from the V-projection
//                    GEMM's epilogue), B = P straight from the S^T accumulator registers - no LDS round
//                    trip, no cross-lane traffic: the K tile is stored with row bits 2<->3 swapped so that
//                    a lane's 8 consecutive P registers are 8 consecutive keys.
//   KV tiles stream through an LDS RING filled by asynchronous LDS-DMA (buffer_load_dwordx4 ... lds, no VGPR staging):
//   tiles t+1 (and t+2) are in flight while tile t is consumed; counted `s_waitcnt vmcnt`, one raw s_barrier
//   per tile.  The LDS image is lane-linear, so K rows are padded to an ODD number of 16-byte chunks (the pad chunk
//   reads out of range = 0), V^T rows are XOR-swizzled (conflict-free ds_read_b128 either way); fragment reads
//   are inline-asm ds_read_b128 with hand-counted lgkmcnt (hipcc would drain vmcnt(0) before every C++ LDS
//   read while a glds is in flight).
// Two KV segments: [self / context keys of the batch row] ++ [a bank shared by seg1_div consecutive
// batch rows] = the ReferenceNet read path (mutual_self_attention.py:238-241) without materialising the
// F-times-repeated bank or the concatenated K/V.  Head dims 40/80/160 are zero-padded to 48/80/160 (bf16).
#include <stdlib.h>
#include "common.h"

static constexpr int ATT_THREADS = 256, BQ = 128, TK = 64;

__device__ __attribute__((aligned(16))) unsigned int g_att_ones_bf16[4] = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
__device__ __attribute__((aligned(16))) unsigned int g_att_ones_f32[4] = {0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u};
__device__ __attribute__((aligned(16))) unsigned int g_att_ones_f16[4] = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
template <typename T> struct OnesBits;   // 1.0 in T, replicated to 32 bits
template <> struct OnesBits<float> { static constexpr unsigned value = 0x3f800000u; };
template <> struct OnesBits<bf16_t> { static constexpr unsigned value = 0x3f803f80u; };
template <> struct OnesBits<f16_t> { static constexpr unsigned value = 0x3c003c00u; };

template <typename T, int DCH>
struct AttCfg {
  static constexpr int V = TT<T>::VEC;
  static constexpr int DPAD = DCH * V;
  static constexpr int NT = (DPAD + 31) / 32;
  static constexpr int DCHP = DCH + 1;                         // odd chunk count per K row
  static constexpr int VCH = TK * (int)sizeof(T) / 16;         // real 16-B chunks per V^T row
  static constexpr int VCHP = VCH;                             // V^T rows are not padded: XOR chunk swizzle instead (vkey)
  static constexpr int VRPB = 256 / (VCH * 16) > 0 ? 256 / (VCH * 16) : 1;   // V^T rows per 256-byte LDS bank row
  // 16-byte chunk c of V^T row n lives at chunk position c ^ vkey(n): the 16 rows of a ds_read_b128 lane group hit 16
  // distinct 16-byte slots (same scheme as the GEMM operand swizzle)
  __device__ static __forceinline__ int vkey(int n) { return (n / VRPB) & (VCH - 1); }
  static constexpr int KROW = DCHP * 16, VROW = VCHP * 16;     // bytes
  static constexpr int K_CHUNKS = TK * DCHP;
  static constexpr int STEPS = 32 / (2 * V);                   // mma16 steps per 32-key sub-tile
};

// (a __device__ function, not a call inside the kernel's lambdas: with the builtin in the kernel template's own body the HOST pass
// silently drops the kernel's stub - every launch then fails to link)
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, __attribute__((address_space(3))) unsigned char* lds_dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, lds_dst, 16, voff, 0, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ uint4 lds_read16(unsigned addr) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
template <int OFF> __device__ __forceinline__ uint4 lds_read16_at(unsigned addr) {   // addr + OFF with OFF in the instruction (16-bit field)
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
  uint4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ float max3f(float a, float b, float c) {   // no NaNs in play: skip fmaxf's canonicalisation
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float max2f(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int swap23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

// G = glds per wave per tile, NSR = ring depth, QT = 32-query tiles per wave (2 halves the LDS fragment traffic per MFMA)
// RES = the resident-context variant (below): its own instantiation - the walk over q tiles costs ~20 registers, which the
// self-attention launches (3 waves per SIMD at d = 40) do not have
// (four blocks per CU at d <= 48 - 128 registers, 2-deep ring - spills 32-130 bytes per lane: not offered)
template <typename T, int DCH, int G, int NSR, int QT, bool RES>
__global__ __launch_bounds__(ATT_THREADS, (QT == 2 ? 2 : (DCH <= 6 ? 3 : (DCH <= 10 ? 2 : 1)))) void attention_kernel(const emo_attention_params p, int stage_bytes, int order_mode, int q_rep_arg) {
  const int q_rep = RES ? q_rep_arg : 1;
  using Cfg = AttCfg<T, DCH>;
  constexpr int V = Cfg::V, NT = Cfg::NT, KROW = Cfg::KROW, VROW = Cfg::VROW, STEPS = Cfg::STEPS;
  constexpr int DCHP = Cfg::DCHP, VCH = Cfg::VCH, VCHP = Cfg::VCHP, K_CHUNKS = Cfg::K_CHUNKS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  // XCD-aware work order: block L of the flat launch runs on XCD L % 8.  A "chunk" = the q-tiles of one (b, head): they share
  // that (b, head)'s K / V, so a chunk stays on ONE XCD (one L2 fill instead of eight).  Chunks are dealt round-robin over
  // the XCDs (chunk c -> XCD c % 8) - NOT as contiguous runs: under CFG the uncond batch rows have half the keys of the
  // cond rows (no bank segment), and contiguous runs would hand whole XCDs the short rows (measured +28 % on the
  // two-segment launch).  Needs (B * heads) % 8 == 0, else the plain order.
  int qt, head, b;
  {
    const int nqt = ((p.Lq + BQ * QT - 1) / (BQ * QT) + q_rep - 1) / q_rep;   // q-tile GROUPS per (b, head): a block walks q_rep tiles
    const int chunks = p.B * p.heads, L = blockIdx.x;
    int hb;
    if ((chunks & 7) == 0 && order_mode >= 2) {
      const int x = L & 7, idx = L >> 3;
      qt = idx % nqt;
      hb = (idx / nqt) * 8 + x;
      if (order_mode == 3) hb = chunks - 1 - hb;   // longest rows (cond: two KV segments) first, the short uncond rows fill the tail
    } else if (order_mode == 1) {
      const int total = gridDim.x;
      const int qn = total >> 3, rn = total & 7, x = L & 7, idx = L >> 3;
      const int wi = (x < rn ? x * (qn + 1) : rn * (qn + 1) + (x - rn) * qn) + idx;
      qt = wi % nqt;
      hb = wi / nqt;
    } else {
      qt = L % nqt;
      hb = L / nqt;
    }
    head = hb % p.heads;
    b = hb / p.heads;
  }
  const int d = p.d;
  const int dch_real = d / V;  // d*sizeof(T) % 16 == 0 checked on the host

  // ---- zero the whole ring once: V^T pad rows (n >= d) are never written by the loader and must be 0
  // ... except V^T row d, which is all ONES when the head dim leaves a pad row (d < NT*32: 40, 80): row d of O^T = V^T P^T
  // is then the softmax denominator sum_k P[q][k], accumulated by the matrix pipe instead of 32 VALU adds per tile
  // Only the V^T PAD rows need it: every K chunk and every V^T row < d is rewritten by the loader for each tile (pad chunks from
  // the zero page), so the prologue touches (NT*32 - d) rows per slot instead of the whole ring - it was 48 KB of LDS stores per
  // block, a visible share of the two-tile launches against the 77-key text context.
  const bool ones_row = d < NT * 32;
  {
    const int pad_begin = K_CHUNKS * 16 + d * VROW, pad_bytes = (NT * 32 - d) * VROW;    // rows d .. NT*32-1 of the V^T region
    const int ones_end = pad_begin + VCH * 16;
    const unsigned one_bits = OnesBits<T>::value;
    for (int i = tid * 16; i < NSR * pad_bytes; i += ATT_THREADS * 16) {
      const int slot = i / pad_bytes, off = pad_begin + (i - slot * pad_bytes);
      const unsigned v = (ones_row && off < ones_end) ? one_bits : 0u;
      *(uint4*)(smem + slot * stage_bytes + off) = make_uint4(v, v, v, v);
    }
  }

  // ---- loader: LDS-DMA through BUFFER descriptors (buffer_load_dwordx4 ... offen lds).  Request #g of this wave writes the LDS
  // chunk positions (g*4 + wave)*64 + lane of the stage: [0, K_CHUNKS) = K rows (row = pos / DCHP holds key (row&32)|swap23(row&31)),
  // then d V^T rows of VCHP chunks, then pad rows.  K_CHUNKS and d * VCHP are multiples of 64, so one wave-level request lies
  // entirely in K rows, in V^T rows, or behind them: its descriptor (K slab | V^T slab of this (batch row, head) | the 16-byte
  // ones constant) and its per-tile advance are WAVE-UNIFORM (SGPRs); a lane carries one 32-bit byte offset per request, bumped by
  // one v_add per tile.  What the 64-bit-pointer loader did with per-lane pointer selects now falls out of the hardware range
  // check (it applies to the vector offset): pad chunks and pad rows carry an offset beyond every slab and read 0; K rows
  // >= Lk of a ragged last tile lie behind the descriptor's extent ((Lk-1) * ldk + d elements) and read 0 without a special path.
  constexpr unsigned OOB = 0xC0000000u;       // stays out of range under < 1 GB of per-tile advances
  constexpr int SZ = (int)sizeof(T);
  int ld_kind[G];                             // wave-uniform: 0 = K rows, 1 = V^T rows, 2 = behind them (ones row / zero pad)
  unsigned ld_off[G];                         // this lane's running byte offset into the request's slab
#pragma unroll
  for (int g = 0; g < G; g++) {
    const int p0 = (g * 4 + wave) * 64;
    ld_kind[g] = p0 < K_CHUNKS ? 0 : (p0 - K_CHUNKS < d * VCHP ? 1 : 2);
  }
  const unsigned* ones_u = std::is_same<T, float>::value ? g_att_ones_f32 : (std::is_same<T, bf16_t>::value ? g_att_ones_bf16 : g_att_ones_f16);
  // ---- tile list over the (up to two) KV segments
  const int nseg = (p.k1 != nullptr && b >= p.seg1_first_batch) ? 2 : 1;
  const int tiles0 = (p.Lk0 + TK - 1) / TK;
  const int tiles1 = nseg == 2 ? (p.Lk1 + TK - 1) / TK : 0;
  const int ntiles = tiles0 + tiles1;
  const int kb0 = b / p.seg0_div, kb1 = nseg == 2 ? (p.seg1_row ? p.seg1_row[0] : b / p.seg1_div - p.seg1_skip) : 0;
  const T* kbase0 = (const T*)p.k0 + (int64_t)kb0 * p.Lk0 * p.ldk0 + head * d;
  const T* vbase0 = (const T*)p.v0t + ((int64_t)kb0 * p.heads * d + (int64_t)head * d) * p.ldv0t;
  const T* kbase1 = nseg == 2 ? (const T*)p.k1 + (int64_t)kb1 * p.Lk1 * p.ldk1 + head * d : nullptr;
  const T* vbase1 = nseg == 2 ? (const T*)p.v1t + ((int64_t)kb1 * p.heads * d + (int64_t)head * d) * p.ldv1t : nullptr;

  __amdgpu_buffer_rsrc_t ld_rsrc[G];
  int ld_inc[G];
  auto setup_segment = [&](bool s1) {
    const int64_t ldk = s1 ? p.ldk1 : p.ldk0, ldvt = s1 ? p.ldv1t : p.ldv0t;
    const int Lk = s1 ? p.Lk1 : p.Lk0;
    const T* kbase = s1 ? kbase1 : kbase0;
    const T* vbase = s1 ? vbase1 : vbase0;
    // (extents < 2^31 bytes: checked on the host)
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, (int)(((int64_t)(Lk - 1) * ldk + d) * SZ), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, (int)((int64_t)d * ldvt * SZ), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)ones_u, 0, 16, 0x00020000);
#pragma unroll
    for (int g = 0; g < G; g++) {
      const int pos = (g * 4 + wave) * 64 + lane;
      if (ld_kind[g] == 0) {
        const int row = pos / DCHP, c = pos % DCHP;
        ld_rsrc[g] = rk; ld_inc[g] = (int)(TK * ldk * SZ);
        ld_off[g] = c < dch_real ? (unsigned)((((row & 32) | swap23(row & 31)) * ldk + c * V) * SZ) : OOB;
      } else if (ld_kind[g] == 1) {
        const int qv = pos - K_CHUNKS, n = qv / VCHP, c = (qv % VCHP) ^ Cfg::vkey(n);
        ld_rsrc[g] = rv; ld_inc[g] = TK * SZ;
        ld_off[g] = (unsigned)((n * ldvt + c * V) * SZ);
      } else {
        const int n = (pos - K_CHUNKS) / VCHP;
        ld_rsrc[g] = ro; ld_inc[g] = 0;
        ld_off[g] = (ones_row && n == d) ? 0u : OOB;     // the all-ones row (rewritten where the rounds cover it), zeros behind it
      }
    }
  };
  setup_segment(false);
  auto issue = [&](int t, int slot) {
    const bool s1 = t >= tiles0;
    if (s1 && t == tiles0) setup_segment(true);
    const int k0 = (s1 ? t - tiles0 : t) * TK;
    const int Lk = s1 ? p.Lk1 : p.Lk0;
    __attribute__((address_space(3))) unsigned char* st = (__attribute__((address_space(3))) unsigned char*)smem + slot * stage_bytes + wave * 1024;
    if (k0 + TK <= Lk) {
#pragma unroll
      for (int g = 0; g < G; g++) {
        lds_dma16(ld_rsrc[g], st + g * 4096, ld_off[g]);
        ld_off[g] += ld_inc[g];
      }
    } else {   // last tile of the segment: V^T chunks wholly past Lk read 0 (the chunk straddling Lk is cleaned in LDS below)
#pragma unroll
      for (int g = 0; g < G; g++) {
        unsigned off = ld_off[g];
        if (ld_kind[g] == 1) {
          const int qv = (g * 4 + wave) * 64 + lane - K_CHUNKS, n = qv / VCHP, c = (qv % VCHP) ^ Cfg::vkey(n);
          if (k0 + c * V >= Lk) off = OOB;
        }
        lds_dma16(ld_rsrc[g], st + g * 4096, off);
      }
    }
  };
  const float c_exp = p.scale * 1.4426950408889634f;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

  // ---- fragment reads: one address per lane and tile, everything else rides in the instructions' offset fields
  //   K  : row (st*32 + l31), chunks 2*kk + half            -> base + st*32*KROW + kk*32
  //   V^T: row (nt*32 + l31), chunk ((st*4 + 2*sp [+ half]) scaled to the element size) ^ vkey(row); vkey repeats every 32 rows and
  //        the chunk field of the address is XORed in place (rows and stages are multiples of the 16 * VCH byte row)
  static_assert((K_CHUNKS * 16) % (VCH * 16) == 0 && (32 / Cfg::VRPB) % VCH == 0, "V^T read addressing");
  const unsigned k_lane = l31 * KROW + half * 16;
  const unsigned v_lane = l31 * VROW + ((((8 * half * SZ) >> 4) ^ Cfg::vkey(l31)) * 16);
  auto read_k_frags = [&](unsigned ks, uint4 (&kf)[2][DCH / 2]) {
    const unsigned a = ks + k_lane;
    static_for<2>([&](auto ST) {
      static_for<DCH / 2>([&](auto KK) {
        constexpr int st = decltype(ST)::value, kk = decltype(KK)::value;
        kf[st][kk] = lds_read16_at<st * 32 * KROW + kk * 32>(a);
      });
    });
  };
  auto read_v_frags = [&](unsigned vs, auto ST, uint4 (&vf)[STEPS][NT]) {
    constexpr int st = decltype(ST)::value;
    static_for<STEPS>([&](auto SP) {
      constexpr int sp = decltype(SP)::value;
      constexpr int key0 = st * 32 + 16 * ((sp * V) >> 3) + ((sp * V) & 7);     // + 8 * half: in v_lane
      const unsigned a = (vs + v_lane) ^ (unsigned)(((key0 * SZ) >> 4) * 16);
      static_for<NT>([&](auto N) {
        constexpr int nt = decltype(N)::value;
        vf[sp][nt] = lds_read16_at<nt * 32 * VROW>(a);
      });
      if constexpr (STEPS * NT > 15) {           // lgkmcnt is a 4-bit counter: drain per step for the widest heads
        wait_lgkmcnt<0>();
        __builtin_amdgcn_sched_barrier(0);
      }
    });
  };
  if ((lds_base & (VROW - 1)) != 0) __builtin_trap();   // the XOR above needs the dynamic LDS base on a V^T row boundary (no static LDS in front)

  __syncthreads();   // ring zeroed before the first async write lands (plain LDS stores: lgkmcnt drained by the barrier)
#pragma unroll
  for (int s = 0; s < NSR - 1; s++)
    if (s < ntiles) issue(s, s);

  // RESIDENT mode (q_rep > 1; the host picks it only when every KV tile of the launch fits the ring, i.e. ntiles <= NSR - the
  // 77-key text / audio context): the K / V^T tiles are requested ONCE and stay in their slots while the block walks q_rep
  // consecutive 128-query tiles of the same (b, head) - the ring fill, the pad clean-up and the barriers are paid by the first
  // of them only (a q tile against 77 keys is 2 short MFMA rounds: the launch was all prologue).
  const int qt_grp = qt;
  // ---- Q fragments: QT tiles of 32 query rows per wave; row q, chunks (2*kk + half).  The NEXT q tile's fragments of a
  // resident walk are requested while the current tile is multiplied (a tile against 77 keys is shorter than the load latency).
  constexpr int BQW = 32 * QT;              // query rows per wave
  uint4 qf_next[QT][DCH / 2];
  auto load_q = [&](int qtile) {
#pragma unroll
    for (int t = 0; t < QT; t++) {
      const int qr = qtile * (4 * BQW) + wave * BQW + t * 32 + l31;
      const bool ok = qr < p.Lq;
      const T* qrow = (const T*)p.q + ((int64_t)b * p.Lq + (ok ? qr : 0)) * p.ldq + head * d;
#pragma unroll
      for (int kk = 0; kk < DCH / 2; kk++) {
        const int c = 2 * kk + half;
        qf_next[t][kk] = (ok && c < dch_real) ? *(const uint4*)(qrow + c * V) : make_uint4(0, 0, 0, 0);
      }
    }
  };
  load_q(qt_grp * q_rep);
  // The Q fragments are ordinary global loads.  hipcc's wait-count pass cannot count past an LDS-DMA: with Q still "pending"
  // at the head of the tile loop it put an `s_waitcnt vmcnt(0)` in front of the first Q.K^T MFMA of EVERY tile, i.e. every
  // wave waited for the ring slot it had just requested (tile t+NSR-1) before touching tile t - the ring never ran ahead.
  // Consuming the fragments here retires them once, next to the prologue requests the first tile needs anyway.
  // (not in the resident variant: all of its tiles are requested by the prologue and waited for once)
  if constexpr (!RES) {
#pragma unroll
    for (int t = 0; t < QT; t++)
#pragma unroll
      for (int kk = 0; kk < DCH / 2; kk++)
        asm volatile("" : "+v"(qf_next[t][kk].x), "+v"(qf_next[t][kk].y), "+v"(qf_next[t][kk].z), "+v"(qf_next[t][kk].w));
  }
#pragma unroll 1
  for (int qi = 0; qi < q_rep; qi++) {
  qt = qt_grp * q_rep + qi;
  if (qi > 0 && qt * (4 * BQW) >= p.Lq) break;
  const bool fill = qi == 0;               // this pass requests / waits for / cleans the tiles
  int qrow_idx[QT];
  bool q_ok[QT];
  uint4 qf[QT][DCH / 2];
#pragma unroll
  for (int t = 0; t < QT; t++) {
    qrow_idx[t] = qt * (4 * BQW) + wave * BQW + t * 32 + l31;
    q_ok[t] = qrow_idx[t] < p.Lq;
#pragma unroll
    for (int kk = 0; kk < DCH / 2; kk++) qf[t][kk] = qf_next[t][kk];
  }
  if constexpr (RES) { if (qi + 1 < q_rep) load_q(qt + 1); }    // (rows past Lq read row 0 of the batch row: in range, never stored)

  f32x16 o[QT][NT];
#pragma unroll
  for (int t = 0; t < QT; t++)
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
      for (int r = 0; r < 16; r++) o[t][nt][r] = 0.f;
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int t = 0; t < QT; t++) { m_run[t] = -1e30f; l_run[t] = 0.f; }

  // ---- one 64-key tile (ring slot at byte offset st_off) against the wave's query rows: S^T for both 32-key sub-tiles, ONE
  // online-softmax update per query row per tile, then O^T += V^T P^T.  FAST = an interior tile of its segment: no key mask,
  // no V^T clean-up, both sub-tiles; (k0, Lk) matter to the other tiles only.  The V^T fragment reads of the first sub-tile go
  // out right behind the Q.K^T MFMAs (they fill the MFMA result latency the max chain would otherwise wait out in s_nops).
  auto tile = [&](unsigned st_off, int k0, int Lk, auto FAST) {
    constexpr bool fast = decltype(FAST)::value;
    const unsigned ks_base = lds_base + st_off, vs_base = ks_base + K_CHUNKS * 16;
    bool ragged = false, second = true;
    if constexpr (!fast) {
      ragged = k0 + TK > Lk;               // wave-uniform
      second = k0 + 32 < Lk;               // ragged last tile (the 77-key context): the second sub-tile may hold no key
      if (fill && ragged && (Lk % V) != 0) {
        // ragged last tile whose final 16-byte V^T chunk straddles Lk: zero the columns >= Lk in LDS (their P is
        // exactly 0, but 0 * garbage must not become NaN).  Rare (context length 77); costs one extra barrier.
        const int cpart = (Lk - k0) / V, efirst = (Lk - k0) % V;
        unsigned char* vs = smem + st_off + K_CHUNKS * 16;
        for (int n = tid; n < d; n += ATT_THREADS) {
          uint4* ptr = (uint4*)(vs + n * VROW + (cpart ^ Cfg::vkey(n)) * 16);
          float f[V];
          unpack16<T>(*ptr, f);
#pragma unroll
          for (int e = 0; e < V; e++) if (e >= efirst) f[e] = 0.f;
          *ptr = pack16<T>(f);
        }
        __syncthreads();
      }
    }
    uint4 kf[2][DCH / 2];
    read_k_frags(ks_base, kf);
    wait_lgkmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
    uint4 pf[QT][2][STEPS];                        // P^T fragments (B operand of the PV MFMAs)
    uint4 vf0[STEPS][NT];
    constexpr bool early_v = STEPS * NT <= 8 && QT == 1;   // (the widest heads drain lgkmcnt per step and keep their reads next to the MFMAs; two query tiles per wave have no registers to spare)
#pragma unroll
    for (int t = 0; t < QT; t++) {
      f32x16 s0, s1;
#pragma unroll
      for (int r = 0; r < 16; r++) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < DCH / 2; kk++) {
        s0 = mma16<T>(kf[0][kk], qf[t][kk], s0);
        s1 = mma16<T>(kf[1][kk], qf[t][kk], s1);
      }
      if constexpr (early_v) {
        if (t == QT - 1) {
          __builtin_amdgcn_sched_barrier(0);
          read_v_frags(vs_base, std::integral_constant<int, 0>{}, vf0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (!fast) {
        if (ragged) {   // keys >= Lk
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int key = k0 + 16 * (r >> 3) + 8 * half + (r & 7);
            if (key >= Lk) s0[r] = -1e30f;
            if (key + 32 >= Lk) s1[r] = -1e30f;
          }
        }
      }
      // The max chain below is inline asm (v_max3_f32 has no builtin), and hipcc pads no hazard whose consumer sits inside an asm
      // string: an MFMA result read too early by a VALU instruction is stale, silently (with one k-step - d = 8 - the chain read
      // the accumulators 4 instructions behind their MFMAs and the output changed from run to run).  One compiler-visible VALU
      // read of the LAST accumulator written makes the hazard recogniser wait out the MFMA; everything behind it is safe.
      // (an identity v_mov_b32_dpp - quad_perm [0,1,2,3] - the optimiser cannot fold away)
      s1[0] = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(s1[0]), 0xE4, 0xF, 0xF, true));
      // row maximum over the lane's 32 scores: two chains (half the dependent latency), then the other half-wave's 32 keys of the
      // same query row through v_permlane32_swap (VALU; __shfl_xor is an LDS round trip + 6 address instructions)
      float mxa = max3f(s0[0], s1[0], s0[1]), mxb = max3f(s0[8], s1[8], s0[9]);
#pragma unroll
      for (int r = 1; r < 7; r++) { mxa = max3f(mxa, s1[r], s0[r + 1]); mxb = max3f(mxb, s1[r + 8], s0[r + 9]); }
      float mx = max3f(mxa, mxb, s1[7]);
      mx = max2f(mx, s1[15]);
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = max2f(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      }
      // THRESHOLDED online softmax: the running reference m_run of a query row moves only when the tile's maximum exceeds it by
      // more than ATT_TAU in log2 units.  Any reference common to all of a row's terms gives the same quotient O / l, so this is
      // exact; what it bounds is the size of the probabilities, P <= 2^ATT_TAU = 256 (bf16 / f16 round P relatively, and the sums
      // are f32).  With "moves whenever the maximum grows at all" one of a wave's 32 rows moved in about every second interior
      // tile - 32 multiplies of the O accumulators, an exponential and the ballot on an issue-bound kernel; now the branch is
      // taken in a row's first tile(s) and practically never again.
      constexpr float ATT_TAU = 8.0f;
      const float m_old = m_run[t];
      const bool grow = (mx - m_old) * c_exp > ATT_TAU;
      float alpha = 1.0f;
      if (__builtin_amdgcn_ballot_w64(grow) != 0) {
        const float m_new = grow ? mx : m_old;
        alpha = __builtin_amdgcn_exp2f((m_old - m_new) * c_exp);   // raw v_exp_f32: arguments are <= 0, no denormal care
        m_run[t] = m_new;
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
          for (int r = 0; r < 16; r++) o[t][nt][r] *= alpha;
      }
      const float m_sc = m_run[t] * c_exp;
      float p0[16], p1[16];
      {
        const v2f cc = {c_exp, c_exp}, mm = {m_sc, m_sc};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const v2f a = {s0[r], s0[r + 1]}, b = {s1[r], s1[r + 1]};
          const v2f ea = a * cc - mm, eb = b * cc - mm;      // v_pk_fma_f32
          p0[r] = __builtin_amdgcn_exp2f(ea.x); p0[r + 1] = __builtin_amdgcn_exp2f(ea.y);
          p1[r] = __builtin_amdgcn_exp2f(eb.x); p1[r + 1] = __builtin_amdgcn_exp2f(eb.y);
        }
      }
      if (!ones_row) {   // no pad row (d = 160): the denominator is summed on the VALU
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) psum += p0[r] + p1[r];
        l_run[t] = l_run[t] * alpha + psum;
      }
#pragma unroll
      for (int sp = 0; sp < STEPS; sp++) {
        const int r0 = sp * V;
        if constexpr (sizeof(T) == 2) {
          pf[t][0][sp] = make_uint4(pack2<T>(p0[r0], p0[r0 + 1]), pack2<T>(p0[r0 + 2], p0[r0 + 3]), pack2<T>(p0[r0 + 4], p0[r0 + 5]),
                                    pack2<T>(p0[r0 + 6], p0[r0 + 7]));
          pf[t][1][sp] = make_uint4(pack2<T>(p1[r0], p1[r0 + 1]), pack2<T>(p1[r0 + 2], p1[r0 + 3]), pack2<T>(p1[r0 + 4], p1[r0 + 5]),
                                    pack2<T>(p1[r0 + 6], p1[r0 + 7]));
        } else {
          pf[t][0][sp] = make_uint4(__float_as_uint(p0[r0]), __float_as_uint(p0[r0 + 1]), __float_as_uint(p0[r0 + 2]), __float_as_uint(p0[r0 + 3]));
          pf[t][1][sp] = make_uint4(__float_as_uint(p1[r0]), __float_as_uint(p1[r0 + 1]), __float_as_uint(p1[r0 + 2]), __float_as_uint(p1[r0 + 3]));
        }
      }
    }
    // ---- O^T += V^T . P^T : per sub-tile the V^T fragments are read once and feed the QT query tiles; the second sub-tile's reads
    // go out in front of the first one's MFMAs
    uint4 vf1[STEPS][NT];
    if constexpr (!early_v) read_v_frags(vs_base, std::integral_constant<int, 0>{}, vf0);
    wait_lgkmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (early_v) {
      if (fast || second) read_v_frags(vs_base, std::integral_constant<int, 1>{}, vf1);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int sp = 0; sp < STEPS; sp++)
#pragma unroll
      for (int t = 0; t < QT; t++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) o[t][nt] = mma16<T>(vf0[sp][nt], pf[t][0][sp], o[t][nt]);
    if (fast || second) {
      if constexpr (!early_v) read_v_frags(vs_base, std::integral_constant<int, 1>{}, vf1);
      wait_lgkmcnt<0>();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int sp = 0; sp < STEPS; sp++)
#pragma unroll
        for (int t = 0; t < QT; t++)
#pragma unroll
          for (int nt = 0; nt < NT; nt++) o[t][nt] = mma16<T>(vf1[sp][nt], pf[t][1][sp], o[t][nt]);
    }
  };

  // ---- the tile loop.  Interior tiles run the FAST body with the ring position and the loader state kept INCREMENTALLY: the
  // general body recomputed slot, segment, key range and request geometry per tile (~55 scalar instructions, among them a
  // kernel-argument load and two divisions by the ring depth), and on this kernel every instruction counts: the SIMD's issue
  // time is the sum of its waves' instructions (profiles/r04*_attention_pmc.txt: MFMA + VALU + SALU + LDS + s_nop active cycles
  // add up to ~90 % of the SIMD cycles whatever the schedule).  A tile is SPECIAL when it is the last of its segment (mask /
  // clean-up) or when the tile it requests (NSR-1 ahead) is the last of a segment, the first of the bank segment (new
  // descriptors) or does not exist; at most 2 * NSR + 1 tiles of a launch.
  constexpr int AHEAD = NSR - 1;
  const unsigned ring_bytes = (unsigned)NSR * (unsigned)stage_bytes;
  unsigned c_off = 0, i_off = (unsigned)(AHEAD % NSR) * (unsigned)stage_bytes;      // consumer / request slot (byte offsets)
  auto next_special = [&](int t) {
    int c = ntiles - 1 - AHEAD;
    const int cand[3] = {tiles0 - 1 - AHEAD, tiles0 - AHEAD, tiles0 - 1};
#pragma unroll
    for (int i = 0; i < 3; i++) if (cand[i] >= t && cand[i] < c) c = cand[i];
    return c > t ? c : t;
  };
  for (int t = 0; t < ntiles;) {
    if constexpr (NSR >= 2 && !RES) {
      const int fast_end = next_special(t);
      for (; t < fast_end; t++) {
        wait_vmcnt<(NSR - 2) * G>();
        __builtin_amdgcn_s_barrier();
        {   // request tile t + AHEAD: an interior tile of the segment being streamed
          __attribute__((address_space(3))) unsigned char* st = (__attribute__((address_space(3))) unsigned char*)smem + i_off + wave * 1024;
#pragma unroll
          for (int g = 0; g < G; g++) {
            lds_dma16(ld_rsrc[g], st + g * 4096, ld_off[g]);
            ld_off[g] += ld_inc[g];
          }
        }
        tile(c_off, 0, 0, std::true_type{});
        c_off += stage_bytes; if (c_off == ring_bytes) c_off = 0;
        i_off += stage_bytes; if (i_off == ring_bytes) i_off = 0;
      }
      if (t >= ntiles) break;
    }
    if (fill) {
      if constexpr (NSR == 1) issue(t, 0);
      const int rem = ntiles - 1 - t;
      if constexpr (NSR >= 3) {
        if (rem >= NSR - 2) wait_vmcnt<(NSR - 2) * G>();
        else wait_vmcnt<0>();
      } else {
        wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
      if constexpr (NSR > 1) {
        if (t + AHEAD < ntiles) issue(t + AHEAD, (t + AHEAD) % NSR);
      }
    }
    {
      const bool s1 = t >= tiles0;
      tile(c_off, (s1 ? t - tiles0 : t) * TK, s1 ? p.Lk1 : p.Lk0, std::false_type{});
    }
    if constexpr (NSR == 1) { if (q_rep == 1) __builtin_amdgcn_s_barrier(); }   // synchronous ring: nobody may still read slot 0 (resident: one tile, never rewritten)
    c_off += stage_bytes; if (c_off == ring_bytes) c_off = 0;
    i_off += stage_bytes; if (i_off == ring_bytes) i_off = 0;
    t++;
  }

  // ---- normalise and store: lane holds O[q][n], n = nt*32 + 8*(r>>2) + 4*half + (r&3)
#pragma unroll
  for (int t = 0; t < QT; t++) {
    float l_tot;
    if (ones_row) {
      // row d of O^T: tile d/32, local row rr = d%32 = 8*(r>>2) + 4*half + (r&3)
      const int nt_d = d >> 5, rr = d & 31, r_d = ((rr >> 3) << 2) | (rr & 3), half_d = (rr >> 2) & 1;
      float lsel = 0.f;
#pragma unroll
      for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++)
          if (nt == nt_d && r == r_d) lsel = o[t][nt][r];
      l_tot = __shfl(lsel, half_d * 32 + l31, 64);
    } else {
      l_tot = l_run[t] + __shfl_xor(l_run[t], 32, 64);
    }
    const float inv = 1.0f / l_tot;
    if (q_ok[t]) {
      T* orow = (T*)p.out + ((int64_t)b * p.Lq + qrow_idx[t]) * p.ldo + head * d;
#pragma unroll
      for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int n0 = nt * 32 + 8 * g + 4 * half;
          if (n0 < d) {
            float v0 = o[t][nt][4 * g] * inv, v1 = o[t][nt][4 * g + 1] * inv, v2 = o[t][nt][4 * g + 2] * inv, v3 = o[t][nt][4 * g + 3] * inv;
            if constexpr (sizeof(T) == 2) {
              *(uint2*)(orow + n0) = make_uint2(pack2<T>(v0, v1), pack2<T>(v2, v3));
            } else {
              *(float4*)(orow + n0) = make_float4(v0, v1, v2, v3);
            }
          }
        }
    }
  }
  }   // q tiles of this block
}

template <typename T, int DCH, int G>
struct AttLaunch {
  using Cfg = AttCfg<T, DCH>;
  static constexpr int NEED = Cfg::K_CHUNKS + Cfg::NT * 32 * Cfg::VCHP;
  static constexpr int STAGE_CHUNKS = G * 256 > NEED ? G * 256 : NEED;
  static constexpr int STAGE_BYTES = STAGE_CHUNKS * 16;
  // ring depth: as deep as LDS allows while keeping >= 2 workgroups per CU when possible
  static constexpr int NSR = 3 * STAGE_BYTES <= 80 * 1024 ? 3 : (2 * STAGE_BYTES <= 160 * 1024 ? 2 : 1);
  static_assert(NSR * STAGE_BYTES <= 160 * 1024, "attention stage does not fit LDS");
};

template <typename T, int DCH, int G, int QT>
static int launch_attention3(const emo_attention_params& p, hipStream_t st) {
  using L = AttLaunch<T, DCH, G>;
  auto kern = attention_kernel<T, DCH, G, L::NSR, QT, false>;
  auto kern_res = attention_kernel<T, DCH, G, L::NSR, QT, true>;
  constexpr int lds = L::NSR * L::STAGE_BYTES;
  if (lds > 64 * 1024) {
    static bool once = false;  // idempotent attribute; benign race
    if (!once) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)kern_res, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return emo_fail(EMO_ERR_HIP, "emo_attention: hipFuncSetAttribute: %s", hipGetErrorString(e));
      once = true;
    }
  }
  // resident mode (kernel comment): every KV tile of every batch row fits the ring - the text / audio context
  const int nqt_all = (p.Lq + BQ * QT - 1) / (BQ * QT);
  const int tiles_max = (p.Lk0 + TK - 1) / TK + (p.k1 ? (p.Lk1 + TK - 1) / TK : 0);
  int q_rep = 1;
  if (tiles_max <= L::NSR) {
#ifdef EMO_ATT_QREP_ENV   // tools/bench/xattn_bench.py sweep builds only (build_variant.sh -DEMO_ATT_QREP_ENV): never in the product
    static const int forced_env = getenv("EMO_ATT_QREP") ? atoi(getenv("EMO_ATT_QREP")) : 0;
    const int forced = forced_env > nqt_all ? nqt_all : forced_env;
#else
    constexpr int forced = 0;
#endif
    const int64_t chunks = (int64_t)p.heads * p.B;
    // up to 4 q tiles per block while that leaves >= 160 blocks (80 -> 66 us at B=24 Lq=4096 d=40, 37 -> 32 at Lq=1024 d=80,
    // 32 -> 25 at Lq=256 d=160; 8 per block: equal / slower - tools/bench/xattn_bench.py)
    for (q_rep = 4; q_rep > 1 && (q_rep > nqt_all || ((nqt_all + q_rep - 1) / q_rep) * chunks < 160); q_rep >>= 1) {}
    if (forced > 0) q_rep = forced;
  }
  const int64_t nblk = (int64_t)((nqt_all + q_rep - 1) / q_rep) * p.heads * p.B;
  if (nblk >= (1ll << 31)) return emo_fail(EMO_ERR_BAD_SHAPE, "emo_attention: too many blocks");
  dim3 grid((unsigned)nblk);
  // order_mode 2: (b, head) chunks round-robin over the XCDs; 3: the same from the LAST batch row backwards - under CFG the cond
  // rows (second half of the batch) carry the bank segment and run twice as long as the uncond rows: started first, the short rows
  // fill the tail (1443 -> 1429 us at the 64x64 level, 143 -> 138.5 at 32x32; equal without a bank segment)
  const int order_mode = (p.k1 != nullptr && p.seg1_first_batch > 0) ? 3 : 2;
  if (q_rep > 1) kern_res<<<grid, ATT_THREADS, lds, st>>>(p, L::STAGE_BYTES, order_mode, q_rep);
  else kern<<<grid, ATT_THREADS, lds, st>>>(p, L::STAGE_BYTES, order_mode, 1);
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

template <typename T, int DCH, int G>
static int launch_attention2(const emo_attention_params& p, hipStream_t st) {
  // QT = 2 (two 32-query tiles per wave: K / V^T fragments, requests and loop overhead shared by 64 query rows - ~136 instead of
  // ~170 instructions per query tile and KV tile) needs 256 registers at d = 40 (+ 20 bytes of scratch) = two waves per SIMD:
  // round 3 measured it 1.7x slower on the old instruction stream, round 4 on the lean one 1.5-3.5 % FASTER (1294 -> 1275 us,
  // 775 -> 748 us, profiles/r04m_attention_qt2.txt) - inside the run-to-run spread of a whole step, with a spill; not enabled.
  return launch_attention3<T, DCH, G, 1>(p, st);
}

// DPREV = chunk count of the next smaller head-dim class: this class serves dch in (DPREV, DCH]
template <typename T, int DCH, int DPREV>
static int launch_attention(const emo_attention_params& p, hipStream_t st) {
  using Cfg = AttCfg<T, DCH>;
  // chunks the loader must cover for THIS head dim (K rows + d V^T rows), rounded up to whole 256-lane rounds
  const int g = (Cfg::K_CHUNKS + p.d * Cfg::VCHP + 255) / 256;
  constexpr int GMIN = (Cfg::K_CHUNKS + (DPREV + 1) * Cfg::V * Cfg::VCHP + 255) / 256;
  constexpr int GMAX = (Cfg::K_CHUNKS + Cfg::DPAD * Cfg::VCHP + 255) / 256;
#define EMO_ATT_G(GV)                                                            \
  case GV:                                                                       \
    if constexpr (GV >= GMIN && GV <= GMAX) return launch_attention2<T, DCH, GV>(p, st); \
    break;
  switch (g) {
    EMO_ATT_G(1) EMO_ATT_G(2) EMO_ATT_G(3) EMO_ATT_G(4) EMO_ATT_G(5) EMO_ATT_G(6) EMO_ATT_G(7) EMO_ATT_G(8)
    EMO_ATT_G(9) EMO_ATT_G(10) EMO_ATT_G(11) EMO_ATT_G(12) EMO_ATT_G(13) EMO_ATT_G(14) EMO_ATT_G(15) EMO_ATT_G(16)
    EMO_ATT_G(17) EMO_ATT_G(18) EMO_ATT_G(19) EMO_ATT_G(20) EMO_ATT_G(21) EMO_ATT_G(22)
    default: break;
  }
#undef EMO_ATT_G
  return emo_fail(EMO_ERR_UNSUPPORTED, "emo_attention: loader rounds %d out of range for head dim %d", g, p.d);
}

template <typename T>
static int dispatch_attention(const emo_attention_params& p, hipStream_t st) {
  constexpr int V = TT<T>::VEC;
  const int dch = p.d / V;
  if (dch <= 2) return launch_attention<T, 2, 0>(p, st);
  if (dch <= 4) return launch_attention<T, 4, 2>(p, st);
  if (dch <= 6) return launch_attention<T, 6, 4>(p, st);
  if (dch <= 10) return launch_attention<T, 10, 6>(p, st);
  if (dch <= 20) return launch_attention<T, 20, 10>(p, st);
  if constexpr (sizeof(T) == 4) {
    if (dch <= 40) return launch_attention<T, 40, 20>(p, st);
  }
  return emo_fail(EMO_ERR_UNSUPPORTED, "emo_attention: head dim %d too large", p.d);
}

extern "C" int emo_attention(const emo_attention_params* pp, void* stream) {
  EMO_CHECK(pp, EMO_ERR_NULL, "emo_attention: null params");
  const emo_attention_params& p = *pp;
  EMO_CHECK(p.q && p.k0 && p.v0t && p.out, EMO_ERR_NULL, "emo_attention: null pointer");
  EMO_CHECK(emo_dtype_ok(p.dtype), EMO_ERR_BAD_DTYPE, "emo_attention: dtype %d", p.dtype);
  const int V = emo_dtype_vec(p.dtype);
  EMO_CHECK(p.B > 0 && p.Lq > 0 && p.Lk0 > 0 && p.heads > 0 && p.d > 0, EMO_ERR_BAD_SHAPE, "emo_attention: bad shape");
  EMO_CHECK(p.d % V == 0, EMO_ERR_BAD_SHAPE, "emo_attention: head dim %d must be a multiple of %d", p.d, V);
  EMO_CHECK(p.ldq % V == 0 && p.ldk0 % V == 0 && p.ldv0t % V == 0 && p.ldo % 4 == 0, EMO_ERR_BAD_SHAPE, "emo_attention: leading dims");
  EMO_CHECK(p.ldv0t >= p.Lk0, EMO_ERR_BAD_SHAPE, "emo_attention: ldv0t < Lk0");
  EMO_CHECK(p.heads <= 65535 && p.B <= 65535, EMO_ERR_BAD_SHAPE, "emo_attention: grid limits");
  EMO_CHECK(p.seg0_div >= 1, EMO_ERR_BAD_SHAPE, "emo_attention: seg0_div must be >= 1");
  if (p.k1 && !p.seg1_row) {
    EMO_CHECK(p.seg1_skip >= 0 && p.seg1_first_batch / (p.seg1_div > 0 ? p.seg1_div : 1) >= p.seg1_skip, EMO_ERR_BAD_SHAPE,
              "emo_attention: seg1_skip %d exceeds the bank rows skipped by seg1_first_batch %d", p.seg1_skip, p.seg1_first_batch);
  }
  if (p.k1) {
    EMO_CHECK(p.v1t && p.Lk1 > 0 && (p.seg1_div > 0 || p.seg1_row) && p.ldk1 % V == 0 && p.ldv1t % V == 0 && p.ldv1t >= p.Lk1, EMO_ERR_BAD_SHAPE,
              "emo_attention: segment-1 geometry");
  }
  {   // the loader addresses one (batch row, head) slab of K and of V^T through 32-bit buffer offsets
    const int64_t sz = p.dtype == EMO_F32 ? 4 : 2, lim = (int64_t)1 << 30;
    EMO_CHECK(((int64_t)p.Lk0 + 64) * p.ldk0 * sz < lim && (int64_t)p.d * p.ldv0t * sz < lim, EMO_ERR_BAD_SHAPE,
              "emo_attention: a K / V^T slab of segment 0 exceeds 1 GB (Lk0 %d, ldk0 %lld, ldv0t %lld)", p.Lk0, (long long)p.ldk0, (long long)p.ldv0t);
    if (p.k1)
      EMO_CHECK(((int64_t)p.Lk1 + 64) * p.ldk1 * sz < lim && (int64_t)p.d * p.ldv1t * sz < lim, EMO_ERR_BAD_SHAPE,
                "emo_attention: a K / V^T slab of segment 1 exceeds 1 GB (Lk1 %d, ldk1 %lld, ldv1t %lld)", p.Lk1, (long long)p.ldk1, (long long)p.ldv1t);
  }
  hipStream_t st = as_stream(stream);
  EMO_DISPATCH(p.dtype, "emo_attention", return dispatch_attention<T>(p, st));
  return EMO_OK;
}
