// attention.hip - flash-style spatial / cross attention for gfx950: softmax(q k^T * scale) v with the
// score matrix kept on chip (8.6 G score elements per forward if materialised).
// Replaces CrossAttention._attention (orig_attention.py:655-684: baddbmm -> softmax -> bmm) and
// xformers.ops.memory_efficient_attention (models/motionmodule.py:300, models/videonet.py:62,117).
//
// Design (wave64, 32x32 MFMA), per workgroup = 4 waves x 32 query rows, KV tiles of 64 keys in LDS:
//   S^T = K . Q^T    A = K tile rows from LDS (ds_read_b128, rows padded to an odd number of 16-B slots),
//                    B = Q fragments held in registers for the whole kernel.  Output lane <-> query
//                    column, so the online-softmax state (m, l) and the rescale are lane-local.
//   O^T = V^T . P^T  A = V^T tile from LDS (keys contiguous: V arrives pre-transposed from the V-projection
//                    GEMM's epilogue), B = P straight from the S^T accumulator registers - no LDS round
//                    trip, no cross-lane traffic: the K tile is stored with row bits 2<->3 swapped so that
//                    a lane's 8 consecutive P registers are 8 consecutive keys.
// Two KV segments: [self / context keys of the batch row] ++ [a bank shared by seg1_div consecutive
// batch rows] = the ReferenceNet read path (mutual_self_attention.py:238-241) without materialising the
// F-times-repeated bank or the concatenated K/V.  Head dims 40/80/160 are zero-padded to 48/80/160 (bf16).
#include "common.h"

static constexpr int ATT_THREADS = 256, BQ = 128, TK = 64;

template <typename T, int DCH>
struct AttCfg {
  static constexpr int V = TT<T>::VEC;
  static constexpr int DPAD = DCH * V;
  static constexpr int NT = (DPAD + 31) / 32;
  static constexpr int KROW = DCH * 16 + 16;                  // bytes; (DCH+1) odd -> conflict-free b128 reads
  static constexpr int VROW = TK * (int)sizeof(T) + 16;       // bytes; odd number of 16-B slots
  static constexpr int K_BYTES = TK * KROW;
  static constexpr int V_BYTES = NT * 32 * VROW;
  static constexpr int LDS_BYTES = K_BYTES + V_BYTES;
  static constexpr int STEPS = 32 / (2 * V);                  // mma16 steps per 32-key sub-tile
};

__device__ __forceinline__ int swap23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

template <typename T, int DCH>
__global__ __launch_bounds__(ATT_THREADS) void attention_kernel(const emo_attention_params p) {
  using Cfg = AttCfg<T, DCH>;
  constexpr int V = Cfg::V, NT = Cfg::NT, KROW = Cfg::KROW, VROW = Cfg::VROW, STEPS = Cfg::STEPS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;
  unsigned char* Vs = smem + Cfg::K_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int qt = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  const int d = p.d;
  const int dch_real = d / V;  // d*sizeof(T) % 16 == 0 checked on the host

  // ---- Q fragments: row q, chunks (2*kk + half)
  const int q = qt * BQ + wave * 32 + l31;
  const bool q_ok = q < p.Lq;
  const T* qrow = (const T*)p.q + ((int64_t)b * p.Lq + (q_ok ? q : 0)) * p.ldq + head * d;
  uint4 qf[DCH / 2];
#pragma unroll
  for (int kk = 0; kk < DCH / 2; kk++) {
    const int c = 2 * kk + half;
    qf[kk] = (q_ok && c < dch_real) ? *(const uint4*)(qrow + c * V) : make_uint4(0, 0, 0, 0);
  }

  // ---- zero the pad regions of the LDS tiles once (pad chunk of K rows, pad rows of V^T)
  for (int i = tid; i < TK * (DCH - dch_real); i += ATT_THREADS) {
    int r = i / (DCH - dch_real), c = dch_real + i % (DCH - dch_real);
    *(uint4*)(Ks + r * KROW + c * 16) = make_uint4(0, 0, 0, 0);
  }
  constexpr int VCH = TK * (int)sizeof(T) / 16;  // 16-B chunks per V^T row
  for (int i = tid; i < (NT * 32 - d) * VCH; i += ATT_THREADS) {
    int r = d + i / VCH, c = i % VCH;
    *(uint4*)(Vs + r * VROW + c * 16) = make_uint4(0, 0, 0, 0);
  }

  f32x16 o[NT];
#pragma unroll
  for (int nt = 0; nt < NT; nt++)
#pragma unroll
    for (int r = 0; r < 16; r++) o[nt][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float c_exp = p.scale * 1.4426950408889634f;

  const int nseg = (p.k1 != nullptr && b >= p.seg1_first_batch) ? 2 : 1;
  for (int seg = 0; seg < nseg; seg++) {
    const int Lk = seg == 0 ? p.Lk0 : p.Lk1;
    const int64_t ldk = seg == 0 ? p.ldk0 : p.ldk1;
    const int64_t ldvt = seg == 0 ? p.ldv0t : p.ldv1t;
    const int kb = seg == 0 ? b / p.seg0_div : b / p.seg1_div;
    const T* kbase = (const T*)(seg == 0 ? p.k0 : p.k1) + (int64_t)kb * Lk * ldk + head * d;
    const T* vbase = (const T*)(seg == 0 ? p.v0t : p.v1t) + ((int64_t)kb * p.heads * d + (int64_t)head * d) * ldvt;
    for (int k0 = 0; k0 < Lk; k0 += TK) {
      __syncthreads();  // previous tile fully consumed
      // ---- stage K tile [64 keys][d] (row = sub-tile*32 + swap23(key&31)) and V^T tile [d][64 keys]
      for (int i = tid; i < TK * dch_real; i += ATT_THREADS) {
        const int key = i / dch_real, c = i % dch_real;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (k0 + key < Lk) v = *(const uint4*)(kbase + (int64_t)(k0 + key) * ldk + c * V);
        *(uint4*)(Ks + ((key & 32) + swap23(key & 31)) * KROW + c * 16) = v;
      }
      for (int i = tid; i < d * VCH; i += ATT_THREADS) {
        const int n = i / VCH, c = i % VCH;
        const int key0 = k0 + c * V;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (key0 < Lk) {
          v = *(const uint4*)(vbase + (int64_t)n * ldvt + key0);
          if (key0 + V > Lk) {  // partial chunk: zero the keys beyond Lk (their P is 0, 0*garbage must not be NaN)
            float f[V];
            unpack16<T>(v, f);
#pragma unroll
            for (int e = 0; e < V; e++) if (key0 + e >= Lk) f[e] = 0.f;
            v = pack16<T>(f);
          }
        }
        *(uint4*)(Vs + n * VROW + c * 16) = v;
      }
      __syncthreads();

#pragma unroll
      for (int st = 0; st < 2; st++) {
        if (k0 + st * 32 >= Lk) break;
        // ---- S^T sub-tile: 32 keys x 32 queries
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; r++) s[r] = 0.f;
        const unsigned char* krow = Ks + (st * 32 + l31) * KROW + half * 16;
#pragma unroll
        for (int kk = 0; kk < DCH / 2; kk++) s = mma16<T>(*(const uint4*)(krow + kk * 32), qf[kk], s);
        // ---- online softmax (lane-local query; the other half of the keys lives in lane^32)
        float mx = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int key = k0 + st * 32 + 16 * (r >> 3) + 8 * half + (r & 7);
          if (key >= Lk) s[r] = -1e30f;
          mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = exp2f((m_run - m_new) * c_exp);
        m_run = m_new;
        float psum = 0.f;
        float pr[16];
#pragma unroll
        for (int r = 0; r < 16; r++) { pr[r] = exp2f((s[r] - m_new) * c_exp); psum += pr[r]; }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
          for (int r = 0; r < 16; r++) o[nt][r] *= alpha;
        // ---- O^T += V^T . P^T
#pragma unroll
        for (int sp = 0; sp < STEPS; sp++) {
          const int r0 = sp * V;
          uint4 pf;
          if constexpr (sizeof(T) == 2) {
            pf = make_uint4(pack_bf2(pr[r0], pr[r0 + 1]), pack_bf2(pr[r0 + 2], pr[r0 + 3]), pack_bf2(pr[r0 + 4], pr[r0 + 5]),
                            pack_bf2(pr[r0 + 6], pr[r0 + 7]));
          } else {
            pf = make_uint4(__float_as_uint(pr[r0]), __float_as_uint(pr[r0 + 1]), __float_as_uint(pr[r0 + 2]),
                            __float_as_uint(pr[r0 + 3]));
          }
          const int key_off = st * 32 + 16 * (r0 >> 3) + 8 * half + (r0 & 7);
          const unsigned char* vrow = Vs + l31 * VROW + key_off * (int)sizeof(T);
#pragma unroll
          for (int nt = 0; nt < NT; nt++) o[nt] = mma16<T>(*(const uint4*)(vrow + nt * 32 * VROW), pf, o[nt]);
        }
      }
    }
  }

  // ---- normalise and store: lane holds O[q][n], n = nt*32 + 8*(r>>2) + 4*half + (r&3)
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (q_ok) {
    T* orow = (T*)p.out + ((int64_t)b * p.Lq + q) * p.ldo + head * d;
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int n0 = nt * 32 + 8 * g + 4 * half;
        if (n0 < d) {
          float v0 = o[nt][4 * g] * inv, v1 = o[nt][4 * g + 1] * inv, v2 = o[nt][4 * g + 2] * inv, v3 = o[nt][4 * g + 3] * inv;
          if constexpr (sizeof(T) == 2) {
            *(uint2*)(orow + n0) = make_uint2(pack_bf2(v0, v1), pack_bf2(v2, v3));
          } else {
            *(float4*)(orow + n0) = make_float4(v0, v1, v2, v3);
          }
        }
      }
  }
}

template <typename T, int DCH>
static int launch_attention(const emo_attention_params& p, hipStream_t st) {
  using Cfg = AttCfg<T, DCH>;
  auto kern = attention_kernel<T, DCH>;
  if (Cfg::LDS_BYTES > 64 * 1024) {
    static bool once = false;  // idempotent attribute; benign race
    if (!once) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
      if (e != hipSuccess) return emo_fail(EMO_ERR_HIP, "emo_attention: hipFuncSetAttribute: %s", hipGetErrorString(e));
      once = true;
    }
  }
  dim3 grid((p.Lq + BQ - 1) / BQ, p.heads, p.B);
  kern<<<grid, ATT_THREADS, Cfg::LDS_BYTES, st>>>(p);
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

template <typename T>
static int dispatch_attention(const emo_attention_params& p, hipStream_t st) {
  constexpr int V = TT<T>::VEC;
  const int dch = p.d / V;
  if (dch <= 2) return launch_attention<T, 2>(p, st);
  if (dch <= 4) return launch_attention<T, 4>(p, st);
  if (dch <= 6) return launch_attention<T, 6>(p, st);
  if (dch <= 10) return launch_attention<T, 10>(p, st);
  if (dch <= 20) return launch_attention<T, 20>(p, st);
  if (dch <= 40) return launch_attention<T, 40>(p, st);
  return emo_fail(EMO_ERR_UNSUPPORTED, "emo_attention: head dim %d too large", p.d);
}

extern "C" int emo_attention(const emo_attention_params* pp, void* stream) {
  EMO_CHECK(pp, EMO_ERR_NULL, "emo_attention: null params");
  const emo_attention_params& p = *pp;
  EMO_CHECK(p.q && p.k0 && p.v0t && p.out, EMO_ERR_NULL, "emo_attention: null pointer");
  EMO_CHECK(p.dtype == EMO_F32 || p.dtype == EMO_BF16, EMO_ERR_BAD_DTYPE, "emo_attention: dtype %d", p.dtype);
  const int V = p.dtype == EMO_F32 ? 4 : 8;
  EMO_CHECK(p.B > 0 && p.Lq > 0 && p.Lk0 > 0 && p.heads > 0 && p.d > 0, EMO_ERR_BAD_SHAPE, "emo_attention: bad shape");
  EMO_CHECK(p.d % V == 0, EMO_ERR_BAD_SHAPE, "emo_attention: head dim %d must be a multiple of %d", p.d, V);
  EMO_CHECK(p.ldq % V == 0 && p.ldk0 % V == 0 && p.ldv0t % V == 0 && p.ldo % 4 == 0, EMO_ERR_BAD_SHAPE, "emo_attention: leading dims");
  EMO_CHECK(p.ldv0t >= p.Lk0, EMO_ERR_BAD_SHAPE, "emo_attention: ldv0t < Lk0");
  EMO_CHECK(p.heads <= 65535 && p.B <= 65535, EMO_ERR_BAD_SHAPE, "emo_attention: grid limits");
  EMO_CHECK(p.seg0_div >= 1, EMO_ERR_BAD_SHAPE, "emo_attention: seg0_div must be >= 1");
  if (p.k1) {
    EMO_CHECK(p.v1t && p.Lk1 > 0 && p.seg1_div > 0 && p.ldk1 % V == 0 && p.ldv1t % V == 0 && p.ldv1t >= p.Lk1, EMO_ERR_BAD_SHAPE,
              "emo_attention: segment-1 geometry");
  }
  hipStream_t st = as_stream(stream);
  return p.dtype == EMO_F32 ? dispatch_attention<float>(p, st) : dispatch_attention<bf16_t>(p, st);
}
