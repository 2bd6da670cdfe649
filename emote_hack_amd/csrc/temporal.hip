// temporal.hip - AnimateDiff motion-module temporal self-attention (motion_module.py:275-334,
// models/motionmodule.py:300).  Tokens are "(b f) d c -> (b d) f c": for every pixel and head a tiny
// F x F attention (F <= 32) over frames.  HBM / latency bound, so no MFMA reshaping:
// the kernel is organised around coalesced 16-byte row segments and LDS.
//
// One workgroup = P pixels x `hpb` heads of one batch element:
//   stage   Q,K,V[f][p][hpb*d] for all F frames into LDS with fully coalesced row-segment loads (the
//           einops transposes of the reference are folded into this indexing - nothing is permuted in HBM)
//   phase A one work item per (p, head, 3 x 3 block of (fq, fk)): score = scale * <q, k>   -> LDS S
//   phase B one thread per (p, head, fq): softmax over fk in f32
//   phase C one thread per (p, head, fq, 16-byte channel vector): out = sum_fk P * V, stored straight to the
//           output rows (coalesced).
#include "common.h"

static constexpr int TA_THREADS = 256;

// (dot16<T>: <q, k> over one 16-byte chunk pair - common.h)

// All index arithmetic is thread-fixed or incremental: the first version decomposed a flat index with runtime div / mod in
// every loop iteration and was VALU-bound on that (124 us for 252 MB at the 64x64 level).
template <typename T>
__global__ __launch_bounds__(TA_THREADS) void temporal_attention_kernel(const T* __restrict__ qkv, int64_t ldqkv, T* __restrict__ out,
                                                                        int64_t ldo, int F, int HW, int C, int d, int hpb, int P,
                                                                        float scale, int JW, int FP) {
  // JW = min(P * CT / V, 256): threads per staged row;  FP = 16 or 32 >= F: padded score grid side
  constexpr int V = TT<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int CT = hpb * d;                       // channels handled by this block
  const int ROW = P * CT * (int)sizeof(T) + 16; // bytes per frame row (+16: break the power-of-two stride)
  unsigned char* Qs = smem;
  unsigned char* Ks = Qs + F * ROW;
  unsigned char* Vs = Ks + F * ROW;
  float* S = (float*)(Vs + F * ROW);            // [P * hpb][F][F]

  const int tid = threadIdx.x;
  const int pix0 = blockIdx.x * P, hg = blockIdx.y, b = blockIdx.z;
  const int c0 = hg * CT;
  const int CTV = CT / V, PC = P * CTV;         // 16-byte vectors per staged row
  // JW threads share a row (JW = P * CT / V when that is <= 256: one div per thread, every lane busy; else 256)
  const int jl = tid % JW, rstep = TA_THREADS / JW, r0 = tid / JW;   // r0 >= rstep: idle tail lanes

  // ---- stage: rows (which, f) of the block's [3][F] rows, vector j = p * CTV + cv; thread -> (j, f = r0, r0 + rstep, ..):
  // the q / k / v vectors of up to 4 frames (12 loads) are in flight per thread before the first LDS store
  for (int j0 = 0; j0 < PC; j0 += JW) {
    const int j = j0 + jl;
    if (j >= PC || r0 >= rstep) break;
    const int pp = j / CTV, cv = j - pp * CTV;
    const bool pix_ok = pix0 + pp < HW;
    const T* src = qkv + ((int64_t)(b * F + r0) * HW + (pix_ok ? pix0 + pp : 0)) * ldqkv + c0 + cv * V;
    const int64_t fstep = (int64_t)rstep * HW * ldqkv;
    unsigned char* dst = smem + j * 16;
#pragma unroll 4
    for (int f = r0; f < F; f += rstep) {
      uint4 vq = make_uint4(0, 0, 0, 0), vk = vq, vv = vq;
      if (pix_ok) { vq = *(const uint4*)src; vk = *(const uint4*)(src + C); vv = *(const uint4*)(src + 2 * C); }
      *(uint4*)(dst + f * ROW) = vq;
      *(uint4*)(dst + (F + f) * ROW) = vk;
      *(uint4*)(dst + (2 * F + f) * ROW) = vv;
      src += fstep;
    }
  }
  __syncthreads();

  // ---- phase A: scores.  Work item -> (pixel-head pair, 3 x 3 block of (fq, fk)): 6 row chunks from LDS feed 9 dot products
  // (one score per thread re-read the q and k rows of every head 12 times over: 327 KB of LDS reads per 23 KB staged - the
  // kernel was co-bound by LDS, not by HBM latency, `profiles/` r02: persistent + prefetched variant no faster)
  const int dv = d / V, NPH = P * hpb;
  {
    const int nb = (F + 2) / 3, per = nb * nb;
    for (int it = tid; it < NPH * per; it += TA_THREADS) {
      const int ph = it / per, r = it - ph * per, bq = r / nb, bk = r - bq * nb;
      const unsigned char* qp[3];
      const unsigned char* kp[3];
#pragma unroll
      for (int u = 0; u < 3; u++) {
        const int fq = bq * 3 + u < F ? bq * 3 + u : F - 1, fk = bk * 3 + u < F ? bk * 3 + u : F - 1;
        qp[u] = Qs + fq * ROW + ph * d * (int)sizeof(T);
        kp[u] = Ks + fk * ROW + ph * d * (int)sizeof(T);
      }
      float acc[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
      for (int c = 0; c < dv; c++) {
        uint4 qv[3], kv[3];
#pragma unroll
        for (int u = 0; u < 3; u++) { qv[u] = *(const uint4*)(qp[u] + c * 16); kv[u] = *(const uint4*)(kp[u] + c * 16); }
#pragma unroll
        for (int u = 0; u < 3; u++)
#pragma unroll
          for (int w = 0; w < 3; w++) acc[u][w] = dot16<T>(qv[u], kv[w], acc[u][w]);
      }
      float* sp = S + ph * F * F;
#pragma unroll
      for (int u = 0; u < 3; u++)
#pragma unroll
        for (int w = 0; w < 3; w++) {
          const int fq = bq * 3 + u, fk = bk * 3 + w;
          if (fq < F && fk < F) sp[fq * F + fk] = acc[u][w] * scale;
        }
    }
  }
  __syncthreads();

  // ---- phase B: softmax rows; probabilities are cast to the value dtype (orig_attention.py:675)
  for (int i = tid; i < NPH * F; i += TA_THREADS) {
    float* row = S + i * F;
    float mx = -1e30f;
    for (int k = 0; k < F; k++) mx = fmaxf(mx, row[k]);
    float sum = 0.f;
    for (int k = 0; k < F; k++) { float e = __expf(row[k] - mx); row[k] = e; sum += e; }
    const float inv = 1.0f / sum;
    for (int k = 0; k < F; k++) {
      float w = row[k] * inv;
      w = round_through<T>(w);
      row[k] = w;
    }
  }
  __syncthreads();

  // ---- phase C: out = P . V: thread -> vector j of output rows fq = r0, r0 + rstep, ...
  for (int j0 = 0; j0 < PC; j0 += JW) {
    const int j = j0 + jl;
    if (j >= PC || r0 >= rstep) break;
    const int pp = j / CTV, cv = j - pp * CTV;
    if (pix0 + pp >= HW) continue;
    const int hh = (cv * V) / d;
    const float* pbase = S + (pp * hpb + hh) * F * F;
    const unsigned char* vp = Vs + j * 16;
    // up to 4 output rows fq = r0 + u * rstep share every V vector read
    for (int fq0 = r0; fq0 < F; fq0 += 4 * rstep) {
      float acc[4][V];
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int e = 0; e < V; e++) acc[u][e] = 0.f;
      for (int fk = 0; fk < F; fk++) {
        float vv[V];
        unpack16<T>(*(const uint4*)(vp + fk * ROW), vv);
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int fq = fq0 + u * rstep;
          const float w = fq < F ? pbase[fq * F + fk] : 0.f;
#pragma unroll
          for (int e = 0; e < V; e++) acc[u][e] += w * vv[e];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int fq = fq0 + u * rstep;
        if (fq < F) *(uint4*)(out + ((int64_t)(b * F + fq) * HW + pix0 + pp) * ldo + c0 + cv * V) = pack16<T>(acc[u]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ MFMA formulation (2-byte types, F <= 16)
// The kernel above spends ~400 VALU instructions per (pixel, head) on 12 x 12 x 40 dot products and is bound by instruction
// issue, not by HBM (2.7 TB/s at the 64x64 level).  Here ONE WAVE owns one (batch row, pixel, head) and the two contractions
// run on the matrix pipe with operands that need no re-layout:
//   S^T = K Q^T   v_mfma_f32_16x16x32: lane (l & 15 = frame, l >> 4 = group of 8 channels) loads its A (K row) and B (Q row)
//                 fragment straight from the qkv rows - 16 bytes per lane, no LDS; frames >= F and channels >= d read as 0.
//                 The accumulator holds S^T[fk = 4 (l >> 4) + r][fq = l & 15]: the softmax over fk is 4 registers x 4 lane
//                 groups (two v_permlane swaps), and the four probabilities of a lane ARE the B operand of the K = 16 MFMA.
//   O^T = V^T P^T v_mfma_f32_16x16x16: A = V^T[n][fk] = V[fk][n] - the V rows go to a wave-private LDS image [frame][d] by
//                 LDS-DMA (lane-linear = row-major, no registers) and ds_read_b64_tr_b16 hands each lane the four frames of its
//                 channel; pad frames are zero rows written once.  O^T[n = 4 (l >> 4) + r][fq]: 4 consecutive channels per
//                 lane -> one 8-byte store per 16 channels.
// ~80 instructions per (pixel, head) instead of ~400.  F <= 32 as two 16-frame blocks; f32 (validation mode: exact-f32 MFMAs have
// another operand layout) keeps the kernel above.
typedef __bf16 ta_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ta_f16x8 __attribute__((ext_vector_type(8)));
typedef short ta_s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 ta_f16x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ f32x4 ta_mma32(const uint4& a, const uint4& b, f32x4 c);
template <> __device__ __forceinline__ f32x4 ta_mma32<bf16_t>(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ta_bf16x8, a), __builtin_bit_cast(ta_bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 ta_mma32<f16_t>(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ta_f16x8, a), __builtin_bit_cast(ta_f16x8, b), c, 0, 0, 0);
}
template <typename T> __device__ __forceinline__ f32x4 ta_mma16(const uint2& a, const uint2& b, f32x4 c);
template <> __device__ __forceinline__ f32x4 ta_mma16<bf16_t>(const uint2& a, const uint2& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(ta_s16x4, a), __builtin_bit_cast(ta_s16x4, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 ta_mma16<f16_t>(const uint2& a, const uint2& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(ta_f16x4, a), __builtin_bit_cast(ta_f16x4, b), c, 0, 0, 0);
}
// all-reduce over the four 16-lane groups of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48): v_permlane16_swap exchanges the odd groups
// of its first operand with the even groups of the second, v_permlane32_swap the upper half with the lower half
template <typename OP> __device__ __forceinline__ float ta_allreduce_groups(float x, OP op) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float y = op(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
  return op(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

static constexpr int TAM_WAVES = 4;
// NFB = 16-frame blocks: 1 (F <= 16) or 2 (F <= 32: S^T is 2 x 2 blocks, the softmax runs over both key blocks in the lane)
template <typename T, int NFB>
__global__ __launch_bounds__(64 * TAM_WAVES) void temporal_attention_mfma_kernel(const T* __restrict__ qkv, int64_t ldqkv, T* __restrict__ out,
                                                                                 int64_t ldo, int F, int HW, int C, int d, int heads,
                                                                                 float c_exp, int64_t items, int wave_lds) {
  static_assert(sizeof(T) == 2, "MFMA temporal attention: 2-byte element types");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, g = lane >> 4;
  const int dch = d / 8, vrow = d * 2;               // 16-byte chunks / bytes per V row
  unsigned char* vimg = smem + wave * wave_lds;       // [16 * NFB frames][d] image of this wave
  const unsigned vimg_a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)vimg;
  // pad frames F .. 16 NFB - 1: zero rows, written once (the DMA below never touches them); + the slack behind the image that the
  // transpose reads of the last channel block run into (their results are rows n >= d of O^T: never stored, but keep them finite)
  for (int off = F * vrow + lane * 16; off < wave_lds; off += 64 * 16) *(uint4*)(vimg + off) = make_uint4(0, 0, 0, 0);
  bool f_ok[NFB];
#pragma unroll
  for (int fb = 0; fb < NFB; fb++) f_ok[fb] = fb * 16 + fr < F;
  const int nvec = F * dch;                           // 16-byte vectors of one (pixel, head)'s V rows
  const int ksteps = (d + 31) / 32, nblocks = (d + 15) / 16;
  // address of this lane's piece of the 4 x 16 block that ds_read_b64_tr_b16 transposes: frame 4 g + (fr >> 2), channels (fr & 3) * 4 ..
  const unsigned tr_lane = vimg_a + (4 * g + (fr >> 2)) * vrow + (fr & 3) * 8;
  for (int64_t item = (int64_t)blockIdx.x * TAM_WAVES + wave; item < items; item += (int64_t)gridDim.x * TAM_WAVES) {
    const int head = (int)(item % heads);
    const int64_t bp = item / heads;                  // batch row * HW + pixel
    const int pix = (int)(bp % HW), b = (int)(bp / HW);
    const int64_t row0 = (int64_t)b * F * HW + pix;   // row of frame 0; frame f: + f * HW
    const T* base = qkv + head * d;
    // ---- V rows -> LDS image (LDS-DMA: vector j = r * 64 + lane lands at byte j * 16 = frame (j / dch), chunk (j % dch))
    for (int j0 = 0; j0 < nvec; j0 += 64) {
      const int j = j0 + lane;
      if (j < nvec) {
        const int fk = j / dch, c = j - fk * dch;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (row0 + (int64_t)fk * HW) * ldqkv + 2 * C + c * 8),
                                         (__attribute__((address_space(3))) void*)(vimg + j0 * 16), 16, 0, 0);
      }
    }
    // ---- S^T = K Q^T: blocks (kb, qb) of 16 x 16
    const T* qrow[NFB];
#pragma unroll
    for (int fb = 0; fb < NFB; fb++) qrow[fb] = base + (row0 + (int64_t)(f_ok[fb] ? fb * 16 + fr : 0) * HW) * ldqkv;
    f32x4 s[NFB][NFB];
#pragma unroll
    for (int kb = 0; kb < NFB; kb++)
#pragma unroll
      for (int qb = 0; qb < NFB; qb++) s[kb][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < ksteps; ks++) {
      const int c = ks * 4 + g;
      uint4 kf[NFB], qf[NFB];
#pragma unroll
      for (int fb = 0; fb < NFB; fb++) {
        kf[fb] = make_uint4(0, 0, 0, 0); qf[fb] = kf[fb];
        if (f_ok[fb] && c < dch) { qf[fb] = *(const uint4*)(qrow[fb] + c * 8); kf[fb] = *(const uint4*)(qrow[fb] + C + c * 8); }
      }
#pragma unroll
      for (int kb = 0; kb < NFB; kb++)
#pragma unroll
        for (int qb = 0; qb < NFB; qb++) s[kb][qb] = ta_mma32<T>(kf[kb], qf[qb], s[kb][qb]);
    }
    // ---- softmax over fk = 16 kb + 4 g + r (rows of S^T) for the lane's fq = 16 qb + fr; probabilities rounded to T
    // (orig_attention.py:675)
    uint2 pb[NFB][NFB];
#pragma unroll
    for (int qb = 0; qb < NFB; qb++) {
      float p[NFB][4];
      float mx = -1e30f;
#pragma unroll
      for (int kb = 0; kb < NFB; kb++)
#pragma unroll
        for (int r = 0; r < 4; r++) { p[kb][r] = (kb * 16 + 4 * g + r < F) ? s[kb][qb][r] : -1e30f; mx = fmaxf(mx, p[kb][r]); }
      mx = ta_allreduce_groups(mx, [](float x, float y) { return fmaxf(x, y); });
      float sum = 0.f;
#pragma unroll
      for (int kb = 0; kb < NFB; kb++)
#pragma unroll
        for (int r = 0; r < 4; r++) { p[kb][r] = __builtin_amdgcn_exp2f((p[kb][r] - mx) * c_exp); sum += p[kb][r]; }
      sum = ta_allreduce_groups(sum, [](float x, float y) { return x + y; });
      const float inv = 1.0f / sum;
#pragma unroll
      for (int kb = 0; kb < NFB; kb++) pb[kb][qb] = make_uint2(pack2<T>(p[kb][0] * inv, p[kb][1] * inv), pack2<T>(p[kb][2] * inv, p[kb][3] * inv));
    }
    // ---- O^T = V^T P^T, 16 channels at a time
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's own DMA: the image is complete
    T* orow[NFB];
#pragma unroll
    for (int qb = 0; qb < NFB; qb++) orow[qb] = out + (row0 + (int64_t)(f_ok[qb] ? qb * 16 + fr : 0) * HW) * ldo + head * d + 4 * g;
    for (int nb = 0; nb < nblocks; nb++) {
      uint2 vf[NFB];
#pragma unroll
      for (int kb = 0; kb < NFB; kb++)
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=&v"(vf[kb]) : "v"(tr_lane + kb * 16 * vrow + nb * 32) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int qb = 0; qb < NFB; qb++) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NFB; kb++) o = ta_mma16<T>(vf[kb], pb[kb][qb], o);
        if (f_ok[qb] && nb * 16 + 4 * g < d) *(uint2*)(orow[qb] + nb * 16) = make_uint2(pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3]));
      }
    }
    // (every transpose read above was waited for: the next item's DMA may overwrite the image)
  }
}

template <typename T>
static void launch_temporal_mfma(const void* qkv, int64_t ldqkv, void* out, int64_t ldo, int B, int F, int HW, int heads, int d, float scale,
                                 hipStream_t st) {
  const int nfb = F <= 16 ? 1 : 2, C = heads * d;
  const int64_t items = (int64_t)B * HW * heads;
  const int wave_lds = 16 * nfb * d * 2 + 64;                    // [16 nfb frames][d] + slack for the last channel block's transpose reads
  int64_t gx = (items + TAM_WAVES - 1) / TAM_WAVES;
  if (gx > 256 * 8) gx = 256 * 8;                                // 8 blocks of 4 waves per CU, each wave walks its items
  const float c_exp = scale * 1.4426950408889634f;
  if (nfb == 1)
    temporal_attention_mfma_kernel<T, 1><<<(unsigned)gx, 64 * TAM_WAVES, TAM_WAVES * wave_lds, st>>>((const T*)qkv, ldqkv, (T*)out, ldo, F, HW, C, d, heads, c_exp,
                                                                                                       items, wave_lds);
  else
    temporal_attention_mfma_kernel<T, 2><<<(unsigned)gx, 64 * TAM_WAVES, TAM_WAVES * wave_lds, st>>>((const T*)qkv, ldqkv, (T*)out, ldo, F, HW, C, d, heads, c_exp,
                                                                                                       items, wave_lds);
}

extern "C" int emo_temporal_attention(const void* qkv, int64_t ldqkv, void* out, int64_t ldo, int B, int F, int HW, int heads, int d,
                                      float scale, int dtype, void* stream) {
  EMO_CHECK(qkv && out, EMO_ERR_NULL, "emo_temporal_attention: null pointer");
  EMO_CHECK(emo_dtype_ok(dtype), EMO_ERR_BAD_DTYPE, "emo_temporal_attention: dtype %d", dtype);
  const int V = emo_dtype_vec(dtype), esz = dtype == EMO_F32 ? 4 : 2;
  EMO_CHECK(B > 0 && F > 0 && F <= 32 && HW > 0 && heads > 0 && d > 0, EMO_ERR_BAD_SHAPE, "emo_temporal_attention: B=%d F=%d HW=%d", B, F, HW);
  EMO_CHECK(d % V == 0 && ldqkv % V == 0 && ldo % V == 0, EMO_ERR_BAD_SHAPE, "emo_temporal_attention: d=%d must be a multiple of %d", d, V);
  const int C = heads * d;
  EMO_CHECK(ldqkv >= 3 * C && ldo >= C, EMO_ERR_BAD_SHAPE, "emo_temporal_attention: leading dims");
  // MFMA formulation (one wave per (batch row, pixel, head)); F <= 32 checked above; its four wave-private V images must fit the 64 KB a
  // launch gets without an attribute (d = 256 with F > 16 does not: the kernel below serves it)
  if (dtype != EMO_F32 && d % 8 == 0 && d <= 256 && TAM_WAVES * (16 * (F <= 16 ? 1 : 2) * d * 2 + 64) <= 64 * 1024) {
    if (dtype == EMO_BF16) launch_temporal_mfma<bf16_t>(qkv, ldqkv, out, ldo, B, F, HW, heads, d, scale, as_stream(stream));
    else launch_temporal_mfma<f16_t>(qkv, ldqkv, out, ldo, B, F, HW, heads, d, scale, as_stream(stream));
    EMO_LAUNCH_CHECK();
    return EMO_OK;
  }
  // choose heads-per-block / pixels-per-block so the Q,K,V stage stays under ~48 KB
  const int budget = 48 * 1024;
  int hpb = heads;
  while (hpb > 1 && (hpb % 2 == 0) && 3 * F * (hpb * d * esz + 16) + hpb * F * F * 4 > budget) hpb /= 2;   // stage + score matrix
  int P = 1;
  while (P < 8 && P * 2 <= HW && 3 * F * (2 * P * hpb * d * esz + 16) + 2 * P * hpb * F * F * 4 <= budget) P *= 2;
  const size_t lds = (size_t)3 * F * (P * hpb * d * esz + 16) + (size_t)P * hpb * F * F * 4;
  EMO_CHECK(lds <= 64 * 1024, EMO_ERR_UNSUPPORTED, "emo_temporal_attention: LDS %zu", lds);
  EMO_CHECK(B <= 65535 && heads / hpb <= 65535, EMO_ERR_BAD_SHAPE, "emo_temporal_attention: grid limits");
  dim3 grid((HW + P - 1) / P, heads / hpb, B);
  const int JW = P * hpb * d / V < TA_THREADS ? P * hpb * d / V : TA_THREADS;
  const int FP = F <= 16 ? 16 : 32;
  hipStream_t st = as_stream(stream);
  EMO_DISPATCH(dtype, "emo_temporal_attention",
               (temporal_attention_kernel<T><<<grid, TA_THREADS, lds, st>>>((const T*)qkv, ldqkv, (T*)out, ldo, F, HW, C, d, hpb, P, scale, JW, FP)));
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}
