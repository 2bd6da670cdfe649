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

extern "C" int emo_temporal_attention(const void* qkv, int64_t ldqkv, void* out, int64_t ldo, int B, int F, int HW, int heads, int d,
                                      float scale, int dtype, void* stream) {
  EMO_CHECK(qkv && out, EMO_ERR_NULL, "emo_temporal_attention: null pointer");
  EMO_CHECK(emo_dtype_ok(dtype), EMO_ERR_BAD_DTYPE, "emo_temporal_attention: dtype %d", dtype);
  const int V = emo_dtype_vec(dtype), esz = dtype == EMO_F32 ? 4 : 2;
  EMO_CHECK(B > 0 && F > 0 && F <= 32 && HW > 0 && heads > 0 && d > 0, EMO_ERR_BAD_SHAPE, "emo_temporal_attention: B=%d F=%d HW=%d", B, F, HW);
  EMO_CHECK(d % V == 0 && ldqkv % V == 0 && ldo % V == 0, EMO_ERR_BAD_SHAPE, "emo_temporal_attention: d=%d must be a multiple of %d", d, V);
  const int C = heads * d;
  EMO_CHECK(ldqkv >= 3 * C && ldo >= C, EMO_ERR_BAD_SHAPE, "emo_temporal_attention: leading dims");
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
