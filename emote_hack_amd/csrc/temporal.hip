// temporal.hip - AnimateDiff motion-module temporal self-attention (motion_module.py:275-334,
// models/motionmodule.py:300).  Tokens are "(b f) d c -> (b d) f c": for every pixel and head a tiny
// F x F attention (F <= 32) over frames.  HBM / latency bound, so no MFMA reshaping:
// the kernel is organised around coalesced 16-byte row segments and LDS.
//
// One workgroup = P pixels x `hpb` heads of one batch element:
//   stage   Q,K,V[f][p][hpb*d] for all F frames into LDS with fully coalesced row-segment loads (the
//           einops transposes of the reference are folded into this indexing - nothing is permuted in HBM)
//   phase A one thread per (p, head, fq, fk): score = scale * <q, k>   -> LDS S
//   phase B one thread per (p, head, fq): softmax over fk in f32
//   phase C one thread per (p, head, fq, 16-byte channel vector): out = sum_fk P * V, stored straight to the
//           output rows (coalesced).
#include "common.h"

static constexpr int TA_THREADS = 256;

template <typename T>
__global__ __launch_bounds__(TA_THREADS) void temporal_attention_kernel(const T* __restrict__ qkv, int64_t ldqkv, T* __restrict__ out,
                                                                        int64_t ldo, int F, int HW, int C, int d, int hpb, int P,
                                                                        float scale) {
  constexpr int V = TT<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int CT = hpb * d;                       // channels handled by this block
  const int ROW = P * CT * (int)sizeof(T) + 16; // bytes per frame row (+16: break the power-of-two stride)
  unsigned char* Qs = smem;
  unsigned char* Ks = Qs + F * ROW;
  unsigned char* Vs = Ks + F * ROW;
  float* S = (float*)(Vs + F * ROW);            // [P][hpb][F][F]

  const int tid = threadIdx.x;
  const int pix0 = blockIdx.x * P, hg = blockIdx.y, b = blockIdx.z;
  const int c0 = hg * CT;
  const int CTV = CT / V;

  // ---- stage
  for (int i = tid; i < 3 * F * P * CTV; i += TA_THREADS) {
    int cv = i % CTV; int r = i / CTV;
    int p = r % P; r /= P;
    int f = r % F; int which = r / F;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (pix0 + p < HW)
      v = *(const uint4*)(qkv + ((int64_t)(b * F + f) * HW + pix0 + p) * ldqkv + which * C + c0 + cv * V);
    *(uint4*)(smem + (which * F + f) * ROW + (p * CT + cv * V) * (int)sizeof(T)) = v;
  }
  __syncthreads();

  // ---- phase A: scores
  const int dv = d / V;
  for (int i = tid; i < P * hpb * F * F; i += TA_THREADS) {
    int fk = i % F; int r = i / F;
    int fq = r % F; r /= F;
    int hh = r % hpb; int p = r / hpb;
    const unsigned char* qp = Qs + fq * ROW + (p * CT + hh * d) * (int)sizeof(T);
    const unsigned char* kp = Ks + fk * ROW + (p * CT + hh * d) * (int)sizeof(T);
    float acc = 0.f;
    for (int c = 0; c < dv; c++) {
      float a[V], k[V];
      unpack16<T>(*(const uint4*)(qp + c * 16), a);
      unpack16<T>(*(const uint4*)(kp + c * 16), k);
#pragma unroll
      for (int e = 0; e < V; e++) acc += a[e] * k[e];
    }
    S[i] = acc * scale;
  }
  __syncthreads();

  // ---- phase B: softmax rows
  for (int i = tid; i < P * hpb * F; i += TA_THREADS) {
    float* row = S + (int64_t)i * F;
    float mx = -1e30f;
    for (int k = 0; k < F; k++) mx = fmaxf(mx, row[k]);
    float sum = 0.f;
    for (int k = 0; k < F; k++) { float e = __expf(row[k] - mx); row[k] = e; sum += e; }
    const float inv = 1.0f / sum;
    for (int k = 0; k < F; k++) row[k] *= inv;
  }
  __syncthreads();

  // ---- phase C: out = P . V
  for (int i = tid; i < F * P * CTV; i += TA_THREADS) {
    int cv = i % CTV; int r = i / CTV;
    int p = r % P; int fq = r / P;
    if (pix0 + p >= HW) continue;
    const int hh = (cv * V) / d;
    const float* prow = S + ((int64_t)(p * hpb + hh) * F + fq) * F;
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; e++) acc[e] = 0.f;
    for (int fk = 0; fk < F; fk++) {
      float vv[V];
      unpack16<T>(*(const uint4*)(Vs + fk * ROW + (p * CT + cv * V) * (int)sizeof(T)), vv);
      float w = prow[fk];
      if constexpr (sizeof(T) == 2) w = bf2f(f2bf(w));  // probabilities are cast to the value dtype (orig_attention.py:675)
#pragma unroll
      for (int e = 0; e < V; e++) acc[e] += w * vv[e];
    }
    *(uint4*)(out + ((int64_t)(b * F + fq) * HW + pix0 + p) * ldo + c0 + cv * V) = pack16<T>(acc);
  }
}

extern "C" int emo_temporal_attention(const void* qkv, int64_t ldqkv, void* out, int64_t ldo, int B, int F, int HW, int heads, int d,
                                      float scale, int dtype, void* stream) {
  EMO_CHECK(qkv && out, EMO_ERR_NULL, "emo_temporal_attention: null pointer");
  EMO_CHECK(dtype == EMO_F32 || dtype == EMO_BF16, EMO_ERR_BAD_DTYPE, "emo_temporal_attention: dtype %d", dtype);
  const int V = dtype == EMO_F32 ? 4 : 8, esz = dtype == EMO_F32 ? 4 : 2;
  EMO_CHECK(B > 0 && F > 0 && F <= 32 && HW > 0 && heads > 0 && d > 0, EMO_ERR_BAD_SHAPE, "emo_temporal_attention: B=%d F=%d HW=%d", B, F, HW);
  EMO_CHECK(d % V == 0 && ldqkv % V == 0 && ldo % V == 0, EMO_ERR_BAD_SHAPE, "emo_temporal_attention: d=%d must be a multiple of %d", d, V);
  const int C = heads * d;
  EMO_CHECK(ldqkv >= 3 * C && ldo >= C, EMO_ERR_BAD_SHAPE, "emo_temporal_attention: leading dims");
  // choose heads-per-block / pixels-per-block so the Q,K,V stage stays under ~48 KB
  const int budget = 48 * 1024;
  int hpb = heads;
  while (hpb > 1 && (hpb % 2 == 0) && 3 * F * (hpb * d * esz + 16) > budget) hpb /= 2;
  int P = 1;
  while (P < 8 && P * 2 <= HW && 3 * F * (2 * P * hpb * d * esz + 16) + 2 * P * hpb * F * F * 4 <= budget) P *= 2;
  const size_t lds = (size_t)3 * F * (P * hpb * d * esz + 16) + (size_t)P * hpb * F * F * 4;
  EMO_CHECK(lds <= 64 * 1024, EMO_ERR_UNSUPPORTED, "emo_temporal_attention: LDS %zu", lds);
  EMO_CHECK(B <= 65535 && heads / hpb <= 65535, EMO_ERR_BAD_SHAPE, "emo_temporal_attention: grid limits");
  dim3 grid((HW + P - 1) / P, heads / hpb, B);
  hipStream_t st = as_stream(stream);
  if (dtype == EMO_F32)
    temporal_attention_kernel<float><<<grid, TA_THREADS, lds, st>>>((const float*)qkv, ldqkv, (float*)out, ldo, F, HW, C, d, hpb, P, scale);
  else
    temporal_attention_kernel<bf16_t><<<grid, TA_THREADS, lds, st>>>((const bf16_t*)qkv, ldqkv, (bf16_t*)out, ldo, F, HW, C, d, hpb, P, scale);
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}
