// norm.hip - GroupNorm (joint 5-D or per-frame statistics) + SiLU, and LayerNorm (+ temporal PE add)
// over NHWC rows.  HBM-bound kernels: every global access is a 16-byte vector on a fully coalesced row.
//
// GroupNorm: reference semantics are nn.GroupNorm applied to a 5-D tensor in the resnets
// (resnet.py:180,191 -> statistics joint over C/G x F x H x W) and to per-frame 4-D tensors in the
// transformers (attention.py:124, motion_module.py:147).  Both are "instances of S consecutive rows".
//   pass 1  gn_partial_kernel : block = (instance n, row-chunk); per-channel sum / sumsq in registers,
//                               fixed-order LDS reduction -> partial (sum, sumsq) per group  (deterministic)
//   pass 2  gn_finalize_kernel: combine partials in double -> (mean, rstd)
//   pass 3  gn_apply_kernel   : y = (x-mean)*rstd*gamma+beta, optional SiLU
#include "common.h"

static constexpr int GN_THREADS = 256;
static constexpr int GN_MAXJ = 3;  // column vectors per thread: C <= 256*3*VEC

static inline int gn_nsplit(int64_t S) {
  int64_t n = (S + 15) / 16;
  if (n > 1024) n = 1024;
  if (n < 1) n = 1;
  return (int)n;
}
extern "C" size_t emo_groupnorm_workspace_bytes(int N, int64_t S, int C, int G) {
  (void)C;
  return (size_t)N * gn_nsplit(S) * G * 2 * sizeof(float);
}

template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_partial_kernel(const T* __restrict__ x, int ldx, float* __restrict__ partials,
                                                                int64_t S, int C, int G, int nsplit) {
  constexpr int V = TT<T>::VEC;
  extern __shared__ float lds[];  // [RP][C][2]
  const int n = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
  const int CV = C / V;
  const int64_t rows_per = (S + nsplit - 1) / nsplit;
  const int64_t s0 = sp * rows_per;
  int64_t s1 = s0 + rows_per; if (s1 > S) s1 = S;
  const int tid = threadIdx.x;
  const T* base = x + (int64_t)n * S * ldx;
  float sum[GN_MAXJ][V], sq[GN_MAXJ][V];
#pragma unroll
  for (int j = 0; j < GN_MAXJ; j++)
#pragma unroll
    for (int e = 0; e < V; e++) { sum[j][e] = 0.f; sq[j][e] = 0.f; }
  int RP;   // row slots in the block
  if (CV <= GN_THREADS) {
    RP = GN_THREADS / CV;
    const int r = tid / CV, cv = tid % CV;
    if (r < RP) {
      int64_t s = s0 + r;
      for (; s + 3 * RP < s1; s += 4 * RP) {   // 4 independent 16-byte loads in flight per thread
        uint4 v0 = *(const uint4*)(base + s * ldx + cv * V);
        uint4 v1 = *(const uint4*)(base + (s + RP) * ldx + cv * V);
        uint4 v2 = *(const uint4*)(base + (s + 2 * RP) * ldx + cv * V);
        uint4 v3 = *(const uint4*)(base + (s + 3 * RP) * ldx + cv * V);
        float f[V];
        unpack16<T>(v0, f);
#pragma unroll
        for (int e = 0; e < V; e++) { sum[0][e] += f[e]; sq[0][e] += f[e] * f[e]; }
        unpack16<T>(v1, f);
#pragma unroll
        for (int e = 0; e < V; e++) { sum[0][e] += f[e]; sq[0][e] += f[e] * f[e]; }
        unpack16<T>(v2, f);
#pragma unroll
        for (int e = 0; e < V; e++) { sum[0][e] += f[e]; sq[0][e] += f[e] * f[e]; }
        unpack16<T>(v3, f);
#pragma unroll
        for (int e = 0; e < V; e++) { sum[0][e] += f[e]; sq[0][e] += f[e] * f[e]; }
      }
      for (; s < s1; s += RP) {
        float f[V];
        unpack16<T>(*(const uint4*)(base + s * ldx + cv * V), f);
#pragma unroll
        for (int e = 0; e < V; e++) { sum[0][e] += f[e]; sq[0][e] += f[e] * f[e]; }
      }
      float* dst = lds + ((int64_t)r * C + cv * V) * 2;
#pragma unroll
      for (int e = 0; e < V; e++) { dst[2 * e] = sum[0][e]; dst[2 * e + 1] = sq[0][e]; }
    }
  } else {
    RP = 1;
    for (int64_t s = s0; s < s1; s++) {
#pragma unroll
      for (int j = 0; j < GN_MAXJ; j++) {
        int cv = tid + j * GN_THREADS;
        if (cv < CV) {
          float f[V];
          unpack16<T>(*(const uint4*)(base + s * ldx + cv * V), f);
#pragma unroll
          for (int e = 0; e < V; e++) { sum[j][e] += f[e]; sq[j][e] += f[e] * f[e]; }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < GN_MAXJ; j++) {
      int cv = tid + j * GN_THREADS;
      if (cv < CV) {
        float* dst = lds + (int64_t)(cv * V) * 2;
#pragma unroll
        for (int e = 0; e < V; e++) { dst[2 * e] = sum[j][e]; dst[2 * e + 1] = sq[j][e]; }
      }
    }
  }
  __syncthreads();
  const int cpg = C / G;
  for (int g = tid; g < G; g += GN_THREADS) {
    float a = 0.f, b = 0.f;
    for (int r = 0; r < RP; r++)
      for (int c = g * cpg; c < (g + 1) * cpg; c++) { a += lds[((int64_t)r * C + c) * 2]; b += lds[((int64_t)r * C + c) * 2 + 1]; }
    float* o = partials + (((int64_t)n * nsplit + sp) * G + g) * 2;
    o[0] = a; o[1] = b;
  }
}

// one wavefront per (instance, group): lanes stride over the row-chunk partials, f64 accumulate, shuffle reduce
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partials, float* __restrict__ stats, int N, int G,
                                                          int nsplit, double count, float eps) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= N * G) return;
  const int n = i / G, g = i % G;
  double a = 0.0, b = 0.0;
  for (int sp = lane; sp < nsplit; sp += 64) {
    const float* p = partials + (((int64_t)n * nsplit + sp) * G + g) * 2;
    a += (double)p[0]; b += (double)p[1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
  if (lane == 0) {
    double mean = a / count;
    double var = b / count - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[2 * i] = (float)mean;
    stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ stats,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              T* __restrict__ y, int ldy, int64_t S, int C, int G, int nsplit, int silu) {
  constexpr int V = TT<T>::VEC;
  const int n = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
  const int CV = C / V, cpg = C / G;
  const int64_t rows_per = (S + nsplit - 1) / nsplit;
  const int64_t s0 = sp * rows_per;
  int64_t s1 = s0 + rows_per; if (s1 > S) s1 = S;
  const int tid = threadIdx.x;
  const T* xb = x + (int64_t)n * S * ldx;
  T* yb = y + (int64_t)n * S * ldy;
  const float* st = stats + (int64_t)n * G * 2;
  if (CV <= GN_THREADS) {
    const int RP = GN_THREADS / CV, r = tid / CV, cv = tid % CV;
    if (r >= RP) return;
    float a[V], b[V];
#pragma unroll
    for (int e = 0; e < V; e++) {
      int c = cv * V + e, g = c / cpg;
      float mean = st[2 * g], rstd = st[2 * g + 1];
      a[e] = rstd * gamma[c]; b[e] = beta[c] - mean * a[e];
    }
    int64_t s = s0 + r;
    for (; s + RP < s1; s += 2 * RP) {   // 2 rows in flight per thread
      uint4 v0 = *(const uint4*)(xb + s * ldx + cv * V);
      uint4 v1 = *(const uint4*)(xb + (s + RP) * ldx + cv * V);
      float f[V], h[V];
      unpack16<T>(v0, f);
      unpack16<T>(v1, h);
#pragma unroll
      for (int e = 0; e < V; e++) {
        float v = f[e] * a[e] + b[e]; f[e] = silu ? silu_f(v) : v;
        float u = h[e] * a[e] + b[e]; h[e] = silu ? silu_f(u) : u;
      }
      *(uint4*)(yb + s * ldy + cv * V) = pack16<T>(f);
      *(uint4*)(yb + (s + RP) * ldy + cv * V) = pack16<T>(h);
    }
    for (; s < s1; s += RP) {
      float f[V];
      unpack16<T>(*(const uint4*)(xb + s * ldx + cv * V), f);
#pragma unroll
      for (int e = 0; e < V; e++) { float v = f[e] * a[e] + b[e]; f[e] = silu ? silu_f(v) : v; }
      *(uint4*)(yb + s * ldy + cv * V) = pack16<T>(f);
    }
  } else {
    for (int j = 0; j < GN_MAXJ; j++) {
      int cv = tid + j * GN_THREADS;
      if (cv >= CV) break;
      float a[V], b[V];
#pragma unroll
      for (int e = 0; e < V; e++) {
        int c = cv * V + e, g = c / cpg;
        float mean = st[2 * g], rstd = st[2 * g + 1];
        a[e] = rstd * gamma[c]; b[e] = beta[c] - mean * a[e];
      }
      for (int64_t s = s0; s < s1; s++) {
        float f[V];
        unpack16<T>(*(const uint4*)(xb + s * ldx + cv * V), f);
#pragma unroll
        for (int e = 0; e < V; e++) { float v = f[e] * a[e] + b[e]; f[e] = silu ? silu_f(v) : v; }
        *(uint4*)(yb + s * ldy + cv * V) = pack16<T>(f);
      }
    }
  }
}

static int gn_check(const char* who, int N, int64_t S, int C, int G, int ld, int dtype) {
  EMO_CHECK(dtype == EMO_F32 || dtype == EMO_BF16, EMO_ERR_BAD_DTYPE, "%s: dtype %d", who, dtype);
  int V = dtype == EMO_F32 ? 4 : 8;
  EMO_CHECK(N > 0 && S > 0 && C > 0 && G > 0 && C % G == 0, EMO_ERR_BAD_SHAPE, "%s: N=%d S=%lld C=%d G=%d", who, N, (long long)S, C, G);
  EMO_CHECK(C % V == 0 && ld % V == 0 && ld >= C, EMO_ERR_BAD_SHAPE, "%s: C=%d ld=%d must be multiples of %d", who, C, ld, V);
  EMO_CHECK(C / V <= GN_THREADS * GN_MAXJ, EMO_ERR_UNSUPPORTED, "%s: C=%d too wide", who, C);
  return EMO_OK;
}

extern "C" int emo_groupnorm_stats(const void* x, int ldx, float* stats, void* partials, int N, int64_t S, int C, int G, float eps,
                                   int dtype, void* stream) {
  EMO_CHECK(x && stats && partials, EMO_ERR_NULL, "emo_groupnorm_stats: null pointer");
  int rc = gn_check("emo_groupnorm_stats", N, S, C, G, ldx, dtype);
  if (rc) return rc;
  const int nsplit = gn_nsplit(S);
  const int V = dtype == EMO_F32 ? 4 : 8, CV = C / V;
  const int RP = CV <= GN_THREADS ? GN_THREADS / CV : 1;
  const size_t lds = (size_t)RP * C * 2 * sizeof(float);
  EMO_CHECK(lds <= 64 * 1024, EMO_ERR_UNSUPPORTED, "emo_groupnorm_stats: LDS %zu", lds);
  hipStream_t st = as_stream(stream);
  if (dtype == EMO_F32) gn_partial_kernel<float><<<N * nsplit, GN_THREADS, lds, st>>>((const float*)x, ldx, (float*)partials, S, C, G, nsplit);
  else gn_partial_kernel<bf16_t><<<N * nsplit, GN_THREADS, lds, st>>>((const bf16_t*)x, ldx, (float*)partials, S, C, G, nsplit);
  EMO_LAUNCH_CHECK();
  gn_finalize_kernel<<<(N * G + 3) / 4, 256, 0, st>>>((const float*)partials, stats, N, G, nsplit, (double)S * (C / G), eps);
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

extern "C" int emo_groupnorm_apply(const void* x, int ldx, const float* stats, const float* gamma, const float* beta, void* y,
                                   int ldy, int N, int64_t S, int C, int G, int silu, int dtype, void* stream) {
  EMO_CHECK(x && stats && gamma && beta && y, EMO_ERR_NULL, "emo_groupnorm_apply: null pointer");
  int rc = gn_check("emo_groupnorm_apply", N, S, C, G, ldx, dtype);
  if (rc) return rc;
  rc = gn_check("emo_groupnorm_apply", N, S, C, G, ldy, dtype);
  if (rc) return rc;
  // more, smaller chunks than the stats pass: this pass is pure streaming
  int64_t want = (S + 7) / 8; if (want > 1024) want = 1024; if (want < 1) want = 1;
  const int nsplit = (int)want;
  hipStream_t st = as_stream(stream);
  if (dtype == EMO_F32) gn_apply_kernel<float><<<N * nsplit, GN_THREADS, 0, st>>>((const float*)x, ldx, stats, gamma, beta, (float*)y, ldy, S, C, G, nsplit, silu);
  else gn_apply_kernel<bf16_t><<<N * nsplit, GN_THREADS, 0, st>>>((const bf16_t*)x, ldx, stats, gamma, beta, (bf16_t*)y, ldy, S, C, G, nsplit, silu);
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

// ------------------------------------------------------------------------------------------ LayerNorm
// A wavefront normalises 64/LPR rows at once: LPR lanes (a power of two) share a row, each lane holds up to
// LN_MAXV 16-byte vectors of it in registers (lane-strided, so a row group reads contiguous 16*LPR-byte runs).
// Two-pass mean / variance like torch (no E[x^2]-mean^2 cancellation); reductions are log2(LPR) shuffles.
// Optional fused temporal positional-encoding add (motion_module.py:246-248 applied after the norm, :282-283).
static constexpr int LN_MAXV = 5;

template <typename T, int LPR>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ y, int ldy, int64_t M, int C,
                                                        float eps, const float* __restrict__ pe, int rows_per_frame, int frames) {
  constexpr int V = TT<T>::VEC;
  constexpr int RPW = 64 / LPR;  // rows per wavefront
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, rsel = lane / LPR;
  const int CV = C / V;
  const float invC = 1.0f / (float)C;
  for (int64_t m0 = ((int64_t)blockIdx.x * 4 + wave) * RPW; m0 < M; m0 += (int64_t)gridDim.x * 4 * RPW) {
    const int64_t m = m0 + rsel;
    const bool ok = m < M;
    float f[LN_MAXV][V];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; j++) {
      const int cv = sub + LPR * j;
      if (ok && cv < CV) {
        unpack16<T>(*(const uint4*)(x + m * ldx + cv * V), f[j]);
#pragma unroll
        for (int e = 0; e < V; e++) s += f[j][e];
      }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s * invC;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; j++) {
      const int cv = sub + LPR * j;
      if (ok && cv < CV) {
#pragma unroll
        for (int e = 0; e < V; e++) { float d = f[j][e] - mean; q += d * d; }
      }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q * invC + eps);
    const float* pe_row = (pe && ok) ? pe + (int64_t)((m / rows_per_frame) % frames) * C : nullptr;
#pragma unroll
    for (int j = 0; j < LN_MAXV; j++) {
      const int cv = sub + LPR * j;
      if (ok && cv < CV) {
        float g[V], b[V], o[V];
        *(float4*)&g[0] = *(const float4*)(gamma + cv * V);
        *(float4*)&b[0] = *(const float4*)(beta + cv * V);
        if constexpr (V == 8) {
          *(float4*)&g[4] = *(const float4*)(gamma + cv * V + 4);
          *(float4*)&b[4] = *(const float4*)(beta + cv * V + 4);
        }
#pragma unroll
        for (int e = 0; e < V; e++) {
          float v = (f[j][e] - mean) * rstd * g[e] + b[e];
          if (pe_row) {
            // the reference adds pe to the LN output tensor (in compute dtype) => round first in bf16 mode
            if constexpr (sizeof(T) == 2) v = bf2f(f2bf(v));
            v += pe_row[cv * V + e];
          }
          o[e] = v;
        }
        *(uint4*)(y + m * ldy + cv * V) = pack16<T>(o);
      }
    }
  }
}

template <typename T>
static void launch_layernorm(const T* x, int ldx, const float* gamma, const float* beta, T* y, int ldy, int64_t M, int C, float eps,
                             const float* pe, int rpf, int frames, hipStream_t st) {
  const int CV = C / TT<T>::VEC;
  int lpr = 1;
  while (lpr * LN_MAXV < CV) lpr *= 2;
  const int rpw = 64 / lpr;
  int64_t g = (M + 4 * rpw - 1) / (4 * rpw); if (g > 256 * 16) g = 256 * 16;
#define EMO_LN(L) layernorm_kernel<T, L><<<(int)g, 256, 0, st>>>(x, ldx, gamma, beta, y, ldy, M, C, eps, pe, rpf, frames)
  switch (lpr) {
    case 1: EMO_LN(1); break;
    case 2: EMO_LN(2); break;
    case 4: EMO_LN(4); break;
    case 8: EMO_LN(8); break;
    case 16: EMO_LN(16); break;
    case 32: EMO_LN(32); break;
    default: EMO_LN(64); break;
  }
#undef EMO_LN
}

extern "C" int emo_layernorm(const void* x, int ldx, const float* gamma, const float* beta, void* y, int ldy, int64_t M, int C,
                             float eps, const float* pe, int rows_per_frame, int frames, int dtype, void* stream) {
  EMO_CHECK(x && gamma && beta && y, EMO_ERR_NULL, "emo_layernorm: null pointer");
  EMO_CHECK(dtype == EMO_F32 || dtype == EMO_BF16, EMO_ERR_BAD_DTYPE, "emo_layernorm: dtype %d", dtype);
  const int V = dtype == EMO_F32 ? 4 : 8;
  EMO_CHECK(M > 0 && C > 0 && C % V == 0 && ldx % V == 0 && ldy % V == 0 && ldx >= C && ldy >= C, EMO_ERR_BAD_SHAPE,
            "emo_layernorm: M=%lld C=%d ldx=%d ldy=%d", (long long)M, C, ldx, ldy);
  EMO_CHECK(C / V <= 64 * LN_MAXV, EMO_ERR_UNSUPPORTED, "emo_layernorm: C=%d too wide", C);
  EMO_CHECK(!pe || (rows_per_frame > 0 && frames > 0), EMO_ERR_BAD_SHAPE, "emo_layernorm: pe needs rows_per_frame/frames");
  EMO_CHECK(((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0, EMO_ERR_BAD_SHAPE, "emo_layernorm: gamma/beta alignment");
  hipStream_t st = as_stream(stream);
  if (dtype == EMO_F32) launch_layernorm<float>((const float*)x, ldx, gamma, beta, (float*)y, ldy, M, C, eps, pe, rows_per_frame, frames, st);
  else launch_layernorm<bf16_t>((const bf16_t*)x, ldx, gamma, beta, (bf16_t*)y, ldy, M, C, eps, pe, rows_per_frame, frames, st);
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}
