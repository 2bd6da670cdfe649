// norm.hip - GroupNorm (joint 5-D or per-frame statistics) + SiLU, and LayerNorm (+ temporal PE add)
// over NHWC rows.  HBM-bound kernels: every global access is a 16-byte vector on a fully coalesced row.
//
// GroupNorm: reference semantics are nn.GroupNorm applied to a 5-D tensor in the resnets
// (resnet.py:180,191 -> statistics joint over C/G x F x H x W) and to per-frame 4-D tensors in the
// transformers (attention.py:124, motion_module.py:147).  Both are "instances of S consecutive rows".
//   pass 1  gn_stats_kernel : block = (instance n, row-chunk); per-channel sum / sumsq in registers,
//                             fixed-order LDS reduction -> partial (sum, sumsq) per (chunk, group)  (deterministic)
//   pass 2  gn_apply_kernel : prologue: every block combines the <= 256 chunk partials of its instance in f64, in the same
//                             fixed order -> (mean, rstd) in LDS (no third launch, no inter-block fence: an agent-scope
//                             release on this part writes back the whole L2);  y = (x-mean)*rstd*gamma+beta, optional SiLU
#include "common.h"

static constexpr int GN_THREADS = 256;
static constexpr int GN_MAXJ = 3;  // column vectors per thread: C <= 256*3*VEC

// Geometry shared by both passes.  Channels wider than one block's 256 column vectors are cut into NC column parts
// of whole groups (C=2560: two parts of 16 groups), each part handled by its own blocks (blockIdx.y).  A block holds
// RP = 256 / (column vectors per part) row slots; a thread streams 4-16 rows (more once the block budget binds).
struct GnGeom { int NC, Cp, Gp, RP, nsplit_stats, nsplit_apply; };
static inline GnGeom gn_geom(int N, int64_t S, int C, int G, int V) {
  GnGeom g;
  g.NC = 1;
  while ((C / g.NC) / V > GN_THREADS && g.NC < 8 && G % (g.NC * 2) == 0 && (C / (g.NC * 2)) % V == 0) g.NC *= 2;
  g.Cp = C / g.NC; g.Gp = G / g.NC;
  const int CVp = g.Cp / V;
  g.RP = CVp <= GN_THREADS ? GN_THREADS / CVp : 1;
  // 16 rows per thread; tensors too small to give every CU a block that way go down to 8 / 4 rows per thread (shorter
  // dependent-load chains).  <= 1024 blocks per launch (the apply pass pays its statistics prologue per block) and
  // <= 256 chunk partials per instance (every apply block re-reduces them).
  auto chunks = [&](int64_t hard_cap) {
    int64_t n = 1;
    for (int rpt = 16; rpt >= 4; rpt >>= 1) {
      const int64_t rows_per_block = (int64_t)g.RP * rpt;
      n = (S + rows_per_block - 1) / rows_per_block;
      if (n * N * g.NC >= 256) break;
    }
    int64_t want = 1024 / ((int64_t)N * g.NC);
    if (want < 1) want = 1;
    if (n > want) n = want;
    if (n > hard_cap) n = hard_cap;
    if (n < 1) n = 1;
    return (int)n;
  };
  g.nsplit_stats = chunks(256);
  g.nsplit_apply = chunks(1 << 20);
  return g;
}
extern "C" size_t emo_groupnorm_workspace_bytes(int N, int64_t S, int C, int G) {
  // the chunk count depends on the element width through RP; size for the larger of the two
  const GnGeom a = gn_geom(N, S, C, G, 8), b = gn_geom(N, S, C, G, 4);
  const int ns = a.nsplit_stats > b.nsplit_stats ? a.nsplit_stats : b.nsplit_stats;
  return (size_t)N * ns * G * 2 * sizeof(float);
}

// pass 1: per-(instance, row chunk) group sums
template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const T* __restrict__ x, int ldx, float* __restrict__ partials,
                                                              int64_t S, int C, int G, int G_all, int nsplit) {
  // C, G: this column part's channels / groups (blockIdx.y selects the part); G_all: groups of the whole tensor
  constexpr int V = TT<T>::VEC;
  extern __shared__ float lds[];  // [RP][C][2]
  const int n = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
  x += (int64_t)blockIdx.y * C;
  const int CV = C / V;
  const int64_t rows_per = (S + nsplit - 1) / nsplit;
  const int64_t s0 = sp * rows_per;
  int64_t s1 = s0 + rows_per; if (s1 > S) s1 = S;
  const int tid = threadIdx.x;
  const T* base = x + (int64_t)n * S * ldx;
  int RP;   // row slots in the block
  if (CV <= GN_THREADS) {
    RP = GN_THREADS / CV;
    const int r = tid / CV, cv = tid % CV;
    if (r < RP) {
      float sum[V], sq[V];
#pragma unroll
      for (int e = 0; e < V; e++) { sum[e] = 0.f; sq[e] = 0.f; }
      const T* col = base + cv * V;
      int64_t s = s0 + r;
      for (; s + 7 * RP < s1; s += 8 * RP) {   // 8 independent 16-byte loads in flight per thread
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = *(const uint4*)(col + (s + u * RP) * ldx);
#pragma unroll
        for (int u = 0; u < 8; u++) {
          float f[V];
          unpack16<T>(v[u], f);
#pragma unroll
          for (int e = 0; e < V; e++) { sum[e] += f[e]; sq[e] += f[e] * f[e]; }
        }
      }
      for (; s < s1; s += RP) {
        float f[V];
        unpack16<T>(*(const uint4*)(col + s * ldx), f);
#pragma unroll
        for (int e = 0; e < V; e++) { sum[e] += f[e]; sq[e] += f[e] * f[e]; }
      }
      float* dst = lds + ((int64_t)r * C + cv * V) * 2;
#pragma unroll
      for (int e = 0; e < V; e++) { dst[2 * e] = sum[e]; dst[2 * e + 1] = sq[e]; }
    }
  } else {
    RP = 1;
    float sum[GN_MAXJ][V], sq[GN_MAXJ][V];
#pragma unroll
    for (int j = 0; j < GN_MAXJ; j++)
#pragma unroll
      for (int e = 0; e < V; e++) { sum[j][e] = 0.f; sq[j][e] = 0.f; }
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
  // 2G outputs (sum, sumsq per group), TPO adjacent lanes each: fixed-order strided partial sums + shuffle tree
  const int cpg = C / G;
  int TPO = 1;
  while (TPO * 2 * (2 * G) <= GN_THREADS && TPO < 64) TPO *= 2;
  float* pout = partials + (((int64_t)n * nsplit + sp) * G_all + (int64_t)blockIdx.y * G) * 2;
  for (int o0 = 0; o0 < 2 * G; o0 += GN_THREADS / TPO) {
    const int o = o0 + tid / TPO, part = tid % TPO;
    float a = 0.f;
    if (o < 2 * G) {
      const int g = o >> 1, which = o & 1, ne = RP * cpg;
      for (int e = part; e < ne; e += TPO) {
        const int r = e / cpg, c = g * cpg + e % cpg;
        a += lds[((int64_t)r * C + c) * 2 + which];
      }
    }
    for (int d = TPO / 2; d > 0; d >>= 1) a += __shfl_xor(a, d, 64);
    if (o < 2 * G && part == 0) pout[o] = a;
  }
}

static constexpr int GN_MAXG = 128;   // groups held in the apply pass's LDS statistics table

template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ partials,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              T* __restrict__ y, int ldy, int64_t S, int C, int G, int G_all,
                                                              int nsplit_stats, int nsplit, double count, float eps, int silu) {
  constexpr int V = TT<T>::VEC;
  __shared__ float s_stats[GN_MAXG * 2];
  const int n = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
  x += (int64_t)blockIdx.y * C; y += (int64_t)blockIdx.y * C;
  gamma += (int64_t)blockIdx.y * C; beta += (int64_t)blockIdx.y * C;
  const int CV = C / V, cpg = C / G;
  const int64_t rows_per = (S + nsplit - 1) / nsplit;
  const int64_t s0 = sp * rows_per;
  int64_t s1 = s0 + rows_per; if (s1 > S) s1 = S;
  const int tid = threadIdx.x;
  const T* xb = x + (int64_t)n * S * ldx;
  T* yb = y + (int64_t)n * S * ldy;
  // the loads that do not depend on the statistics are issued BEFORE the statistics prologue: gamma / beta and the first
  // 4 rows of this thread (one memory round trip overlapped instead of three in a row)
  const bool narrow = CV <= GN_THREADS;
  const int RP = narrow ? GN_THREADS / CV : 1, r = narrow ? tid / CV : 0, cv = narrow ? tid % CV : 0;
  const bool active = narrow && r < RP;
  const T* xc = xb + cv * V;
  T* yc = yb + cv * V;
  float gm[V], bt[V];
  uint4 v[4];
  int64_t s = s0 + r;
  bool have = active && s + 3 * RP < s1;
  if (active) {
#pragma unroll
    for (int e = 0; e < V; e += 4) {
      *(float4*)(gm + e) = *(const float4*)(gamma + cv * V + e);
      *(float4*)(bt + e) = *(const float4*)(beta + cv * V + e);
    }
  }
  if (have) {
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = *(const uint4*)(xc + (s + u * RP) * ldx);
  }
  // ---- prologue: (mean, rstd) of this instance's groups from the chunk partials: TPG adjacent lanes per group, each a
  // strided f64 sum, then a shuffle tree - the same order in every block
  {
    int TPG = 1;
    while (TPG * 2 * G <= GN_THREADS && TPG < 64) TPG *= 2;
    const float* pin = partials + ((int64_t)n * nsplit_stats * G_all + (int64_t)blockIdx.y * G) * 2;
    for (int g0 = 0; g0 < G; g0 += GN_THREADS / TPG) {
      const int g = g0 + tid / TPG, part = tid % TPG;
      double a = 0.0, b = 0.0;
      if (g < G)
        for (int q = part; q < nsplit_stats; q += TPG) {
          const float2 pv = *(const float2*)(pin + ((int64_t)q * G_all + g) * 2);
          a += (double)pv.x; b += (double)pv.y;
        }
      for (int d = TPG / 2; d > 0; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
      if (g < G && part == 0) {
        const double mean = a / count;
        double var = b / count - mean * mean;
        if (var < 0.0) var = 0.0;
        s_stats[2 * g] = (float)mean;
        s_stats[2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
      }
    }
    __syncthreads();
  }
  if (narrow) {
    if (!active) return;
    float a[V], b[V];
#pragma unroll
    for (int e = 0; e < V; e++) {
      const int g = (cv * V + e) / cpg;
      const float mean = s_stats[2 * g], rstd = s_stats[2 * g + 1];
      a[e] = rstd * gm[e]; b[e] = fmaf(-mean, a[e], bt[e]);
    }
    while (have) {   // the next 4 rows are requested before the current 4 are normalised and stored
      const int64_t sn = s + 4 * RP;
      const bool hn = sn + 3 * RP < s1;
      uint4 w[4];
      if (hn) {
#pragma unroll
        for (int u = 0; u < 4; u++) w[u] = *(const uint4*)(xc + (sn + u * RP) * ldx);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        float f[V];
        unpack16<T>(v[u], f);
#pragma unroll
        for (int e = 0; e < V; e++) { const float t = f[e] * a[e] + b[e]; f[e] = silu ? silu_f(t) : t; }
        *(uint4*)(yc + (s + u * RP) * ldy) = pack16<T>(f);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = w[u];
      s = sn; have = hn;
    }
    for (; s < s1; s += RP) {
      float f[V];
      unpack16<T>(*(const uint4*)(xc + s * ldx), f);
#pragma unroll
      for (int e = 0; e < V; e++) { const float t = f[e] * a[e] + b[e]; f[e] = silu ? silu_f(t) : t; }
      *(uint4*)(yc + s * ldy) = pack16<T>(f);
    }
  } else {
    for (int j = 0; j < GN_MAXJ; j++) {
      int cv = tid + j * GN_THREADS;
      if (cv >= CV) break;
      float a[V], b[V];
#pragma unroll
      for (int e = 0; e < V; e++) {
        int c = cv * V + e, g = c / cpg;
        float mean = s_stats[2 * g], rstd = s_stats[2 * g + 1];
        a[e] = rstd * gamma[c]; b[e] = fmaf(-mean, a[e], beta[c]);
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
  EMO_CHECK(emo_dtype_ok(dtype), EMO_ERR_BAD_DTYPE, "%s: dtype %d", who, dtype);
  int V = emo_dtype_vec(dtype);
  EMO_CHECK(N > 0 && S > 0 && C > 0 && G > 0 && C % G == 0, EMO_ERR_BAD_SHAPE, "%s: N=%d S=%lld C=%d G=%d", who, N, (long long)S, C, G);
  EMO_CHECK(C % V == 0 && ld % V == 0 && ld >= C, EMO_ERR_BAD_SHAPE, "%s: C=%d ld=%d must be multiples of %d", who, C, ld, V);
  EMO_CHECK(C / V <= GN_THREADS * GN_MAXJ, EMO_ERR_UNSUPPORTED, "%s: C=%d too wide", who, C);
  EMO_CHECK(G <= GN_MAXG, EMO_ERR_UNSUPPORTED, "%s: G=%d groups (max %d)", who, G, GN_MAXG);
  return EMO_OK;
}

extern "C" int emo_groupnorm_stats(const void* x, int ldx, void* partials, int N, int64_t S, int C, int G, int dtype, void* stream) {
  EMO_CHECK(x && partials, EMO_ERR_NULL, "emo_groupnorm_stats: null pointer");
  int rc = gn_check("emo_groupnorm_stats", N, S, C, G, ldx, dtype);
  if (rc) return rc;
  const GnGeom gg = gn_geom(N, S, C, G, emo_dtype_vec(dtype));
  const size_t lds = (size_t)gg.RP * gg.Cp * 2 * sizeof(float);
  EMO_CHECK(lds <= 64 * 1024, EMO_ERR_UNSUPPORTED, "emo_groupnorm_stats: LDS %zu", lds);
  hipStream_t st = as_stream(stream);
  const dim3 grid((unsigned)(N * gg.nsplit_stats), (unsigned)gg.NC);
  EMO_DISPATCH(dtype, "emo_groupnorm_stats", (gn_stats_kernel<T><<<grid, GN_THREADS, lds, st>>>((const T*)x, ldx, (float*)partials, S, gg.Cp, gg.Gp, G, gg.nsplit_stats)));
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

extern "C" int emo_groupnorm_apply(const void* x, int ldx, const void* partials, const float* gamma, const float* beta, void* y,
                                   int ldy, int N, int64_t S, int C, int G, float eps, int silu, int dtype, void* stream) {
  EMO_CHECK(x && partials && gamma && beta && y, EMO_ERR_NULL, "emo_groupnorm_apply: null pointer");
  int rc = gn_check("emo_groupnorm_apply", N, S, C, G, ldx, dtype);
  if (rc) return rc;
  rc = gn_check("emo_groupnorm_apply", N, S, C, G, ldy, dtype);
  if (rc) return rc;
  EMO_CHECK(((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0, EMO_ERR_BAD_SHAPE, "emo_groupnorm_apply: gamma/beta alignment");
  const GnGeom gg = gn_geom(N, S, C, G, emo_dtype_vec(dtype));
  const double count = (double)S * (C / G);
  hipStream_t st = as_stream(stream);
  const dim3 grid((unsigned)(N * gg.nsplit_apply), (unsigned)gg.NC);
  EMO_DISPATCH(dtype, "emo_groupnorm_apply",
               (gn_apply_kernel<T><<<grid, GN_THREADS, 0, st>>>((const T*)x, ldx, (const float*)partials, gamma, beta, (T*)y, ldy, S, gg.Cp, gg.Gp, G,
                                                                gg.nsplit_stats, gg.nsplit_apply, count, eps, silu)));
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

// The coefficient half of the apply pass for consumers that normalise on the fly (the halo conv, emo_gemm_params.gn_coef): one
// block per instance runs the SAME statistics prologue as gn_apply_kernel - per column part, TPG lanes per group, strided f64
// sums + shuffle tree - and writes scale = rstd * gamma, shift = beta - mean * scale, channel pairs interleaved (s, s, b, b).
__global__ __launch_bounds__(GN_THREADS) void gn_coeffs_kernel(const float* __restrict__ partials, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ coef,
                                                               int C, int G_all, int NC, int nsplit_stats, double count, float eps) {
  __shared__ float s_stats[GN_MAXG * 2];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int G = G_all / NC;
  int TPG = 1;
  while (TPG * 2 * G <= GN_THREADS && TPG < 64) TPG *= 2;
  for (int y = 0; y < NC; y++) {
    const float* pin = partials + ((int64_t)n * nsplit_stats * G_all + (int64_t)y * G) * 2;
    for (int g0 = 0; g0 < G; g0 += GN_THREADS / TPG) {
      const int g = g0 + tid / TPG, part = tid % TPG;
      double a = 0.0, b = 0.0;
      if (g < G)
        for (int q = part; q < nsplit_stats; q += TPG) {
          const float2 pv = *(const float2*)(pin + ((int64_t)q * G_all + g) * 2);
          a += (double)pv.x; b += (double)pv.y;
        }
      for (int d = TPG / 2; d > 0; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
      if (g < G && part == 0) {
        const double mean = a / count;
        double var = b / count - mean * mean;
        if (var < 0.0) var = 0.0;
        s_stats[2 * (y * G + g)] = (float)mean;
        s_stats[2 * (y * G + g) + 1] = (float)(1.0 / sqrt(var + (double)eps));
      }
    }
  }
  __syncthreads();
  const int cpg = C / G_all;
  for (int c = tid; c < C; c += GN_THREADS) {
    const int g = c / cpg;
    const float mean = s_stats[2 * g], rstd = s_stats[2 * g + 1];
    const float a = rstd * gamma[c];
    float* dst = coef + (int64_t)n * 2 * C + (c >> 1) * 4 + (c & 1);
    dst[0] = a;
    dst[2] = fmaf(-mean, a, beta[c]);   // (the very expression of the apply kernels: one fma, whatever the compiler would contract)
  }
}

extern "C" int emo_groupnorm_coeffs(const void* partials, const float* gamma, const float* beta, float* coef, int N,
                                    int64_t S, int C, int G, float eps, int dtype, void* stream) {
  EMO_CHECK(partials && gamma && beta && coef, EMO_ERR_NULL, "emo_groupnorm_coeffs: null pointer");
  int rc = gn_check("emo_groupnorm_coeffs", N, S, C, G, C, dtype);
  if (rc) return rc;
  const GnGeom gg = gn_geom(N, S, C, G, emo_dtype_vec(dtype));
  const double count = (double)S * (C / G);
  gn_coeffs_kernel<<<(unsigned)N, GN_THREADS, 0, as_stream(stream)>>>((const float*)partials, gamma, beta, coef, C, G, gg.NC,
                                                                      gg.nsplit_stats, count, eps);
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

// ------------------------------------------------------------------------------------------ GroupNorm in one launch
// Instances small enough that one block holds (instance, slab of whole groups) in registers: one read, statistics through
// LDS in the same fixed order as the two-pass path (f32 per-thread partials, f64 mean / variance), normalise from the
// registers, one write.  The 8x8 / 16x16 / 32x32 levels' norms are all of this kind (M <= 24576 rows per-frame or
// M <= 6144 joint): two dependent launches and a second read of x become one launch.
static constexpr int GN1_MAXR = 16, GN1_MAXT = 1024, GN1_MAXGPB = 16, GN1_MINBLOCKS = 32, GN1_MAXELEMS = 32768;
struct Gn1Geom { int ok, GPB, Wc, NT, R; };
static inline Gn1Geom gn1_geom(int N, int64_t S, int C, int G, int V) {
  Gn1Geom g{0, 0, 0, 0, 0};
  if (N <= 0 || S <= 0 || G <= 0 || C % G) return g;
  const int cpg = C / G;
  int gpb = 1;
  while (gpb <= GN1_MAXGPB && ((gpb * cpg) % V || G % gpb)) gpb++;
  if (gpb > GN1_MAXGPB) return g;
  g.GPB = gpb; g.Wc = gpb * cpg;
  const int CVW = g.Wc / V;
  if (CVW > 256 || (int64_t)N * (G / gpb) < GN1_MINBLOCKS) return g;
  // measured (profiles/r04n_groupnorm_one_launch.txt): a slab is Wc * sizeof(T) contiguous bytes per row, so a block's loads are
  // short row segments; past ~32 K elements per block the two coalesced passes are faster again.  (Wider slabs - 160 / 320 bytes per
  // row, as many rows as 16 x 1024 threads hold - measured 23.4 vs 27.5 us at N=24 S=1024 C=640 and 14.3 vs 16.3 at N=24 S=256 C=1280,
  // 21.7 vs 19.3 at C=320: +-0.05 ms per step, not taken; 24 rows per thread spill.)
  if (S * g.Wc > GN1_MAXELEMS) return g;
  for (int nt = 256; nt <= GN1_MAXT; nt *= 2) {   // short load chains: <= 4 rows per thread if any block size gives that
    const int64_t rows = (S + nt / CVW - 1) / (nt / CVW);
    const int64_t lim = nt == GN1_MAXT ? GN1_MAXR : 4;
    if (rows <= lim) {
      // (the DYNAMIC LDS [RP][Wc][2] floats must fit the 64 KB a launch gets without hipFuncAttributeMaxDynamicSharedMemorySize; the 2 KB of
      // static tables come on top - 65280 + 2176 bytes at N = 2, S = 768, C = 1280 is a geometry of the bench and launches)
      if ((size_t)(nt / CVW) * g.Wc * 2 * sizeof(float) > 64 * 1024) return g;
      g.NT = nt; g.R = rows <= 2 ? 2 : rows <= 4 ? 4 : rows <= 8 ? 8 : 16; g.ok = 1;
      return g;
    }
  }
  return g;
}

template <typename T, int R>
__global__ __launch_bounds__(GN1_MAXT) void gn_one_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, T* __restrict__ y, int ldy, int S, int Wc,
                                                          int GPB, int slabs, double count, float eps, int silu) {
  constexpr int V = TT<T>::VEC;
  extern __shared__ float lds[];   // [RP][Wc][2]
  __shared__ float s_stats[GN1_MAXGPB * 2];
  __shared__ float s_wave[GN1_MAXGPB * 16 * 2];
  const int n = blockIdx.x / slabs, sl = blockIdx.x % slabs;
  const int CVW = Wc / V, NT = blockDim.x, RP = NT / CVW;
  const int tid = threadIdx.x, r = tid / CVW, cv = tid % CVW;
  const bool active = r < RP;
  const int c0 = sl * Wc + cv * V;
  const T* xc = x + (int64_t)n * S * ldx + c0;
  T* yc = y + (int64_t)n * S * ldy + c0;
  uint4 v[R];
  float gm[V], bt[V];
  if (active) {
#pragma unroll
    for (int k = 0; k < R; k++) {
      const int s = r + k * RP;
      v[k] = s < S ? *(const uint4*)(xc + (int64_t)s * ldx) : make_uint4(0u, 0u, 0u, 0u);   // zeros add nothing to the sums
    }
#pragma unroll
    for (int e = 0; e < V; e += 4) {
      *(float4*)(gm + e) = *(const float4*)(gamma + c0 + e);
      *(float4*)(bt + e) = *(const float4*)(beta + c0 + e);
    }
    float sum[V], sq[V];
#pragma unroll
    for (int e = 0; e < V; e++) { sum[e] = 0.f; sq[e] = 0.f; }
#pragma unroll
    for (int k = 0; k < R; k++) {
      float f[V];
      unpack16<T>(v[k], f);
#pragma unroll
      for (int e = 0; e < V; e++) { sum[e] += f[e]; sq[e] += f[e] * f[e]; }
    }
    float* dst = lds + ((int64_t)r * Wc + cv * V) * 2;
#pragma unroll
    for (int e = 0; e < V; e += 2) *(float4*)(dst + 2 * e) = make_float4(sum[e], sq[e], sum[e + 1], sq[e + 1]);
  }
  __syncthreads();
  // per group: every thread sums a strided share of the [RP][cpg] partials, shuffle tree per wave, wave results through
  // LDS, one thread finishes in f64 - the same order on every launch
  {
    const int cpg = Wc / GPB, ne = RP * cpg, nw = NT / 64, wv = tid / 64, ln = tid % 64;
    for (int g = 0; g < GPB; g++) {
      float a = 0.f, b = 0.f;
      for (int e = tid; e < ne; e += NT) {
        const int rr = e / cpg, c = g * cpg + e % cpg;
        const float2 p = *(const float2*)(lds + ((int64_t)rr * Wc + c) * 2);
        a += p.x; b += p.y;
      }
      for (int d = 32; d > 0; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
      if (ln == 0) { s_wave[(g * 16 + wv) * 2] = a; s_wave[(g * 16 + wv) * 2 + 1] = b; }
    }
    __syncthreads();
    if (tid < GPB) {
      double sa = 0.0, sb = 0.0;
      for (int w = 0; w < nw; w++) { sa += (double)s_wave[(tid * 16 + w) * 2]; sb += (double)s_wave[(tid * 16 + w) * 2 + 1]; }
      const double mean = sa / count;
      double var = sb / count - mean * mean;
      if (var < 0.0) var = 0.0;
      s_stats[2 * tid] = (float)mean;
      s_stats[2 * tid + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
  __syncthreads();
  if (!active) return;
  float a[V], b[V];
  {
    const int cpg = Wc / GPB;
#pragma unroll
    for (int e = 0; e < V; e++) {
      const int g = (cv * V + e) / cpg;
      const float mean = s_stats[2 * g], rstd = s_stats[2 * g + 1];
      a[e] = rstd * gm[e]; b[e] = fmaf(-mean, a[e], bt[e]);
    }
  }
#pragma unroll
  for (int k = 0; k < R; k++) {
    const int s = r + k * RP;
    if (s < S) {
      float f[V];
      unpack16<T>(v[k], f);
#pragma unroll
      for (int e = 0; e < V; e++) { const float t = f[e] * a[e] + b[e]; f[e] = silu ? silu_f(t) : t; }
      *(uint4*)(yc + (int64_t)s * ldy) = pack16<T>(f);
    }
  }
}

extern "C" int emo_groupnorm_one_launch_ok(int N, int64_t S, int C, int G, int dtype) {
  if (!emo_dtype_ok(dtype)) return 0;
  const int V = emo_dtype_vec(dtype);
  if (N <= 0 || S <= 0 || C <= 0 || G <= 0 || C % G || C % V || C / V > GN_THREADS * GN_MAXJ || G > GN_MAXG) return 0;   // (gn_check's limits)
  return gn1_geom(N, S, C, G, V).ok;
}

extern "C" int emo_groupnorm(const void* x, int ldx, const float* gamma, const float* beta, void* y, int ldy, int N, int64_t S, int C,
                             int G, float eps, int silu, int dtype, void* stream) {
  EMO_CHECK(x && gamma && beta && y, EMO_ERR_NULL, "emo_groupnorm: null pointer");
  int rc = gn_check("emo_groupnorm", N, S, C, G, ldx, dtype);
  if (rc) return rc;
  rc = gn_check("emo_groupnorm", N, S, C, G, ldy, dtype);
  if (rc) return rc;
  EMO_CHECK(((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0, EMO_ERR_BAD_SHAPE, "emo_groupnorm: gamma/beta alignment");
  const int V = emo_dtype_vec(dtype);
  const Gn1Geom gg = gn1_geom(N, S, C, G, V);
  EMO_CHECK(gg.ok, EMO_ERR_UNSUPPORTED, "emo_groupnorm: N=%d S=%lld C=%d G=%d does not fit one block per (instance, group slab); "
            "use emo_groupnorm_stats + emo_groupnorm_apply", N, (long long)S, C, G);
  const int RP = gg.NT / (gg.Wc / V);
  const size_t lds = (size_t)RP * gg.Wc * 2 * sizeof(float);
  const int slabs = G / gg.GPB;
  const double count = (double)S * (C / G);
  hipStream_t st = as_stream(stream);
  const dim3 grid((unsigned)(N * slabs));
#define GN1_LAUNCH(RR) \
  EMO_DISPATCH(dtype, "emo_groupnorm", (gn_one_kernel<T, RR><<<grid, gg.NT, lds, st>>>((const T*)x, ldx, gamma, beta, (T*)y, ldy, (int)S, \
                                                                                       gg.Wc, gg.GPB, slabs, count, eps, silu)))
  if (gg.R == 2) { GN1_LAUNCH(2); }
  else if (gg.R == 4) { GN1_LAUNCH(4); }
  else if (gg.R == 8) { GN1_LAUNCH(8); }
  else { GN1_LAUNCH(16); }
#undef GN1_LAUNCH
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

// ------------------------------------------------------------------------------------------ GroupNorm folded into a Linear
// GN(x) W^T + b = x W'_n^T + b'_n per instance n (emo_hip.h emo_groupnorm_fold_linear): the per-frame GroupNorm in front of
// proj_in (attention.py:124,135-146; motion_module.py:147-151) has no activation behind it, so its scale / shift move into a
// per-instance copy of the weights - N x Cout x C elements (4.9 MB at the 64x64 level against 2 x 63 MB of activation traffic
// for the apply pass it replaces).  block = (instance, 8 output rows); a wave per row, lanes over 16-byte column vectors.
static constexpr int GNF_ROWS = 8, GNF_MAXC = 2560;   // (8: two rows per wave - the kernel is a chain of dependent loads, not bandwidth)

template <typename T>
__global__ __launch_bounds__(256) void gn_fold_linear_kernel(const float* __restrict__ partials, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const T* __restrict__ W,
                                                             const float* __restrict__ bias, T* __restrict__ w_out,
                                                             float* __restrict__ rowbias, int C, int G, int nsplit_stats, double count,
                                                             float eps, int Cout) {
  constexpr int V = TT<T>::VEC;
  __shared__ float s_stats[GN_MAXG * 2];
  __shared__ float s_scale[GNF_MAXC], s_mean[GNF_MAXC], s_beta[GNF_MAXC];
  const int n = blockIdx.x, o0 = blockIdx.y * GNF_ROWS, tid = threadIdx.x, cpg = C / G;
  {   // (mean, rstd) per group: the apply pass's combine - TPG adjacent lanes per group, strided f64 sums, shuffle tree
    int TPG = 1;
    while (TPG * 2 * G <= 256 && TPG < 64) TPG *= 2;
    const float* pin = partials + (int64_t)n * nsplit_stats * G * 2;
    for (int g0 = 0; g0 < G; g0 += 256 / TPG) {
      const int g = g0 + tid / TPG, part = tid % TPG;
      double a = 0.0, b = 0.0;
      if (g < G)
        for (int q = part; q < nsplit_stats; q += TPG) {
          const float2 pv = *(const float2*)(pin + ((int64_t)q * G + g) * 2);
          a += (double)pv.x; b += (double)pv.y;
        }
      for (int d = TPG / 2; d > 0; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
      if (g < G && part == 0) {
        const double mean = a / count;
        double var = b / count - mean * mean;
        if (var < 0.0) var = 0.0;
        s_stats[2 * g] = (float)mean;
        s_stats[2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
      }
    }
    __syncthreads();
  }
  for (int c = tid; c < C; c += 256) {
    const int g = c / cpg;
    s_scale[c] = gamma[c] * s_stats[2 * g + 1];
    s_mean[c] = s_stats[2 * g];
    s_beta[c] = beta[c];
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  for (int o = o0 + wave; o < o0 + GNF_ROWS && o < Cout; o += 4) {
    const T* wr = W + (int64_t)o * C;
    T* wo = w_out + ((int64_t)n * Cout + o) * C;
    float acc = 0.f;
    for (int c = lane * V; c < C; c += 64 * V) {
      float f[V];
      unpack16<T>(*(const uint4*)(wr + c), f);
#pragma unroll
      for (int e = 0; e < V; e++) {
        const float t = round_through<T>(f[e] * s_scale[c + e]);   // the value the GEMM multiplies by
        acc += f[e] * s_beta[c + e] - s_mean[c + e] * t;
        f[e] = t;
      }
      *(uint4*)(wo + c) = pack16<T>(f);
    }
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if (lane == 0) rowbias[(int64_t)n * Cout + o] = acc + (bias ? bias[o] : 0.f);
  }
}

extern "C" int emo_groupnorm_fold_linear(const void* partials, const float* gamma, const float* beta, const void* W, const float* bias,
                                         void* w_out, float* rowbias_out, int N, int64_t S, int C, int G, int Cout, float eps, int dtype,
                                         void* stream) {
  EMO_CHECK(partials && gamma && beta && W && w_out && rowbias_out, EMO_ERR_NULL, "emo_groupnorm_fold_linear: null pointer");
  int rc = gn_check("emo_groupnorm_fold_linear", N, S, C, G, C, dtype);
  if (rc) return rc;
  EMO_CHECK(Cout > 0 && C <= GNF_MAXC, EMO_ERR_UNSUPPORTED, "emo_groupnorm_fold_linear: C=%d Cout=%d", C, Cout);
  EMO_CHECK(((uintptr_t)W % 16) == 0 && ((uintptr_t)w_out % 16) == 0, EMO_ERR_BAD_SHAPE, "emo_groupnorm_fold_linear: W alignment");
  const GnGeom gg = gn_geom(N, S, C, G, emo_dtype_vec(dtype));
  const double count = (double)S * (C / G);
  const dim3 grid((unsigned)N, (unsigned)((Cout + GNF_ROWS - 1) / GNF_ROWS));
  hipStream_t st = as_stream(stream);
  EMO_DISPATCH(dtype, "emo_groupnorm_fold_linear",
               (gn_fold_linear_kernel<T><<<grid, 256, 0, st>>>((const float*)partials, gamma, beta, (const T*)W, bias, (T*)w_out, rowbias_out, C, G,
                                                              gg.nsplit_stats, count, eps, Cout)));
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
            v = round_through<T>(v);
            v += pe_row[cv * V + e];
          }
          o[e] = v;
        }
        *(uint4*)(y + m * ldy + cv * V) = pack16<T>(o);
      }
    }
  }
}

// Statistics half of the LayerNorm for the GEMM fold: stats[m] = (mean, rstd), same two-pass arithmetic and lane mapping.
template <typename T, int LPR>
__global__ __launch_bounds__(256) void layernorm_stats_kernel(const T* __restrict__ x, int ldx, float* __restrict__ stats, int64_t M, int C,
                                                              float eps) {
  constexpr int V = TT<T>::VEC;
  constexpr int RPW = 64 / LPR;
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
    if (ok && sub == 0) *(float2*)(stats + 2 * m) = make_float2(mean, rsqrtf(q * invC + eps));
  }
}

extern "C" int emo_layernorm_stats(const void* x, int ldx, float* stats, int64_t M, int C, float eps, int dtype, void* stream) {
  EMO_CHECK(x && stats, EMO_ERR_NULL, "emo_layernorm_stats: null pointer");
  EMO_CHECK(emo_dtype_ok(dtype), EMO_ERR_BAD_DTYPE, "emo_layernorm_stats: dtype %d", dtype);
  const int V = emo_dtype_vec(dtype);
  EMO_CHECK(M > 0 && C > 0 && C % V == 0 && ldx % V == 0 && ldx >= C, EMO_ERR_BAD_SHAPE, "emo_layernorm_stats: M=%lld C=%d ldx=%d", (long long)M, C, ldx);
  EMO_CHECK(C / V <= 64 * LN_MAXV, EMO_ERR_UNSUPPORTED, "emo_layernorm_stats: C=%d too wide", C);
  EMO_CHECK(((uintptr_t)stats % 8) == 0, EMO_ERR_BAD_SHAPE, "emo_layernorm_stats: stats alignment");
  hipStream_t st = as_stream(stream);
  const int CV = C / V;
  int lpr = 1;
  while (lpr * LN_MAXV < CV) lpr *= 2;
  const int rpw = 64 / lpr;
  int64_t g = (M + 4 * rpw - 1) / (4 * rpw); if (g > 256 * 16) g = 256 * 16;
#define EMO_LNS(L) EMO_DISPATCH(dtype, "emo_layernorm_stats", (layernorm_stats_kernel<T, L><<<(int)g, 256, 0, st>>>((const T*)x, ldx, stats, M, C, eps)))
  switch (lpr) {
    case 1: EMO_LNS(1); break;
    case 2: EMO_LNS(2); break;
    case 4: EMO_LNS(4); break;
    case 8: EMO_LNS(8); break;
    case 16: EMO_LNS(16); break;
    case 32: EMO_LNS(32); break;
    default: EMO_LNS(64); break;
  }
#undef EMO_LNS
  EMO_LAUNCH_CHECK();
  return EMO_OK;
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
  EMO_CHECK(emo_dtype_ok(dtype), EMO_ERR_BAD_DTYPE, "emo_layernorm: dtype %d", dtype);
  const int V = emo_dtype_vec(dtype);
  EMO_CHECK(M > 0 && C > 0 && C % V == 0 && ldx % V == 0 && ldy % V == 0 && ldx >= C && ldy >= C, EMO_ERR_BAD_SHAPE,
            "emo_layernorm: M=%lld C=%d ldx=%d ldy=%d", (long long)M, C, ldx, ldy);
  EMO_CHECK(C / V <= 64 * LN_MAXV, EMO_ERR_UNSUPPORTED, "emo_layernorm: C=%d too wide", C);
  EMO_CHECK(!pe || (rows_per_frame > 0 && frames > 0), EMO_ERR_BAD_SHAPE, "emo_layernorm: pe needs rows_per_frame/frames");
  EMO_CHECK(((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0, EMO_ERR_BAD_SHAPE, "emo_layernorm: gamma/beta alignment");
  hipStream_t st = as_stream(stream);
  EMO_DISPATCH(dtype, "emo_layernorm", (launch_layernorm<T>((const T*)x, ldx, gamma, beta, (T*)y, ldy, M, C, eps, pe, rows_per_frame, frames, st)));
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}
