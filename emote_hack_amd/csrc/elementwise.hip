// elementwise.hip - layout conversion, channel concat, residual add, dtype convert, SiLU, timestep
// sinusoid, and the fused sampler step.  All HBM-bound: 16-byte vector accesses, grid-stride.
#include <stdarg.h>
#include "common.h"

thread_local char emo_err_buf[256] = {0};
int emo_fail(int code, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(emo_err_buf, sizeof(emo_err_buf), fmt, ap); va_end(ap);
  return code;
}
extern "C" int emo_version(void) { return 200; }   // round 2: emo_attention_params.seg1_row, LayerNorm-folded GEMM
extern "C" const char* emo_last_error_string(void) { return emo_err_buf; }

static inline int grid_for(int64_t work, int block) {
  int64_t g = (work + block - 1) / block;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

// ---------------------------------------------------------------- (B,C,F,H,W) f32 <-> rows
// Tile transpose through LDS: a block moves a [32 channels x 64 pixels] tile so both sides are coalesced.
template <typename T>
__global__ __launch_bounds__(256) void ncfhw_to_rows_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int C,
                                                            int F, int HW, int Cpad, int ldo) {
  __shared__ float tile[32][65];
  const int ptiles = (HW + 63) / 64, ctiles = (Cpad + 31) / 32;
  const int64_t ntiles = (int64_t)B * F * ptiles * ctiles;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    int ct = t % ctiles; int64_t r = t / ctiles;
    int pt = r % ptiles; r /= ptiles;
    int f = r % F; int b = r / F;
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {
      int c = i / 64, p = i % 64;
      int cc = ct * 32 + c, pp = pt * 64 + p;
      float v = 0.f;
      if (cc < C && pp < HW) v = x[(((int64_t)b * C + cc) * F + f) * HW + pp];
      tile[c][p] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {
      int p = i / 32, c = i % 32;
      int cc = ct * 32 + c, pp = pt * 64 + p;
      if (cc < Cpad && pp < HW) TT<T>::st(&y[((int64_t)(b * F + f) * HW + pp) * ldo + cc], tile[c][p]);
    }
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(256) void rows_to_ncfhw_kernel(const T* __restrict__ x, float* __restrict__ y, int B, int C,
                                                            int F, int HW, int ldi) {
  __shared__ float tile[32][65];
  const int ptiles = (HW + 63) / 64, ctiles = (C + 31) / 32;
  const int64_t ntiles = (int64_t)B * F * ptiles * ctiles;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    int ct = t % ctiles; int64_t r = t / ctiles;
    int pt = r % ptiles; r /= ptiles;
    int f = r % F; int b = r / F;
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {
      int p = i / 32, c = i % 32;
      int cc = ct * 32 + c, pp = pt * 64 + p;
      float v = 0.f;
      if (cc < C && pp < HW) v = TT<T>::ld(&x[((int64_t)(b * F + f) * HW + pp) * ldi + cc]);
      tile[c][p] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {
      int c = i / 64, p = i % 64;
      int cc = ct * 32 + c, pp = pt * 64 + p;
      if (cc < C && pp < HW) y[(((int64_t)b * C + cc) * F + f) * HW + pp] = tile[c][p];
    }
    __syncthreads();
  }
}

extern "C" int emo_ncfhw_to_rows(const float* x, void* y, int B, int C, int F, int H, int W, int Cpad, int ldo,
                                 int dtype, void* stream) {
  EMO_CHECK(x && y, EMO_ERR_NULL, "emo_ncfhw_to_rows: null pointer");
  EMO_CHECK(B > 0 && C > 0 && F > 0 && H > 0 && W > 0 && Cpad >= C && ldo >= Cpad, EMO_ERR_BAD_SHAPE,
            "emo_ncfhw_to_rows: bad shape B=%d C=%d F=%d H=%d W=%d Cpad=%d ldo=%d", B, C, F, H, W, Cpad, ldo);
  int HW = H * W;
  int64_t ntiles = (int64_t)B * F * ((HW + 63) / 64) * ((Cpad + 31) / 32);
  int grid = (int)(ntiles < 4096 ? ntiles : 4096);
  EMO_DISPATCH(dtype, "emo_ncfhw_to_rows", (ncfhw_to_rows_kernel<T><<<grid, 256, 0, as_stream(stream)>>>(x, (T*)y, B, C, F, HW, Cpad, ldo)));
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

extern "C" int emo_rows_to_ncfhw(const void* x, float* y, int B, int C, int F, int H, int W, int ldi, int dtype,
                                 void* stream) {
  EMO_CHECK(x && y, EMO_ERR_NULL, "emo_rows_to_ncfhw: null pointer");
  EMO_CHECK(B > 0 && C > 0 && F > 0 && H > 0 && W > 0 && ldi >= C, EMO_ERR_BAD_SHAPE, "emo_rows_to_ncfhw: bad shape");
  int HW = H * W;
  int64_t ntiles = (int64_t)B * F * ((HW + 63) / 64) * ((C + 31) / 32);
  int grid = (int)(ntiles < 4096 ? ntiles : 4096);
  EMO_DISPATCH(dtype, "emo_rows_to_ncfhw", (rows_to_ncfhw_kernel<T><<<grid, 256, 0, as_stream(stream)>>>((const T*)x, y, B, C, F, HW, ldi)));
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

// ---------------------------------------------------------------- copy columns / add (16-byte vectors)
template <typename T>
__global__ __launch_bounds__(256) void copy_cols_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int coff,
                                                        int64_t M, int C) {
  constexpr int V = TT<T>::VEC;
  const int cv = C / V;
  const int64_t total = M * cv;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t m = i / cv; int c = (int)(i % cv) * V;
    *(uint4*)(y + m * ldy + coff + c) = *(const uint4*)(x + m * ldx + c);
  }
}
extern "C" int emo_copy_cols(const void* x, int ldx, void* y, int ldy, int coff, int64_t M, int C, int dtype, void* stream) {
  EMO_CHECK(x && y, EMO_ERR_NULL, "emo_copy_cols: null pointer");
  int V = emo_dtype_vec(dtype);
  EMO_CHECK(emo_dtype_ok(dtype), EMO_ERR_BAD_DTYPE, "emo_copy_cols: dtype %d", dtype);
  EMO_CHECK(M > 0 && C > 0 && C % V == 0 && coff % V == 0 && ldx % V == 0 && ldy % V == 0 && coff + C <= ldy && C <= ldx,
            EMO_ERR_BAD_SHAPE, "emo_copy_cols: C=%d coff=%d ldx=%d ldy=%d must be multiples of %d", C, coff, ldx, ldy, V);
  int grid = grid_for(M * (C / V), 256);
  EMO_DISPATCH(dtype, "emo_copy_cols", (copy_cols_kernel<T><<<grid, 256, 0, as_stream(stream)>>>((const T*)x, ldx, (T*)y, ldy, coff, M, C)));
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, int lda, const T* __restrict__ b, int ldb, float alpha,
                                                  T* __restrict__ y, int ldy, int64_t M, int C) {
  constexpr int V = TT<T>::VEC;
  const int cv = C / V;
  const int64_t total = M * cv;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t m = i / cv; int c = (int)(i % cv) * V;
    float fa[V], fb[V];
    unpack16<T>(*(const uint4*)(a + m * lda + c), fa);
    unpack16<T>(*(const uint4*)(b + m * ldb + c), fb);
#pragma unroll
    for (int j = 0; j < V; j++) fa[j] += alpha * fb[j];
    *(uint4*)(y + m * ldy + c) = pack16<T>(fa);
  }
}
extern "C" int emo_add(const void* a, int lda, const void* b, int ldb, float alpha, void* y, int ldy, int64_t M, int C,
                       int dtype, void* stream) {
  EMO_CHECK(a && b && y, EMO_ERR_NULL, "emo_add: null pointer");
  EMO_CHECK(emo_dtype_ok(dtype), EMO_ERR_BAD_DTYPE, "emo_add: dtype %d", dtype);
  int V = emo_dtype_vec(dtype);
  EMO_CHECK(M > 0 && C > 0 && C % V == 0 && lda % V == 0 && ldb % V == 0 && ldy % V == 0, EMO_ERR_BAD_SHAPE, "emo_add: bad shape");
  int grid = grid_for(M * (C / V), 256);
  EMO_DISPATCH(dtype, "emo_add", (add_kernel<T><<<grid, 256, 0, as_stream(stream)>>>((const T*)a, lda, (const T*)b, ldb, alpha, (T*)y, ldy, M, C)));
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

// ---------------------------------------------------------------- convert / silu (scalar tails allowed)
__device__ __forceinline__ float round_through_half(float f) { return (float)(_Float16)f; }  // IEEE half, RNE

template <typename S, typename D>
__global__ __launch_bounds__(256) void convert_kernel(const S* __restrict__ s, D* __restrict__ d, int64_t n, int fp16_round) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = TT<S>::ld(s + i);
    if (fp16_round) v = round_through_half(v);
    TT<D>::st(d + i, v);
  }
}
extern "C" int emo_convert(const void* src, int sdt, void* dst, int ddt, int64_t n, int fp16_round, void* stream) {
  EMO_CHECK(src && dst, EMO_ERR_NULL, "emo_convert: null pointer");
  EMO_CHECK(n > 0, EMO_ERR_BAD_SHAPE, "emo_convert: n=%lld", (long long)n);
  int grid = grid_for(n, 256);
  hipStream_t st = as_stream(stream);
  EMO_CHECK(emo_dtype_ok(sdt) && emo_dtype_ok(ddt), EMO_ERR_BAD_DTYPE, "emo_convert: dtypes %d -> %d", sdt, ddt);
  EMO_DISPATCH(sdt, "emo_convert", {
    using S = T;
    const S* sp = (const S*)src;
    if (ddt == EMO_F32) convert_kernel<S, float><<<grid, 256, 0, st>>>(sp, (float*)dst, n, fp16_round);
    else if (ddt == EMO_BF16) convert_kernel<S, bf16_t><<<grid, 256, 0, st>>>(sp, (bf16_t*)dst, n, fp16_round);
    else convert_kernel<S, f16_t><<<grid, 256, 0, st>>>(sp, (f16_t*)dst, n, fp16_round);
  });
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void silu_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    TT<T>::st(y + i, silu_f(TT<T>::ld(x + i)));
}
extern "C" int emo_silu(const void* x, void* y, int64_t n, int dtype, void* stream) {
  EMO_CHECK(x && y, EMO_ERR_NULL, "emo_silu: null pointer");
  EMO_CHECK(n > 0, EMO_ERR_BAD_SHAPE, "emo_silu: n");
  int grid = grid_for(n, 256);
  EMO_DISPATCH(dtype, "emo_silu", (silu_kernel<T><<<grid, 256, 0, as_stream(stream)>>>((const T*)x, (T*)y, n)));
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

// ---------------------------------------------------------------- timestep sinusoid
// embeddings.py:46-63: emb = t.float() * exp(exponent); [sin | cos], flipped to [cos | sin] when
// flip_sin_to_cos.  `freqs` = exp(-ln(max_period) * arange(half) / (half - shift)) is a constant f32
// table built once on the host exactly as the reference builds it, so the angle t*freq is the same
// f32 product (an on-device expf would move the k=0 angle of t=981 by ~1e-4 rad).
template <typename T>
__global__ void timestep_kernel(const int64_t* __restrict__ ts, const float* __restrict__ freqs, T* __restrict__ out, int B,
                                int dim, int flip) {
  const int half = dim / 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * dim; i += gridDim.x * blockDim.x) {
    int b = i / dim, j = i % dim;
    float v = 0.f;
    if (j < 2 * half) {
      int k = j % half;
      bool second = j >= half;
      float ang = (float)ts[b] * freqs[k];
      bool want_cos = flip ? !second : second;
      v = want_cos ? cosf(ang) : sinf(ang);
    }
    TT<T>::st(out + i, v);
  }
}
extern "C" int emo_timestep_embedding(const int64_t* ts, const float* freqs, void* out, int B, int dim, int flip, int dtype,
                                      void* stream) {
  EMO_CHECK(ts && out && freqs, EMO_ERR_NULL, "emo_timestep_embedding: null pointer");
  EMO_CHECK(B > 0 && dim > 1, EMO_ERR_BAD_SHAPE, "emo_timestep_embedding: B=%d dim=%d", B, dim);
  int grid = grid_for((int64_t)B * dim, 256);
  EMO_DISPATCH(dtype, "emo_timestep_embedding", (timestep_kernel<T><<<grid, 256, 0, as_stream(stream)>>>(ts, freqs, (T*)out, B, dim, flip)));
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

// ---------------------------------------------------------------- sampler: CFG + window average + step
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x = (x ^ (x >> 16)) * 0x7FEB352Du;
  x = (x ^ (x >> 15)) * 0x846CA68Bu;
  x = x ^ (x >> 16);
  return x;
}
__global__ __launch_bounds__(256) void cfg_step_kernel(const float* __restrict__ np, const float* __restrict__ counter,
                                                       float* __restrict__ lat, float* __restrict__ eps_out, int C, int F, int HW,
                                                       float gs, float c_x, float c_eps, float c_n, uint32_t seed, uint32_t step) {
  const int64_t n = (int64_t)C * F * HW;
  const uint32_t key = mix32(seed ^ mix32(step + 0x9E3779B9u));
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int f = (int)((i / HW) % F);
    float inv = 1.0f / counter[f];
    float eps = np[i] * inv;                        // noise_pred / counter (EMOAnimationPipeline.py:813)
    if (gs > 1.0f) {                                // do_classifier_free_guidance = guidance_scale > 1.0 (:622); else np is [1][n]
      const float cc = np[n + i] * inv;
      eps = eps + gs * (cc - eps);                  // :814
    }
    float x = c_x * lat[i] + c_eps * eps;
    if (c_n != 0.f) {
      uint32_t h1 = mix32(((uint32_t)i * 2u + 0u) ^ key), h2 = mix32(((uint32_t)i * 2u + 1u) ^ key);
      float u1 = ((float)h1 + 1.0f) * 2.3283064365386963e-10f;  // (0,1]
      float u2 = (float)h2 * 2.3283064365386963e-10f;
      x += c_n * sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
    }
    lat[i] = x;
    if (eps_out) eps_out[i] = eps;
  }
}
extern "C" int emo_cfg_step(const float* np, const float* counter, float* lat, float* eps_out, int C, int F, int HW,
                            float gs, float c_x, float c_eps, float c_n, uint32_t seed, uint32_t step, void* stream) {
  EMO_CHECK(np && counter && lat, EMO_ERR_NULL, "emo_cfg_step: null pointer");
  EMO_CHECK(C > 0 && F > 0 && HW > 0, EMO_ERR_BAD_SHAPE, "emo_cfg_step: bad shape");
  cfg_step_kernel<<<grid_for((int64_t)C * F * HW, 256), 256, 0, as_stream(stream)>>>(np, counter, lat, eps_out, C, F, HW, gs, c_x,
                                                                                   c_eps, c_n, seed, step);
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void accumulate_window_kernel(const T* __restrict__ pred, int ld, float* __restrict__ np,
                                                                float* __restrict__ counter, const int32_t* __restrict__ frames,
                                                                int nf, int C, int F, int HW, int add_counter) {
  const int64_t n = (int64_t)nf * HW * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % C); int64_t r = i / C;
    int p = (int)(r % HW); int j = (int)(r / HW);
    int f = frames[j];
    if (f < 0) continue;   // position dropped by the host: an earlier duplicate of a frame inside one window
    np[((int64_t)c * F + f) * HW + p] += TT<T>::ld(pred + r * ld + c);
  }
  if (add_counter && blockIdx.x == 0 && threadIdx.x < nf && frames[threadIdx.x] >= 0) counter[frames[threadIdx.x]] += 1.0f;
}
extern "C" int emo_accumulate_window(const void* pred, int ld, float* np, float* counter, const int32_t* frames, int nf, int C,
                                     int F, int HW, int add_counter, int dtype, void* stream) {
  EMO_CHECK(pred && np && counter && frames, EMO_ERR_NULL, "emo_accumulate_window: null pointer");
  EMO_CHECK(nf > 0 && nf <= 256 && C > 0 && ld >= C, EMO_ERR_BAD_SHAPE, "emo_accumulate_window: bad shape");
  int grid = grid_for((int64_t)nf * HW * C, 256);
  EMO_DISPATCH(dtype, "emo_accumulate_window", (accumulate_window_kernel<T><<<grid, 256, 0, as_stream(stream)>>>((const T*)pred, ld, np, counter, frames, nf, C, F, HW, add_counter)));
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}
