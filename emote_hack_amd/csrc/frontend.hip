// frontend.hip - the small kernels either side of the denoising loop:
//   emo_softmax_rows   : softmax(scale * x) over the columns of a materialised score matrix - the single-head, 512-wide
//                        attention of the VAE mid block (diffusers AutoencoderKL, called at EMOAnimationPipeline.py:291-307,
//                        402-414) runs as two MFMA GEMMs around it (head dim 512 is outside the flash kernel's register
//                        budget, and at one frame per call the 32 MB score matrix is cheap)
//   emo_audio_windows  : per audio frame, the wav2vec features of frames [f-m, f+n] zero-padded at the ends
//                        (Net.py:649-667 Wav2VecFeatureExtractor.extract_features_from_wav) - pure indexing, bit-exact
//   emo_rows_to_video  : decoded NHWC rows -> (B, C, F, H, W) f32 with video = (x / 2 + 0.5).clamp(0, 1)
//                        (EMOAnimationPipeline.py:303-306)
//   emo_channelnorm    : per-CHANNEL normalisation over the rows of a sequence (nn.GroupNorm(C, C) on (1, C, T): the first conv
//                        layer of the wav2vec2 feature extractor, transformers Wav2Vec2GroupNormConvLayer) + affine + optional
//                        erf-GELU; chunk partials in f32, re-reduced in f64 in a fixed order by every apply block (deterministic)
// All HBM-bound and tiny; 16-byte accesses where the geometry allows.
#include "common.h"

static inline int fgrid(int64_t work, int block) {
  int64_t g = (work + block - 1) / block;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

// one wavefront per row; two passes over the row (online max / sum, then normalise).  N is a few thousand (h*w of a latent)
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy, int64_t M,
                                                           int N, float scale) {
  constexpr int V = TT<T>::VEC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float c = scale * 1.4426950408889634f;
  for (int64_t m = (int64_t)blockIdx.x * 4 + wave; m < M; m += (int64_t)gridDim.x * 4) {
    const T* xr = x + m * ldx;
    T* yr = y + m * ldy;
    float mx = -3.0e38f, sum = 0.f;
    const int NV = N / V;
    for (int i = lane; i < NV; i += 64) {
      float f[V];
      unpack16<T>(*(const uint4*)(xr + i * V), f);
      float lm = f[0];
#pragma unroll
      for (int e = 1; e < V; e++) lm = fmaxf(lm, f[e]);
      const float nm = fmaxf(mx, lm);
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < V; e++) s += exp2f((f[e] - nm) * c);
      sum = sum * exp2f((mx - nm) * c) + s;
      mx = nm;
    }
    for (int i = NV * V + lane; i < N; i += 64) {   // ragged tail
      const float v = TT<T>::ld(xr + i), nm = fmaxf(mx, v);
      sum = sum * exp2f((mx - nm) * c) + exp2f((v - nm) * c);
      mx = nm;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float om = __shfl_xor(mx, o, 64), os = __shfl_xor(sum, o, 64);
      const float nm = fmaxf(mx, om);
      sum = sum * exp2f((mx - nm) * c) + os * exp2f((om - nm) * c);
      mx = nm;
    }
    const float inv = 1.0f / sum;
    for (int i = lane; i < NV; i += 64) {
      float f[V];
      unpack16<T>(*(const uint4*)(xr + i * V), f);
#pragma unroll
      for (int e = 0; e < V; e++) f[e] = exp2f((f[e] - mx) * c) * inv;
      *(uint4*)(yr + i * V) = pack16<T>(f);
    }
    for (int i = NV * V + lane; i < N; i += 64) TT<T>::st(yr + i, exp2f((TT<T>::ld(xr + i) - mx) * c) * inv);
  }
}

extern "C" int emo_softmax_rows(const void* x, int64_t ldx, void* y, int64_t ldy, int64_t M, int N, float scale, int dtype, void* stream) {
  EMO_CHECK(x && y, EMO_ERR_NULL, "emo_softmax_rows: null pointer");
  EMO_CHECK(emo_dtype_ok(dtype), EMO_ERR_BAD_DTYPE, "emo_softmax_rows: dtype %d", dtype);
  const int V = emo_dtype_vec(dtype);
  EMO_CHECK(M > 0 && N > 0 && ldx >= N && ldy >= N && ldx % V == 0 && ldy % V == 0, EMO_ERR_BAD_SHAPE,
            "emo_softmax_rows: M=%lld N=%d ldx=%lld ldy=%lld", (long long)M, N, (long long)ldx, (long long)ldy);
  EMO_CHECK(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0, EMO_ERR_BAD_SHAPE, "emo_softmax_rows: 16-byte alignment");
  EMO_DISPATCH(dtype, "emo_softmax_rows", (softmax_rows_kernel<T><<<fgrid(M, 4), 256, 0, as_stream(stream)>>>((const T*)x, ldx, (T*)y, ldy, M, N, scale)));
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

// out[t][j][:] = feats[t - m + j][:] if 0 <= t - m + j < T else 0     (j = 0 .. m + n)
template <typename T>
__global__ __launch_bounds__(256) void audio_windows_kernel(const T* __restrict__ feats, T* __restrict__ out, int Tn, int D, int m, int n) {
  const int wlen = m + n + 1;
  const int64_t total = (int64_t)Tn * wlen * D;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const int64_t r = i / D;
    const int j = (int)(r % wlen), t = (int)(r / wlen);
    const int src = t - m + j;
    out[i] = (src >= 0 && src < Tn) ? feats[(int64_t)src * D + d] : (T)0;
  }
}

extern "C" int emo_audio_windows(const void* feats, void* out, int T_, int D, int m, int n, int dtype, void* stream) {
  EMO_CHECK(feats && out, EMO_ERR_NULL, "emo_audio_windows: null pointer");
  EMO_CHECK(T_ > 0 && D > 0 && m >= 0 && n >= 0, EMO_ERR_BAD_SHAPE, "emo_audio_windows: T=%d D=%d m=%d n=%d", T_, D, m, n);
  EMO_CHECK(emo_dtype_ok(dtype), EMO_ERR_BAD_DTYPE, "emo_audio_windows: dtype %d", dtype);
  const int64_t total = (int64_t)T_ * (m + n + 1) * D;
  EMO_DISPATCH(dtype, "emo_audio_windows", (audio_windows_kernel<T><<<fgrid(total, 256), 256, 0, as_stream(stream)>>>((const T*)feats, (T*)out, T_, D, m, n)));
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

// rows ((b f) h w, ld >= C) -> (B, C, F, H, W) f32, y = clamp(x * mul + add, lo, hi)
template <typename T>
__global__ __launch_bounds__(256) void rows_to_video_kernel(const T* __restrict__ x, int64_t ld, float* __restrict__ y, int B, int C, int F,
                                                            int HW, float mul, float add, float lo, float hi) {
  const int64_t total = (int64_t)B * C * F * HW;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW);
    int64_t r = i / HW;
    const int f = (int)(r % F); r /= F;
    const int c = (int)(r % C);
    const int b = (int)(r / C);
    const float v = TT<T>::ld(x + ((int64_t)(b * F + f) * HW + p) * ld + c) * mul + add;
    y[i] = fminf(fmaxf(v, lo), hi);
  }
}

extern "C" int emo_rows_to_video(const void* x, int64_t ld, float* y, int B, int C, int F, int HW, float mul, float add, float lo, float hi,
                                 int dtype, void* stream) {
  EMO_CHECK(x && y, EMO_ERR_NULL, "emo_rows_to_video: null pointer");
  EMO_CHECK(B > 0 && C > 0 && F > 0 && HW > 0 && ld >= C, EMO_ERR_BAD_SHAPE, "emo_rows_to_video: bad shape");
  EMO_CHECK(emo_dtype_ok(dtype), EMO_ERR_BAD_DTYPE, "emo_rows_to_video: dtype %d", dtype);
  const int64_t total = (int64_t)B * C * F * HW;
  EMO_DISPATCH(dtype, "emo_rows_to_video", (rows_to_video_kernel<T><<<fgrid(total, 256), 256, 0, as_stream(stream)>>>((const T*)x, ld, y, B, C, F, HW, mul, add, lo, hi)));
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

// ---- FaceLocator pieces (Net.py:819-855): 2x2 max pooling over NHWC rows, bilinear upsampling of the logits map
template <typename T>
__global__ __launch_bounds__(256) void maxpool2x2_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy, int n_img, int H,
                                                         int W, int C) {
  const int Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)n_img * Ho * Wo * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int xo = (int)(r % Wo); r /= Wo;
    const int yo = (int)(r % Ho);
    const int img = (int)(r / Ho);
    const T* p = x + (((int64_t)img * H + 2 * yo) * W + 2 * xo) * ldx + c;
    const float v = fmaxf(fmaxf(TT<T>::ld(p), TT<T>::ld(p + ldx)), fmaxf(TT<T>::ld(p + (int64_t)W * ldx), TT<T>::ld(p + (int64_t)(W + 1) * ldx)));
    TT<T>::st(y + (((int64_t)img * Ho + yo) * Wo + xo) * ldy + c, v);
  }
}

extern "C" int emo_maxpool2x2(const void* x, int64_t ldx, void* y, int64_t ldy, int n_img, int H, int W, int C, int dtype, void* stream) {
  EMO_CHECK(x && y, EMO_ERR_NULL, "emo_maxpool2x2: null pointer");
  EMO_CHECK(n_img > 0 && H >= 2 && W >= 2 && C > 0 && ldx >= C && ldy >= C, EMO_ERR_BAD_SHAPE, "emo_maxpool2x2: bad shape");
  EMO_CHECK(emo_dtype_ok(dtype), EMO_ERR_BAD_DTYPE, "emo_maxpool2x2: dtype %d", dtype);
  const int64_t total = (int64_t)n_img * (H / 2) * (W / 2) * C;
  EMO_DISPATCH(dtype, "emo_maxpool2x2", (maxpool2x2_kernel<T><<<fgrid(total, 256), 256, 0, as_stream(stream)>>>((const T*)x, ldx, (T*)y, ldy, n_img, H, W, C)));
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

// F.interpolate(mode='bilinear', align_corners=False) of rows ((n) h w, ld >= C) to (n, C, Ho, Wo) f32:
// src = (dst + 0.5) * in / out - 0.5 clamped at 0, neighbours i0 = floor(src), i1 = min(i0 + 1, in - 1)
template <typename T>
__global__ __launch_bounds__(256) void bilinear_kernel(const T* __restrict__ x, int64_t ld, float* __restrict__ y, int n_img, int C, int h, int w,
                                                       int Ho, int Wo) {
  const int64_t total = (int64_t)n_img * C * Ho * Wo;
  const float sy = (float)h / (float)Ho, sx = (float)w / (float)Wo;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xo = (int)(i % Wo);
    int64_t r = i / Wo;
    const int yo = (int)(r % Ho); r /= Ho;
    const int c = (int)(r % C);
    const int img = (int)(r / C);
    const float fy = fmaxf(((float)yo + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf(((float)xo + 0.5f) * sx - 0.5f, 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + 1 < h ? y0 + 1 : h - 1, x1 = x0 + 1 < w ? x0 + 1 : w - 1;
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const T* b = x + (int64_t)img * h * w * ld + c;
    const float v00 = TT<T>::ld(b + ((int64_t)y0 * w + x0) * ld), v01 = TT<T>::ld(b + ((int64_t)y0 * w + x1) * ld);
    const float v10 = TT<T>::ld(b + ((int64_t)y1 * w + x0) * ld), v11 = TT<T>::ld(b + ((int64_t)y1 * w + x1) * ld);
    y[i] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
  }
}

extern "C" int emo_bilinear_to_nchw(const void* x, int64_t ld, float* y, int n_img, int C, int h, int w, int Ho, int Wo, int dtype, void* stream) {
  EMO_CHECK(x && y, EMO_ERR_NULL, "emo_bilinear_to_nchw: null pointer");
  EMO_CHECK(n_img > 0 && C > 0 && h > 0 && w > 0 && Ho > 0 && Wo > 0 && ld >= C, EMO_ERR_BAD_SHAPE, "emo_bilinear_to_nchw: bad shape");
  EMO_CHECK(emo_dtype_ok(dtype), EMO_ERR_BAD_DTYPE, "emo_bilinear_to_nchw: dtype %d", dtype);
  const int64_t total = (int64_t)n_img * C * Ho * Wo;
  EMO_DISPATCH(dtype, "emo_bilinear_to_nchw", (bilinear_kernel<T><<<fgrid(total, 256), 256, 0, as_stream(stream)>>>((const T*)x, ld, y, n_img, C, h, w, Ho, Wo)));
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}


// ---------------------------------------------------------------------------------------------------------------- channel norm
// x rows (S, C): y[s, c] = act((x[s, c] - mean_c) * rstd_c * gamma_c + beta_c), statistics over the S rows of each channel
// (biased variance, like nn.GroupNorm).  Pass 1: block (chunk, column slab of 64 channels) -> partial (sum, sumsq) per channel.
// Pass 2: every block re-reduces the chunk partials of its slab in f64 in chunk order, then normalises its rows.
static constexpr int CN_COLS = 64, CN_ROWS_T = 4;   // 256 threads = 64 channels x 4 row lanes

template <typename T>
__global__ __launch_bounds__(256) void channelnorm_stats_kernel(const T* __restrict__ x, int64_t ldx, float* __restrict__ part, int64_t S, int C, int nchunk) {
  const int c = blockIdx.y * CN_COLS + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int64_t per = (S + nchunk - 1) / nchunk, s0 = blockIdx.x * per, s1 = s0 + per < S ? s0 + per : S;
  float sum = 0.f, sq = 0.f;
  if (c < C)
    for (int64_t s_ = s0 + rl; s_ < s1; s_ += CN_ROWS_T) {
      const float v = TT<T>::ld(x + s_ * ldx + c);
      sum += v; sq = fmaf(v, v, sq);
    }
  __shared__ float sh[2][CN_ROWS_T][CN_COLS];
  sh[0][rl][threadIdx.x & 63] = sum; sh[1][rl][threadIdx.x & 63] = sq;
  __syncthreads();
  if (rl == 0 && c < C) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int r = 0; r < CN_ROWS_T; r++) { a += sh[0][r][threadIdx.x]; b += sh[1][r][threadIdx.x]; }
    part[((int64_t)blockIdx.x * C + c) * 2] = a;
    part[((int64_t)blockIdx.x * C + c) * 2 + 1] = b;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void channelnorm_apply_kernel(const T* __restrict__ x, int64_t ldx, const float* __restrict__ part,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y,
                                                                int64_t ldy, int64_t S, int C, int nchunk, float eps, int act) {
  const int cl = threadIdx.x & 63, c = blockIdx.y * CN_COLS + cl, rl = threadIdx.x >> 6;
  __shared__ float sc[CN_COLS], sf[CN_COLS];
  if (rl == 0 && c < C) {
    double a = 0.0, b = 0.0;
    for (int k = 0; k < nchunk; k++) { a += (double)part[((int64_t)k * C + c) * 2]; b += (double)part[((int64_t)k * C + c) * 2 + 1]; }
    const double mean = a / (double)S;
    double var = b / (double)S - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    sc[cl] = rstd * gamma[c];
    sf[cl] = beta[c] - (float)mean * rstd * gamma[c];
  }
  __syncthreads();
  if (c >= C) return;
  const float k = sc[cl], o = sf[cl];
  for (int64_t s_ = (int64_t)blockIdx.x * CN_ROWS_T + rl; s_ < S; s_ += (int64_t)gridDim.x * CN_ROWS_T) {
    float v = fmaf(TT<T>::ld(x + s_ * ldx + c), k, o);
    if (act == 1) v = gelu_for<T>(v);
    TT<T>::st(y + s_ * ldy + c, v);
  }
}

extern "C" size_t emo_channelnorm_workspace_bytes(int64_t S, int C) {
  int64_t n = (S + 255) / 256;
  if (n > 64) n = 64;
  if (n < 1) n = 1;
  return (size_t)n * C * 2 * sizeof(float);
}
extern "C" int emo_channelnorm(const void* x, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy, int64_t S, int C, float eps,
                               int act, void* workspace, int dtype, void* stream) {
  EMO_CHECK(x && gamma && beta && y && workspace, EMO_ERR_NULL, "emo_channelnorm: null pointer");
  EMO_CHECK(S > 0 && C > 0 && ldx >= C && ldy >= C && (act == 0 || act == 1), EMO_ERR_BAD_SHAPE, "emo_channelnorm: S=%lld C=%d act=%d", (long long)S, C, act);
  int64_t n = (S + 255) / 256;
  if (n > 64) n = 64;
  if (n < 1) n = 1;
  const dim3 g1((unsigned)n, (unsigned)((C + CN_COLS - 1) / CN_COLS));
  int64_t nb = (S + CN_ROWS_T * 16 - 1) / (CN_ROWS_T * 16);
  if (nb > 512) nb = 512;
  if (nb < 1) nb = 1;
  const dim3 g2((unsigned)nb, g1.y);
  hipStream_t st = as_stream(stream);
  EMO_DISPATCH(dtype, "emo_channelnorm", (channelnorm_stats_kernel<T><<<g1, 256, 0, st>>>((const T*)x, ldx, (float*)workspace, S, C, (int)n)));
  EMO_LAUNCH_CHECK();
  EMO_DISPATCH(dtype, "emo_channelnorm", (channelnorm_apply_kernel<T><<<g2, 256, 0, st>>>((const T*)x, ldx, (const float*)workspace, gamma, beta, (T*)y, ldy, S, C,
                                                                                               (int)n, eps, act)));
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}
