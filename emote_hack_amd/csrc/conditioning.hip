// conditioning.hip - the small EMO conditioning ops: speed-bucket encoders,
// embedding gather, ReLU/tanh, per-batch broadcast add.  All tiny and latency-bound.
#include "common.h"

static inline int cgrid(int64_t n) { int64_t g = (n + 255) / 256; return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g)); }

// act: 0 = SiLU, 1 = ReLU, 2 = tanh, 3 = erf-GELU (the wav2vec2 front-end's activation)
template <typename T>
__global__ void act_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, int kind) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = TT<T>::ld(x + i);
    v = kind == 0 ? silu_f(v) : (kind == 1 ? fmaxf(v, 0.f) : (kind == 2 ? tanhf(v) : gelu_for<T>(v)));
    TT<T>::st(y + i, v);
  }
}
extern "C" int emo_act(const void* x, void* y, int64_t n, int kind, int dtype, void* stream) {
  EMO_CHECK(x && y, EMO_ERR_NULL, "emo_act: null pointer");
  EMO_CHECK(n > 0 && kind >= 0 && kind <= 3, EMO_ERR_BAD_SHAPE, "emo_act: n=%lld kind=%d", (long long)n, kind);
  EMO_DISPATCH(dtype, "emo_act", (act_kernel<T><<<cgrid(n), 256, 0, as_stream(stream)>>>((const T*)x, (T*)y, n, kind)));
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

// SpeedEncoder.encode_speed (Net.py:231-247): out[b, i] = tanh((v[b] - center[i]) / radius[i] * 3), f32 math
template <typename T>
__global__ void speed_encode_kernel(const float* __restrict__ v, const float* __restrict__ centers, const float* __restrict__ radii,
                                    T* __restrict__ out, int B, int nb) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * nb) return;
  int b = i / nb, k = i % nb;
  TT<T>::st(out + i, tanhf((v[b] - centers[k]) / radii[k] * 3.0f));
}
extern "C" int emo_speed_encode(const float* v, const float* centers, const float* radii, void* out, int B, int nb, int dtype,
                                void* stream) {
  EMO_CHECK(v && centers && radii && out, EMO_ERR_NULL, "emo_speed_encode: null pointer");
  EMO_CHECK(B > 0 && nb > 0, EMO_ERR_BAD_SHAPE, "emo_speed_encode: bad shape");
  EMO_DISPATCH(dtype, "emo_speed_encode", (speed_encode_kernel<T><<<cgrid((int64_t)B * nb), 256, 0, as_stream(stream)>>>(v, centers, radii, (T*)out, B, nb)));
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

// SpeedController.map_speed_to_bucket (train_stage_3_speedlayers.py:42-47): argmin_i |v - centers[i]|, INT bit-exact:
// f32 |v - c|, strict '<' keeps the FIRST minimum (torch.argmin tie rule), out-of-range values clamp to the end buckets.
__global__ void speed_bucket_kernel(const float* __restrict__ v, const float* __restrict__ centers, int32_t* __restrict__ idx, int B, int nb) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float best = fabsf(v[b] - centers[0]);
  int bi = 0;
  for (int i = 1; i < nb; i++) {
    float d = fabsf(v[b] - centers[i]);
    if (d < best) { best = d; bi = i; }
  }
  idx[b] = bi;
}
extern "C" int emo_speed_bucket(const float* v, const float* centers, int32_t* idx, int B, int nb, void* stream) {
  EMO_CHECK(v && centers && idx, EMO_ERR_NULL, "emo_speed_bucket: null pointer");
  EMO_CHECK(B > 0 && nb > 0, EMO_ERR_BAD_SHAPE, "emo_speed_bucket: bad shape");
  speed_bucket_kernel<<<cgrid(B), 256, 0, as_stream(stream)>>>(v, centers, idx, B, nb);
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

// nn.Embedding lookup: out[b, :] = table[idx[b], :]
template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ table, const int32_t* __restrict__ idx, T* __restrict__ out, int B, int D, int rows) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)B * D; i += (int64_t)gridDim.x * blockDim.x) {
    int b = (int)(i / D), c = (int)(i % D);
    int r = idx[b];
    r = r < 0 ? 0 : (r >= rows ? rows - 1 : r);
    out[i] = table[(int64_t)r * D + c];
  }
}
extern "C" int emo_gather_rows(const void* table, const int32_t* idx, void* out, int B, int D, int rows, int dtype, void* stream) {
  EMO_CHECK(table && idx && out, EMO_ERR_NULL, "emo_gather_rows: null pointer");
  EMO_CHECK(B > 0 && D > 0 && rows > 0, EMO_ERR_BAD_SHAPE, "emo_gather_rows: bad shape");
  EMO_DISPATCH(dtype, "emo_gather_rows", (gather_rows_kernel<T><<<cgrid((int64_t)B * D), 256, 0, as_stream(stream)>>>((const T*)table, idx, (T*)out, B, D, rows)));
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

// y[m, c] = x[m, c] + rb[m / rows_per_batch, c]   (EMOStage3.forward: output + speed_embed[..., None, None],
// train_stage_3_speedlayers.py:268-269, in NHWC rows)
template <typename T>
__global__ void add_rowbias_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ rb, int ldr, T* __restrict__ y, int ldy,
                                   int64_t M, int C, int rows_per_batch) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < M * C; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t m = i / C; int c = (int)(i % C);
    TT<T>::st(y + m * ldy + c, TT<T>::ld(x + m * ldx + c) + TT<T>::ld(rb + (m / rows_per_batch) * ldr + c));
  }
}
extern "C" int emo_add_rowbias(const void* x, int ldx, const void* rb, int ldr, void* y, int ldy, int64_t M, int C, int rows_per_batch,
                               int dtype, void* stream) {
  EMO_CHECK(x && rb && y, EMO_ERR_NULL, "emo_add_rowbias: null pointer");
  EMO_CHECK(M > 0 && C > 0 && rows_per_batch > 0, EMO_ERR_BAD_SHAPE, "emo_add_rowbias: bad shape");
  EMO_DISPATCH(dtype, "emo_add_rowbias", (add_rowbias_kernel<T><<<cgrid(M * C), 256, 0, as_stream(stream)>>>((const T*)x, ldx, (const T*)rb, ldr, (T*)y, ldy, M, C, rows_per_batch)));
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}
