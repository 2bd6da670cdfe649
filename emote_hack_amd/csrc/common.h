// common.h - shared device/host helpers for the gfx950 kernels (wave64, MFMA 32x32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>
#include <utility>
#include "../../include/emo_hip.h"

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef _Float16 f16_t;         // IEEE half (a distinct C++ type: the kernels are templated on the element type)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

extern thread_local char emo_err_buf[256];
int emo_fail(int code, const char* fmt, ...);

#define EMO_CHECK(cond, code, ...) \
  do { if (!(cond)) return emo_fail((code), __VA_ARGS__); } while (0)

#define EMO_LAUNCH_CHECK()                                                        \
  do { hipError_t e_ = hipGetLastError();                                         \
       if (e_ != hipSuccess) return emo_fail(EMO_ERR_HIP, "%s:%d hip launch: %s", \
                                             __FILE__, __LINE__, hipGetErrorString(e_)); } while (0)

__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
// f32 -> bf16 (round to nearest even): the native cast lowers to v_cvt_pk_bf16_f32 on gfx950 (one VALU op
// per PAIR instead of ~5 integer ops per value)
typedef __bf16 bf16x2_native __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  bf16x2_native v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}

template <typename T> struct TT;
template <> struct TT<float> {
  static constexpr int VEC = 4;
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
  __device__ static __forceinline__ float ld_val(float v) { return v; }
};
template <> struct TT<bf16_t> {
  static constexpr int VEC = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
  __device__ static __forceinline__ float ld_val(float v) { return bf2f(f2bf(v)); }
};

template <> struct TT<f16_t> {
  static constexpr int VEC = 8;
  __device__ static __forceinline__ float ld(const f16_t* p) { return (float)*p; }
  __device__ static __forceinline__ void st(f16_t* p, float v) { *p = (f16_t)v; }
  __device__ static __forceinline__ float ld_val(float v) { return (float)(f16_t)v; }
};
typedef _Float16 f16x2_native __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) {   // round to nearest even (v_cvt_f16_f32), not pkrtz
  f16x2_native v = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
// two f32 -> one packed 32-bit word of T (T = bf16_t or f16_t)
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack2<bf16_t>(float lo, float hi) { return pack_bf2(lo, hi); }
template <> __device__ __forceinline__ uint32_t pack2<f16_t>(float lo, float hi) { return pack_h2(lo, hi); }
// four 2-byte elements (8 bytes) -> f32
template <typename T> __device__ __forceinline__ void unpack4(const uint2& v, float* o);
template <> __device__ __forceinline__ void unpack4<bf16_t>(const uint2& v, float* o) {
  o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
  o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack4<f16_t>(const uint2& v, float* o) {
  const f16x2_native a = __builtin_bit_cast(f16x2_native, v.x), b = __builtin_bit_cast(f16x2_native, v.y);
  o[0] = (float)a.x; o[1] = (float)a.y; o[2] = (float)b.x; o[3] = (float)b.y;
}
// round an f32 through T (the reference casts attention probabilities to the value dtype)
template <typename T> __device__ __forceinline__ float round_through(float v) { return TT<T>::ld_val(v); }

// unpack a 16-byte vector of T into floats (VEC of them)
template <typename T> __device__ __forceinline__ void unpack16(const uint4& v, float* out);
template <> __device__ __forceinline__ void unpack16<float>(const uint4& v, float* o) {
  o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y); o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const uint4& v, float* o) {
  o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
  o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
  o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack16<f16_t>(const uint4& v, float* o) {
  const f16x2_native a = __builtin_bit_cast(f16x2_native, v.x), b = __builtin_bit_cast(f16x2_native, v.y);
  const f16x2_native c = __builtin_bit_cast(f16x2_native, v.z), d = __builtin_bit_cast(f16x2_native, v.w);
  o[0] = (float)a.x; o[1] = (float)a.y; o[2] = (float)b.x; o[3] = (float)b.y;
  o[4] = (float)c.x; o[5] = (float)c.y; o[6] = (float)d.x; o[7] = (float)d.y;
}
template <typename T> __device__ __forceinline__ uint4 pack16(const float* in);
template <> __device__ __forceinline__ uint4 pack16<float>(const float* i) {
  return make_uint4(__float_as_uint(i[0]), __float_as_uint(i[1]), __float_as_uint(i[2]), __float_as_uint(i[3]));
}
template <> __device__ __forceinline__ uint4 pack16<bf16_t>(const float* i) {
  return make_uint4(pack_bf2(i[0], i[1]), pack_bf2(i[2], i[3]), pack_bf2(i[4], i[5]), pack_bf2(i[6], i[7]));
}

template <> __device__ __forceinline__ uint4 pack16<f16_t>(const float* i) {
  return make_uint4(pack_h2(i[0], i[1]), pack_h2(i[2], i[3]), pack_h2(i[4], i[5]), pack_h2(i[6], i[7]));
}

// One "mma16" step: A and B fragments are 16 bytes per lane (bf16: 8 k-values, f32: 4 k-values) of the
// rows/cols (lane&31), k-half (lane>>5).  bf16 -> one v_mfma_f32_32x32x16_bf16 (K=16);
// f32 -> four v_mfma_f32_32x32x2_f32 (K=8, exact f32 fma chain).  Any k<->lane assignment is valid as
// long as A and B agree, which they do because both fragments come from the same chunk index.
template <typename T> __device__ __forceinline__ f32x16 mma16(const uint4& a, const uint4& b, f32x16 c);
template <> __device__ __forceinline__ f32x16 mma16<bf16_t>(const uint4& a, const uint4& b, f32x16 c) {
  union { uint4 u; s16x8 s; } ua, ub; ua.u = a; ub.u = b;
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, ua.s),
                                                 __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, ub.s), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16 mma16<f16_t>(const uint4& a, const uint4& b, f32x16 c) {
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16 mma16<float>(const uint4& a, const uint4& b, f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  return c;
}
// acc + <a, b> over one 16-byte chunk pair (bf16 / f16: v_dot2_f32_*; f32: fma chain)
template <typename T> __device__ __forceinline__ float dot16(const uint4& a, const uint4& b, float acc);
template <> __device__ __forceinline__ float dot16<float>(const uint4& a, const uint4& b, float acc) {
  acc += __uint_as_float(a.x) * __uint_as_float(b.x);
  acc += __uint_as_float(a.y) * __uint_as_float(b.y);
  acc += __uint_as_float(a.z) * __uint_as_float(b.z);
  acc += __uint_as_float(a.w) * __uint_as_float(b.w);
  return acc;
}
template <> __device__ __forceinline__ float dot16<f16_t>(const uint4& a, const uint4& b, float acc) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a.x), __builtin_bit_cast(h2, b.x), acc, false);   // v_dot2_f32_f16
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a.y), __builtin_bit_cast(h2, b.y), acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a.z), __builtin_bit_cast(h2, b.z), acc, false);
  acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a.w), __builtin_bit_cast(h2, b.w), acc, false);
  return acc;
}
template <> __device__ __forceinline__ float dot16<bf16_t>(const uint4& a, const uint4& b, float acc) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a.x), __builtin_bit_cast(bf2, b.x), acc, false);   // v_dot2_f32_bf16
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a.y), __builtin_bit_cast(bf2, b.y), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a.z), __builtin_bit_cast(bf2, b.z), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a.w), __builtin_bit_cast(bf2, b.w), acc, false);
  return acc;
}
// 1.0 in T replicated over a 16-byte chunk (sum of a chunk = dot16(chunk, ones))
template <typename T> __device__ __forceinline__ uint4 ones16();
template <> __device__ __forceinline__ uint4 ones16<float>() { return make_uint4(0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u); }
template <> __device__ __forceinline__ uint4 ones16<bf16_t>() { return make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u); }
template <> __device__ __forceinline__ uint4 ones16<f16_t>() { return make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u); }

// k-values consumed by one mma16 step
template <typename T> struct MmaK { static constexpr int value = 2 * TT<T>::VEC; };

// C/D layout of the 32x32 MFMA: lane l, reg r -> col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)
__device__ __forceinline__ int mfma_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// dtype dispatch: binds the element type to `T` and runs the statement(s); unknown dtype -> EMO_ERR_BAD_DTYPE
#define EMO_DISPATCH(dtype, who, ...)                                                              \
  do {                                                                                             \
    switch (dtype) {                                                                               \
      case EMO_F32: { using T = float; __VA_ARGS__; } break;                                      \
      case EMO_BF16: { using T = bf16_t; __VA_ARGS__; } break;                                    \
      case EMO_F16: { using T = f16_t; __VA_ARGS__; } break;                                      \
      default: return emo_fail(EMO_ERR_BAD_DTYPE, "%s: dtype %d", who, (int)(dtype));             \
    }                                                                                              \
  } while (0)
static inline bool emo_dtype_ok(int dtype) { return dtype == EMO_F32 || dtype == EMO_BF16 || dtype == EMO_F16; }
static inline int emo_dtype_vec(int dtype) { return dtype == EMO_F32 ? 4 : 8; }

// x * sigmoid(x) with the hardware reciprocal (1 ulp) instead of an IEEE division (~10 VALU slots): the GroupNorm + SiLU apply
// pass spends ~40 % of its time in VALU at 8 elements per 16-byte load
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// erf-GELU for the bf16 epilogues: Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7, far below a bf16 ulp),
// ~12 VALU ops instead of ocml erff's ~40 - the GEGLU epilogue is VALU-bound otherwise.
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __frcp_rn(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * __expf(-z * z);
  const float erf_v = copysignf(erf_abs, x);
  return 0.5f * x * (1.0f + erf_v);
}
template <typename T> __device__ __forceinline__ float gelu_for(float x) {
  if constexpr (sizeof(T) == 2) return gelu_erf_fast(x);
  else return gelu_erf_f(x);
}
// v * gelu(x) for two (value, gate) pairs - the GEGLU product of the bf16 epilogues - without transcendentals:
// gelu(x) = x * (0.5 + E(x)), E(x) = 0.5 erf(x / sqrt 2) ~ x * P(x^2) on |x| <= 4 (degree-6 minimax in x^2, fitted to the gelu
// error: |error| <= 1.9e-4 absolute, a bf16 ulp at 0.05), clamped to +-0.5.  Beyond the fitted range x * P(x^2) keeps growing
// (>= 0.50001 at 4, monotone, +-inf at overflow), so the clamp alone gives gelu -> x / 0 exactly in the tails: no |x|, no input
// clamp, and the product v * x * (0.5 + E) stays in packed f32 math - 16 issue slots per pair with the rounding (ocml erff: ~40
// per VALUE; the rcp + exp form ~23; the first polynomial version, |x| * E(|x|) + x / 2 and a separate multiply: 21 per pair).
// The 256x256 GEGLU epilogue is VALU-bound: 287 -> 266-273 us at M=98304 N=2560 K=320 (profiles/r04q_geglu_form.txt).
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void geglu_poly2(float va, float vb, float xa, float xb, float& oa, float& ob) {
  const v2f_t x = {xa, xb}, v = {va, vb};
  const v2f_t u = x * x;
  v2f_t p = {2.2781060593501935e-08f, 2.2781060593501935e-08f};
  p = p * u + v2f_t{-1.598583253326069e-06f, -1.598583253326069e-06f};
  p = p * u + v2f_t{4.7955145419109613e-05f, 4.7955145419109613e-05f};
  p = p * u + v2f_t{-8.140119025483727e-04f, -8.140119025483727e-04f};
  p = p * u + v2f_t{8.772371336817741e-03f, 8.772371336817741e-03f};
  p = p * u + v2f_t{-6.457307189702988e-02f, -6.457307189702988e-02f};
  p = p * u + v2f_t{3.978833258152008e-01f, 3.978833258152008e-01f};
  const v2f_t e = x * p;
  const v2f_t s = v2f_t{__builtin_amdgcn_fmed3f(e.x, -0.5f, 0.5f), __builtin_amdgcn_fmed3f(e.y, -0.5f, 0.5f)} + v2f_t{0.5f, 0.5f};
  const v2f_t o = (v * x) * s;
  oa = o.x; ob = o.y;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [0, N) - keeps every array index a constant
// expression (runtime-indexed register arrays are demoted to scratch memory by hipcc)
template <int... Is, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f)); }

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
