// valu_rates.hip - issue cost (cycles per wave-instruction) of the VALU ops the attention softmax is made of, on gfx950:
//   hipcc --offload-arch=gfx950 -O3 tools/bench/micro/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
// One block of W waves per CU-SIMD-set; each wave runs N x 16 independent instructions of one kind; s_memtime around the loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP16(X) X X X X X X X X X X X X X X X X

template <int KIND>
__global__ void k(float* out, uint64_t* cyc, int iters, float seed) {
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = seed + threadIdx.x * 1e-3f + i;
  float w0 = seed, w1 = seed * 0.5f;
  typedef float v2 __attribute__((ext_vector_type(2)));
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
  f32x16 acc = {0};
  bf8 a = {0}, b = {0};
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if constexpr (KIND == 0) {          // v_exp_f32 x16
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
    } else if constexpr (KIND == 1) {   // v_fma_f32 x16
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(w0), "v"(w1));
    } else if constexpr (KIND == 2) {   // v_pk_fma_f32 x8 (16 values)
#pragma unroll
      for (int i = 0; i < 16; i += 2) { v2 x = {v[i], v[i + 1]}; v2 c = {w0, w0}, d = {w1, w1};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d)); v[i] = x.x; v[i + 1] = x.y; }
    } else if constexpr (KIND == 3) {   // v_max3_f32 x16
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(w0), "v"(w1));
    } else if constexpr (KIND == 4) {   // v_cvt_pk_bf16_f32 x16
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w0));
    } else if constexpr (KIND == 5) {   // v_exp_f16 x16
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_exp_f16 %0, %0" : "+v"(v[i]));
    } else if constexpr (KIND == 6) {   // 8 exp + 8 fma interleaved (co-issue?)
#pragma unroll
      for (int i = 0; i < 16; i += 2) { asm volatile("v_exp_f32 %0, %0" : "+v"(v[i])); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i + 1]) : "v"(w0), "v"(w1)); }
    } else if constexpr (KIND == 7) {   // 4 MFMA 32x32x16 alone
#pragma unroll
      for (int i = 0; i < 4; i++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    } else if constexpr (KIND == 8) {   // 4 MFMA each followed by 4 exp (does the exp ride under the MFMA?)
#pragma unroll
      for (int i = 0; i < 4; i++) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        asm volatile("v_exp_f32 %0, %0" : "+v"(v[4 * i])); asm volatile("v_exp_f32 %0, %0" : "+v"(v[4 * i + 1]));
        asm volatile("v_exp_f32 %0, %0" : "+v"(v[4 * i + 2])); asm volatile("v_exp_f32 %0, %0" : "+v"(v[4 * i + 3])); }
    } else if constexpr (KIND == 9) {   // 4 MFMA each followed by 7 fma
#pragma unroll
      for (int i = 0; i < 4; i++) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 7; j++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(4 * i + j) & 15]) : "v"(w0), "v"(w1)); }
    } else if constexpr (KIND == 10) {  // v_mul_f32 x16
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w0));
    } else if constexpr (KIND == 11) {  // v_rcp_f32 x16
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += v[i];
  for (int i = 0; i < 16; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND> void run(const char* name, int n_instr, int waves_per_simd) {
  const int threads = 256 * waves_per_simd, iters = 2000;
  float* out; uint64_t* cyc;
  hipMalloc(&out, 256 * threads * 4); hipMalloc(&cyc, 256 * 64 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND><<<256, threads>>>(out, cyc, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<KIND><<<256, threads>>>(out, cyc, iters, 1.0f);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  uint64_t h[4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  // wall clock at ~2.4 GHz -> SIMD cycles per instruction (each SIMD runs waves_per_simd waves)
  const double cyc_wall = ms * 1e-3 * 2.4e9 / ((double)iters * n_instr * waves_per_simd);
  printf("%-34s waves/SIMD %d: %.2f SIMD-cycles per wave-instruction (wall @2.4GHz), counter/instr %.2f (100 MHz ticks x24?)\n", name, waves_per_simd, cyc_wall,
         (double)h[0] / ((double)iters * n_instr));
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 2; w++) {
    run<1>("v_fma_f32", 16, w); run<10>("v_mul_f32", 16, w); run<2>("v_pk_fma_f32 (per instr)", 8, w); run<0>("v_exp_f32", 16, w); run<5>("v_exp_f16", 16, w);
    run<11>("v_rcp_f32", 16, w); run<3>("v_max3_f32", 16, w); run<4>("v_cvt_pk_bf16_f32", 16, w); run<6>("8 exp + 8 fma interleaved (per instr)", 16, w);
    run<7>("mfma 32x32x16 bf16 alone", 4, w); run<8>("mfma + 4 exp (per group)", 4, w); run<9>("mfma + 7 fma (per group)", 4, w);
  }
  return 0;
}
