// gemm8p.hip - a standalone 256x256 bf16 GEMM main loop in the PHASE structure of the guide's 8-phase template
// (cdna_hip_programming.md "The 256^2 8-phase template"), to measure what that structure is worth on this part against
// the product's k-step loop (gemm_impl.h) BEFORE re-basing the product on it.  C[M][N] = A[M][K] . W[N][K]^T, bf16 in,
// f32 accumulate, bf16 out; M, N multiples of 256, K a multiple of 64.  Not product code.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bench/micro/gemm8p.hip -o tools/bench/micro/gemm8p
//   tools/bench/micro/gemm8p [M N K]...
//
// Structure: 8 waves = 2 (M) x 4 (N), a wave owns 128 x 64 of C as 8 x 4 fragments of v_mfma_f32_16x16x32_bf16.
// A K-tile (64 k) is 4 phases = the wave's 4 quadrants of 64 x 32: (A0,B0) (A0,B1) (A1,B1) (A1,B0).  A phase is
//     LOAD section: this phase's fragment reads (12 / 4 / 8 / 0 ds_read_b128) + 2 LDS-DMA requests of a later K-tile
//     s_barrier ; s_waitcnt lgkmcnt(0)
//     MFMA section: 16 MFMAs, nothing else
//     s_barrier
// and the two wave groups (wr = 0 / 1: one wave per SIMD each) run ONE barrier apart: while group 0 multiplies, group 1
// loads, and vice versa - the matrix pipe always has exactly one wave per SIMD with all its operands in registers.
// LDS: 2 buffers x (A 256 rows + W 256 rows) x 128 B = 128 KB, lane-linear 1 KB DMA pieces of 8 rows, 16-byte chunk
// position XOR (row / 2) & 7 on the source address and on the fragment read (conflict-free for the 16x16x32 lane groups).
// Restaging follows the LAST read of each quarter of a buffer (barrier numbers in the comments of the loop):
//     A0 rows, B0 rows: last read in phase 1 -> requested for K-tile k+2 in phases 3 and 4 of K-tile k
//     B1 rows         : last read in phase 2 -> requested in phase 1 of K-tile k+1
//     A1 rows         : last read in phase 3 -> requested in phase 2 of K-tile k+1
// and phase 4 of K-tile k+1 waits for all of K-tile k+2 with vmcnt(4) (the two newer pieces belong to K-tile k+3).
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;

#define GLDS16(g, l) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g), (__attribute__((address_space(3))) void*)(l), 16, 0, 0)
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint4 lds_read16(unsigned addr) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
__device__ __forceinline__ f32x4 mma(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

static constexpr int BM = 256, BN = 256, BKB = 128;          // K-tile: 128 bytes = 64 bf16
static constexpr int A_BYTES = BM * BKB, BUF_BYTES = (BM + BN) * BKB, LDS_BYTES = 2 * BUF_BYTES;

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

template <bool STAGGER>
__global__ __launch_bounds__(512, 1) void gemm8p_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, bf16_t* __restrict__ C,
                                                        int M, int N, int K, int gm) {
  extern __shared__ __attribute__((aligned(128))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  // ---- tile of this block: XCD-aware order (block b runs on XCD b % 8; an XCD owns a contiguous run of tiles), bands of gm
  // tile rows with m fastest inside a band (the product's order, gemm_impl.h)
  const int tiles_m = M / BM, tiles_n = N / BN, tiles_all = tiles_m * tiles_n;
  int t;
  {
    const int i = blockIdx.x, qn = tiles_all >> 3, rn = tiles_all & 7, x = i & 7, idx = i >> 3;
    t = (x < rn ? x * (qn + 1) : rn * (qn + 1) + (x - rn) * qn) + idx;
  }
  int tm, tn;
  if (gm <= 1) { tm = t / tiles_n; tn = t - tm * tiles_n; }
  else {
    const int per = gm * tiles_n, g = t / per, r = t - g * per, m0 = g * gm;
    const int gsz = tiles_m - m0 < gm ? tiles_m - m0 : gm;
    tn = r / gsz; tm = m0 + r - tn * gsz;
  }
  const int64_t bm = (int64_t)tm * BM;
  const int bn = tn * BN;

  // ---- loader: a quarter (A0 / A1 / B0 / B1) is 128 rows = 16 pieces of 8 rows; wave w requests pieces 2w, 2w+1.
  //   A0: rows g*128 + r (g = 0, 1; r < 64)    A1: + 64          (group g's first / second 64 rows)
  //   B0: rows c*64 + r (c = 0..3; r < 32)     B1: + 32          (wave column c's first / second 32 columns)
  const int lrow = lane >> 3, lpos = lane & 7;
  const bf16_t* src[4][2];     // [quarter][piece]: source of this lane's 16 bytes at K-tile 0
  unsigned dst[4][2];          // LDS byte offset inside a buffer
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int pr = (2 * wave + j) * 8 + lrow;                     // row inside the quarter, 0..127
    const int ra0 = (pr >> 6) * 128 + (pr & 63), rb0 = (pr >> 5) * 64 + (pr & 31);
    const int rows[4] = {ra0, ra0 + 64, rb0, rb0 + 32};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int r = rows[q];
      const int c = lpos ^ swz(r);                                // logical 16-byte chunk stored at position lpos
      src[q][j] = q < 2 ? A + (bm + r) * (int64_t)K + c * 8 : W + (int64_t)(bn + r) * K + c * 8;
      dst[q][j] = (q < 2 ? 0 : A_BYTES) + (r - lrow) * BKB + lane * 16;   // piece base (8 rows) + lane-linear
    }
  }
  auto request = [&](auto Q, int kt, int buf) {    // quarter Q of K-tile kt into buffer buf (2 pieces per wave)
    constexpr int q = decltype(Q)::value;
#pragma unroll
    for (int j = 0; j < 2; j++) GLDS16(src[q][j] + (int64_t)kt * 64, lds + buf * BUF_BYTES + dst[q][j]);
  };

  // ---- fragment addresses: lane (i = lane % 16, kb = lane / 16) reads row (base + i), chunk ks * 4 + kb
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  const int fi = lane & 15, kb = lane >> 4;
  unsigned fa[8], fb[4];   // byte offsets inside a buffer at ks = 0: A row fragments 0..7 (16 rows each), W row fragments 0..3
  int ka[8], kbk[4];       // swizzle keys
#pragma unroll
  for (int f = 0; f < 8; f++) { const int r = wr * 128 + f * 16 + fi; fa[f] = r * BKB; ka[f] = swz(r); }
#pragma unroll
  for (int f = 0; f < 4; f++) { const int r = wc * 64 + f * 16 + fi; fb[f] = A_BYTES + r * BKB; kbk[f] = swz(r); }

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K / 64;
  // ---- prologue: K-tiles 0 and 1 complete, then the barrier that publishes them; group 1 then falls one barrier behind
  request(std::integral_constant<int, 0>{}, 0, 0); request(std::integral_constant<int, 2>{}, 0, 0);
  request(std::integral_constant<int, 3>{}, 0, 0); request(std::integral_constant<int, 1>{}, 0, 0);
  if (nk > 1) {
    request(std::integral_constant<int, 0>{}, 1, 1); request(std::integral_constant<int, 2>{}, 1, 1);
    request(std::integral_constant<int, 3>{}, 1, 1); request(std::integral_constant<int, 1>{}, 1, 1);
  }
  wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  if (STAGGER && wr == 1) __builtin_amdgcn_s_barrier();

  uint4 ra[4][2], rb0[2][2], rb1[2][2];   // A half (4 row fragments x 2 k-steps), B0 / B1 (2 fragments x 2 k-steps)
  auto read_a = [&](unsigned st, int half_) {
#pragma unroll
    for (int f = 0; f < 4; f++)
#pragma unroll
      for (int ks = 0; ks < 2; ks++) ra[f][ks] = lds_read16(st + fa[half_ * 4 + f] + (((ks * 4 + kb) ^ ka[half_ * 4 + f]) << 4));
  };
  auto read_b = [&](unsigned st, int half_, uint4 (&r)[2][2]) {
#pragma unroll
    for (int f = 0; f < 2; f++)
#pragma unroll
      for (int ks = 0; ks < 2; ks++) r[f][ks] = lds_read16(st + fb[half_ * 2 + f] + (((ks * 4 + kb) ^ kbk[half_ * 2 + f]) << 4));
  };
  auto mfma_quadrant = [&](int ah, int bh, const uint4 (&b)[2][2]) {   // 16 MFMAs: rows ah*4..+4, columns bh*2..+2 (acc = W_frag x A_frag)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int f = 0; f < 4; f++)
#pragma unroll
        for (int g = 0; g < 2; g++) acc[ah * 4 + f][bh * 2 + g] = mma(b[g][ks], ra[f][ks], acc[ah * 4 + f][bh * 2 + g]);
    __builtin_amdgcn_s_setprio(0);
  };
  // barrier numbers: group 0's load section of phase p of K-tile k ends at barrier 8k + 2(p-1), its MFMA section at +1;
  // group 1 is one barrier later.  A quarter's reads are complete (lgkmcnt(0)) right behind the barrier that ends the load
  // section, i.e. for BOTH groups before barrier 8k + 2p; a request into that quarter's place is issued behind barrier 8k + 2p + 1.
  for (int kt = 0; kt < nk; kt++) {
    const int buf = kt & 1;
    const unsigned st = lds_base + buf * BUF_BYTES;
    const bool more2 = kt + 2 < nk, more1 = kt + 1 < nk;
    // ---- phase 1: (A0, B0); request B1 of K-tile kt+1 (its place was last read in phase 2 of K-tile kt-1)
    read_b(st, 0, rb0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(st, 0);
    if (more1 && kt >= 1) request(std::integral_constant<int, 3>{}, kt + 1, buf ^ 1);
    __builtin_amdgcn_s_barrier();
    wait_lgkm0();
    __builtin_amdgcn_sched_barrier(0);
    mfma_quadrant(0, 0, rb0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 2: (A0, B1); request A1 of K-tile kt+1 (last read in phase 3 of K-tile kt-1)
    read_b(st, 1, rb1);
    if (more1 && kt >= 1) request(std::integral_constant<int, 1>{}, kt + 1, buf ^ 1);
    __builtin_amdgcn_s_barrier();
    wait_lgkm0();
    __builtin_amdgcn_sched_barrier(0);
    mfma_quadrant(0, 1, rb1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 3: (A1, B1); request A0 of K-tile kt+2 (last read in phase 1 of THIS K-tile: both groups are past barrier 8k+2)
    read_a(st, 1);
    if (more2) request(std::integral_constant<int, 0>{}, kt + 2, buf);
    __builtin_amdgcn_s_barrier();
    wait_lgkm0();
    __builtin_amdgcn_sched_barrier(0);
    mfma_quadrant(1, 1, rb1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 4: (A1, B0 from registers); request B0 of K-tile kt+2; K-tile kt+1 must be complete for phase 1 of kt+1:
    // its newest pieces are B1 (phase 1) and A1 (phase 2) of this K-tile - everything but the 2 x 2 pieces of K-tile kt+2
    if (more2) { request(std::integral_constant<int, 2>{}, kt + 2, buf); wait_vm<4>(); }
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    mfma_quadrant(1, 0, rb0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
  if (STAGGER && wr == 0) __builtin_amdgcn_s_barrier();

  // ---- epilogue (plain): acc[i][j] = D[n][m], lane holds n = (lane / 16) * 4 + r of row m = lane % 16 -> one 8-byte store
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int64_t m = bm + wr * 128 + i * 16 + fi;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int n = bn + wc * 64 + j * 16 + kb * 4;
      *(uint2*)(C + m * N + n) = make_uint2(pack2(acc[i][j][0], acc[i][j][1]), pack2(acc[i][j][2], acc[i][j][3]));
    }
  }
}

// ------------------------------------------------------------------------------------------------ host
static inline bf16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); }
static inline float bf2f(bf16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <bool ST> static float time_kernel(const bf16_t* A, const bf16_t* W, bf16_t* C, int M, int N, int K, int gm, int iters) {
  HIPCHK(hipFuncSetAttribute((const void*)gemm8p_kernel<ST>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  const int grid = (M / BM) * (N / BN);
  gemm8p_kernel<ST><<<grid, 512, LDS_BYTES>>>(A, W, C, M, N, K, gm);
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipEventRecord(e0));
  for (int i = 0; i < iters; i++) gemm8p_kernel<ST><<<grid, 512, LDS_BYTES>>>(A, W, C, M, N, K, gm);
  HIPCHK(hipEventRecord(e1)); HIPCHK(hipEventSynchronize(e1));
  float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

int main(int argc, char** argv) {
  std::vector<int> shapes;
  for (int i = 1; i + 2 < argc; i += 3) { shapes.push_back(atoi(argv[i])); shapes.push_back(atoi(argv[i + 1])); shapes.push_back(atoi(argv[i + 2])); }
  if (shapes.empty()) shapes = {8192, 8192, 8192, 4096, 4096, 4096, 98304, 2560, 320, 24576, 5120, 640, 6144, 10240, 1280, 98304, 1024, 320, 24576, 1280, 3200};
  for (size_t s = 0; s + 2 < shapes.size(); s += 3) {
    const int M = shapes[s], N = shapes[s + 1], K = shapes[s + 2];
    if (M % 256 || N % 256 || K % 64) { printf("skip %d %d %d\n", M, N, K); continue; }
    std::vector<bf16_t> hA((size_t)M * K), hW((size_t)N * K);
    uint32_t seed = 12345u + (uint32_t)s;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 32768.0f - 1.0f; };   // uniform [-1, 1)
    for (auto& v : hA) v = f2bf(rnd());
    for (auto& v : hW) v = f2bf(rnd());
    bf16_t *dA, *dW, *dC;
    HIPCHK(hipMalloc(&dA, hA.size() * 2)); HIPCHK(hipMalloc(&dW, hW.size() * 2)); HIPCHK(hipMalloc(&dC, (size_t)M * N * 2));
    HIPCHK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    const int tiles_n = N / BN, gm = tiles_n > 12 ? 4 : (tiles_n >= 5 ? 4 : 1);
    // correctness: sampled outputs against a double-precision dot product (transposes and K order would show)
    std::vector<bf16_t> hC((size_t)M * N);
    double worst[2] = {0, 0};
    for (int st = 0; st < 2; st++) {
      HIPCHK(hipMemset(dC, 0xff, (size_t)M * N * 2));
      if (st) time_kernel<true>(dA, dW, dC, M, N, K, gm, 1); else time_kernel<false>(dA, dW, dC, M, N, K, gm, 1);
      HIPCHK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
      uint32_t s2 = 777;
      for (int it = 0; it < 4000; it++) {
        s2 = s2 * 1664525u + 1013904223u; const int m = (int)((s2 >> 4) % (uint32_t)M);
        s2 = s2 * 1664525u + 1013904223u; const int n = (int)((s2 >> 4) % (uint32_t)N);
        double ref = 0;
        for (int k = 0; k < K; k++) ref += (double)bf2f(hA[(size_t)m * K + k]) * (double)bf2f(hW[(size_t)n * K + k]);
        const double got = bf2f(hC[(size_t)m * N + n]);
        const double err = fabs(got - ref) / (fabs(ref) + 0.05 * sqrt((double)K));
        if (!(err <= worst[st])) worst[st] = err;     // (NaN sticks)
      }
    }
    const int iters = K >= 4096 ? 20 : 50;
    const float t0 = time_kernel<false>(dA, dW, dC, M, N, K, gm, iters), t1 = time_kernel<true>(dA, dW, dC, M, N, K, gm, iters);
    const float t0b = time_kernel<false>(dA, dW, dC, M, N, K, gm, iters), t1b = time_kernel<true>(dA, dW, dC, M, N, K, gm, iters);
    const double fl = 2.0 * M * N * K;
    printf("M=%6d N=%5d K=%5d  lockstep %8.1f / %8.1f us (%6.0f TF/s)   staggered %8.1f / %8.1f us (%6.0f TF/s)   rel err %.3g / %.3g\n", M, N, K, t0 * 1e3,
           t0b * 1e3, fl / (t0b * 1e-3) / 1e12, t1 * 1e3, t1b * 1e3, fl / (t1b * 1e-3) / 1e12, worst[0], worst[1]);
    fflush(stdout);
    HIPCHK(hipFree(dA)); HIPCHK(hipFree(dW)); HIPCHK(hipFree(dC));
  }
  return 0;
}
