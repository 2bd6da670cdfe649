// dma_rate.hip - what one CU gets out of the LDS-DMA path (global_load_lds_dwordx4, 1 KB per wave-instruction), gfx950:
//   hipcc --offload-arch=gfx950 -O3 tools/bench/micro/dma_rate.hip -o tools/bench/micro/dma_rate && tools/bench/micro/dma_rate
// 256 blocks x 8 waves (128 KB of LDS: one block per CU, the 256x256 GEMM's occupancy).  Every wave keeps DEPTH 1-KB pieces in
// flight (request DEPTH, then per round: wait for the oldest group, request the next - the GEMM's one-stage-ahead ring is
// DEPTH = 8).  Sources: MODE 0 = the block re-reads its own 64 KB (L2 hits), 1 = GEMM-like (32 CUs of an "XCD" share a 2 MB
// panel set, walking K), 2 = streaming 4 GB (HBM).  Prints bytes / cycle / CU from s_memtime and chip-wide TB/s from events.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#define GLDS16(g, l) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g), (__attribute__((address_space(3))) void*)(l), 16, 0, 0)
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int DEPTH, int MODE>
__global__ __launch_bounds__(512, 1) void k(const unsigned char* __restrict__ src, size_t bytes, int rounds, uint64_t* cyc) {
  extern __shared__ __attribute__((aligned(128))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // piece j of wave w in round r: 1 KB = 8 rows of 128 B
  size_t base;
  size_t step;          // bytes advanced per round by this block
  if (MODE == 0) { base = (size_t)blockIdx.x * 65536; step = 0; }
  else if (MODE == 1) { base = (size_t)(blockIdx.x % 8) * (64u << 20) + (size_t)((blockIdx.x / 8) % 4) * 65536; step = 4 * 65536; }   // 4 distinct panels per XCD, shared by 8 CUs each
  else { base = (size_t)blockIdx.x * 65536; step = (size_t)gridDim.x * 65536; }
  const unsigned char* p = src + base + (size_t)wave * 8192 + lane * 16;
  unsigned char* l = lds + wave * 8192;
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  size_t off = 0;
#pragma unroll
  for (int j = 0; j < DEPTH; j++) GLDS16(p + (off + (size_t)j * 1024) % bytes, l + (j % 8) * 1024);
  for (int r = 1; r < rounds; r++) {
    off += step;
    if (MODE != 0 && base + off + 65536 > bytes) off = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) GLDS16(p + off + (size_t)j * 1024, l + (j % 8) * 1024 + (r & 1) * 65536);
    wait_vm<DEPTH>();
    if (DEPTH <= 8) __builtin_amdgcn_s_barrier();   // the GEMM's per-stage barrier
  }
  wait_vm<0>();
  const uint64_t t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int DEPTH, int MODE> void run(const unsigned char* d, size_t bytes, uint64_t* dc, const char* name) {
  const int rounds = 2000, grid = 256;
  hipFuncSetAttribute((const void*)k<DEPTH, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<DEPTH, MODE><<<grid, 512, 131072>>>(d, bytes, 50, dc);
  hipEventRecord(e0);
  k<DEPTH, MODE><<<grid, 512, 131072>>>(d, bytes, rounds, dc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  uint64_t hc[256]; hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
  double c = 0; for (int i = 0; i < grid; i++) c += hc[i]; c /= grid;
  const double per_block = (double)rounds * 65536.0;
  printf("%-34s depth %2d: %6.1f B/clk/CU (s_memtime ticks: %.0f per 64 KB round)  %6.2f TB/s chip-wide\n", name, DEPTH, per_block / c, c / rounds,
         per_block * grid / (ms * 1e-3) / 1e12);
}

int main() {
  const size_t bytes = (size_t)4 << 30;
  unsigned char* d; hipMalloc(&d, bytes); hipMemset(d, 1, bytes);
  uint64_t* dc; hipMalloc(&dc, 256 * sizeof(uint64_t));
  run<8, 0>(d, bytes, dc, "own 64 KB (L2 hit)");
  run<16, 0>(d, bytes, dc, "own 64 KB (L2 hit)");
  run<8, 1>(d, bytes, dc, "shared panels (GEMM-like)");
  run<16, 1>(d, bytes, dc, "shared panels (GEMM-like)");
  run<8, 2>(d, bytes, dc, "streaming 4 GB (HBM)");
  run<16, 2>(d, bytes, dc, "streaming 4 GB (HBM)");
  return 0;
}
