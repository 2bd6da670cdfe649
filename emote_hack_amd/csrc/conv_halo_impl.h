// conv_halo_impl.h - the halo-reuse 3x3 convolution kernel (stride 1, pad 1) and its launcher; instantiated per element type in
// conv_halo_{f32,bf16,f16}.hip.  Shares the LDS swizzle, the MFMA step and the staged epilogue with the GEMM (gemm_impl.h).
#pragma once
#include "gemm_impl.h"

__device__ __forceinline__ void lds_write16(unsigned addr, const uint4& v) {
  const u32x4_t x = {v.x, v.y, v.z, v.w};
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(x) : "memory");
}
// The LDS reads of this file are inline asm with hand-counted lgkmcnt waits, and hipcc knows nothing of the latency between the two:
// a compiler-generated instruction that consumes a read's result - or a register COPY of it, which the allocator is free to insert
// right behind the read - may land in front of the s_waitcnt asm (it depends on the read, not on the wait) and then works on whatever
// the registers held (seen: v_mov of a ds_read's output three instructions ahead of its wait - profiles/r05*_gn_conv.txt).  Two
// shapes are safe and used below: (a) read and wait in ONE asm statement, for code off the critical path; (b) the first consumers of
// an early read are themselves asm volatile statements taking the registers as plain inputs - volatile asms keep their order, so
// they sit behind the wait, and an input operand needs no copy.
__device__ __forceinline__ uint4 lds_read16_sync(unsigned addr) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  return v;
}
// (b): x0, x1 = the two elements of packed word w as f32; t = x * a + b (one correctly rounded fma, like the compiler's contraction)
template <typename T> __device__ __forceinline__ void unpack2_asm(unsigned w, float& x0, float& x1);
template <> __device__ __forceinline__ void unpack2_asm<bf16_t>(unsigned w, float& x0, float& x1) {
  asm volatile("v_lshlrev_b32 %0, 16, %2\n\tv_and_b32 %1, 0xffff0000, %2" : "=&v"(x0), "=&v"(x1) : "v"(w));
}
template <> __device__ __forceinline__ void unpack2_asm<f16_t>(unsigned w, float& x0, float& x1) {
  asm volatile("v_cvt_f32_f16 %0, %2\n\tv_lshrrev_b32 %1, 16, %2\n\tv_cvt_f32_f16 %1, %1" : "=&v"(x0), "=&v"(x1) : "v"(w));
}
template <> __device__ __forceinline__ void unpack2_asm<float>(unsigned w, float& x0, float& x1) { x0 = __uint_as_float(w); x1 = 0.f; }
__device__ __forceinline__ float fma_asm(float x, unsigned a, unsigned b) {
  float t;
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(x), "v"(a), "v"(b));
  return t;
}

// ------------------------------------------------------------------------------------------ 3x3 conv, halo reuse
// Stride-1 3x3 convolution (resnet.py:30-38 - 44 of the 54 convs of a UNet pass) without the 9x re-read of the im2col
// loader.  A block owns an 8x16 patch of output pixels of one frame (= 128 GEMM rows) x 128 output channels.  K runs
// channel-chunk major: for every 128-byte channel chunk the (8+2)x(16+2) input HALO of the patch is brought into LDS ONCE
// (23 KB instead of 9 x 16 KB of im2col rows) and the 9 taps read their A fragments from it at shifted pixel positions;
// only the weight tile (128 x 128 B per tap) streams per stage.  LDS-DMA traffic per MFMA drops by ~40 % - the
// direct-to-LDS path (~9 TB/s chip-wide measured) is what bounds the im2col kernel.
//   * LDS: 2 halo buffers (chunk c+1 arrives in 6 pieces during the first 6 taps of chunk c) + a 2-deep weight ring
//     = 78 KB -> 2 blocks per CU.  LDS "rows" of the halo are halo pixels; same XOR chunk swizzle as the GEMM.
//   * persistent blocks, continuous loader stream across tiles, weights one stage ahead (requested right behind the
//     barrier), vmcnt(0) + one s_barrier per tap-stage.
//   * epilogue = the GEMM's row-major fused epilogue (bias, temb row bias, residual), rows mapped through the patch.
// Needs Cin % (128 B of channels) == 0, H >= 8, W >= 16 (frames that are not whole patches: overlapped last patch, see tpx below);
// everything else stays on the im2col loader.
// PH_ = patch height: 8 (4 waves, 128 output pixels, 78 KB of LDS, 2 blocks per CU) or 16 (8 waves, 256 pixels, 114 KB,
// 1 block per CU).  The weight tile is the larger stream: 16 KB per tap-stage against 2.6 KB of halo (PH 8) - the 16-row
// patch feeds twice the MFMAs from the same weights, 4.9 KB of LDS-DMA traffic per MFLOP instead of 8.9 (the direct-to-LDS
// path, ~9 TB/s chip-wide, is what the 8-row kernel sits on at 1000-1050 TFLOP/s).
// BN_ = output channels per block: 128, or 64 for the REMAINDER columns of a width that is an odd multiple of 64 (N = 320: two
// 128-column tiles + one of 64 - a third 128-column tile computed 64 columns of zeros, 17 % of the launch; host side: gemm.hip).
// Round 6: with 16-row patches (one block per CU either way: 82 KB of halo buffers) a 192-column block fits - 131 KB of LDS, 96
// accumulator registers (223 in all) - and N = 320 can run as 128 + 192 columns, two launches of full-width blocks instead of three
// with a half-width tail; it pays where a launch of one block per patch is a single round (gemm.hip plans it, profiles/r06k)
template <int PH_, int BN_ = HaloGeom::BN> struct HaloT {
  static constexpr int PH = PH_, PW = HaloGeom::PW, NW = PH_ / 2, HW_ = PW + 2, HPIX = (PH + 2) * (PW + 2);   // 180 / 324 halo pixels
  static constexpr int PIECES = (HPIX + 7) / 8;                 // 1 KB glds pieces of 8 pixels: 23 / 41
  static constexpr int LH = (PIECES + NW - 1) / NW;             // pieces per wave: 6 (the last round is partial)
  static constexpr int HALO_BYTES = PIECES * 1024;              // 23 / 41 KB
  static constexpr int BN = BN_, LB = BN / (8 * NW);            // weight tile rows, glds per wave per stage
  static_assert(BN_ == 128 || BN_ == 64 || (BN_ == 192 && PH_ == 16), "halo conv: 128 or 64 output channels per block (192 with 16-row patches)");
  static constexpr int B_BYTES = BN * KBYTES;                   // 16 KB
  static constexpr int B_OFF = 2 * HALO_BYTES;
  static constexpr int LDS_BYTES = 2 * HALO_BYTES + 2 * B_BYTES;   // 79872 / 116736
  static constexpr int GN_OFF = LDS_BYTES, GN_BYTES = 1024;        // GN variant: the chunk's scales | shifts behind the weight ring
  static_assert(LH == 6, "the halo pieces ride on taps 0..5");
};

// GN: GroupNorm (+ SiLU) of the INPUT applied to the halo image in LDS (emo_gemm_params.gn_coef; resnet.py:180-183,
// 191-196 norm -> nonlinearity -> conv): A holds the raw producer output.  Piece i of a chunk's halo is requested on tap i and has
// landed behind tap i+1's vmcnt(0); the lane that requested a 16-byte slot (one halo pixel x 8 channels) reads it back, applies
// x * scale_c + shift_c and SiLU in f32, rounds and writes it in place - no cross-wave hand-off, the barrier of the chunk's first
// tap publishes it.  Slots of pad pixels (zero page) stay zero: the conv pads the NORMALISED tensor.  The XOR swizzle key of a lane's
// slots is the same for all six pieces ((piece & 1) == (wave & 1)): a lane always works on the same 8 channels of a chunk, whose
// (scale, scale, shift, shift) quads wave 0 brings into LDS on tap 0 (one more direct-to-LDS request).  The arithmetic of a slot is
// spread over the tap's four k-steps, one packed pair per k-step (its coefficient quad read with the fragments), a quarter of it
// behind each MFMA: the VALU work fills issue slots the matrix pipe leaves empty (the loop uses 45-55 % of them).
template <typename T, int PH_, int BN_ = HaloGeom::BN, bool GN = false>
__global__ __launch_bounds__(32 * PH_, 2) void conv3x3_halo_kernel(const emo_gemm_params p) {
  using Halo = HaloT<PH_, BN_>;
  constexpr int V = TT<T>::VEC, BK = KBYTES / (int)sizeof(T);
  constexpr int WTM = 2, WTN = BN_ / 64, NW = Halo::NW, LH = Halo::LH, LB = Halo::LB, BN = Halo::BN;   // two wave columns of WTN x 32 channels
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wvm = wave >> 1, wvn = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;

  // nearest x2 upsampling folded into the halo loader (resnet.py:74-82: the interpolated tensor never exists): the patch
  // grid lives on the UPSAMPLED frame (He x We), a halo pixel (y, x) is read from source pixel (y >> 1, x >> 1)
  const int ups = p.upsample2x ? 1 : 0;
  const int He = p.H << ups, We = p.W_ << ups;
  // patches per frame.  A frame that is not a whole number of patches (24 x 24 at BASELINE configs[4]) gets its LAST patch column /
  // row shifted back inside the frame (x0 = We - PW): the overlapped pixels are computed by two blocks - the same arithmetic in the
  // same order, so both store the same bits - and no store needs a mask.  (The host keeps such frames off in-place epilogues.)
  const int tpx = (We + Halo::PW - 1) / Halo::PW, tpy = (He + Halo::PH - 1) / Halo::PH, tpi = tpx * tpy;
  auto patch_y0 = [&](int rem) { const int y = (rem / tpx) * Halo::PH; return y + Halo::PH > He ? He - Halo::PH : y; };
  auto patch_x0 = [&](int rem) { const int x = (rem % tpx) * Halo::PW; return x + Halo::PW > We ? We - Halo::PW : x; };
  const int tiles_m = (int)(p.M / ((int64_t)He * We)) * tpi;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_all = tiles_m * tiles_n;
  const int G = gridDim.x;
  auto tile_of = [&](int i) {
    const int qn = tiles_all >> 3, rn = tiles_all & 7, x = i & 7, idx = i >> 3;
    return (x < rn ? x * (qn + 1) : rn * (qn + 1) + (x - rn) * qn) + idx;
  };
  const int nchunks = p.Cin / BK;
  const int nk = nchunks * 9;            // tap-stages per tile

  const T* __restrict__ A = (const T*)p.A;
  const T* __restrict__ W = (const T*)p.W;
  const T* zero = (const T*)g_zero_page;
  const int lrow = lane / CPR, lchunk = lane % CPR;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;

  // ---- halo loader: piece pi = i*NW + wave covers halo pixels pi*8 .. +8 (lane -> pixel pi*8 + lane/8, 16-byte slot lane%8).
  // The swizzled channel chunk a lane fetches into its slot, lchunk ^ swz(pixel), is the SAME for all of its pieces - and for its rows
  // of the weight tile: swz(8 * piece + lrow) = ((piece & 1) * 4 + lrow / 2) and piece & 1 == wave & 1 (NW is even).
  // Requests go through buffer resources (one wave-uniform descriptor + one 32-bit lane offset + a scalar offset for the chunk /
  // tap): pad pixels and weight rows past N carry an offset outside the descriptor's range and read 0 by the hardware range check -
  // no 64-bit pointers to keep, advance or select against a zero page.  The activation descriptor spans ONE frame.
  static_assert(NW % 2 == 0, "one swizzle key per lane");
  const int klog = lchunk ^ swz(8 * (wave & 1) + lrow);
  bool h_piece[LH];
#pragma unroll
  for (int i = 0; i < LH; i++) h_piece[i] = i * NW + wave < Halo::PIECES;
  unsigned h_off[LH];                    // byte offset of the source of slot i inside the frame (chunk 0); bit 31: pad pixel
  __amdgpu_buffer_rsrc_t h_rs;
  const unsigned frame_bytes = (unsigned)((((int64_t)p.H * p.W_ - 1) * p.lda + p.Cin) * (int64_t)sizeof(T));
  int h_iter = blockIdx.x, h_c = 0, h_count = 0;   // halo loader: tile, chunk, running chunk counter (buffer = count % 2)
  // GN: validity of the slots this lane requested for the chunk in flight (bit i: piece i is an in-image pixel; latched at request
  // time - setup_halo moves on to the next tile before the last piece is transformed), which pieces were requested at all
  // (wave-uniform), and the coefficient row of the tile the loader is on
  unsigned g_ok = 0, g_req = 0;
  int64_t g_row = 0;
  auto setup_halo = [&](int iter) {
    const int tile = tile_of(iter);
    const int tm = tile / tiles_n;
    const int img = tm / tpi, rem = tm % tpi;
    if constexpr (GN) g_row = (int64_t)(img / p.gn_imgs_per_inst) * 2 * p.Cin;
    h_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (int64_t)img * p.H * p.W_ * p.lda), 0, (int)frame_bytes, 0x00020000);
    const int y0 = patch_y0(rem) - 1, x0 = patch_x0(rem) - 1;
#pragma unroll
    for (int i = 0; i < LH; i++) {
      const int hp = (i * NW + wave) * 8 + lrow, hy = hp / Halo::HW_, hx = hp - hy * Halo::HW_;
      const int iy = y0 + hy, ix = x0 + hx;
      const bool ok = hp < Halo::HPIX && iy >= 0 && iy < He && ix >= 0 && ix < We;   // (pad pixels of the last piece: hp >= HPIX)
      h_off[i] = ok ? (unsigned)(((iy >> ups) * p.W_ + (ix >> ups)) * (int)p.lda) * (unsigned)sizeof(T) + klog * 16 : 0x80000000u;
    }
  };
  auto issue_halo = [&](auto I) {
    constexpr int i = decltype(I)::value;
    if (h_piece[i]) {
      glds16_buffer(h_rs, lds + (h_count & 1) * Halo::HALO_BYTES + (i * NW + wave) * 1024, h_off[i], (unsigned)h_c * KBYTES);
      if constexpr (GN) {
        g_ok = (g_ok & ~(1u << i)) | ((h_off[i] >> 31 ^ 1u) << i);
        g_req |= 1u << i;
      }
    }
  };
  // GN: the chunk's coefficient quads (2 * BK floats) travel like everything else in this loop - one direct-to-LDS request by
  // wave 0 on tap 0 (16 bytes = one channel pair per lane; no compiler-tracked global load sits between the counted vmcnt waits);
  // the lanes read the quad of pair kk of their channel group in k-step kk of taps 1..6
  auto issue_gn = [&]() {
    if (wave == 0) {
      const float* src = p.gn_coef + g_row + (int64_t)h_c * 2 * BK + lane * 4;
      EMO_GLDS16(lane * 2 < BK ? (const void*)src : (const void*)zero, lds + Halo::GN_OFF);
    }
  };
  const unsigned g_caddr = lds_base + Halo::GN_OFF + klog * (V / 2) * 16;
  const bool gn_silu = GN && p.gn_silu != 0;
  // one whole slot at once (stream prologue and the f32 validation mode): the arithmetic of gn_apply_kernel, element by element
  auto gn_slot_now = [&](int i, unsigned buf) {
    const unsigned a = lds_base + buf * Halo::HALO_BYTES + (i * NW + wave) * 1024 + lane * 16;
    uint4 v = lds_read16_sync(a);
    float f[V];
    unpack16<T>(v, f);
    static_for<V / 2>([&](auto J) {
      constexpr int j = decltype(J)::value;
      const uint4 c = lds_read16_sync(g_caddr + j * 16);
      const float t0 = f[2 * j] * __uint_as_float(c.x) + __uint_as_float(c.z), t1 = f[2 * j + 1] * __uint_as_float(c.y) + __uint_as_float(c.w);
      f[2 * j] = gn_silu ? silu_f(t0) : t0;
      f[2 * j + 1] = gn_silu ? silu_f(t1) : t1;
    });
    const uint4 o = pack16<T>(f);
    lds_write16(a, ((g_ok >> i) & 1) ? o : make_uint4(0, 0, 0, 0));   // (the slot of a pad pixel holds zeros and keeps them)
  };
  auto advance_halo = [&]() {   // after the last piece of a chunk
    h_count++;
    if (++h_c >= nchunks) {
      h_c = 0;
      h_iter += G;
      if (h_iter < tiles_all) setup_halo(h_iter);
    }
  };

  // ---- weight loader: stage (c, t) of a tile reads W[n][t*Cin + c*BK ..+BK); the descriptor spans the tile's rows
  unsigned b_off[LB];
#pragma unroll
  for (int i = 0; i < LB; i++) b_off[i] = (unsigned)(((i * NW + wave) * (64 / CPR) + lrow) * p.K) * (unsigned)sizeof(T) + klog * 16;
  __amdgpu_buffer_rsrc_t b_rs;
  int l_iter = blockIdx.x, l_c = 0, l_t = 0;
  auto setup_b = [&](int iter) {
    const int tile = tile_of(iter);
    const int lbn = (tile % tiles_n) * BN;
    const int rows = p.N - lbn < BN ? p.N - lbn : BN;
    b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (int64_t)lbn * p.K), 0, (int)((int64_t)rows * p.K * (int64_t)sizeof(T)), 0x00020000);
  };
  auto issue_b = [&](int slot) {
    const unsigned koff = (unsigned)(l_t * p.Cin + l_c * BK) * (unsigned)sizeof(T);
#pragma unroll
    for (int i = 0; i < LB; i++) glds16_buffer(b_rs, lds + Halo::B_OFF + slot * Halo::B_BYTES + (i * NW + wave) * 1024, b_off[i], koff);
  };
  auto advance_b = [&]() {
    if (++l_t >= 9) {
      l_t = 0;
      if (++l_c >= nchunks) {
        l_c = 0;
        l_iter += G;
        if (l_iter < tiles_all) setup_b(l_iter);
      }
    }
  };

  // ---- fragment addressing
  int hp0[WTM];   // halo pixel of tap (0,0) for this lane's output pixel of MFMA tile row i
#pragma unroll
  for (int i = 0; i < WTM; i++) hp0[i] = ((wvm * WTM + i) * 2 + (l31 >> 4)) * Halo::HW_ + (l31 & 15);
  unsigned fb_off[WTN][KSTEPS];
#pragma unroll
  for (int j = 0; j < WTN; j++) {
    const int r = wvn * 32 * WTN + j * 32 + l31;
#pragma unroll
    for (int kk = 0; kk < KSTEPS; kk++) fb_off[j][kk] = Halo::B_OFF + r * KBYTES + (((kk * 2 + half) ^ swz(r)) * 16);
  }

  T* __restrict__ C = (T*)p.C;
  const T* __restrict__ R = (const T*)p.residual;
  emo_gemm_params pe = p;     // the accumulators start at bias + temb row bias: the epilogue sees neither
  pe.bias = nullptr;
  pe.rowbias = nullptr;
  const bool lds_epi = sizeof(T) == 2 && (p.N & 7) == 0 && (p.ldc & 7) == 0 && (!R || (p.ldr & 7) == 0);

  // ---- stream prologue: halo of the first chunk, weights of the first stage
  setup_halo(h_iter);
  if constexpr (GN) issue_gn();
  static_for<LH>([&](auto I) { issue_halo(I); });
  advance_halo();
  setup_b(l_iter);
  issue_b(0);
  advance_b();
  if constexpr (GN) {   // the first chunk's halo is normalised before the loop's first barrier publishes it
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    static_for<LH>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if (h_piece[i]) gn_slot_now(i, 0);
    });
    wait_lgkmcnt<0>();
  }

  int gs = 0, gc = 0;   // running tap-stage / chunk counters of the MFMA loop (weight slot = gs % 2, halo buffer = gc % 2)
  for (int c_iter = blockIdx.x; c_iter < tiles_all; c_iter += G) {
    const int c_tile = tile_of(c_iter);
    f32x16 acc[WTM][WTN];
    init_acc_bias<WTM, WTN>(acc, p.bias, (c_tile % tiles_n) * BN + wvn * 32 * WTN, half, p.N);   // zeros without a bias
    if (p.rowbias) {   // temb row bias (resnet.py:188): one row per frame, a patch lies inside one frame
      const int tm0 = c_tile / tiles_n;
      const int64_t m0 = (int64_t)(tm0 / tpi) * He * We;
      const float* rb = p.rowbias + (m0 / p.rows_per_batch) * p.ld_rowbias;
      const int wnb = (c_tile % tiles_n) * BN + wvn * 32 * WTN;
#pragma unroll
      for (int j = 0; j < WTN; j++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int n0 = wnb + j * 32 + 8 * g + 4 * half;
          if (n0 < p.N) {
            const float4 b4 = *(const float4*)(rb + n0);
#pragma unroll
            for (int i = 0; i < WTM; i++) { acc[i][j][4 * g] += b4.x; acc[i][j][4 * g + 1] += b4.y; acc[i][j][4 * g + 2] += b4.z; acc[i][j][4 * g + 3] += b4.w; }
          }
        }
    }

    for (int c = 0; c < nchunks; c++, gc++) {
      const unsigned stH = lds_base + (gc & 1) * Halo::HALO_BYTES;
      for (int t = 0; t < 9; t++, gs++) {
        wait_vmcnt<0>();                  // this stage's weights (and, at t == 0, the whole halo) have landed
        __builtin_amdgcn_s_barrier();     // ... for every wave; everyone is done with the previous stage's slot
        // next stage's weights, then one piece of the next chunk's halo (pieces 0..5 ride on taps 0..5)
        if (l_iter < tiles_all) { issue_b((gs + 1) & 1); advance_b(); }
        if constexpr (GN) {
          if (t == 0) { g_req = 0; if (h_iter < tiles_all) issue_gn(); }
        }
        if (t < LH && h_iter < tiles_all) {
          switch (t) {
            case 0: issue_halo(std::integral_constant<int, 0>{}); break;
            case 1: issue_halo(std::integral_constant<int, 1>{}); break;
            case 2: issue_halo(std::integral_constant<int, 2>{}); break;
            case 3: issue_halo(std::integral_constant<int, 3>{}); break;
            case 4: issue_halo(std::integral_constant<int, 4>{}); break;
            default: issue_halo(std::integral_constant<int, 5>{}); break;
          }
          if (t == LH - 1) advance_halo();
        }
        const unsigned stB = lds_base + (gs & 1) * Halo::B_BYTES;
        // A fragment addresses of this tap: halo pixel (y + ky, x + kx)
        const int toff = (t / 3) * Halo::HW_ + (t % 3);
        unsigned fa_base[WTM], fa_key[WTM];
#pragma unroll
        for (int i = 0; i < WTM; i++) {
          const int hp = hp0[i] + toff;
          fa_base[i] = stH + hp * KBYTES;
          fa_key[i] = swz(hp);
        }
        // GN: the slot this lane requested on the previous tap (it landed behind this tap's vmcnt(0)) is normalised during this tap
        bool g_do = false;
        unsigned g_addr = 0, g_valid = 0;
        if constexpr (GN) {
          const int ti = t > 0 ? t - 1 : 0;
          g_do = t >= 1 && t <= LH && ((g_req >> ti) & 1) != 0;
          g_addr = lds_base + ((gc + 1) & 1) * Halo::HALO_BYTES + (ti * NW + wave) * 1024 + lane * 16;
          g_valid = (g_ok >> ti) & 1;
          if constexpr (sizeof(T) == 4) {          // f32 validation mode: the whole slot at once
            if (g_do) gn_slot_now(ti, (gc + 1) & 1);
            g_do = false;
          }
        }
        constexpr int NMMA = WTM * WTN, NRD = WTM + WTN;
        auto ksteps = [&](auto GT) {
          constexpr bool G_ = decltype(GT)::value;
          uint4 fa[2][WTM], fb[2][WTN];
          uint4 g_v = make_uint4(0, 0, 0, 0), g_o = make_uint4(0, 0, 0, 0), g_q[2];
          float g_t0 = 0.f, g_t1 = 0.f, g_e0 = 0.f, g_e1 = 0.f;
          if constexpr (G_) { g_v = lds_read16(g_addr); g_q[0] = lds_read16(g_caddr); }
#pragma unroll
          for (int i = 0; i < WTM; i++) fa[0][i] = lds_read16(fa_base[i] + ((half ^ fa_key[i]) << 4));
#pragma unroll
          for (int j = 0; j < WTN; j++) fb[0][j] = lds_read16(stB + fb_off[j][0]);
          // a quarter of the arithmetic of packed pair kk of the slot: st 0 unpack + scale / shift, 1 exp, 2 reciprocal, 3 product + rounding
          auto gn_stage = [&](auto KK, auto ST) {
            constexpr int kk = decltype(KK)::value, st = decltype(ST)::value;
            if constexpr (G_ && sizeof(T) == 2) {
              if constexpr (st == 0) {
                // (asm: the first consumers of g_v / g_q - read before the k-step's lgkmcnt wait - must not be movable in front of it)
                const unsigned w = kk == 0 ? g_v.x : kk == 1 ? g_v.y : kk == 2 ? g_v.z : g_v.w;
                float x0, x1;
                unpack2_asm<T>(w, x0, x1);
                const uint4& c = g_q[kk & 1];
                g_t0 = fma_asm(x0, c.x, c.z);
                g_t1 = fma_asm(x1, c.y, c.w);
                // (pure VALU code has no place of its own in the instruction stream - hipcc's DAG linearisation gathers all of it
                // behind the k-step's last MFMA, where it waits out four back-to-back MFMA issues; an empty asm with the stage's
                // results as operands keeps every quarter behind ITS MFMA, in the issue slots the matrix pipe leaves free)
                asm volatile("" : "+v"(g_t0), "+v"(g_t1));
              } else if constexpr (st == 1) {
                g_e0 = __builtin_amdgcn_exp2f(-1.4426950408889634f * g_t0);
                g_e1 = __builtin_amdgcn_exp2f(-1.4426950408889634f * g_t1);
                asm volatile("" : "+v"(g_e0), "+v"(g_e1));
              } else if constexpr (st == 2) {
                g_e0 = __builtin_amdgcn_rcpf(1.0f + g_e0);
                g_e1 = __builtin_amdgcn_rcpf(1.0f + g_e1);
                asm volatile("" : "+v"(g_e0), "+v"(g_e1));
              } else {
                const float o0 = gn_silu ? g_t0 * g_e0 : g_t0, o1 = gn_silu ? g_t1 * g_e1 : g_t1;
                unsigned r = g_valid ? pack2<T>(o0, o1) : 0u;     // (the slot of a pad pixel holds zeros and keeps them)
                asm volatile("" : "+v"(r));
                if constexpr (kk == 0) g_o.x = r; else if constexpr (kk == 1) g_o.y = r; else if constexpr (kk == 2) g_o.z = r; else g_o.w = r;
              }
            }
          };
          static_for<KSTEPS>([&](auto KK) {
            constexpr int kk = decltype(KK)::value, cur = kk & 1, nxt = cur ^ 1;
            wait_lgkmcnt<0>();
            __builtin_amdgcn_sched_barrier(0);
            constexpr int n_rd = (kk + 1 < KSTEPS) ? NRD : 0;
            static_for<NMMA>([&](auto Q) {
              constexpr int q = decltype(Q)::value, i = q / WTN, j = q % WTN;
              acc[i][j] = mma16<T>(fb[cur][j], fa[cur][i], acc[i][j]);   // rows = n, lane = m
              __builtin_amdgcn_sched_barrier(0);
              if constexpr (q < n_rd) {
                if constexpr (q < WTM) fa[nxt][q] = lds_read16(fa_base[q] + ((((kk + 1) * 2 + half) ^ fa_key[q]) << 4));
                else fb[nxt][q - WTM] = lds_read16(stB + fb_off[q - WTM][kk + 1]);
              }
              if constexpr (G_ && q == 0 && kk + 1 < KSTEPS && sizeof(T) == 2) g_q[nxt] = lds_read16(g_caddr + (kk + 1) * 16);
              static_for<4 / NMMA>([&](auto R) { gn_stage(KK, std::integral_constant<int, q * (4 / NMMA) + decltype(R)::value>{}); });
              __builtin_amdgcn_sched_barrier(0);
            });
            // (64-channel blocks: 2 MFMAs per k-step carry 2 of the 3 fragment reads of the next one)
            static_for<(n_rd > NMMA ? n_rd - NMMA : 0)>([&](auto Q) {
              constexpr int q = decltype(Q)::value + NMMA;
              if constexpr (q < WTM) fa[nxt][q] = lds_read16(fa_base[q] + ((((kk + 1) * 2 + half) ^ fa_key[q]) << 4));
              else fb[nxt][q - WTM] = lds_read16(stB + fb_off[q - WTM][kk + 1]);
            });
            __builtin_amdgcn_sched_barrier(0);
          });
          if constexpr (G_) lds_write16(g_addr, g_o);   // (complete behind the next tap's first lgkmcnt(0), two barriers before it is read)
        };
        if (g_do) ksteps(std::true_type{});
        else ksteps(std::false_type{});
      }
    }

    // ---- epilogue: MFMA tile row i of this wave = patch rows 2*(wvm*2+i), +1 (16 pixels each)
    const int tm = c_tile / tiles_n;
    const int img = tm / tpi, rem = tm % tpi;
    const int y0 = patch_y0(rem), x0 = patch_x0(rem);
    const int wn0 = (c_tile % tiles_n) * BN + wvn * 32 * WTN;
    bool staged = false;
    if constexpr (sizeof(T) == 2) {
      if (lds_epi) {
        // the halo buffer of the chunk just finished is free once every wave is past its last tap; the halo loader
        // rewrites it only behind the next stage's barrier
        __builtin_amdgcn_s_barrier();
        // (rows of this wave's tile: patch rows (wvm*WTM + i)*2 + r/16, pixels r%16 - the linear-rows epilogue with a row pitch)
        const int64_t wm0 = ((int64_t)img * He + y0 + wvm * WTM * 2) * We + x0;
        const float no_ln[WTM] = {1.f, 1.f};
        const unsigned xb = lds_base + ((gc + 1) & 1) * Halo::HALO_BYTES;
        // (bias and the temb row bias are already in the accumulators)
        if (R != nullptr || p.out_scale != 1.0f)
          epilogue_lds<T, WTM, WTN, NW, Halo::HALO_BYTES, false, false, true, false, Halo::PW>(acc, pe, wm0, wn0, wave, lane, xb, C, R, no_ln, We);
        else
          epilogue_lds<T, WTM, WTN, NW, Halo::HALO_BYTES, false, false, false, false, Halo::PW>(acc, pe, wm0, wn0, wave, lane, xb, C, R, no_ln, We);
        staged = true;
      }
    }
    if (!staged) {
#pragma unroll
      for (int i = 0; i < WTM; i++) {
        const int y = y0 + (wvm * WTM + i) * 2 + (l31 >> 4), x = x0 + (l31 & 15);
        const int64_t m = ((int64_t)img * He + y) * We + x;
        epilogue_row<T, WTN>(acc[i], pe, m, true, wn0, half, C, R);
      }
    }
  }
}

template <typename T, int PH_, int BN_, bool GN> static int launch_halo(const emo_gemm_params& p, int64_t gx, hipStream_t st) {
  using Halo = HaloT<PH_, BN_>;
  constexpr int lds_bytes = Halo::LDS_BYTES + (GN ? Halo::GN_BYTES : 0);
  static bool once = false;
  if (!once) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3_halo_kernel<T, PH_, BN_, GN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return emo_fail(EMO_ERR_HIP, "emo_gemm: hipFuncSetAttribute(halo conv): %s", hipGetErrorString(e));
    once = true;
  }
  conv3x3_halo_kernel<T, PH_, BN_, GN><<<(unsigned)gx, 64 * Halo::NW, lds_bytes, st>>>(p);
  EMO_LAUNCH_CHECK();
  return EMO_OK;
}

// bn = output channels per block (128, or 64: the remainder launch of a width that is an odd multiple of 64)
template <typename T> int gemm_run_halo(const emo_gemm_params& p, int ph, int bn, int64_t gx, hipStream_t st) {
  if (p.gn_coef) {
    if (bn == 64) return ph == 16 ? launch_halo<T, 16, 64, true>(p, gx, st) : launch_halo<T, 8, 64, true>(p, gx, st);
    return ph == 16 ? launch_halo<T, 16, 128, true>(p, gx, st) : launch_halo<T, 8, 128, true>(p, gx, st);
  }
  if (bn == 64) return ph == 16 ? launch_halo<T, 16, 64, false>(p, gx, st) : launch_halo<T, 8, 64, false>(p, gx, st);
  if (bn == 192) {
    if (ph != 16) return emo_fail(EMO_ERR_UNSUPPORTED, "emo_gemm: 192-channel halo blocks need 16-row patches");
    return launch_halo<T, 16, 192, false>(p, gx, st);
  }
  return ph == 16 ? launch_halo<T, 16, 128, false>(p, gx, st) : launch_halo<T, 8, 128, false>(p, gx, st);
}
