// The U-Net stem: Conv2d(Cin <= 3, 64, k=3, p=1) of reference unet/unet.py:16 applied to the NCHW fp32 input batch
// (utils/training.py:45), bf16.  The general 3x3 kernels see this layer as Cin padded to 32 channels behind a layout pass
// (NCHW fp32 -> NHWC bf16, 134 MB written and read again) and spend 29/32 of their matrix work on zeros: 59 + 105 us.
// Its real shape is an im2col GEMM with K = 9 Cin <= 27: ONE 16x16x32 MFMA per 16 pixels and 16 channels.
//
// Here a wave streams 16-pixel blocks with no LDS and no barrier: the weights of all 64 output channels for the whole K
// are 16 registers (read once from the fp32 OIHW parameter, whose row IS the im2col order k = ci * 9 + tap, and rounded to
// bf16 exactly like the packed copies); the B operand is gathered straight from the NCHW fp32 image -- lane (pixel, k
// block) fetches its eight (channel, tap) values with bounds checks (neighbouring lanes and taps hit the same cache lines)
// and rounds them to bf16, the same values the layout pass would have produced; 4 MFMAs; the result leaves from the
// accumulators with the channel rows of each pair of 16-channel blocks interleaved, so a lane stores 8 consecutive channels
// of its pixel (16 bytes).  BatchNorm statistics stay in registers for the whole kernel (one partial row per workgroup).
// The padded NHWC bf16 copy of the input that the weight-gradient pass reads is written on the way (side output).
#include <stdlib.h>
#include "common.hpp"
#include "segk_internal.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
  const int x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
  return v + __int_as_float(x);
}
// sum over the 16 lanes of a DPP row, result in every lane: xor 1, xor 2 (quad permutes), half mirror, mirror
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x141>(v);
  v = dpp_add<0x140>(v);
  return v;
}
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  return cvt_pk_bf16(a, b);      // compiler-visible (common.hpp): an asm conversion behind an MFMA reads too early
}

constexpr int STEM_WAVES = 8;
#ifndef STEM_WG_PER_CU
#define STEM_WG_PER_CU 2
#endif

__global__ __launch_bounds__(STEM_WAVES * 64, 4) void stem_stream_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                         bf16_t* __restrict__ z, bf16_t* __restrict__ xn,
                                                                         float* __restrict__ stats, int B, int H, int W,
                                                                         int Cin, long nblk) {
  __shared__ float red[STEM_WAVES][64][2];
  const int lane = threadIdx.x & 63, lc = lane & 15, lq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int K = Cin * 9;
  const long HW = (long)H * W;

  // ---- weights: A operand rows = channels (pairs of 16-channel blocks interleaved), lane (row lc, k block lq)
  u32x4 wf[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int p = g >> 1, half = g & 1;
    const int n_row = p * 32 + 8 * (lc >> 2) + 4 * half + (lc & 3);
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = lq * 8 + e;
      f[e] = k < K ? w[(size_t)n_row * K + k] : 0.f;
    }
    wf[g] = (u32x4){pk_bf16(f[0], f[1]), pk_bf16(f[2], f[3]), pk_bf16(f[4], f[5]), pk_bf16(f[6], f[7])};
  }
  // ---- the lane's eight (channel, tap) gathers: offset from the pixel's own element of channel 0, and (dy, dx)
  // 8 bits per gather e (four per register): (dy + 1) | (dx + 1) << 2 | ci << 4 | real << 6 -- the offset ci * HW + dy * W + dx is
  // rebuilt per block from these (two multiply-adds) instead of living in eight registers beside two blocks of gathers
  unsigned gpk[2] = {0u, 0u};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = lq * 8 + e;
    const int ci = k / 9, tap = k - ci * 9;
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    const unsigned bits = k < K ? (unsigned)((dy + 1) | ((dx + 1) << 2) | (ci << 4) | (1 << 6)) : 0x05u;   // padding: dy = dx = 0
    gpk[e >> 2] |= bits << (8 * (e & 3));
  }

  float s1[2][8], s2[2][8];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[p][j] = 0.f; s2[p][j] = 0.f; }

  const int W16 = W / 16;
  // 32-bit block arithmetic (nblk < 2^31 by the launcher's shape check): a 64-bit division is a ~130-instruction loop.
  // The loop is bound by the latency of its gathers (16 waves per CU, one memory round trip per 16-pixel block and wave): two
  // blocks are fetched per pass.  A register set holds the raw gathers (from an address clamped to the pixel itself where the
  // tap lies outside the image) and the validity bits that zero them when the block is consumed -- a select at fetch time
  // would wait for the loads right there.
  const unsigned step = gridDim.x * STEM_WAVES;
  struct Blk { unsigned row; int xx; unsigned ok; float g[8]; };
  auto fetch = [&](unsigned blk, Blk& k) {
    k.row = blk / (unsigned)W16;                   // b * H + y
    k.xx = (int)(blk - k.row * W16) * 16 + lc;
    const unsigned b = k.row / (unsigned)H;
    const int y = (int)(k.row - b * H);
    // 32-bit element index off the uniform base pointer (one address register per gather instead of two; the launcher's
    // shape check keeps B * Cin * H * W below 2^31)
    const unsigned base = (b * (unsigned)Cin) * (unsigned)HW + (unsigned)y * (unsigned)W + (unsigned)k.xx;
    unsigned ok = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned bits = gpk[e >> 2] >> (8 * (e & 3));
      const int dy = (int)(bits & 3u) - 1, dx = (int)((bits >> 2) & 3u) - 1, ci = (int)((bits >> 4) & 3u);
      const bool v = ((bits >> 6) & 1u) & ((unsigned)(y + dy) < (unsigned)H) & ((unsigned)(k.xx + dx) < (unsigned)W);
      ok |= (v ? 1u : 0u) << e;
      const int off = ci * (int)HW + dy * W + dx;
      k.g[e] = x[base + (unsigned)(v ? off : 0)];   // unconditional (a conditional load compiles to branch + load + wait)
    }
    k.ok = ok;
  };
  auto work = [&](const Blk& k) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = ((k.ok >> e) & 1u) ? k.g[e] : 0.f;
    const u32x4 bf = {pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[g]), __builtin_bit_cast(bf16x8, bf),
                                                       (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    const size_t opix = (size_t)k.row * W + k.xx;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const f32x4 a0 = acc[2 * p], a1 = acc[2 * p + 1];
      const float o[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1[p][j] += o[j]; s2[p][j] = fmaf(o[j], o[j], s2[p][j]); }
      const u32x4 ov = {pk_bf16(o[0], o[1]), pk_bf16(o[2], o[3]), pk_bf16(o[4], o[5]), pk_bf16(o[6], o[7])};
      *(u32x4*)(z + opix * 64 + p * 32 + lq * 8) = ov;
    }
    if (xn != nullptr) {                           // padded NHWC copy of the input (32 channels: Cin real, zeros behind);
      u32x4 c = {0u, 0u, 0u, 0u};                  // not the training path (the weight gradient gathers from NCHW itself)
      if (lq == 0) {
        const unsigned b = k.row / (unsigned)H;
        const float* px = x + (size_t)b * Cin * HW + (size_t)(k.row - b * H) * W + k.xx;
        const float c0 = px[0], c1 = Cin > 1 ? px[HW] : 0.f, c2 = Cin > 2 ? px[2 * HW] : 0.f;
        c.x = pk_bf16(c0, c1);
        c.y = pk_bf16(c2, 0.f);
      }
      *(u32x4*)(xn + opix * 32 + lq * 8) = c;
    }
  };
  {
    // two blocks per pass: both blocks' gathers are issued before either is consumed (one memory round trip for 32 pixels);
    // blocks are consumed in index order, so the statistics add up in the old order, bit-identical
    const unsigned nb = (unsigned)nblk;
    unsigned blk = blockIdx.x * STEM_WAVES + wave;
    for (; blk < nb; blk += 2 * step) {
      const bool two = blk + step < nb;             // (wave-uniform) an odd last block is fetched twice, multiplied once
      Blk ka, kb;
      fetch(blk, ka);
      fetch(two ? blk + step : blk, kb);
      __builtin_amdgcn_sched_barrier(0);
      work(ka);
      if (two) work(kb);
    }
  }
  if (stats == nullptr) return;
  // ---- statistics: over the 16 pixel lanes of each DPP row, then over the workgroup's waves in wave order (bit-stable)
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a = row16_sum(s1[p][j]), q = row16_sum(s2[p][j]);
      if (lc == 0) {
        red[wave][p * 32 + lq * 8 + j][0] = a;
        red[wave][p * 32 + lq * 8 + j][1] = q;
      }
    }
  __syncthreads();
  if (threadIdx.x < 64) {
    float a = 0.f, q = 0.f;
#pragma unroll
    for (int wv = 0; wv < STEM_WAVES; ++wv) { a += red[wv][threadIdx.x][0]; q += red[wv][threadIdx.x][1]; }
    ((float2*)stats)[(size_t)blockIdx.x * 64 + threadIdx.x] = make_float2(a, q);
  }
}

// ------------------------------------------------------------------------------------------------
// The stem's weight gradient, dW[n][k] = sum over pixels of dz[p][n] * col[p][k] with k = ci * 9 + tap the im2col index
// (the OIHW order of the parameter): a [64 x P] . [P x 27] contraction over PIXELS.  The general kernel runs it on the
// 32-channel-padded NHWC copy of the input (455 MB of traffic with the halo, 9 taps x MFMAs on 29 zero channels: 100 us);
// here a wave streams 16-pixel steps: its 16 x 64 dz values go through a 2 KB wave-private LDS slot and come back
// transposed (ds_read_b64_tr_b16: 8 consecutive pixels of one channel per lane), the im2col operand is gathered straight
// from the NCHW fp32 image (8 consecutive x positions of one (channel, tap) = 32 contiguous bytes per lane), two
// 32x32x16 MFMAs per step accumulate the whole 64 x 32 gradient in 32 registers; the eight waves of a workgroup add up
// through LDS in a fixed order and write one slab per workgroup for segk_wgrad_reduce (taps = 1, CA = 9 Cin).
struct __attribute__((packed, aligned(4))) F4U { float v[4]; };    // 16-byte load at 4-byte alignment

__global__ __launch_bounds__(STEM_WAVES * 64, 2) void stem_wgrad_kernel(const float* __restrict__ x, const bf16_t* __restrict__ dz,
                                                                        float* __restrict__ slabs, int B, int H, int W, int Cin,
                                                                        long nblk) {
  __shared__ __attribute__((aligned(16))) char lds[4 * 64 * 32 * 4];       // 32 KB: dz slots (2 KB per wave), then the reduction
  typedef __attribute__((address_space(3))) s16x4* lds_v4;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int K = Cin * 9;
  const long HW = (long)H * W;
  char* const slot = lds + wave * 2048;

  // B operand (im2col): lane (column j = lane % 32, pixel half hh = lane / 32)
  const int j = lane & 31, hh = lane >> 5;
  const int ci = j / 9, tap = j - ci * 9;
  const int dy = tap / 3 - 1, dx = tap % 3 - 1;
  const bool jok = j < K;
  // A operand (dz transposed): the fragment addressing of wgrad_kernel (two transposing reads per 32-channel block)
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
  const int coff = (16 * (g & 1) + 4 * p) * 2, xa = 8 * (g >> 1) + q;
  // dz staging: lane -> (pixel lp, 16-byte piece lc4) of each 32-channel block
  const int lp = lane >> 2, lc4 = lane & 3;

  f32x16 acc[2];
#pragma unroll
  for (int nh = 0; nh < 2; ++nh)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nh][r] = 0.f;

  const int W16 = W / 16;
  const long step = (long)gridDim.x * STEM_WAVES;
  for (long blk = (long)blockIdx.x * STEM_WAVES + wave; blk < nblk; blk += step) {
    const long row = blk / W16;                    // b * H + y
    const int x0 = (int)(blk - row * W16) * 16;
    const long b = row / H;
    const int y = (int)(row - b * H);
    // ---- dz: 16 pixels x 64 channels -> LDS, natural [block][pixel][64 B]
    const bf16_t* dp = dz + ((size_t)row * W + x0 + lp) * 64 + lc4 * 8;
    const uint4 d0 = *(const uint4*)dp, d1 = *(const uint4*)(dp + 32);
    // ---- im2col column j, pixels x0 + 8 hh .. + 7 of image row y + dy
    float v[8];
    const int xs = x0 + 8 * hh + dx;
    const bool rok = jok & ((unsigned)(y + dy) < (unsigned)H);
    const float* src = x + ((size_t)(b * Cin + ci) * H + (y + dy)) * W + xs;
    if (rok && xs >= 0 && xs + 8 <= W) {
      const F4U a = *(const F4U*)src, c = *(const F4U*)(src + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = a.v[e]; v[4 + e] = c.v[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (rok && (unsigned)(xs + e) < (unsigned)W) ? src[e] : 0.f;
    }
    *(uint4*)(slot + lp * 64 + lc4 * 16) = d0;
    *(uint4*)(slot + 1024 + lp * 64 + lc4 * 16) = d1;
    const u32x4 fbv = {pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
    const bf16x8 fb = __builtin_bit_cast(bf16x8, fbv);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");         // the wave's own LDS stores before its transposing reads
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(slot + nh * 1024 + xa * 64 + coff));
      const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(slot + nh * 1024 + (xa + 4) * 64 + coff));
      const bf16x8 fa = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
      acc[nh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[nh], 0, 0, 0);
    }
  }

  // ---- the eight waves add up in a fixed order: 7..4 -> 3..0, 3..2 -> 1..0, 1 -> 0 (32 KB of LDS: four wave images)
  float* const red = (float*)lds;
  const int col = lane & 31, lh = lane >> 5;
  auto put = [&](int slot_i) {
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = 32 * nh + (r & 3) + 8 * (r >> 2) + 4 * lh;
        red[(slot_i * 64 + n) * 32 + col] = acc[nh][r];
      }
  };
  auto add = [&](int slot_i) {
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = 32 * nh + (r & 3) + 8 * (r >> 2) + 4 * lh;
        acc[nh][r] += red[(slot_i * 64 + n) * 32 + col];
      }
  };
  __syncthreads();                                 // every wave is done with its dz slot
  if (wave >= 4) put(wave - 4);
  __syncthreads();
  if (wave < 4) add(wave);
  __syncthreads();
  if (wave == 2 || wave == 3) put(wave - 2);
  __syncthreads();
  if (wave < 2) add(wave);
  __syncthreads();
  if (wave == 1) put(0);
  __syncthreads();
  if (wave == 0) {
    add(0);
    float* const slab = slabs + (size_t)blockIdx.x * 64 * 32;
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = 32 * nh + (r & 3) + 8 * (r >> 2) + 4 * lh;
        slab[n * 32 + col] = acc[nh][r];
      }
  }
}

}  // namespace

// workgroups (= slabs [64][32] fp32) of the stem weight-gradient kernel, or 0 where it does not apply
int segk_stem_wgrad_slabs(int B, int H, int W, int Cin, int Cout, int dtype) {
  static const bool off = getenv("SEGK_NO_STEM_WGRAD") != nullptr;      // A/B switch
  if (off) return 0;
  const int rows = segk_stem_rows(B, H, W, Cin, Cout, dtype);           // the same shape conditions as the forward
  if (rows <= 0) return 0;
  const long nblk = (long)B * H * (W / 16);
  long g = (nblk + STEM_WAVES - 1) / STEM_WAVES;
  const long cap = (long)segk_num_cus() * 2;
  if (g > cap) g = cap;
  return (int)g;
}

int segk_stem_wgrad_launch(const float* x, const void* dz, float* slabs, int B, int H, int W, int Cin, int Cout,
                           hipStream_t st) {
  SEGK_REQUIRE(x && dz && slabs, "stem_wgrad: null pointer");
  const int g = segk_stem_wgrad_slabs(B, H, W, Cin, Cout, SEGK_DT_BF16);
  SEGK_REQUIRE(g > 0, "stem_wgrad: shape not served (bf16, 1..3 input channels, 64 output channels, W a multiple of 16)");
  const long nblk = (long)B * H * (W / 16);
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3(g), dim3(STEM_WAVES * 64), 0, st, x, (const bf16_t*)dz, slabs, B, H, W, Cin, nblk);
  SEGK_CHECK_LAUNCH("stem_wgrad");
  return 0;
}

// workgroups (= rows of BatchNorm partials) of the stem kernel for this problem, or 0 where it does not apply
int segk_stem_rows(int B, int H, int W, int Cin, int Cout, int dtype) {
  static const bool off = getenv("SEGK_NO_STEM") != nullptr;            // A/B switch
  if (off || dtype != SEGK_DT_BF16 || B <= 0 || H <= 0 || W <= 0 || W % 16 != 0 || Cin < 1 || Cin > 3 || Cout != 64) return 0;
  if ((long long)B * H * W * 64 >= 2147483647LL * 16 || (long long)B * Cin * H * W >= 2147483647LL) return 0;
  const long nblk = (long)B * H * (W / 16);
  long g = (nblk + STEM_WAVES - 1) / STEM_WAVES;
  const long cap = (long)segk_num_cus() * STEM_WG_PER_CU;   // 8-wave workgroups per CU: the gathers want many in flight
  if (g > cap) g = cap;
  return (int)(g > 1024 ? 1024 : g);               // <= 1024 rows: the one-block statistics finalisation
}

int segk_stem_launch(const float* x, const float* w, void* z, void* xn, float* stats, int B, int H, int W, int Cin, int Cout,
                     hipStream_t st) {
  SEGK_REQUIRE(x && w && z, "stem3x3: null pointer");
  const int g = segk_stem_rows(B, H, W, Cin, Cout, SEGK_DT_BF16);
  SEGK_REQUIRE(g > 0, "stem3x3: shape not served (bf16, 1..3 input channels, 64 output channels, W a multiple of 16)");
  const long nblk = (long)B * H * (W / 16);
  hipLaunchKernelGGL(stem_stream_kernel, dim3(g), dim3(STEM_WAVES * 64), 0, st, x, w, (bf16_t*)z, (bf16_t*)xn, stats, B, H, W, Cin,
                     nblk);
  SEGK_CHECK_LAUNCH("stem3x3");
  return 0;
}
