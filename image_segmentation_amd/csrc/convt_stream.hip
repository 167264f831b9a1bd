// ConvTranspose2d(k=2, s=2) forward for the SHORT-K levels of the U-Net up path (reference unet/unet.py:59: up3 256->128,
// up4 128->64), bf16.  As a GEMM it is [P x Cin] . [Cin x 4 Cout] with a pixel-shuffle store: 34 GFLOP for 0.2-0.4 GB of
// traffic, i.e. bound by the store of the 2x up-sampled output.  The generic kernel runs it as synchronous 128-pixel
// tiles through LDS (load, barrier, 4-8 k-steps, staged epilogue: 90 / 152 us against 40 / 80 us of HBM time) and the
// producer/consumer GEMM is no better at this K (one workgroup per CU exposes every unit boundary).
//
// This kernel has no LDS, no barrier and no unit boundary: the weights are small enough (Cin x 4 Cout <= 256 KB) to live in
// REGISTERS, sliced over the waves of a workgroup -- a wave keeps the 16x16x32 A-operand fragments of its output channels
// for ALL of K (128 VGPRs: K / 32 k-steps x NBW 16-channel blocks) -- and streams 16-pixel blocks: K / 32 sixteen-byte
// loads per lane straight from global memory (64 B per pixel and k-step, the next block in flight under the MFMAs of the
// current one), K / 32 x NBW MFMAs, and the result leaves from the accumulators.  The MFMA is oriented channels x pixels
// and the channel rows of each PAIR of 16-channel blocks are interleaved (block 2p row 4q+j = channel 32p + 8q + j, block
// 2p+1 row 4q+j = channel 32p + 8q + 4 + j), so a lane ends with 8 consecutive channels of one output pixel: one 16-byte
// store per pair, 64 contiguous bytes per pixel from the four lanes of a column.
#include <stdlib.h>
#include "common.hpp"
#include "segk_internal.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// KS = K / 32 k-steps, NBW = 16-channel blocks per wave (KS * NBW = 32 fragments = 128 weight registers).
// MODE 0: forward, K = Cin, N = 4 Cout (tap-major), x [B,H,W,Cin] -> out [B,2H,2W,Cout] pixel-shuffled (+ bias4).
// MODE 1: data gradient, K = 4 Cout (tap-major: k-step ks reads tap ks / (Cout/32) of the 2x2 output pixels of an input
//         pixel), N = Cin: x := dout [B,2H,2W,Cout] -> out := din [B,H,W,Cin]; `Cout` is the layer's Cout in both modes.
template <int KS, int NBW, int MODE>
__global__ __launch_bounds__(512, 2) void convt_stream_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp,
                                                              const float* __restrict__ bias4, bf16_t* __restrict__ out,
                                                              int B, int H, int W, int Cout, int nblocks16) {
  constexpr int K = KS * 32;
  const int lane = threadIdx.x & 63, lc = lane & 15, lq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int N = MODE == 0 ? 4 * Cout : NBW * 16 * 2, N16 = N / 16;      // MODE 1: N = Cin = 128 (two waves cover it)
  const int wpc = N16 / NBW;                       // waves that cover all N
  const int streams = 8 / wpc;                     // independent pixel streams of the workgroup
  const int wn = wave % wpc, stream = wave / wpc;
  const int g0 = wn * NBW;                         // first 16-channel block of this wave

  // ---- weights: A operand (rows = channels), fragment [ks][nb]: lane (row lc, k-block lq) holds 8 consecutive k
  u32x4 wf[KS][NBW];
  float bs[NBW][4];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const int g = g0 + nb, p = g >> 1, half = g & 1;
    const int n_row = p * 32 + 8 * (lc >> 2) + 4 * half + (lc & 3);      // channel (tap-major n) of operand row lc
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      wf[ks][nb] = *(const u32x4*)(wp + ((size_t)ks * N + n_row) * 32 + lq * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) bs[nb][j] = (MODE == 0 && bias4) ? bias4[p * 32 + 8 * lq + 4 * half + j] : 0.f;   // rows 4 lq + j of D
  }

  // 32-bit block arithmetic (nblocks16 is an int): 64-bit divisions by W16 / H are ~130-instruction loops, two per block
  const unsigned total = (unsigned)nblocks16;      // 16-pixel blocks: W % 16 == 0, so a block never leaves its image row
  const unsigned step = gridDim.x * streams;
  unsigned pb = blockIdx.x * streams + stream;
  if (pb >= total) return;
  u32x4 xf0[KS], xf1[KS];                          // ping-pong by code, not by index (a run-time index would put them in scratch)
  const int W16 = W / 16;
  auto load_x = [&](unsigned blk, u32x4 (&f)[KS]) {
    if constexpr (MODE == 0) {
      const bf16_t* src = x + ((size_t)blk * 16 + lc) * K + lq * 8;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) f[ks] = *(const u32x4*)(src + ks * 32);
    } else {
      const unsigned row = blk / (unsigned)W16;    // b * H + y of the INPUT grid
      const int xi = (int)(blk - row * W16) * 16 + lc;
      const unsigned b = row / (unsigned)H;
      const int y = (int)(row - b * H);
      const int cpt = Cout / 32;                   // k-steps per tap
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int tap = ks / cpt, cc = ks - tap * cpt;
        const size_t opix = ((size_t)(b * 2 * H + 2 * y + (tap >> 1))) * (2 * W) + 2 * xi + (tap & 1);
        f[ks] = *(const u32x4*)(x + opix * Cout + cc * 32 + lq * 8);
      }
    }
  };
  // one 16-pixel block: the next block's loads go out first, then K/32 x NBW MFMAs, then NBW/2 sixteen-byte stores per lane
  auto block = [&](unsigned blk, const u32x4 (&cur)[KS], u32x4 (&nxt)[KS]) {
    if (blk + step < total) load_x(blk + step, nxt);
    f32x4 acc[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) acc[nb] = (f32x4){bs[nb][0], bs[nb][1], bs[nb][2], bs[nb][3]};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb)
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ks][nb]),
                                                          __builtin_bit_cast(bf16x8, cur[ks]), acc[nb], 0, 0, 0);
    // input pixel of this lane's column: block blk = (b, y, x0 / 16)
    const unsigned row = blk / (unsigned)W16;      // b * H + y
    const int x0 = (int)(blk - row * W16) * 16 + lc;
    const unsigned b = row / (unsigned)H;
    const int y = (int)(row - b * H);
#pragma unroll
    for (int pp = 0; pp < NBW / 2; ++pp) {
      const int n_base = ((g0 >> 1) + pp) * 32;    // 32 consecutive n of one tap
      const int tap = n_base / Cout, co = n_base - tap * Cout + 8 * lq;
      const f32x4 a0 = acc[2 * pp], a1 = acc[2 * pp + 1];
      u32x4 v;
      v.x = cvt_pk_bf16(a0[0], a0[1]);
      v.y = cvt_pk_bf16(a0[2], a0[3]);
      v.z = cvt_pk_bf16(a1[0], a1[1]);
      v.w = cvt_pk_bf16(a1[2], a1[3]);
      if constexpr (MODE == 0) {
        const size_t opix = ((size_t)(b * 2 * H + 2 * y + (tap >> 1))) * (2 * W) + 2 * x0 + (tap & 1);
        *(u32x4*)(out + opix * Cout + co) = v;
      } else {
        *(u32x4*)(out + ((size_t)row * W + x0) * N + n_base + 8 * lq) = v;
      }
    }
  };
  load_x(pb, xf0);
  for (;;) {
    block(pb, xf0, xf1);
    pb += step;
    if (pb >= total) break;
    block(pb, xf1, xf0);
    pb += step;
    if (pb >= total) break;
  }
}

template <int KS, int NBW, int MODE>
int launch(const void* x, const void* wp, const float* bias4, void* out, int B, int H, int W, int Cout, int n16, hipStream_t st) {
  const long nblk = (long)B * H * (W / 16);
  const int wpc = n16 / NBW, streams = 8 / wpc;
  long g = (nblk + streams - 1) / streams;
  const long cap = (long)segk_num_cus();           // one 8-wave workgroup per CU (two waves per SIMD at <= 256 registers)
  if (g > cap) g = cap;
  hipLaunchKernelGGL((convt_stream_kernel<KS, NBW, MODE>), dim3((int)g), dim3(512), 0, st, (const bf16_t*)x, (const bf16_t*)wp,
                     bias4, (bf16_t*)out, B, H, W, Cout, (int)nblk);
  SEGK_CHECK_LAUNCH("convt_stream");
  return 0;
}

}  // namespace

// the streaming kernel serves this ConvTranspose forward (bf16, Cin 128 or 256, whole 16-pixel blocks per image row, the
// channel count a multiple of 32 whose 16-channel blocks divide evenly over the eight waves)
int segk_convt_stream_ok(int B, int H, int W, int Cin, int Cout, int dtype) {
  static const bool off = getenv("SEGK_NO_CONVT_STREAM") != nullptr;     // A/B switch (tools/kbench.py convt)
  if (off || dtype != SEGK_DT_BF16 || B <= 0 || H <= 0 || W <= 0 || W % 16 != 0 || Cout <= 0 || Cout % 32 != 0) return 0;
  if ((long long)B * H * W * 4 >= 2147483647LL) return 0;
  if (Cin != 128 && Cin != 256) return 0;
  const int NBW = Cin == 128 ? 8 : 4, n16 = 4 * Cout / 16;
  if (n16 % NBW != 0) return 0;
  const int wpc = n16 / NBW;
  return (wpc == 1 || wpc == 2 || wpc == 4 || wpc == 8) ? 1 : 0;
}

int segk_convt_stream_launch(const void* x, const void* wp, const float* bias4, void* out, int B, int H, int W, int Cin,
                             int Cout, hipStream_t st) {
  SEGK_REQUIRE(x && wp && out, "convt_stream: null pointer");
  SEGK_REQUIRE(segk_convt_stream_ok(B, H, W, Cin, Cout, SEGK_DT_BF16), "convt_stream: shape not served");
  const int n16 = 4 * Cout / 16;
  return Cin == 128 ? launch<4, 8, 0>(x, wp, bias4, out, B, H, W, Cout, n16, st)
                    : launch<8, 4, 0>(x, wp, bias4, out, B, H, W, Cout, n16, st);
}

// ... and this ConvTranspose data gradient (bf16, Cin 128, Cout 64: the up4 level -- K = 4 Cout = 256, N = Cin = 128)
int segk_convt_stream_dgrad_ok(int B, int H, int W, int Cin, int Cout, int dtype) {
  static const bool off = getenv("SEGK_NO_CONVT_STREAM") != nullptr;
  if (off || dtype != SEGK_DT_BF16 || B <= 0 || H <= 0 || W <= 0 || W % 16 != 0) return 0;
  if ((long long)B * H * W * 4 >= 2147483647LL) return 0;
  return (Cin == 128 && Cout == 64) ? 1 : 0;
}

int segk_convt_stream_dgrad_launch(const void* dout, const void* wd, void* din, int B, int H, int W, int Cin, int Cout,
                                   hipStream_t st) {
  SEGK_REQUIRE(dout && wd && din, "convt_stream_dgrad: null pointer");
  SEGK_REQUIRE(segk_convt_stream_dgrad_ok(B, H, W, Cin, Cout, SEGK_DT_BF16), "convt_stream_dgrad: shape not served");
  return launch<8, 4, 1>(dout, wd, nullptr, din, B, H, W, Cout, Cin / 16, st);
}
