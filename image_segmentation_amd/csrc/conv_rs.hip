// Register-stationary 3x3 convolution for the narrow, high-resolution bf16 layers of the U-Net (Cin <= 64, output
// channel tiles of 64: the 3->64 stem, the 64->64 convs of down1/up4 and their data gradients, the 64->128 concat
// gradient; reference unet/unet.py:16,19 at the 256x256 level).  These layers move 0.5-0.8 GB for 0.15 TFLOP: they are
// bound by HBM and by everything that keeps a CU from streaming, so the kernel is built around three things:
//
//   * the WEIGHTS LIVE IN REGISTERS.  K = 9*Cin is tiny (<= 576): a wave keeps the 16x16x32 A-operand fragments of its
//     32 output channels for all 9 taps x 2 chunks in 144 VGPRs for the whole (persistent) kernel.  LDS then holds
//     activations only, the MFMA loop reads ONE fragment per two MFMAs (the weight-stationary kernel it replaces read
//     three per two and was LDS-bandwidth bound), and 130 KB of LDS are free for a three-deep ring of input tiles;
//   * the INPUT ARRIVES BY LDS-DMA (global_load_lds, 1 KiB per wave-instruction, no staging registers, zero page for
//     out-of-image halo pixels), two tiles ahead of the one being multiplied, so every CU always has ~90 KB of loads in
//     flight.  The LDS image is unpadded (64 B per pixel and chunk); a DMA wave-instruction writes 64 consecutive 16-byte
//     slots, so bank-conflict freedom comes from WHICH piece each lane fetches: piece q of patch pixel p sits at
//       (p >> 3) * 512 + (q >> 1) * 256 + ((6 p + q) & 15) * 16
//     -- within every aligned group of 8 pixels the slot (6p + q) mod 16 is a bijection, and a 16x16x32 operand read
//     (16 consecutive pixels x 4 pieces) hits 16 distinct slots per 16-lane group for ANY starting pixel, i.e. for all
//     nine tap shifts;
//   * the OUTPUT LEAVES FROM REGISTERS.  The MFMA is oriented channels x pixels (A = weights, B = patch) and the channel
//     rows are permuted so that a lane ends up with 8 consecutive channels of one pixel: one 16-byte global store per 16
//     pixels, no LDS staging tile, no epilogue barriers.  BatchNorm statistics are reduced over the 16 pixel lanes with
//     DPP adds and accumulated per wave in LDS across all its tiles (one partial row per wave and kernel).
//
// One s_barrier per tile.  PRO (second conv of a DoubleConv block): the BatchNorm+ReLU of the previous layer is applied
// in LDS by the wave that fetched the piece (after its own vmcnt wait), which also writes the transformed activation
// to `act_out` for the weight-gradient pass.
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"
#include "segk_internal.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

// out-of-image pixels are fetched from here (a device array of the code object: nothing is allocated)
__device__ __attribute__((aligned(256))) unsigned char g_rs_zero_page[256];

constexpr int RS_PW = 34, RS_NPIX = 340, RS_NBLK = 22, RS_CHB = RS_NBLK * 1024;

// compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N-1 (inline-asm immediates need constants)
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// LDS reads of DMA-written data go through inline asm: hipcc orders every LDS load it can see behind ALL pending
// global_load_lds (s_waitcnt vmcnt(0)-like), which would drain the tiles in flight once per tile.  The statement's
// destination is only valid after the lgkmcnt wait the caller places (lds_wait<N>), and nothing may be scheduled across
// that wait (sched_barrier).
template <int OFF> __device__ __forceinline__ void lds_rd128(uint4& d, int addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
// LDS stores inside the tile loop go through inline asm as well: hipcc orders a visible LDS store behind every pending
// LDS-DMA too (write-after-write), i.e. an s_waitcnt vmcnt(0) that also waits for the global stores just issued.
template <int OFF> __device__ __forceinline__ void lds_wr128(int addr, const uint4& v) {
  const u32x4 t = {v.x, v.y, v.z, v.w};
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(t), "n"(OFF) : "memory");
}

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

// Diagnostic build only (-DSEGK_RS_STAMPS, tools/stamp_build.sh): per-wave cycle sums of the loop's phases, written over
// the statistics buffer by lane 0 of every wave; the shipped library contains no stamp.
#ifdef SEGK_RS_STAMPS
#define RS_STAMP(i)                                                                       \
  do {                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    unsigned long long t_;                                                                \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
    stamp_sum[i] += t_ - stamp_last;                                                      \
    stamp_last = t_;                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  } while (0)
#else
#define RS_STAMP(i) do {} while (0)
#endif

// The lane id, recomputed where it is needed (volatile: not hoisted): values derived from it in the DMA / transform
// sections then do not occupy registers across the MFMA loop, where 144 weight registers leave little room.
__device__ __forceinline__ int fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// 8 floats -> 8 bf16 (round to nearest even, NaN-preserving), one v_cvt_pk_bf16_f32 per pair
__device__ __forceinline__ uint4 pack8_bf16(const float (&v)[8]) {
  uint4 o;
  o.x = cvt_pk_bf16(v[0], v[1]);
  o.y = cvt_pk_bf16(v[2], v[3]);
  o.z = cvt_pk_bf16(v[4], v[5]);
  o.w = cvt_pk_bf16(v[6], v[7]);
  return o;
}

#ifndef RS_ABL
#define RS_ABL 0      // diagnostic builds: bit 0 = no output stores, bit 1 = no DMA, bit 2 = no MFMA loop (timing only)
#endif

template <int NCH, bool PRO>
__global__ __launch_bounds__(512, 2) void conv_rs_kernel(const ConvArgs a) {
  typedef bf16_t T;
  constexpr int TILEB = NCH * RS_CHB;
  constexpr int NPIECE = NCH * RS_NBLK;              // DMA pieces (1 KiB) per tile
  constexpr int NPW = (NPIECE + 7) / 8;              // per wave: waves below NPIECE % 8 move NPW pieces, the others NPW - 1
  constexpr int NFULL = NPIECE % 8 == 0 ? 8 : NPIECE % 8;
  constexpr int TAB_OFF = 3 * TILEB;                 // PRO: [NCH][4 pieces][8 scale | 8 shift]
  constexpr int NSTEP = NCH * 9;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = wave & 3, jh = wave >> 2;             // pixel slab (2 tile rows) and channel half of the wave
  const int col = lane & 15, g = lane >> 4;
  const int H = a.H, W = a.W;

  // ---- work: the channel tile is fixed per workgroup, pixel tiles are walked with stride GW inside the XCD's range
  const int NT = a.Ntot >> 6;
  const int tpi = a.tiles_x * a.tiles_y, MT = a.B * tpi;
  const int xcd = blockIdx.x & 7, wgi = blockIdx.x >> 3, GWX = gridDim.x >> 3;
  const int nt = wgi % NT, slot = wgi / NT, GW = GWX / NT;
  const int mpx = (MT + 7) >> 3;
  int mt = xcd * mpx + slot;
  const int mt_end = min(MT, (xcd + 1) * mpx);
  const int n0 = nt * 64;
  const int srow = (xcd * GW + slot) * 4 + s;         // this wave's row of the statistics partials
  float2* const stat_dst = a.stats ? (float2*)a.stats + (size_t)srow * a.Ntot + n0 + 32 * jh + 8 * g : nullptr;
  if (mt >= mt_end) {                                 // nothing to do: the partial rows still have to exist
    if (stat_dst && col == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) stat_dst[i] = make_float2(0.f, 0.f);
    }
    return;
  }
  auto decode = [&](int m, int& b, int& y0, int& x0) {
    b = m / tpi;
    const int trem = m - b * tpi;
    const int tyi = trem / a.tiles_x;
    y0 = tyi * 8;
    x0 = (trem - tyi * a.tiles_x) * 32;
  };

  // ---- resident weights: A operand (16 channels x 32 k) of the wave's two channel blocks, every tap and chunk.
  // Row m = 4 g' + i of block cb is channel 8 g' + 4 cb + i, so that accumulator row 4 g + i (held by lane row g) of
  // the two blocks are 8 consecutive channels.
  u32x4 wf[NCH][9][2];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int n = n0 + 32 * jh + 8 * (col >> 2) + 4 * cb + (col & 3);
        wf[c][t][cb] = *(const u32x4*)((const char*)a.w + ((size_t)(c * 9 + t) * a.Ntot + n) * 64 + g * 16);
      }

  // ---- LDS bookkeeping
  if (PRO) {
    if (tid < NCH * 32) {
      const int c = tid >> 5, r = tid & 31, q = r >> 3, i = r & 7;
      float* tab = (float*)(smem + TAB_OFF + (c * 4 + q) * 64);
      tab[i] = a.scale[c * 32 + q * 8 + i];
      tab[8 + i] = a.shift[c * 32 + q * 8 + i];
    }
  }

  // ---- DMA: lane -> (pixel pl of the 16-pixel block, piece q) by inverting the slot map (file header).  Piece i of the
  // wave is block bi = wave + 8 i of the tile (chunk bi / 22, pixels 16 (bi % 22) ..).  poff[i]: byte offset of the
  // lane's 16 bytes from the chunk's tile origin (patch pixel (0,0) = image pixel (y0-1, x0-1)); pixels past the patch
  // (the tail of the last block) re-read the last patch pixel into LDS slots nobody reads.
  auto dma_lane = [](int l, int& dq, int& dpl) {
    const int drow = l >> 4, dslot = l & 15;
    dq = ((drow & 1) << 1) | (dslot & 1);
    dpl = ((drow >> 1) << 3) | ((3 * (((dslot - (dslot & 1) - 2 * (drow & 1)) & 15) >> 1)) & 7);
  };
  int CAs = a.CA, CBs = a.CB;                        // pinned in scalar registers (hipcc otherwise re-loads them from
  asm volatile("" : "+s"(CAs), "+s"(CBs));           // the kernel-argument segment for every DMA piece)
  const int nchA = CAs >> 5;
  unsigned poff[NPW];
  {
  int dq, dpl;
  dma_lane(lane, dq, dpl);
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int bi = min(wave + 8 * i, NPIECE - 1);
    const int chunk = bi / RS_NBLK, blk = bi - chunk * RS_NBLK;
    const int p = min(16 * blk + dpl, RS_NPIX - 1);
    const int py = p / RS_PW, px = p - py * RS_PW;
    const int C = chunk < nchA ? CAs : CBs;
    poff[i] = (unsigned)((py * W + px) * C) * 2u + dq * 16;
  }
  }
  auto chunk_base = [&](int chunk, int b, int y0, int x0) -> const char* {   // wave-uniform
    const T* base;
    int C;
    if (chunk < nchA) { base = (const T*)a.srcA + chunk * 32; C = CAs; }
    else { base = (const T*)a.srcB + (chunk - nchA) * 32; C = CBs; }
    return (const char*)(base + ((ptrdiff_t)(b * H + y0 - 1) * W + (x0 - 1)) * C);
  };
  auto lane_pixel = [&](int bi, int dpl, int& py, int& px) {  // patch coordinates of the lane's pixel in block bi
    const int blk = bi - (bi / RS_NBLK) * RS_NBLK;
    const int p = 16 * blk + dpl;
    py = (p * 241) >> 13;                             // p / 34 for p < 352
    px = p - py * RS_PW;
  };
  auto issue_tile = [&](int m, int bsel) {
    int b, y0, x0;
    decode(m, b, y0, x0);
    const bool interior = (y0 > 0) && (y0 + 8 < H) && (x0 > 0) && (x0 + 32 < W);   // the whole patch lies in the image
    const char* const cb0 = chunk_base(0, b, y0, x0);
    const char* const cb1 = NCH > 1 ? chunk_base(1, b, y0, x0) : cb0;
    int dq = 0, dpl = 0, fl = 0;
    if (!interior) { fl = fresh_lane(); dma_lane(fl, dq, dpl); }
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int bi = wave + 8 * i;                    // wave-uniform
      if (bi >= NPIECE) break;
      const char* const cb = (bi >= RS_NBLK) ? cb1 : cb0;
      char* const dst = smem + bsel * TILEB + bi * 1024;
      unsigned po = poff[i];
      asm volatile("" : "+v"(po));                    // keeps hipcc from hoisting 64-bit (base + offset) sums out of the
                                                      // tile loop (five register pairs, spilled around the MFMA loop)
      if (RS_ABL & 2) continue;
      if (interior) {
        __builtin_amdgcn_global_load_lds((glb_vp)(cb + po), (lds_vp)dst, 16, 0, 0);
      } else {
        int py, px;
        lane_pixel(bi, dpl, py, px);
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        const bool ok = (py < 10) & (gy >= 0) & (gy < H) & (gx >= 0) & (gx < W);
        const char* src = ok ? cb + po : (const char*)g_rs_zero_page + (fl & 15) * 16;
        __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)dst, 16, 0, 0);
      }
    }
  };
  // BatchNorm+ReLU of the producer layer on the wave's own pieces of a landed tile (halo pixels outside the image stay
  // exact zeros), plus the side output of the transformed activation
  auto transform_tile = [&](int m, int bsel) {
    int b, y0, x0;
    decode(m, b, y0, x0);
    const int fl = fresh_lane();
    int dq, dpl;
    dma_lane(fl, dq, dpl);
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int bi = wave + 8 * i;
      if (bi >= NPIECE) break;                        // wave-uniform
      const int chunk = bi / RS_NBLK;
      int py, px;
      lane_pixel(bi, dpl, py, px);
      const int gy = y0 + py - 1, gx = x0 + px - 1;
      const bool ok = (py < 10) & (gy >= 0) & (gy < H) & (gx >= 0) & (gx < W);
      const int posa = bsel * TILEB + bi * 1024 + fl * 16;
      const int taba = TAB_OFF + (chunk * 4 + dq) * 64;
      // two halves of four channels each: few registers are free beside the 144 resident weight registers
      uint4 raw, tsc, tsh, v;
      lds_rd128<0>(raw, posa);
      lds_rd128<0>(tsc, taba);
      lds_rd128<32>(tsh, taba);
      lds_wait<0>();
      __builtin_amdgcn_sched_barrier(0);
      {
        const float f0 = fmaxf(fmaf(bf2f(raw.x & 0xffffu), __uint_as_float(tsc.x), __uint_as_float(tsh.x)), 0.f);
        const float f1 = fmaxf(fmaf(__uint_as_float(raw.x & 0xffff0000u), __uint_as_float(tsc.y), __uint_as_float(tsh.y)), 0.f);
        const float f2 = fmaxf(fmaf(bf2f(raw.y & 0xffffu), __uint_as_float(tsc.z), __uint_as_float(tsh.z)), 0.f);
        const float f3 = fmaxf(fmaf(__uint_as_float(raw.y & 0xffff0000u), __uint_as_float(tsc.w), __uint_as_float(tsh.w)), 0.f);
        v.x = cvt_pk_bf16(f0, f1);
        v.y = cvt_pk_bf16(f2, f3);
      }
      __builtin_amdgcn_sched_barrier(0);
      lds_rd128<16>(tsc, taba);
      lds_rd128<48>(tsh, taba);
      lds_wait<0>();
      __builtin_amdgcn_sched_barrier(0);
      {
        const float f0 = fmaxf(fmaf(bf2f(raw.z & 0xffffu), __uint_as_float(tsc.x), __uint_as_float(tsh.x)), 0.f);
        const float f1 = fmaxf(fmaf(__uint_as_float(raw.z & 0xffff0000u), __uint_as_float(tsc.y), __uint_as_float(tsh.y)), 0.f);
        const float f2 = fmaxf(fmaf(bf2f(raw.w & 0xffffu), __uint_as_float(tsc.z), __uint_as_float(tsh.z)), 0.f);
        const float f3 = fmaxf(fmaf(__uint_as_float(raw.w & 0xffff0000u), __uint_as_float(tsc.w), __uint_as_float(tsh.w)), 0.f);
        v.z = cvt_pk_bf16(f0, f1);
        v.w = cvt_pk_bf16(f2, f3);
      }
      v = ok ? v : make_uint4(0u, 0u, 0u, 0u);
      lds_wr128<0>(posa, v);
      const bool inner = (py >= 1) & (py <= 8) & (px >= 1) & (px <= 32);
      if (a.act_out && nt == 0 && ok && inner)
        *(uint4*)((T*)a.act_out + ((size_t)(b * H + gy) * W + gx) * a.CA + chunk * 32 + dq * 8) = v;
    }
  };
  auto wait_pieces = [&]() {                          // all but the wave's youngest tile of DMA pieces have landed
    if (wave < NFULL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW - 1) : "memory");
  };

  // ---- fragment read addresses.  A fragment is 16 consecutive patch pixels starting at p0 = 68 s + c0 with c0 a
  // compile-time constant of (tile row, half row, tap): lane (col, g) reads piece g of pixel p0 + col.  With
  // u = (c0 & 7) + col + (68 s & 7) the address is abuf[c0 & 7] + (c0 >> 3) * 512 + chunk * CHB (immediate).
  const int pwv = 68 * s;
  int abuf[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int u = k + col + (pwv & 7);
    abuf[k] = ((u >> 3) + (pwv >> 3)) * 512 + (g >> 1) * 256 + ((6 * u + g) & 15) * 16;
  }

  // ---- output: lane (col, g) stores the 8 channels 8 g .. 8 g + 7 of its half for pixel (2 s + rr, 16 xh + col) of
  // the tile: byte offset soff (+ rr * ystep + xh * xstep) from the tile's first pixel in the destination of this channel half
  T* dch;
  int dstride;
  if (n0 + 32 * jh < a.CO1) { dch = (T*)a.out + n0 + 32 * jh; dstride = a.CO1; }     // per 32-channel half
  else { dch = (T*)a.out2 + (n0 + 32 * jh - a.CO1); dstride = a.CO2; }
  const unsigned soff = (unsigned)((2 * s * W + col) * dstride + 8 * g) * 2u;
  const unsigned xstep = 32u * dstride;              // 16 pixels further along x
  const unsigned ystep = 2u * W * dstride;           // one row down

  // ---- prologue: tiles 0 and 1 in flight, tile 0 landed (and transformed)
  int m1 = mt + GW, m2 = mt + 2 * GW;                 // the next two tiles of this workgroup
  const bool class_b = wave >= 4;
  issue_tile(mt, 0);
  if (m1 < mt_end) {
    issue_tile(m1, 1);
    wait_pieces();
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (PRO) {
    __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): the table is written ...
    __builtin_amdgcn_s_barrier();                     // ... by other waves
    transform_tile(mt, 0);
  }
  int bsel = 0;                                       // ring slot of the current tile
  const bool has_stats = (a.stats != nullptr);

#ifdef SEGK_RS_STAMPS
  unsigned long long stamp_sum[6] = {0, 0, 0, 0, 0, 0}, stamp_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
  const unsigned long long stamp_t0 = stamp_last;
#endif
  // ---- a tile is multiplied in two passes (its slab's two tile rows), 16 accumulator registers each: the registers
  // this frees hold the wave's BatchNorm statistics for the WHOLE kernel (per lane: 8 channels x (sum, sum of squares)),
  // so a pass's epilogue is a pack, two stores and 32 accumulating VALU operations -- no cross-lane step, no LDS (an
  // earlier form reduced over lanes and accumulated in LDS per tile: a quarter of the kernel's time).
  //
  // The two waves of a SIMD (w and w + 4) run the phases of a tile in different orders, so that one wave's MFMA stream
  // runs beside its partner's vector-memory issue and VALU work instead of both queueing for the matrix pipe first and
  // for the memory pipe afterwards (the older wave wins the pipe, finishes its passes first and would then idle at the
  // barrier):   class A (waves 0-3):  barrier | pass 0, store | pass 1, store | refill the DMA ring | wait
  //             class B (waves 4-7):  barrier | store pass 1 of the PREVIOUS tile (its 16 accumulators simply stay in
  //                                   registers across the barrier) | refill the ring | pass 0, store | pass 1 | wait
  float s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  f32x4 acc1[2][2];                                   // pass 1 (class B: kept across the barrier)
  int pb = 0, py0 = 0, px0 = 0;
  bool have_prev = false;

  auto run_pass = [&](auto HC, f32x4 (&acc)[2][2]) {
    constexpr int h = decltype(HC)::value;            // tile row 2 s + h of the slab
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int pg = 0; pg < 2; ++pg) acc[cb][pg] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // fragment f = 2 * step + half row lives in ring register f % RING and is fetched RING - 1 fragments ahead of its
    // two MFMAs, into the register the fragment consumed just before has left
#ifndef RS_RING
#define RS_RING 5
#endif
    constexpr int RING = RS_RING;
    uint4 fr[RING];
    auto rd = [&](auto FC) {
      constexpr int f = decltype(FC)::value, ct = f >> 1, pg = f & 1;
      constexpr int c = ct / 9, t = ct - c * 9, dy = t / 3, dx = t - dy * 3;
      constexpr int c0 = RS_PW * (h + dy) + 16 * pg + dx;
      lds_rd128<(c0 >> 3) * 512 + c * RS_CHB>(fr[f % RING], abuf[c0 & 7]);
    };
    constexpr int NFRAG = 2 * NSTEP;
    if (!(RS_ABL & 4)) static_for<0, RING - 1>(rd);
    if (!(RS_ABL & 4)) static_for<0, NFRAG>([&](auto FC) {
      constexpr int f = decltype(FC)::value, ct = f >> 1, pg = f & 1;
      lds_wait<(NFRAG - 1 - f < RING - 2) ? NFRAG - 1 - f : RING - 2>();   // fragment f has arrived (LDS returns in order)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
        acc[cb][pg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
            __builtin_bit_cast(bf16x8, wf[ct / 9][ct % 9][cb]), __builtin_bit_cast(bf16x8, fr[f % RING]), acc[cb][pg], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (f + RING - 1 < NFRAG) rd(std::integral_constant<int, f + RING - 1>{});
    });
    __builtin_amdgcn_sched_barrier(0);
  };
  // epilogue of a pass, from registers: lane (col, g) holds channels 8 g .. 8 g + 7 of its half for pixel
  // (2 s + h, 16 pg + col) of tile (ub, uy0, ux0)
  auto store_pass = [&](auto HC, const f32x4 (&acc)[2][2], int ub, int uy0, int ux0) {
    constexpr int h = decltype(HC)::value;
    char* const tb = (char*)(dch + ((size_t)(ub * H + uy0) * W + ux0) * dstride);    // wave-uniform
    const bool full = (uy0 + 8 <= H) && (ux0 + 32 <= W);
#pragma unroll
    for (int pg = 0; pg < 2; ++pg) {
      float v[8];
      bool in = true;
      if (!full) {                                    // pixels past the image edge: not stored, not in the statistics
        in = (uy0 + 2 * s + h < H) && (ux0 + 16 * pg + col < W);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = in ? acc[0][pg][i] : 0.f; v[4 + i] = in ? acc[1][pg][i] : 0.f; }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = acc[0][pg][i]; v[4 + i] = acc[1][pg][i]; }
      }
      if (has_stats) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { s1[i] += v[i]; s2[i] = fmaf(v[i], v[i], s2[i]); }
      }
      if (!(RS_ABL & 1)) {
        if (in) *(uint4*)(tb + (pg * xstep + h * ystep) + soff) = pack8_bf16(v);
      } else {
        asm volatile("" ::"v"(v[0] + v[7]));
      }
    }
  };
  const std::integral_constant<int, 0> H0{};
  const std::integral_constant<int, 1> H1{};

  for (;;) {
    int ub, uy0, ux0;
    decode(mt, ub, uy0, ux0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // own LDS writes (transform) are done
    RS_STAMP(5);                                      // (loop bookkeeping / transform of the previous iteration)
    __builtin_amdgcn_s_barrier();                     // the tile is complete in LDS for every wave
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    RS_STAMP(0);                                      // barrier wait
    // the ring slot two tiles back is free now (every wave passed this tile's barrier)
    const int bnext = (bsel == 2) ? 0 : bsel + 1, bfree = (bnext == 2) ? 0 : bnext + 1;
    if (class_b) {
      if (have_prev) store_pass(H1, acc1, pb, py0, px0);
      RS_STAMP(4);                                    // epilogue
      if (m2 < mt_end) issue_tile(m2, bfree);
      RS_STAMP(2);                                    // DMA issue
    }
    {
      f32x4 acc0[2][2];
      run_pass(H0, acc0);
      RS_STAMP(1);                                    // MFMA pass
      store_pass(H0, acc0, ub, uy0, ux0);
      RS_STAMP(4);
    }
    run_pass(H1, acc1);
    RS_STAMP(1);
    if (!class_b) {
      store_pass(H1, acc1, ub, uy0, ux0);
      RS_STAMP(4);
      if (m2 < mt_end) issue_tile(m2, bfree);
      RS_STAMP(2);
    } else {
      pb = ub; py0 = uy0; px0 = ux0;
      have_prev = true;
    }
    // ---- the NEXT tile's own pieces have landed: only the pieces of the tile after it may stay in flight (class B
    // issued those before its passes: the two stores of pass 0 are younger, so it also waits for the first two of them,
    // which are a whole tile old by now)
    if (m2 < mt_end) wait_pieces();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RS_STAMP(3);                                      // wait for the next tile's pieces

    if (m1 >= mt_end) break;
    if (PRO) transform_tile(m1, bnext);
    mt = m1; m1 = m2; m2 += GW;
#pragma unroll
    for (int k = 0; k < 8; ++k) abuf[k] += (bnext == 0) ? -2 * TILEB : TILEB;
    bsel = bnext;
  }
  if (class_b && have_prev) store_pass(H1, acc1, pb, py0, px0);   // the deferred pass of the last tile

#ifdef SEGK_RS_STAMPS
  if (has_stats && lane == 0) {                       // [workgroup][wave][8]: six phase sums, total, unused
    unsigned long long tend;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tend)::"memory");
    unsigned long long* o = (unsigned long long*)a.stats + ((size_t)blockIdx.x * 8 + wave) * 8;
    for (int i = 0; i < 6; ++i) o[i] = stamp_sum[i];
    o[6] = tend - stamp_t0;
  }
  return;
#endif
  if (has_stats) {                                    // once per kernel: sum over the 16 pixel lanes of each lane row
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      s1[i] = row16_sum(s1[i]);
      s2[i] = row16_sum(s2[i]);
    }
    if (col == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) stat_dst[i] = make_float2(s1[i], s2[i]);
    }
  }
}

template <int NCH, bool PRO>
int launch_rs(ConvArgs a, hipStream_t st) {
  constexpr size_t lds = 3 * (size_t)NCH * RS_CHB + NCH * 256;
  static_assert(lds <= 160 * 1024, "conv_rs: LDS exceeds 160 KiB");
  a.tiles_x = cdiv(a.W, 32);
  a.tiles_y = cdiv(a.H, 8);
  const int NT = a.Ntot / 64;
  int gw, GW;
  segk_conv_rs_grid(a.B, a.H, a.W, NT, &gw, &GW);
  a.persistent = 1;
  auto kern = conv_rs_kernel<NCH, PRO>;
  static bool attr_set[SEGK_MAX_DEVICES] = {};
  const int dev = segk_device_index();
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      SEGK_FAIL(-3, "conv_rs: cannot raise dynamic LDS limit");
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3(8 * gw), dim3(512), lds, st, a);
  SEGK_CHECK_LAUNCH("conv_rs");
  return 0;
}

}  // namespace

// workgroups per XCD (gw, a multiple of the NT channel tiles) and per channel tile (GW); one workgroup per CU at most
void segk_conv_rs_grid(int B, int H, int W, int NT, int* gw_out, int* GW_out) {
  const int MT = B * cdiv(W, 32) * cdiv(H, 8);
  int gw = segk_num_cus() / 8;
  gw -= gw % NT;
  const int need = ((MT + 7) / 8) * NT;
  if (gw > need) gw = need;
  if (gw < NT) gw = NT;
  *gw_out = gw;
  *GW_out = gw / NT;
}

// rows of BatchNorm partial sums the kernel writes: one per wave slab and workgroup of a channel tile
int segk_conv_rs_rows(int B, int H, int W, int n_p) {
  int gw, GW;
  segk_conv_rs_grid(B, H, W, n_p / 64, &gw, &GW);
  return 8 * GW * 4;
}

// bf16, 3x3, one or two 64-byte input chunks, 64-channel output tiles, image wider than 16 pixels
int segk_conv_use_rs(int cin_p, int n_p, int dtype, int W) {
  static const bool off = getenv("SEGK_NO_RS") != nullptr;    // A/B switch for tools/kbench.py (read once, used by the
  // tile-count query and the dispatch alike: the statistics buffer is sized for the kernel that runs)
  return !off && dtype == SEGK_DT_BF16 && (cin_p == 32 || cin_p == 64) && n_p % 64 == 0 && W > 16;
}

int segk_conv_rs_launch(const ConvArgs& a, hipStream_t st) {
  const int nch = (a.CA + a.CB) / 32;
  SEGK_REQUIRE(nch == 1 || nch == 2, "conv_rs: Cin=%d not served", a.CA + a.CB);
  SEGK_REQUIRE(a.Ntot % 64 == 0 && a.CO1 % 32 == 0 && !a.bias, "conv_rs: bad output configuration");
  if (a.scale) return nch == 2 ? launch_rs<2, true>(a, st) : launch_rs<1, true>(a, st);
  return nch == 2 ? launch_rs<2, false>(a, st) : launch_rs<1, false>(a, st);
}
