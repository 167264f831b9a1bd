// Weight-gradient kernels on CDNA4 matrix cores for NHWC activations (gfx950).
//
//   dW[n][tap][k] = sum over pixels p of  dz[p][n] * a[p + offset(tap)][k]
//
//   GEO 0 : Conv2d 3x3 pad 1   (reference unet/unet.py:16,19; clip/clipunet.py:87,90)   9 taps, halo 1
//   GEO 1 : Conv2d 1x1         (clip/clipunet.py:84,122)                                1 tap
//   GEO 2 : ConvTranspose2d(k=2,s=2) (unet.py:59, clipunet.py:83): dz := layer input (coarse grid),
//           a := output gradient on the 2x fine grid, taps (a,c) at stride 2                4 taps
//
// The contraction index is the PIXEL, while NHWC keeps channels contiguous, so both MFMA operands are
// "k-strided".  bf16: tiles are staged in their natural [pixel][32 channels] form (64-byte rows, one LDS
// region per 32-channel block) and fragments are fetched with the gfx950 transposing LDS read
// ds_read_b64_tr_b16, which hands each lane 4 consecutive pixels of one channel: two reads make one
// 32x32x16 operand.  Tap shifts are pixel-row address shifts of the same staged patch (no im2col).
// fp32: exact 32x32x2 MFMA, one ds_read_b32 per operand (lanes = 32 contiguous channels).
// Each wave owns a 32x32 (n,k) block for ALL taps (9 accumulators); split-K over spatial tiles writes
// fp32 slabs that segk_wgrad_reduce sums in fixed order (bit-stable, no atomics).
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"
#include "segk_internal.h"
#include "../../include/segk.h"

namespace {

// compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N-1 (inline-asm immediates need constants)
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <typename T, int GEO, int WC = 1> struct WG {
  static constexpr bool BF = (sizeof(T) == 2);
  // dz rows per tile (16 px wide).  The ConvTranspose 4 x 2 configuration is latency-bound (one 8-wave workgroup per CU,
  // synchronous staging): 8 rows per tile halve its iterations
  static constexpr int R = (GEO == 2) ? (BF ? (WC == 4 ? 8 : 4) : 2) : (BF ? 8 : 4);
  static constexpr int NTAPS = GEO == 0 ? 9 : (GEO == 1 ? 1 : 4);
  static constexpr int PH = GEO == 0 ? R + 2 : (GEO == 1 ? R : 2 * R);
  static constexpr int PW = GEO == 0 ? 18 : (GEO == 1 ? 16 : 32);
  static constexpr int BLKP = 32 * (int)sizeof(T);   // bytes of one pixel inside a 32-channel block
  static constexpr int PPB = BLKP / 16;             // 16-byte pieces per pixel per block
  static constexpr int NDZ = R * 16;                // dz pixels per tile
  static constexpr int NPP = PH * PW;               // patch pixels per tile
};

__device__ __forceinline__ int tap_pixel(int geo, int tap, int r, int x) {
  // patch-pixel index of dz pixel (r, x) under tap
  if (geo == 0) return (r + tap / 3) * 18 + x + tap % 3;
  if (geo == 1) return r * 16 + x;
  return (2 * r + (tap >> 1)) * 32 + 2 * x + (tap & 1);
}

template <typename T, int GEO, int WC, int WI>
__global__ __launch_bounds__((WC * WI > 4 ? WC * WI : 4) * 64, (sizeof(T) == 2 ? 2 : 1)) void wgrad_kernel(const WgradArgs a) {
  using G = WG<T, GEO, WC>;
  using E = ET<T>;
  constexpr int NT = (WC * WI > 4 ? WC * WI : 4) * 64;      // threads: four waves, or one per 32 x 32 block (4 x 2 tiles)
  constexpr int R = G::R, NTAPS = G::NTAPS, BLKP = G::BLKP, PPB = G::PPB;
  constexpr int DZ_PIECES = G::NDZ * WC * PPB, PA_PIECES = G::NPP * WI * PPB;
  constexpr int NDL = (DZ_PIECES + NT - 1) / NT, NPL = (PA_PIECES + NT - 1) / NT;
  constexpr int DZ_BYTES = WC * G::NDZ * BLKP;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const dzs = smem;
  char* const pas = smem + DZ_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = a.H, W = a.W;
  const int tiles_x = (W + 15) >> 4, tiles_y = (H + R - 1) / R;
  const int ntiles = a.B * tiles_y * tiles_x;

  const int KT = (a.CA + a.CB) / (32 * WI), NCT = (a.CD / (32 * WC)) * KT;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int s = lid / NCT, ct = lid - s * NCT;
  const int n0 = (ct / KT) * 32 * WC, k0 = (ct % KT) * 32 * WI;

  const T* src; int SC, kc0;                      // shifted-operand source of this k-tile
  if (k0 < a.CA) { src = (const T*)a.srcA; SC = a.CA; kc0 = k0; }
  else { src = (const T*)a.srcB; SC = a.CB; kc0 = k0 - a.CA; }
  const bool pro = (a.scale != nullptr);
  const int FH = GEO == 2 ? 2 * H : H, FW = GEO == 2 ? 2 * W : W;   // grid of the shifted operand

  // tile-invariant piece decomposition: NT % (W?*PPB) == 0, so a thread's channel slot is fixed
  static_assert(NT % (WC * PPB) == 0 && NT % (WI * PPB) == 0, "channel slot must be thread-invariant");
  const int dcc = tid % (WC * PPB), pcc = tid % (WI * PPB);
  int d_pix[NDL], p_pix[NPL];
#pragma unroll
  for (int i = 0; i < NDL; ++i) {
    const int q = tid + i * NT;
    d_pix[i] = q < DZ_PIECES ? q / (WC * PPB) : -1;
  }
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int q = tid + i * NT;
    p_pix[i] = q < PA_PIECES ? q / (WI * PPB) : -1;
  }
  float psc[E::VEC], psh[E::VEC];
  if (pro) {
#pragma unroll
    for (int j = 0; j < E::VEC; ++j) {
      psc[j] = a.scale[kc0 + pcc * E::VEC + j];
      psh[j] = a.shift[kc0 + pcc * E::VEC + j];
    }
  }

  uint4 dreg[NDL], preg[NPL];
  bool pok[NPL];
  auto prefetch = [&](int t) {
    const int b = t / (tiles_y * tiles_x);
    const int rem = t - b * tiles_y * tiles_x;
    const int y0 = (rem / tiles_x) * R, x0 = (rem % tiles_x) * 16;
#pragma unroll
    for (int i = 0; i < NDL; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);   // out-of-image pixels contribute zeros
      if (d_pix[i] >= 0) {
        const int gy = y0 + (d_pix[i] >> 4), gx = x0 + (d_pix[i] & 15);
        if (gy < H && gx < W)
          v = *(const uint4*)((const T*)a.dz + ((size_t)(b * H + gy) * W + gx) * a.CD + n0 + dcc * E::VEC);
      }
      dreg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      pok[i] = false;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (p_pix[i] >= 0) {
        const int py = p_pix[i] / G::PW, px = p_pix[i] - py * G::PW;
        int gy, gx;
        if (GEO == 0) { gy = y0 + py - 1; gx = x0 + px - 1; }
        else if (GEO == 1) { gy = y0 + py; gx = x0 + px; }
        else { gy = 2 * y0 + py; gx = 2 * x0 + px; }
        if (gy >= 0 && gy < FH && gx >= 0 && gx < FW) {
          pok[i] = true;
          v = *(const uint4*)(src + ((size_t)(b * FH + gy) * FW + gx) * SC + kc0 + pcc * E::VEC);
        }
      }
      preg[i] = v;
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < NDL; ++i)
      if (d_pix[i] >= 0) {
        const int blk = dcc / PPB, c16 = dcc - blk * PPB;
        *(uint4*)(dzs + blk * (G::NDZ * BLKP) + d_pix[i] * BLKP + c16 * 16) = dreg[i];
      }
#pragma unroll
    for (int i = 0; i < NPL; ++i)
      if (p_pix[i] >= 0) {
        const int blk = pcc / PPB, c16 = pcc - blk * PPB;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (pok[i]) {
          v = preg[i];
          if (pro) {
            float f[E::VEC];
            unpack16<T>(v, f);
#pragma unroll
            for (int j = 0; j < E::VEC; ++j) f[j] = fmaxf(fmaf(f[j], psc[j], psh[j]), 0.f);
            v = pack16<T>(f);
          }
        }
        *(uint4*)(pas + blk * (G::NPP * BLKP) + p_pix[i] * BLKP + c16 * 16) = v;
      }
  };

  const bool computes = wave < WC * WI;
  const int wc = wave / WI, wi = wave - wc * WI;
  f32x16 acc[NTAPS];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const char* const dzb = dzs + wc * (G::NDZ * BLKP);
  const char* const pab = pas + wi * (G::NPP * BLKP);

  int t = s;
  if (t < ntiles) prefetch(t);
  for (; t < ntiles; t += a.S) {
    stage();
    __syncthreads();
    if (t + a.S < ntiles) prefetch(t + a.S);
    if (computes) {
      if constexpr (G::BF) {
        // lane -> (group g: channel half gsub, k half h; q = pixel row of the 4x16 block, p = 4-col piece)
        const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
        const int gsub = g & 1, h = g >> 1;
        const int coff = (16 * gsub + 4 * p) * 2;
        if constexpr (GEO == 2) {
          // ConvTranspose: few MFMAs per tile (4 rows x 4 taps), so an LDS round trip in front of every MFMA is most of the
          // tile's time: all dz fragments first, then the four tap fragments of row r+1 are read under the MFMAs of row r
          typedef __attribute__((address_space(3))) s16x4* lds_v4;
          const int xa = 8 * h + q;
          bf16x8 fa[R], fb[2][NTAPS];
          auto rd_b = [&](int r, bf16x8 (&F)[NTAPS]) {
#pragma unroll
            for (int tp = 0; tp < NTAPS; ++tp) {
              const int p0 = tap_pixel(GEO, tp, r, xa), p1 = tap_pixel(GEO, tp, r, xa + 4);
              const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(pab + p0 * BLKP + coff));
              const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(pab + p1 * BLKP + coff));
              F[tp] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
            }
          };
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(dzb + (r * 16 + xa) * BLKP + coff));
            const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(dzb + (r * 16 + xa + 4) * BLKP + coff));
            fa[r] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
          }
          rd_b(0, fb[0]);
#pragma unroll
          for (int r = 0; r < R; ++r) {
            if (r + 1 < R) rd_b(r + 1, fb[(r + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tp = 0; tp < NTAPS; ++tp) acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[r], fb[r & 1][tp], acc[tp], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
      #pragma unroll
        for (int r = 0; r < R; ++r) {
          const int xa = 8 * h + q;   // pixel column of read 0; read 1 is +4
          typedef __attribute__((address_space(3))) s16x4* lds_v4;
          const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(dzb + (r * 16 + xa) * BLKP + coff));
          const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(dzb + (r * 16 + xa + 4) * BLKP + coff));
          const bf16x8 fa = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
          for (int tp = 0; tp < NTAPS; ++tp) {
            const int p0 = tap_pixel(GEO, tp, r, xa), p1 = tap_pixel(GEO, tp, r, xa + 4);
            const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(pab + p0 * BLKP + coff));
            const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(pab + p1 * BLKP + coff));
            const bf16x8 fb = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
            acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[tp], 0, 0, 0);
          }
        }
        }
      } else {
        const int i32 = lane & 31, h = lane >> 5;
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
          for (int kp = 0; kp < 8; ++kp) {
            const int x = 2 * kp + h;
            const float fa = *(const float*)(dzb + (r * 16 + x) * BLKP + i32 * 4);
#pragma unroll
            for (int tp = 0; tp < NTAPS; ++tp) {
              const float fb = *(const float*)(pab + tap_pixel(GEO, tp, r, x) * BLKP + i32 * 4);
              acc[tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[tp], 0, 0, 0);
            }
          }
        }
      }
    }
    __syncthreads();
  }

  if (computes) {
    const int K = a.CA + a.CB;
    const int col = lane & 31, lh = lane >> 5;
    float* const slab = a.slabs + (size_t)s * a.CD * NTAPS * K;
#pragma unroll
    for (int tp = 0; tp < NTAPS; ++tp)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        slab[((size_t)(n0 + wc * 32 + row) * NTAPS + tp) * K + k0 + wi * 32 + col] = acc[tp][r];
      }
  }
}

#ifdef SEGK_WGRAD_STAMPS
__device__ unsigned long long g_wstamps[256 * 8 * 8];   // diagnostic build: [workgroup][wave][phase sums]
#endif
// ------------------------------------------------------------------------------------------------
// bf16 3x3 weight gradient with LDS-DMA staging (global_load_lds: no staging VGPRs, no ds_write pass).
//   U = the un-shifted operand  (8 x 16 pixel tile, WU blocks of 32 channels)
//   V = the tap-shifted operand (10 x 18 patch,     WV blocks), raw, out-of-image pixels read a zero page
// Both tiles are double-buffered in LDS; the DMA of tile t+1 flies under the MFMAs of tile t.  The three
// kx-fragments of a patch row serve taps (ky, kx) of three consecutive dz rows, so they rotate through
// registers and each dz row fetches ONE new patch row (6 transposing reads instead of 18).
// PRO (the second conv of a block: its input is relu(bn(z1))): the roles are swapped so that the operand
// needing the BatchNorm+ReLU transform is the un-shifted one -- the transform is then applied to ONE
// fragment per dz row, in registers, with the lane's own channel constants, and the shifted operand (dz)
// needs no transform and gets exact-zero halos from the zero page.  Swapping A/B in the MFMA keeps the
// accumulator oriented [co][ci]; the tap index mirrors (tap -> 8 - tap).
// WC x WI waves of 32 x 32 blocks make the workgroup's (n, k) tile; 4 x 2 (eight waves, one workgroup per CU) moves
// 0.7 of the LDS-DMA bytes per FLOP of two 2 x 2 workgroups: the deep layers are bound by that L2 -> LDS fill rate.
// RAG: the image is not whole 8 x 16 tiles (H % 8 or W % 16 != 0: the 14 / 28 / 56-pixel levels of the CLIP decoder): a piece's
// pixel is then tested against the tile's valid rows / columns (two compares more per piece) instead of the four
// "this side of the patch is outside" bits, and pixels of the tile beyond the image read the zero page like halo pixels do.
template <int WC, int WI, bool PRO, bool RAG>
__global__ __launch_bounds__((WC * WI > 4 ? WC * WI : 4) * 64, 2) void wgrad_dma_kernel(const WgradArgs a) {
  typedef bf16_t T;
  constexpr int NWV = WC * WI > 4 ? WC * WI : 4;          // waves per workgroup
  constexpr int R = 8, NTAPS = 9, BLKP = 64;
  constexpr int WU = PRO ? WI : WC, WV = PRO ? WC : WI;
  constexpr int NU = R * 16, NV = 192;                    // pixels per block region (patch: 180 used)
  constexpr int UB = WU * NU * BLKP, VB = WV * NV * BLKP, BUF = UB + VB;
  constexpr int NINSTR = WU * 8 + WV * 12;                // 1 KiB DMA wave-instructions per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W;
  const int tiles_x = (W + 15) >> 4, tiles_y = (H + R - 1) / R;
  const int ntiles = a.B * tiles_y * tiles_x;
  const int KT = (a.CA + a.CB) / (32 * WI), NCT = (a.CD / (32 * WC)) * KT;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int s = lid / NCT, ct = lid - s * NCT;
  const int n0 = (ct / KT) * 32 * WC, k0 = (ct % KT) * 32 * WI;

  // operand roles
  const T* srck; int SCk, kc0;                            // the k-tile's source (first or second concat operand)
  if (k0 < a.CA) { srck = (const T*)a.srcA; SCk = a.CA; kc0 = k0; }
  else { srck = (const T*)a.srcB; SCk = a.CB; kc0 = k0 - a.CA; }
  const T* const ubase = PRO ? srck + kc0 : (const T*)a.dz + n0;
  const int UC = PRO ? SCk : a.CD;
  const T* const vbase = PRO ? (const T*)a.dz + n0 : srck + kc0;
  const int VC = PRO ? a.CD : SCk;
  const char* const zeros = (const char*)a.zeros;

  const int lp = lane >> 2, lc = lane & 3;                // pixel-in-chunk / 16-byte piece of the lane
  // ---- DMA issue.  Per tile every wave moves NI = ceil(NINSTR / NWV) one-KiB pieces.  Everything about a piece that does
  // not depend on the tile is computed ONCE per kernel: the lane's byte offset from the tile's first pixel (poff) and five
  // flag bits (which side of the 10 x 18 patch the lane's pixel lies on: top, bottom, left, right; bit 4 = a pixel of the
  // patch region's unused tail).  Per tile the address of a piece is then scalar tile base + poff, and a lane reads the zero
  // page instead when its flags meet the tile's "this side is outside the image" bits: ~8 vector instructions per piece
  // where a general form (any tile may cross the image edge) spends ~40 on
  // divisions, bounds tests and 64-bit multiplies, issued by all eight waves in front of every tile's MFMAs (the RAG
  // instances add two compares per piece for images that are not whole 8 x 16 tiles).
  constexpr int NI = (NINSTR + NWV - 1) / NWV;
  int poff[NI];
  int pyx[RAG ? NI : 1];                                  // RAG: (patch row << 8) | patch column of the lane's pixel (U: r+1, lp+1)
  unsigned pflag[(NI + 5) / 6];
#pragma unroll
  for (int q = 0; q < (NI + 5) / 6; ++q) pflag[q] = 0u;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int j = min(wave + NWV * i, NINSTR - 1);         // wave-uniform
    if (j < WU * 8) {
      const int blk = j >> 3, r = j & 7;
      poff[i] = ((r * W + lp) * UC + blk * 32 + lc * 8) * 2;
      if constexpr (RAG) pyx[i] = ((r + 1) << 8) | (lp + 1);
    } else {
      const int jj = j - WU * 8;
      const int blk = jj / 12, ch = jj - blk * 12;
      const int pp = ch * 16 + lp;
      const int py = pp / 18, px = pp - py * 18;
      poff[i] = (((py - 1) * W + (px - 1)) * VC + blk * 32 + lc * 8) * 2;
      const unsigned f = (py == 0 ? 1u : 0u) | (py == 9 ? 2u : 0u) | (px == 0 ? 4u : 0u) | (px == 17 ? 8u : 0u) |
                         (pp >= 180 ? 16u : 0u);
      pflag[i / 6] |= f << (5 * (i % 6));
      if constexpr (RAG) pyx[i] = (py << 8) | px;
    }
  }
  auto issue_tile = [&](int t, int bufsel) {
    const int bimg = t / (tiles_y * tiles_x);
    const int rem = t - bimg * tiles_y * tiles_x;
    const int y0 = (rem / tiles_x) * R, x0 = (rem % tiles_x) * 16;
    {                                                      // every tile lies inside the image (its halo may not): launcher
      const unsigned out = (y0 == 0 ? 1u : 0u) | (y0 + R >= H ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (x0 + 16 >= W ? 8u : 0u) | 16u;
      const size_t pix0 = (size_t)(bimg * H + y0) * W + x0;
      const char* const ub0 = (const char*)(ubase + pix0 * UC);   // wave-uniform tile bases
      const char* const vb0 = (const char*)(vbase + pix0 * VC);
      char* const bb = smem + bufsel * BUF;
      const char* const zsrc = zeros + lc * 16;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int j = wave + NWV * i;                      // wave-uniform
        if (j >= NINSTR) break;
        const bool is_u = j < WU * 8;
        const int jj = j - WU * 8;
        char* const dst = is_u ? bb + (j >> 3) * (NU * BLKP) + (j & 7) * 1024
                               : bb + UB + (jj / 12) * (NV * BLKP) + (jj % 12) * 1024;
        const char* const base = is_u ? ub0 : vb0;
        bool bad;
        if constexpr (RAG) {                               // rows y0 + py - 1 and columns x0 + px - 1 against the image
          const int py = pyx[i] >> 8, px = pyx[i] & 255;
          bad = (((pflag[i / 6] >> (5 * (i % 6))) & 16u) != 0u) | (py == 0 ? y0 == 0 : py > H - y0) |
                (px == 0 ? x0 == 0 : px > W - x0);
        } else {
          bad = ((pflag[i / 6] >> (5 * (i % 6))) & out) != 0u;
        }
        const char* const src = bad ? zsrc : base + (ptrdiff_t)poff[i];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    }
  };

  const bool computes = wave < WC * WI;
  const int wc = wave / WI, wi = wave - wc * WI;
  const int wu = PRO ? wi : wc, wv = PRO ? wc : wi;
  f32x16 acc[NTAPS];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // lane -> (group g: channel half gsub, k half h; q = pixel row of the 4x16 block, p = 4-col piece)
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
  const int gsub = g & 1, h = g >> 1;
  const int coff = (16 * gsub + 4 * p) * 2;
  const int xa = 8 * h + q;                               // pixel column of read 0; read 1 is +4
  float psc = 1.f, psh = 0.f;
  if (PRO) {
    const int c = kc0 + wu * 32 + 16 * gsub + i16;         // the lane's own channel of the un-shifted operand
    psc = a.scale[c];
    psh = a.shift[c];
  }
  // Fragment reads go through inline asm: hipcc orders every LDS load it can see behind ALL pending LDS-DMA (an
  // s_waitcnt vmcnt(0) in front of the tile's first read), which drained the next tile's DMA before the MFMAs it was
  // meant to fly under.  The destination is valid only after the counted lgkmcnt wait placed by the loop below (LDS
  // returns in order), and nothing is scheduled across those waits (sched_barrier).
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  auto frag_rd = [&](s16x8& dst, int vaddr, auto OFFc) {      // vaddr: lane part + tile buffer base; OFF: pixel offset
    constexpr int OFF = decltype(OFFc)::value;
    s16x4 f0, f1;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f0) : "v"(vaddr), "n"(OFF));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f1) : "v"(vaddr), "n"(OFF + 4 * BLKP));
    dst = __builtin_shufflevector(f0, f1, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  // BatchNorm(scale, shift) + ReLU on the 8 pixels of one channel; vmask bit j = pixel j lies inside the image.
  // Instruction count matters here (VALU issued beside the MFMAs): scalar fp32 fma (packed fp32 is slower next to
  // MFMAs), one bf16 conversion per pair, the ReLU as a packed signed-16-bit max on the bf16 bit patterns
  // (rounding to bf16 is monotonic and sign-preserving, so max(round(x), 0) == round(max(x, 0)) bit for bit),
  // and the image-edge masking only on partial tiles.
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  auto transform = [&](s16x8 v, unsigned vmask) {
    const uint4 raw = __builtin_bit_cast(uint4, v);
    float f[8];
    unpack16<T>(raw, f);
    unsigned o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float x0 = fmaf(f[2 * i], psc, psh), x1 = fmaf(f[2 * i + 1], psc, psh);
      unsigned pk;
      pk = cvt_pk_bf16(x0, x1);
      o[i] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk), (s16x2){0, 0}));
    }
    if (vmask != 0xffu) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        o[i] &= ((vmask >> (2 * i)) & 1 ? 0x0000ffffu : 0u) | ((vmask >> (2 * i + 1)) & 1 ? 0xffff0000u : 0u);
    }
    return __builtin_bit_cast(s16x8, make_uint4(o[0], o[1], o[2], o[3]));
  };

  // Diagnostic build only (-DSEGK_WGRAD_STAMPS, tools/stamp_build.sh): per-wave cycle sums of a tile's phases -- [0] DMA issue +
  // MFMA rows, [1] wait for the next tile's DMA, [2] barrier -- written over the head of the slab buffer; results are garbage.
#ifdef SEGK_WGRAD_STAMPS
  unsigned long long wst[3] = {0, 0, 0}, wlast, wt0;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wlast)::"memory");
  wt0 = wlast;
#define WSTAMP(i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
                       wst[i] += t_ - wlast; wlast = t_; } while (0)
#else
#define WSTAMP(i) do {} while (0)
#endif
  int t = s, cur = 0;
  if (t < ntiles) issue_tile(t, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  WSTAMP(2);
  // The two waves of a SIMD (w and w + NWV/2 of an eight-wave workgroup) issue the next tile's DMA at different points of
  // the tile: the first half in front of its MFMAs, the second half behind the MFMAs of dz row WGRAD_STAGGER_ROW -- one
  // wave's memory issue then runs beside its partner's matrix work instead of both queueing for the vector-memory pipe
  // first and for the matrix pipe afterwards (same idea as the class A / B waves of conv_rs.hip)
#ifndef WGRAD_STAGGER_ROW
#define WGRAD_STAGGER_ROW 1
#endif
  constexpr int SROW = (NWV == 8) ? WGRAD_STAGGER_ROW : -1;
  const bool late_issue = SROW >= 0 && wave >= NWV / 2;
  for (; t < ntiles; t += a.S) {
    const bool has_next = t + a.S < ntiles;
    if (has_next && !late_issue) issue_tile(t + a.S, cur ^ 1);    // DMA of the next tile flies under the MFMAs
    if (computes) {
      unsigned xmask = 0xffu, rows_ok = R;
      if (PRO) {                                           // partial tiles: mask pixels outside the image
        const int bimg = t / (tiles_y * tiles_x);
        const int rem = t - bimg * tiles_y * tiles_x;
        const int y0 = (rem / tiles_x) * R, x0 = (rem % tiles_x) * 16;
        rows_ok = (unsigned)min(R, H - y0);
        xmask = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j)   // fragment element j = pixel column 8h + j of the tile row (two 4-row blocks)
          xmask |= (x0 + 8 * h + j < W ? 1u : 0u) << j;
      }
      // Read schedule of a tile (transposing reads, two per fragment): the three kx-fragments of patch rows 0..2 and
      // the dz row 0 up front; then per dz row r: [dz row r+1] -> 3 MFMAs (ky = 0) -> [patch row r+3, into the
      // registers row r has just left] -> 6 MFMAs (ky = 1, 2).  Reads retire in order, so "all but the N youngest
      // reads" (s_waitcnt lgkmcnt(N)) names exactly what has arrived.
      const int ua = cur * BUF + wu * (NU * BLKP) + xa * BLKP + coff;
      const int va = cur * BUF + UB + wv * (NV * BLKP) + xa * BLKP + coff;
      // the first MFMAs (ky = 0 of dz row 0) need dz row 0 and patch row 0 only: those 8 reads go first, patch rows 1 and 2
      // (12 reads) land under them -- the row loop's first wait leaves 14 reads in flight instead of draining all 20
      s16x8 fv[3][3], fu[2];
      frag_rd(fu[0], ua, std::integral_constant<int, 0>{});
      static_for<0, 9>([&](auto IC) {
        constexpr int y = decltype(IC)::value / 3, kx = decltype(IC)::value % 3;
        frag_rd(fv[y][kx], va, std::integral_constant<int, (y * 18 + kx) * BLKP>{});
      });
      if (PRO) {
        asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");     // dz row 0 (the oldest two of 20 reads; the field holds <= 15)
        __builtin_amdgcn_sched_barrier(0);
        fu[0] = transform(fu[0], rows_ok > 0 ? xmask : 0u);
      }
      static_for<0, R>([&](auto RC) {
        constexpr int r = decltype(RC)::value;
        if constexpr (r + 1 < R) frag_rd(fu[(r + 1) & 1], ua, std::integral_constant<int, (r + 1) * 16 * BLKP>{});
        // dz row r and patch row r are in registers: younger are patch row r+2 (6 reads; at r = 0 patch rows 1 and 2: 12) and
        // dz row r+1 (2)
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((r == 0 ? 12 : 6) + (r + 1 < R ? 2 : 0)) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 fur = __builtin_bit_cast(bf16x8, fu[r & 1]);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {                   // ky = 0 consumes patch row r ...
          const bf16x8 fvr = __builtin_bit_cast(bf16x8, fv[r % 3][kx]);
          const int ti = PRO ? 8 - kx : kx;
          acc[ti] = PRO ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fvr, fur, acc[ti], 0, 0, 0)
                        : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fur, fvr, acc[ti], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (r + 1 < R) {                         // ... whose register slot then takes patch row r+3
          static_for<0, 3>([&](auto KC) {
            constexpr int kx = decltype(KC)::value;
            frag_rd(fv[r % 3][kx], va, std::integral_constant<int, ((r + 3) * 18 + kx) * BLKP>{});
          });
        }
        // patch rows r+1 and r+2 are in registers: younger are dz row r+1 (2) and patch row r+3 (6)
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(r + 1 < R ? 8 : 0) : "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ky = 1; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const bf16x8 fvr = __builtin_bit_cast(bf16x8, fv[(r + ky) % 3][kx]);
            const int ti = PRO ? 8 - (ky * 3 + kx) : ky * 3 + kx;
            acc[ti] = PRO ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fvr, fur, acc[ti], 0, 0, 0)
                          : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fur, fvr, acc[ti], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (r == SROW) {
          if (has_next && late_issue) issue_tile(t + a.S, cur ^ 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (PRO && r + 1 < R) {                  // the next dz row has arrived (only patch row r+3 is younger)
          asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          fu[(r + 1) & 1] = transform(fu[(r + 1) & 1], (unsigned)(r + 1) < rows_ok ? xmask : 0u);
        }
      });
    }
    WSTAMP(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // next tile's DMA has landed
    WSTAMP(1);
    __syncthreads();
    WSTAMP(2);
    cur ^= 1;
  }

  if (computes) {
    const int K = a.CA + a.CB;
    const int col = lane & 31, lh = lane >> 5;
    float* const slab = a.slabs + (size_t)s * a.CD * NTAPS * K;
#pragma unroll
    for (int tp = 0; tp < NTAPS; ++tp)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        slab[((size_t)(n0 + wc * 32 + row) * NTAPS + tp) * K + k0 + wi * 32 + col] = acc[tp][r];
      }
  }
#ifdef SEGK_WGRAD_STAMPS
  {
    unsigned long long t_;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");   // the slab stores are out
    if (lane == 0 && blockIdx.x < 256) {
      unsigned long long* o = g_wstamps + ((size_t)blockIdx.x * 8 + wave) * 8;
      o[0] = wst[0]; o[1] = wst[1]; o[2] = wst[2]; o[3] = t_ - wlast; o[4] = t_ - wt0;
    }
  }
#endif
}

template <int WC, int WI, bool PRO, bool RAG>
int launch_dma(const WgradArgs& a, hipStream_t st) {
  constexpr int WU = PRO ? WI : WC, WV = PRO ? WC : WI;
  constexpr size_t lds = 2 * ((size_t)WU * 128 * 64 + (size_t)WV * 192 * 64);
  static_assert(lds <= 160 * 1024, "wgrad_dma: LDS exceeds 160 KiB");
  const int NCT = (a.CD / (32 * WC)) * ((a.CA + a.CB) / (32 * WI));
  auto kern = wgrad_dma_kernel<WC, WI, PRO, RAG>;
  static bool attr_set[SEGK_MAX_DEVICES] = {};     // per device: the attribute is device state
  const int dev_ = segk_device_index();
  if (!attr_set[dev_]) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      SEGK_FAIL(-3, "wgrad_dma: cannot raise dynamic LDS limit");
    attr_set[dev_] = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.S * NCT), dim3((WC * WI > 4 ? WC * WI : 4) * 64), lds, st, a);
#ifdef SEGK_WGRAD_STAMPS
  { void* sp = nullptr; if (hipGetSymbolAddress(&sp, HIP_SYMBOL(g_wstamps)) == hipSuccess) (void)hipMemcpyAsync(a.slabs, sp, sizeof(unsigned long long) * 256 * 8 * 8, hipMemcpyDeviceToDevice, st); }
#endif
  SEGK_CHECK_LAUNCH("wgrad_dma");
  return 0;
}

template <int WC, int WI>
int launch_dma_pro(const WgradArgs& a, hipStream_t st) {
  const bool rag = a.H % 8 != 0 || a.W % 16 != 0;       // images that are not whole 8 x 16 tiles
  if (rag) return a.scale ? launch_dma<WC, WI, true, true>(a, st) : launch_dma<WC, WI, false, true>(a, st);
  return a.scale ? launch_dma<WC, WI, true, false>(a, st) : launch_dma<WC, WI, false, false>(a, st);
}

template <typename T, int GEO, int WC, int WI>
int launch_cfg(const WgradArgs& a, hipStream_t st) {
  using G = WG<T, GEO, WC>;
  const size_t lds = (size_t)WC * G::NDZ * G::BLKP + (size_t)WI * G::NPP * G::BLKP;
  const int NCT = (a.CD / (32 * WC)) * ((a.CA + a.CB) / (32 * WI));
  auto kern = wgrad_kernel<T, GEO, WC, WI>;
  static bool attr_set[SEGK_MAX_DEVICES] = {};     // per device: the attribute is device state
  const int dev_ = segk_device_index();
  if (!attr_set[dev_]) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      SEGK_FAIL(-3, "wgrad: cannot raise dynamic LDS limit");
    attr_set[dev_] = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.S * NCT), dim3((WC * WI > 4 ? WC * WI : 4) * 64), lds, st, a);
#ifdef SEGK_WGRAD_STAMPS
  { void* sp = nullptr; if (hipGetSymbolAddress(&sp, HIP_SYMBOL(g_wstamps)) == hipSuccess) (void)hipMemcpyAsync(a.slabs, sp, sizeof(unsigned long long) * 256 * 8 * 8, hipMemcpyDeviceToDevice, st); }
#endif
  SEGK_CHECK_LAUNCH("wgrad");
  return 0;
}

template <typename T, int GEO>
int launch_geo(const WgradArgs& a, hipStream_t st) {
  const bool wc2 = a.CD % 64 == 0;
  const bool wi2 = a.CA % 64 == 0 && a.CB % 64 == 0;
  if constexpr (GEO == 0 && sizeof(T) == 2) {
    // bf16 3x3: LDS-DMA kernel (needs the caller's zero page); images that are not whole 8 x 16 tiles take its RAG instances
    if (a.zeros) {
      if (segk_wgrad_wc(a.CD, a.CA, a.CB, 0, SEGK_DT_BF16) == 4) return launch_dma_pro<4, 2>(a, st);
      if (wc2 && wi2) return launch_dma_pro<2, 2>(a, st);
      if (wc2) return launch_dma_pro<2, 1>(a, st);
      if (wi2) return launch_dma_pro<1, 2>(a, st);
      return launch_dma_pro<1, 1>(a, st);
    }
  }
  if constexpr (GEO == 2 && sizeof(T) == 2) {     // ConvTranspose: 128 x 64 tiles halve the operand re-reads (it is traffic-bound)
    if (segk_wgrad_wc(a.CD, a.CA, a.CB, 2, SEGK_DT_BF16) == 4) return launch_cfg<T, GEO, 4, 2>(a, st);
  }
  if (wc2 && wi2) return launch_cfg<T, GEO, 2, 2>(a, st);
  if (wc2) return launch_cfg<T, GEO, 2, 1>(a, st);
  if (wi2) return launch_cfg<T, GEO, 1, 2>(a, st);
  return launch_cfg<T, GEO, 1, 1>(a, st);
}

template <typename T>
int launch_t(const WgradArgs& a, int geo, hipStream_t st) {
  if (geo == 0) return launch_geo<T, 0>(a, st);
  if (geo == 1) return launch_geo<T, 1>(a, st);
  return launch_geo<T, 2>(a, st);
}

}  // namespace

// 32-channel blocks of the dz operand per workgroup (the host sizes the split-K slabs with it: segk_wgrad_split)
int segk_wgrad_wc(int CD, int CA, int CB, int geo, int dtype) {
  static const bool off = getenv("SEGK_WGRAD_NO_WC4") != nullptr;      // A/B switch for tools/kbench.py
  if ((geo == 0 || geo == 2) && dtype == SEGK_DT_BF16 && !off && CD % 128 == 0 && CA % 64 == 0 && CB % 64 == 0) return 4;
  return CD % 64 == 0 ? 2 : 1;
}
// split-K factor over spatial tiles: enough workgroups to fill the chip (two 4-wave workgroups or one 8-wave workgroup
// per CU), never more slabs than tiles
int segk_wgrad_split(int tiles, int CD, int CA, int CB, int geo, int dtype) {
  if (tiles <= 0 || CD <= 0 || CA <= 0 || CB < 0 || CD % 32 || CA % 32 || CB % 32) return 0;   // channel counts are padded
  const int wc = segk_wgrad_wc(CD, CA, CB, geo, dtype);
  const int wi = (CA % 64 == 0 && CB % 64 == 0) ? 2 : 1;
  const int nct = (CD / (32 * wc)) * ((CA + CB) / (32 * wi));
  const int target = wc == 4 ? 256 : 512;     // a 4 x 2 workgroup (eight waves) fills a CU
  if (wc == 4 && geo == 2) tiles = (tiles + 1) / 2;         // its ConvTranspose tiles are 8 rows (segk_wgrad_tiles counts 4-row tiles);
                                                            // a slab whose index exceeds the tile count is written as zeros
  int S = nct < target ? target / nct : 1;
  if (S > tiles) S = tiles;
  return S < 1 ? 1 : S;
}

int segk_wgrad_tiles(int B, int H, int W, int geo, int dtype) {
  const bool bf = dtype == SEGK_DT_BF16;
  const int R = geo == 2 ? (bf ? 4 : 2) : (bf ? 8 : 4);
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  const long long n = (long long)B * cdiv(H, R) * cdiv(W, 16);
  return n > 0x7fffffffLL ? 0 : (int)n;
}

int segk_wgrad_launch(const WgradArgs& a, int geo, int dtype, hipStream_t st) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "wgrad: bad dtype %d", dtype);
  SEGK_REQUIRE(geo >= 0 && geo <= 2, "wgrad: bad geometry %d", geo);
  SEGK_REQUIRE(a.dz && a.srcA && a.slabs, "wgrad: null pointer");
  SEGK_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0, "wgrad: bad shape");
  SEGK_REQUIRE(a.CD > 0 && a.CD % 32 == 0 && a.CA > 0 && a.CA % 32 == 0 && a.CB >= 0 && a.CB % 32 == 0,
               "wgrad: channel counts must be multiples of 32 (CD=%d CA=%d CB=%d)", a.CD, a.CA, a.CB);
  SEGK_REQUIRE((a.CB == 0) == (a.srcB == nullptr), "wgrad: second source mismatch");
  SEGK_REQUIRE(!(a.scale && a.CB) && ((a.scale == nullptr) == (a.shift == nullptr)), "wgrad: bad BN prologue");
  SEGK_REQUIRE(a.S >= 1 && a.S <= 65535, "wgrad: bad split-K factor %d", a.S);
  SEGK_REQUIRE((long long)a.B * a.H * a.W * 4 < 2147483647LL, "wgrad: pixel index overflows int32");
  return dtype == SEGK_DT_BF16 ? launch_t<bf16_t>(a, geo, st) : launch_t<float>(a, geo, st);
}
