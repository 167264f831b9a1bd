// Implicit-GEMM convolution for NHWC activations on CDNA4 matrix cores (gfx950).
//
// One kernel family serves, by geometry/flags:
//   GEO 0 : Conv2d 3x3 pad 1 stride 1   -- forward (reference unet/unet.py:16,19; clip/clipunet.py:87,90)
//                                          and data-gradient (same kernel, flipped+transposed weights)
//   GEO 1 : 1x1 "conv"                  -- ConvTranspose2d(k=2,s=2) forward as GEMM + pixel-shuffle store
//                                          (unet.py:59, clipunet.py:83), its data-gradient as a 2x2
//                                          un-shuffle gather GEMM, and the CLIP 1x1 projections (clipunet.py:84,122)
//
// GEMM view: M = 256 output pixels of one spatial tile (16x16 or 8x32), N = BN output channels,
// K = taps x Cin.  The input patch (tile + halo) of one 64-byte channel chunk is staged ONCE into LDS
// and reused by all 9 taps as shifted row addresses (the 3x3 im2col never exists); weight tiles stream
// through a second double-buffered LDS region.  Staging is register-staged (global -> VGPR -> LDS,
// issue-early / write-late) so the previous layer's BatchNorm+ReLU can be applied on the fly
// (prologue fusion) and so out-of-image halo pixels become exact zeros.  MFMA: bf16 32x32x16 or exact
// fp32 32x32x2 with a byte-identical LDS image (a 16-byte fragment read is 8 bf16 k-values or 4 fp32).
// Epilogue: optional bias, per-channel sum / sum-of-squares partials for training-mode BatchNorm taken
// from the fp32 accumulators (deterministic per-tile partials, no atomics), LDS transpose, 16-byte
// coalesced NHWC stores (optionally split over two destinations, or pixel-shuffled for ConvTranspose).
#include <type_traits>
#include "common.hpp"
#include "segk_internal.h"

// pixel-tile geometry shared by the launcher and the BN-statistics sizing query (api.hip)
int segk_conv_bm(int geo, int unit) { return (unit % 128 == 0 || geo != 0) ? 256 : 128; }
int segk_conv_twl(int bm, int W) { return bm == 128 ? 4 : (W > 16 ? 5 : 4); }   // 8x16 | 8x32 | 16x16 tiles

namespace {

constexpr int PIXB = 80;  // LDS pitch of one pixel's 64-byte K-chunk: +16 B pad -> conflict-free ds_read_b128

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                  acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  // lane (r, h) supplies k-slot h of each of the four 32x32x2 products: floats [4h + j] of the 8-float block
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
};

template <int V> using IC = std::integral_constant<int, V>;

// Tile: BM = WM*MF*32 output pixels (TH x TW) by BN = WN*NF*32 output channels, WM*WN waves.
// PBUF: patch buffers (2 = next chunk staged under the current chunk's MFMAs; 1 = smaller LDS footprint so
// that three 4-wave workgroups share a CU and overlap each other's load / MFMA / store phases).
template <typename T, int GEO, int WM, int WN, int MF, int NF, int PBUF>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_igemm_kernel(const ConvArgs a) {
  using E = ET<T>;
  constexpr int NW = WM * WN, NTHR = NW * 64;
  constexpr int BM = WM * MF * 32, BN = WN * NF * 32;
  constexpr int NTAPS = (GEO == 0) ? 9 : 1;
  constexpr int TPS = (GEO == 0) ? 3 : 1;   // taps per pipeline step (one kernel row)
  constexpr int SPC = (GEO == 0) ? 3 : 1;   // steps per channel chunk
  constexpr int HALO = (GEO == 0) ? 1 : 0;
  constexpr int NPL = (GEO == 0) ? (BM == 256 ? 3 : 4) : (BM * 4 + NTHR - 1) / NTHR;  // patch pieces / thread
  constexpr int NWP = TPS * BN * 4;         // weight 16-byte pieces per step
  constexpr int NWL = (NWP + NTHR - 1) / NTHR;
  static_assert(BM == 256 || BM == 128, "pixel tile is 256 or 128");
  static_assert(NTHR % 4 == 0, "a thread keeps one 16-byte slot of the 64-byte chunk");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int lr = lane & 31, lh = lane >> 5;

  const int twl = a.twl, tw = 1 << twl, th = BM >> twl;
  const int PW = tw + 2 * HALO, PH = th + 2 * HALO;
  const int ROWP = (PW * PIXB + 255) & ~255;
  const int PB = PH * ROWP;
  constexpr int WB = TPS * BN * PIXB;

  // ---- block -> (pixel tile, channel tile); XCD-contiguous so the N-tiles of a pixel tile share an L2
  const int NT = a.Ntot / BN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = lid / NT, nt = lid - mt * NT;
  const int tpi = a.tiles_x * a.tiles_y;
  const int b = mt / tpi;
  const int trem = mt - b * tpi;
  const int tyi = trem / a.tiles_x, txi = trem - tyi * a.tiles_x;
  const int y0 = tyi * th, x0 = txi * tw;
  const int n0 = nt * BN;
  const int H = a.H, W = a.W;

  // ---- per-thread staging assignment (chunk-invariant)
  const int pc = tid & 3;  // 16-byte slot inside the 64-byte chunk (NTHR % 4 == 0: same for every piece)
  int ppix[NPL], plds[NPL];
  {
    const int NP = PH * PW * 4;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int q = tid + i * NTHR;
      plds[i] = -1;
      ppix[i] = -1;
      if (q < NP) {
        const int pix = q >> 2;
        const int py = pix / PW, px = pix - py * PW;
        const int gy = y0 + py - HALO, gx = x0 + px - HALO;
        plds[i] = py * ROWP + px * PIXB + pc * 16;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W)
          ppix[i] = a.unshuf ? ((b * 2 * H + 2 * gy) * 2 * W + 2 * gx) : ((b * H + gy) * W + gx);
      }
    }
  }
  int wsrc[NWL], wlds[NWL];
#pragma unroll
  for (int i = 0; i < NWL; ++i) {
    const int q = tid + i * NTHR;
    wsrc[i] = -1;
    wlds[i] = -1;
    if (q < NWP) {
      const int t = q / (BN * 4), r = q - t * (BN * 4);
      wsrc[i] = t * a.Ntot * 64 + r * 16;            // byte offset from the step's weight base
      wlds[i] = t * (BN * PIXB) + (r >> 2) * PIXB + (r & 3) * 16;
    }
  }

  const int nchA = a.CA / E::CH;
  const int nchunks = a.unshuf ? 4 * nchA : (a.CA + a.CB) / E::CH;
  const int nsteps = nchunks * SPC;
  const bool pro = (a.scale != nullptr);

  uint4 preg[NPL], wreg[2][NWL];
  float psc[E::VEC], psh[E::VEC];

  auto load_patch = [&](int kc) {
    const T* src;
    int C, coff, tapadd = 0;
    if (a.unshuf) {
      const int tap = kc / nchA, cc = kc - tap * nchA;
      src = (const T*)a.srcA; C = a.CA; coff = cc * E::CH;
      tapadd = (tap >> 1) * 2 * W + (tap & 1);
    } else if (kc < nchA) {
      src = (const T*)a.srcA; C = a.CA; coff = kc * E::CH;
    } else {
      src = (const T*)a.srcB; C = a.CB; coff = (kc - nchA) * E::CH;
    }
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);   // every element is always written: keeps preg[] in registers
      if (ppix[i] >= 0) v = *(const uint4*)(src + ((size_t)(ppix[i] + tapadd) * C + coff + pc * E::VEC));
      preg[i] = v;
    }
    if (pro) {
#pragma unroll
      for (int j = 0; j < E::VEC; ++j) {
        psc[j] = a.scale[coff + pc * E::VEC + j];
        psh[j] = a.shift[coff + pc * E::VEC + j];
      }
    }
  };
  auto store_patch = [&](char* pbuf) {
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      if (plds[i] >= 0) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ppix[i] >= 0) {
          v = preg[i];
          if (pro) {  // BatchNorm(scale, shift) + ReLU of the producer layer, applied on load
            float f[E::VEC];
            unpack16<T>(v, f);
#pragma unroll
            for (int j = 0; j < E::VEC; ++j) f[j] = fmaxf(fmaf(f[j], psc[j], psh[j]), 0.f);
            v = pack16<T>(f);
          }
        }
        *(uint4*)(pbuf + plds[i]) = v;
      }
    }
  };
  auto load_w = [&](int s, uint4 (&wr)[NWL]) {
    const int kc = s / SPC, tg = s - kc * SPC;
    const char* wb = (const char*)a.w + ((size_t)(kc * NTAPS + tg * TPS) * a.Ntot + n0) * 64;
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (wsrc[i] >= 0) v = *(const uint4*)(wb + wsrc[i]);
      wr[i] = v;
    }
  };
  auto store_w = [&](char* wbuf, const uint4 (&wr)[NWL]) {
#pragma unroll
    for (int i = 0; i < NWL; ++i)
      if (wlds[i] >= 0) *(uint4*)(wbuf + wlds[i]) = wr[i];
  };

  // ---- per-lane fragment addresses
  int laneA[MF], laneB[NF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    const int m = (wm * MF + mf) * 32 + lr;
    laneA[mf] = (m >> twl) * ROWP + (m & (tw - 1)) * PIXB + lh * 16;
  }
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) laneB[nf] = ((wn * NF + nf) * 32 + lr) * PIXB + lh * 16;

  f32x16 acc[MF][NF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f;

  char* const patch0 = smem;
  char* const wbuf0 = smem + PBUF * PB;

  // one tap (two MFMA k-steps over the 64-byte chunk) against LDS
  auto mma_tap = [&](const char* pb, const char* wb) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 fa[MF], fb[NF];
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) fa[mf] = *(const uint4*)(pb + laneA[mf] + kk * 32);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) fb[nf] = *(const uint4*)(wb + laneB[nf] + kk * 32);
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) Mma<T>::run(fa[mf], fb[nf], acc[mf][nf]);
    }
  };

  if constexpr (GEO == 0) {
    // Software pipeline, weights two steps ahead: at step s the registers of parity (s+1)&1 hold step s+1's
    // weight tile (loaded during step s-1) and are written to LDS BETWEEN this step's MFMA groups, while
    // the loads of step s+2 are issued into the other register set.  One barrier per step.
    load_patch(0);
    load_w(0, wreg[0]);
    store_patch(patch0);
    store_w(wbuf0, wreg[0]);
    if (nsteps > 1) load_w(1, wreg[1]);
    __syncthreads();

    auto do_step = [&](auto TGc, auto PARc, int kc) {
      constexpr int TG = decltype(TGc)::value, PAR = decltype(PARc)::value;
      const int s = kc * 3 + TG;
      const bool more_chunks = (kc + 1 < nchunks);
      if (s + 2 < nsteps) load_w(s + 2, wreg[PAR]);
      if (TG == 0 && more_chunks) load_patch(kc + 1);
      const char* pb = patch0 + (PBUF == 2 ? (kc & 1) * PB : 0) + TG * ROWP;
      const char* wb = wbuf0 + PAR * WB;
      mma_tap(pb, wb);
      if (s + 1 < nsteps) store_w(wbuf0 + (PAR ^ 1) * WB, wreg[PAR ^ 1]);
      mma_tap(pb + PIXB, wb + BN * PIXB);
      if (TG == 2 && PBUF == 2 && more_chunks) store_patch(patch0 + ((kc + 1) & 1) * PB);
      mma_tap(pb + 2 * PIXB, wb + 2 * BN * PIXB);
      __syncthreads();
      if (TG == 2 && PBUF == 1 && more_chunks) {
        store_patch(patch0);
        __syncthreads();
      }
    };
    int kc = 0;
    for (; kc + 1 < nchunks; kc += 2) {   // six steps: register/LDS parities are compile-time constants
      do_step(IC<0>{}, IC<0>{}, kc);
      do_step(IC<1>{}, IC<1>{}, kc);
      do_step(IC<2>{}, IC<0>{}, kc);
      do_step(IC<0>{}, IC<1>{}, kc + 1);
      do_step(IC<1>{}, IC<0>{}, kc + 1);
      do_step(IC<2>{}, IC<1>{}, kc + 1);
    }
    if (kc < nchunks) {
      do_step(IC<0>{}, IC<0>{}, kc);
      do_step(IC<1>{}, IC<1>{}, kc);
      do_step(IC<2>{}, IC<0>{}, kc);
    }
  } else {
    // 1x1 geometry: one tap per chunk; patch + weights staged one step ahead
    load_patch(0);
    load_w(0, wreg[0]);
    store_patch(patch0);
    store_w(wbuf0, wreg[0]);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
      const bool has_next = (s + 1 < nsteps);
      if (has_next) {
        load_w(s + 1, wreg[0]);
        load_patch(s + 1);
      }
      mma_tap(patch0 + (PBUF == 2 ? (s & 1) * PB : 0), wbuf0 + (s & 1) * WB);
      if (PBUF == 1) __syncthreads();
      if (has_next) {
        store_w(wbuf0 + ((s + 1) & 1) * WB, wreg[0]);
        store_patch(patch0 + (PBUF == 2 ? ((s + 1) & 1) * PB : 0));
      }
      __syncthreads();
    }
  }

  // ---- epilogue: bias, BN statistics from fp32 accumulators, LDS transpose, coalesced store
  constexpr int OP = BN * E::ES + 16;
  char* const ot = smem;
  float* const red = (float*)(smem + BM * OP);
  const bool do_stats = (a.stats != nullptr);
  float s1[NF], s2[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    s1[nf] = 0.f;
    s2[nf] = 0.f;
    const int n = (wn * NF + nf) * 32 + lr;
    const float bv = a.bias ? a.bias[n0 + n] : 0.f;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (wm * MF + mf) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float v = acc[mf][nf][r] + bv;
        const bool valid = (y0 + (m >> twl) < H) && (x0 + (m & (tw - 1)) < W);
        if (valid) {
          s1[nf] += v;
          s2[nf] += v * v;
        }
        *(T*)(ot + m * OP + n * E::ES) = from_float<T>(v);
      }
    }
  }
  if (do_stats) {
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      s1[nf] += __shfl_xor(s1[nf], 32);
      s2[nf] += __shfl_xor(s2[nf], 32);
      if (lh == 0) {
        const int n = (wn * NF + nf) * 32 + lr;
        red[(wm * BN + n) * 2 + 0] = s1[nf];
        red[(wm * BN + n) * 2 + 1] = s2[nf];
      }
    }
  }
  __syncthreads();
  if (do_stats && tid < BN) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < WM; ++w) {  // fixed order: bit-stable
      t1 += red[(w * BN + tid) * 2 + 0];
      t2 += red[(w * BN + tid) * 2 + 1];
    }
    float2* dst = (float2*)a.stats + (size_t)mt * a.Ntot + n0 + tid;
    *dst = make_float2(t1, t2);
  }
  constexpr int CPR = BN * E::ES / 16;
  for (int q = tid; q < BM * CPR; q += NTHR) {
    const int m = q / CPR, cc = q - m * CPR;
    const int ty = m >> twl, tx = m & (tw - 1);
    if (y0 + ty >= H || x0 + tx >= W) continue;
    const uint4 v = *(const uint4*)(ot + m * OP + cc * 16);
    const int n = n0 + cc * E::VEC;
    T* dst;
    if (a.shuffle) {  // ConvTranspose2d(k=2,s=2): N = (a*2+c)*Cout + co -> pixel (2y+a, 2x+c)
      const int tq = n0 / a.CO1, co = n - tq * a.CO1;
      const size_t opix = ((size_t)(b * 2 * H + 2 * (y0 + ty) + (tq >> 1))) * (2 * W) + 2 * (x0 + tx) + (tq & 1);
      dst = (T*)a.out + opix * a.CO1 + co;
    } else {
      const size_t pix = ((size_t)(b * H + y0 + ty)) * W + x0 + tx;
      if (n < a.CO1) dst = (T*)a.out + pix * a.CO1 + n;
      else dst = (T*)a.out2 + pix * a.CO2 + (n - a.CO1);
    }
    *(uint4*)dst = v;
  }
}

template <typename T, int GEO, int WM, int WN, int MF, int NF, int PBUF>
int launch_cfg(ConvArgs a, hipStream_t st) {
  using E = ET<T>;
  constexpr int BM = WM * MF * 32, BN = WN * NF * 32, NTHR = WM * WN * 64;
  constexpr int HALO = (GEO == 0) ? 1 : 0;
  constexpr int TPS = (GEO == 0) ? 3 : 1;
  a.twl = segk_conv_twl(BM, a.W);
  const int tw = 1 << a.twl, th = BM >> a.twl;
  a.tiles_x = cdiv(a.W, tw);
  a.tiles_y = cdiv(a.H, th);
  const int PW = tw + 2 * HALO, PH = th + 2 * HALO;
  const int ROWP = (PW * PIXB + 255) & ~255;
  const size_t main_b = PBUF * (size_t)PH * ROWP + 2 * (size_t)TPS * BN * PIXB;
  const size_t epi_b = BM * (size_t)(BN * E::ES + 16) + (size_t)WM * BN * 8;
  const size_t lds = main_b > epi_b ? main_b : epi_b;
  SEGK_REQUIRE(lds <= 160 * 1024, "conv_igemm: LDS %zu exceeds 160 KiB", lds);
  const int grid = a.B * a.tiles_x * a.tiles_y * (a.Ntot / BN);
  auto kern = conv_igemm_kernel<T, GEO, WM, WN, MF, NF, PBUF>;
  static bool attr_set = false;  // idempotent; racing setters write the same value
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      SEGK_FAIL(-3, "conv_igemm: cannot raise dynamic LDS limit");
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), lds, st, a);
  SEGK_CHECK_LAUNCH("conv_igemm");
  return 0;
}

template <typename T, int GEO>
int launch_geo(const ConvArgs& a, hipStream_t st) {
  // BN must divide N; with the pixel-shuffle store a channel tile must not straddle two taps.
  const int unit = a.shuffle ? a.CO1 : a.Ntot;
  if (unit % 128 == 0) return launch_cfg<T, GEO, 4, 2, 2, 2, 2>(a, st);            // 256 px x 128 ch, 8 waves
  if constexpr (GEO == 0) {
    if (unit % 64 == 0) return launch_cfg<T, GEO, 2, 2, 2, 1, 1>(a, st);           // 128 px x 64 ch, 4 waves
    return launch_cfg<T, GEO, 4, 1, 1, 1, 1>(a, st);                               // 128 px x 32 ch, 4 waves
  } else {
    if (unit % 64 == 0) return launch_cfg<T, GEO, 4, 2, 2, 1, 2>(a, st);
    return launch_cfg<T, GEO, 8, 1, 1, 1, 2>(a, st);
  }
}

}  // namespace

int segk_conv_igemm_launch(const ConvArgs& a, int geo, int dtype, hipStream_t st) {
  using F = ET<float>;
  using H = ET<bf16_t>;
  const int CH = (dtype == SEGK_DT_BF16) ? H::CH : F::CH;
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "conv_igemm: bad dtype %d", dtype);
  SEGK_REQUIRE(geo == 0 || geo == 1, "conv_igemm: bad geometry %d", geo);
  SEGK_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0, "conv_igemm: bad shape B=%d H=%d W=%d", a.B, a.H, a.W);
  SEGK_REQUIRE(a.srcA && a.w && a.out, "conv_igemm: null pointer");
  SEGK_REQUIRE(a.CA > 0 && a.CA % CH == 0, "conv_igemm: CA=%d must be a positive multiple of %d", a.CA, CH);
  SEGK_REQUIRE(a.CB >= 0 && a.CB % CH == 0 && (a.CB == 0) == (a.srcB == nullptr), "conv_igemm: bad second source");
  SEGK_REQUIRE(!(a.unshuf && (a.CB || geo != 1)), "conv_igemm: un-shuffle gather needs geo 1, single source");
  SEGK_REQUIRE(!(a.scale && a.CB), "conv_igemm: BN prologue with two sources is unsupported");
  SEGK_REQUIRE((a.scale == nullptr) == (a.shift == nullptr), "conv_igemm: scale/shift must come together");
  SEGK_REQUIRE(a.Ntot > 0 && a.Ntot % 32 == 0, "conv_igemm: N=%d must be a multiple of 32", a.Ntot);
  SEGK_REQUIRE(a.CO1 > 0 && a.CO1 % 32 == 0 && a.CO2 >= 0 && a.CO2 % 32 == 0, "conv_igemm: bad output channels");
  if (a.shuffle) SEGK_REQUIRE(a.Ntot == 4 * a.CO1 && !a.out2 && geo == 1, "conv_igemm: pixel-shuffle needs N=4*Cout");
  else SEGK_REQUIRE(a.Ntot == a.CO1 + a.CO2 && (a.CO2 == 0) == (a.out2 == nullptr), "conv_igemm: N != CO1+CO2");
  SEGK_REQUIRE((long long)a.B * a.H * a.W * 4 < 2147483647LL, "conv_igemm: pixel index overflows int32");
  if (dtype == SEGK_DT_BF16) return geo == 0 ? launch_geo<bf16_t, 0>(a, st) : launch_geo<bf16_t, 1>(a, st);
  return geo == 0 ? launch_geo<float, 0>(a, st) : launch_geo<float, 1>(a, st);
}
