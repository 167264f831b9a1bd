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
#include "common.hpp"
#include "segk_internal.h"

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

template <typename T, int GEO, int WM, int WN, int MF, int NF>
__global__ __launch_bounds__(512, 2) void conv_igemm_kernel(const ConvArgs a) {
  using E = ET<T>;
  constexpr int BN = WN * NF * 32;
  constexpr int NTAPS = (GEO == 0) ? 9 : 1;
  constexpr int TPS = (GEO == 0) ? 3 : 1;   // taps per pipeline step (one kernel row)
  constexpr int SPC = (GEO == 0) ? 3 : 1;   // steps per channel chunk
  constexpr int HALO = (GEO == 0) ? 1 : 0;
  constexpr int NPL = (GEO == 0) ? 3 : 2;   // patch 16-byte pieces per thread per chunk
  constexpr int NWP = TPS * BN * 4;         // weight 16-byte pieces per step
  constexpr int NWL = (NWP + 511) / 512;
  static_assert(WM * WN == 8 && WM * MF * 32 == 256, "tile is 256 pixels x BN, 8 waves");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int lr = lane & 31, lh = lane >> 5;

  const int twl = a.twl, tw = 1 << twl, th = 256 >> twl;
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
  const int pc = tid & 3;  // 16-byte slot inside the 64-byte chunk (512 % 4 == 0: same for every piece)
  int ppix[NPL], plds[NPL];
  {
    const int NP = PH * PW * 4;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int q = tid + i * 512;
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
    const int q = tid + i * 512;
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

  uint4 preg[NPL], wreg[NWL];
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
  auto load_w = [&](int s) {
    const int kc = s / SPC, tg = s - kc * SPC;
    const char* wb = (const char*)a.w + ((size_t)(kc * NTAPS + tg * TPS) * a.Ntot + n0) * 64;
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (wsrc[i] >= 0) v = *(const uint4*)(wb + wsrc[i]);
      wreg[i] = v;
    }
  };
  auto store_w = [&](char* wbuf) {
#pragma unroll
    for (int i = 0; i < NWL; ++i)
      if (wlds[i] >= 0) *(uint4*)(wbuf + wlds[i]) = wreg[i];
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
  char* const wbuf0 = smem + 2 * PB;

  // ---- prologue: stage chunk 0 / step 0
  load_patch(0);
  load_w(0);
  store_patch(patch0);
  store_w(wbuf0);
  __syncthreads();

  for (int s = 0; s < nsteps; ++s) {
    const int kc = s / SPC, tg = s - kc * SPC;
    const bool has_next = (s + 1 < nsteps);
    const bool next_chunk = has_next && (tg == SPC - 1);
    if (has_next) load_w(s + 1);          // global loads fly under the MFMAs below
    if (next_chunk) load_patch(kc + 1);

    const char* pb = patch0 + (kc & 1) * PB + tg * ROWP * (GEO == 0 ? 1 : 0);
    const char* wb = wbuf0 + (s & 1) * WB;
#pragma unroll
    for (int t = 0; t < TPS; ++t) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        uint4 fa[MF], fb[NF];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) fa[mf] = *(const uint4*)(pb + laneA[mf] + t * PIXB + kk * 32);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) fb[nf] = *(const uint4*)(wb + t * (BN * PIXB) + laneB[nf] + kk * 32);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) Mma<T>::run(fa[mf], fb[nf], acc[mf][nf]);
      }
    }
    if (has_next) store_w(wbuf0 + ((s + 1) & 1) * WB);
    if (next_chunk) store_patch(patch0 + ((kc + 1) & 1) * PB);
    __syncthreads();
  }

  // ---- epilogue: bias, BN statistics from fp32 accumulators, LDS transpose, coalesced store
  constexpr int OP = BN * E::ES + 16;
  char* const ot = smem;
  float* const red = (float*)(smem + 256 * OP);
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
  for (int q = tid; q < 256 * CPR; q += 512) {
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

template <typename T, int GEO, int WM, int WN, int MF, int NF>
int launch_cfg(const ConvArgs& a, hipStream_t st) {
  using E = ET<T>;
  constexpr int BN = WN * NF * 32;
  constexpr int HALO = (GEO == 0) ? 1 : 0;
  constexpr int TPS = (GEO == 0) ? 3 : 1;
  const int tw = 1 << a.twl, th = 256 >> a.twl;
  const int PW = tw + 2 * HALO, PH = th + 2 * HALO;
  const int ROWP = (PW * PIXB + 255) & ~255;
  const size_t main_b = 2 * (size_t)PH * ROWP + 2 * (size_t)TPS * BN * PIXB;
  const size_t epi_b = 256 * (size_t)(BN * E::ES + 16) + (size_t)WM * BN * 8;
  const size_t lds = main_b > epi_b ? main_b : epi_b;
  SEGK_REQUIRE(lds <= 160 * 1024, "conv_igemm: LDS %zu exceeds 160 KiB", lds);
  const int grid = a.B * a.tiles_x * a.tiles_y * (a.Ntot / BN);
  auto kern = conv_igemm_kernel<T, GEO, WM, WN, MF, NF>;
  static bool attr_set = false;  // idempotent; racing setters write the same value
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      SEGK_FAIL(-3, "conv_igemm: cannot raise dynamic LDS limit");
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
  SEGK_CHECK_LAUNCH("conv_igemm");
  return 0;
}

template <typename T, int GEO>
int launch_geo(const ConvArgs& a, hipStream_t st) {
  // BN must divide N; with the pixel-shuffle store a channel tile must not straddle two taps.
  const int unit = a.shuffle ? a.CO1 : a.Ntot;
  if (unit % 128 == 0) return launch_cfg<T, GEO, 4, 2, 2, 2>(a, st);
  if (unit % 64 == 0) return launch_cfg<T, GEO, 4, 2, 2, 1>(a, st);
  return launch_cfg<T, GEO, 8, 1, 1, 1>(a, st);
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
  SEGK_REQUIRE(a.twl == 4 || a.twl == 5, "conv_igemm: tile width must be 16 or 32");
  SEGK_REQUIRE(a.tiles_x == cdiv(a.W, 1 << a.twl) && a.tiles_y == cdiv(a.H, 256 >> a.twl), "conv_igemm: tile grid mismatch");
  SEGK_REQUIRE((long long)a.B * a.H * a.W * 4 < 2147483647LL, "conv_igemm: pixel index overflows int32");
  if (dtype == SEGK_DT_BF16) return geo == 0 ? launch_geo<bf16_t, 0>(a, st) : launch_geo<bf16_t, 1>(a, st);
  return geo == 0 ? launch_geo<float, 0>(a, st) : launch_geo<float, 1>(a, st);
}
