// Implicit-GEMM convolution for NHWC activations on CDNA4 matrix cores (gfx950).
//
// One kernel family serves, by geometry/flags:
//   GEO 0 : Conv2d 3x3 pad 1 stride 1   -- forward (reference unet/unet.py:16,19; clip/clipunet.py:87,90)
//                                          and data-gradient (same kernel, flipped+transposed weights)
//   GEO 1 : 1x1 "conv"                  -- ConvTranspose2d(k=2,s=2) forward as GEMM + pixel-shuffle store
//                                          (unet.py:59, clipunet.py:83), its data-gradient as a 2x2
//                                          un-shuffle gather GEMM, and the CLIP 1x1 projections (clipunet.py:84,122)
//
// GEMM view: M = BM output pixels of one spatial tile (TH x TW), N = BN output channels, K = taps x Cin.
// The input patch (tile + halo) of one 64-byte channel chunk is staged ONCE into LDS and reused by all 9
// taps as shifted row addresses (the 3x3 im2col never exists); weight tiles stream through a second,
// double-buffered LDS region.  Staging is register-staged (global -> VGPR -> LDS, issue-early /
// write-late) so the previous layer's BatchNorm+ReLU can be applied on the fly (prologue fusion) and so
// out-of-image halo pixels become exact zeros.  MFMA: bf16 32x32x16 or exact fp32 32x32x2 with a
// byte-identical LDS image (a 16-byte fragment read is 8 bf16 k-values or 4 fp32).
//
// The kernel is PERSISTENT over work units (pixel tile x channel tile): every XCD owns a contiguous range
// of units (neighbouring tiles share halos and weights through that XCD's L2), each workgroup walks its
// XCD's range with a fixed stride, and the loads of the next unit's first patch chunk / weight step are
// issued under the current unit's last MFMA steps, so the HBM latency of a tile start hides behind the
// previous tile's tail and the per-thread addressing set-up is paid once per workgroup, not per tile.
//
// Epilogue: optional bias, per-channel sum / sum-of-squares partials for training-mode BatchNorm taken
// from the fp32 accumulators (deterministic per-tile partials, no atomics), LDS transpose, 16-byte
// coalesced NHWC stores (optionally split over two destinations, or pixel-shuffled for ConvTranspose).
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"
#include "segk_internal.h"

// pixel-tile geometry shared by the launcher and the BN-statistics sizing query (api.hip)
// weight-stationary kernel: at most two 64-byte input chunks and N a multiple of 64 (but not a 128-wide layer
// with a long K, which the MFMA-bound streaming kernel serves better)
int segk_conv_use_ws(int cin_p, int n_p, int dtype) {
  return dtype == SEGK_DT_BF16 && cin_p <= 64 && n_p % 64 == 0;   // bf16 performance mode only
}
// producer/consumer kernel (bf16 3x3, at least two 64-byte input chunks, not a weight-stationary layer): returns
// its channel tile, 128 (256-pixel tiles) or 64 (512-pixel tiles, Cin >= 128: with K = 576 the unit boundary
// dominates and the weight-stationary kernel wins), or 0
int segk_conv_use_pipe(int cin_p, int n_p, int dtype) {
  if (dtype != SEGK_DT_BF16 || segk_conv_use_ws(cin_p, n_p, dtype) || cin_p < 64) return 0;
  if (n_p % 128 == 0) return 128;
  if (n_p % 64 == 0 && cin_p >= 128) return 64;
  return 0;
}
int segk_conv_writes_act(int cin_p, int n_p, int dtype) {   // (conv_rs layers are a subset of these shapes)
  return segk_conv_use_ws(cin_p, n_p, dtype) || segk_conv_use_pipe(cin_p, n_p, dtype) != 0;
}
int segk_conv_bm(int geo, int unit) { return (unit % 128 == 0 || geo != 0) ? 256 : 128; }
int segk_conv_twl(int bm, int W) { return bm == 128 ? 4 : (W > 16 ? 5 : 4); }   // 8x16 | 8x32 | 16x16 tiles

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in registers (SSA)

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

static int num_cus() { return segk_num_cus(); }

// Tile: BM = WM*MF*32 output pixels (TH x TW, TW = 1 << TWL) by BN = WN*NF*32 output channels, WM*WN waves.
// PBUF: patch buffers (2 = next chunk staged under the current chunk's MFMAs; 1 = smaller LDS footprint so
// that two or three 4-wave workgroups share a CU and overlap each other's load / MFMA / store phases).
template <typename T, int GEO, int TWL, int WM, int WN, int MF, int NF, int PBUF, bool PRO>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_igemm_kernel(const ConvArgs a) {
  using E = ET<T>;
  constexpr int NW = WM * WN, NTHR = NW * 64;
  constexpr int BM = WM * MF * 32, BN = WN * NF * 32;
  constexpr int NTAPS = (GEO == 0) ? 9 : 1;
  constexpr int TPS = (GEO == 0) ? 3 : 1;   // taps per pipeline step (one kernel row)
  constexpr int SPC = (GEO == 0) ? 3 : 1;   // steps per channel chunk
  constexpr int HALO = (GEO == 0) ? 1 : 0;
  constexpr int TW = 1 << TWL, TH = BM >> TWL;
  constexpr int PW = TW + 2 * HALO, PH = TH + 2 * HALO;
  constexpr int ROWP = (PW * PIXB + 255) & ~255;   // multiple of 256 B: two-row fragments stay conflict-free
  constexpr int PB = PH * ROWP;
  constexpr int WB = TPS * BN * PIXB;
  constexpr int NP = PH * PW * 4;                  // patch 16-byte pieces per chunk
  constexpr int NPL = (NP + NTHR - 1) / NTHR;
  constexpr int NWP = TPS * BN * 4;                // weight 16-byte pieces per step
  constexpr int NWL = (NWP + NTHR - 1) / NTHR;
  static_assert(BM == 256 || BM == 128, "pixel tile is 256 or 128");
  static_assert(NTHR % 4 == 0, "a thread keeps one 16-byte slot of the 64-byte chunk");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  const int lr = lane & 31, lh = lane >> 5;
  const int H = a.H, W = a.W;

  // ---- work units
  const int NT = a.Ntot / BN;
  const int tpi = a.tiles_x * a.tiles_y;
  const int U = a.B * tpi * NT;
  int u, u_end, GW;
  if (a.persistent) {
    const int xcd = blockIdx.x & 7, upx = (U + 7) >> 3;
    GW = gridDim.x >> 3;
    u = xcd * upx + (blockIdx.x >> 3);
    u_end = min(U, (xcd + 1) * upx);
  } else {
    u = xcd_remap(blockIdx.x, gridDim.x);
    u_end = u + 1;
    GW = 1;
  }
  if (u >= u_end) return;

  // ---- unit-invariant staging assignment.  Straight-line staging: every thread always moves NPL patch
  // pieces and NWL weight pieces; pieces past the tile's count read a valid dummy address and land in a
  // per-thread trash slot behind the staging buffers, so there is no divergent control flow around
  // loads or LDS stores.
  constexpr int MAINB = PBUF * PB + 2 * WB;
  const int trash = MAINB + tid * 16;
  const int pc = tid & 3;  // 16-byte slot inside the 64-byte chunk (NTHR % 4 == 0: same for every piece)
  int plds[NPL], prel[NPL];   // LDS byte offset and packed patch-relative coordinates (py << 8 | px)
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int q = tid + i * NTHR;
    const int pix = q >> 2;
    const int py = pix / PW, px = pix - py * PW;
    plds[i] = (q < NP) ? py * ROWP + px * PIXB + pc * 16 : trash;
    prel[i] = (q < NP) ? ((py << 8) | px) : (0x7fff << 8);  // row 32767: outside every image, never valid
  }
  int wsrc[NWL], wlds[NWL];
#pragma unroll
  for (int i = 0; i < NWL; ++i) {
    const int q = tid + i * NTHR;
    const int t = q / (BN * 4), r = q - t * (BN * 4);
    wsrc[i] = (q < NWP) ? t * a.Ntot * 64 + r * 16 : 0;    // byte offset from the step's weight base
    wlds[i] = (q < NWP) ? PBUF * PB + t * (BN * PIXB) + (r >> 2) * PIXB + (r & 3) * 16 : trash;
  }
  const int nchA = a.CA / E::CH;
  const int nchunks = a.unshuf ? 4 * nchA : (a.CA + a.CB) / E::CH;
  const int nsteps = nchunks * SPC;

  // per-lane fragment addresses
  int laneA[MF], laneB[NF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    const int m = (wm * MF + mf) * 32 + lr;
    laneA[mf] = (m >> TWL) * ROWP + (m & (TW - 1)) * PIXB + lh * 16;
  }
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) laneB[nf] = ((wn * NF + nf) * 32 + lr) * PIXB + lh * 16;

  char* const patch0 = smem;
  char* const wbuf0 = smem + PBUF * PB;

  // ---- unit state
  int ub, uy0, ux0, un0, umt;       // current unit
  auto decode = [&](int uu, int& mt, int& b, int& y0, int& x0, int& n0) {
    mt = uu / NT;
    n0 = (uu - mt * NT) * BN;
    b = mt / tpi;
    const int trem = mt - b * tpi;
    const int tyi = trem / a.tiles_x;
    y0 = tyi * TH;
    x0 = (trem - tyi * a.tiles_x) * TW;
  };
  u32x4 preg[NPL], wreg[NWL];
  unsigned pvalid = 0;              // bit i: preg[i] holds an in-image pixel
  float psc[E::VEC], psh[E::VEC];

  auto load_patch = [&](int kc, int b, int y0, int x0) {
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
    const T* const base = src + coff + pc * E::VEC;
    pvalid = 0;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int gy = y0 + (prel[i] >> 8) - HALO, gx = x0 + (prel[i] & 255) - HALO;
      const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
      const int cy = ok ? gy : y0, cx = ok ? gx : x0;      // clamp to the tile origin: always a valid pixel
      const int lin = a.unshuf ? ((b * 2 * H + 2 * cy) * 2 * W + 2 * cx) : ((b * H + cy) * W + cx);
      preg[i] = *(const u32x4*)(base + (size_t)(lin + tapadd) * C);
      pvalid |= (ok ? 1u : 0u) << i;
    }
    if (PRO) {
#pragma unroll
      for (int j = 0; j < E::VEC; ++j) {
        psc[j] = a.scale[coff + pc * E::VEC + j];
        psh[j] = a.shift[coff + pc * E::VEC + j];
      }
    }
  };
  auto store_patch = [&](int pboff) {
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      u32x4 v = preg[i];
      if (PRO) {  // BatchNorm(scale, shift) + ReLU of the producer layer, applied on load
        float f[E::VEC];
        unpack16<T>(make_uint4(v.x, v.y, v.z, v.w), f);
#pragma unroll
        for (int j = 0; j < E::VEC; ++j) f[j] = fmaxf(fmaf(f[j], psc[j], psh[j]), 0.f);
        const uint4 t = pack16<T>(f);
        v = (u32x4){t.x, t.y, t.z, t.w};
      }
      const bool ok = (pvalid >> i) & 1;   // out-of-image halo pixels are exact zeros (after the transform)
      v = ok ? v : (u32x4){0u, 0u, 0u, 0u};
      const int off = plds[i] + ((plds[i] < MAINB) ? pboff : 0);
      *(u32x4*)(smem + off) = v;
    }
  };
  auto load_w = [&](int s, int n0) {
    const int kc = s / SPC, tg = s - kc * SPC;
    const char* wb = (const char*)a.w + ((size_t)(kc * NTAPS + tg * TPS) * a.Ntot + n0) * 64;
#pragma unroll
    for (int i = 0; i < NWL; ++i) wreg[i] = *(const u32x4*)(wb + wsrc[i]);
  };
  auto store_w = [&](int par) {
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int off = wlds[i] + ((wlds[i] < MAINB) ? par * WB : 0);
      *(u32x4*)(smem + off) = wreg[i];
    }
  };

  f32x16 acc[MF][NF];
  auto zero_acc = [&]() {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f;
  };

  // one pipeline step against LDS: TPS taps x two MFMA k-steps, with the fragment reads of sub-step i+1
  // issued BEFORE the MFMAs of sub-step i (register double buffer) so LDS latency hides under the matrix pipe
  auto mma_step = [&](const char* pb, const char* wb) {
    constexpr int NSUB = TPS * 2;
    uint4 fa[2][MF], fb[2][NF];
    auto rd = [&](int i, uint4 (&A)[MF], uint4 (&Bf)[NF]) {
      const int t = i >> 1, kk = i & 1;
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) A[mf] = *(const uint4*)(pb + laneA[mf] + t * PIXB + kk * 32);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) Bf[nf] = *(const uint4*)(wb + laneB[nf] + t * (BN * PIXB) + kk * 32);
    };
    rd(0, fa[0], fb[0]);
#pragma unroll
    for (int i = 0; i < NSUB; ++i) {
      if (i + 1 < NSUB) rd(i + 1, fa[(i + 1) & 1], fb[(i + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch reads ahead of this sub-step's MFMAs
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) Mma<T>::run(fa[i & 1][mf], fb[i & 1][nf], acc[mf][nf]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- epilogue (entered right after a barrier: nobody reads the staging LDS any more).
  // FULL = the whole tile lies inside the image (the common case): no per-element bounds tests.
  constexpr int OP = BN * E::ES + 16;
  auto epilogue_t = [&](auto FULLc) {
    constexpr bool FULL = decltype(FULLc)::value;
    char* const ot = smem;
    float* const red = (float*)(smem + BM * OP);
    const bool do_stats = (a.stats != nullptr);
    float s1[NF], s2[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      s1[nf] = 0.f;
      s2[nf] = 0.f;
      const int n = (wn * NF + nf) * 32 + lr;
      const float bv = a.bias ? a.bias[un0 + n] : 0.f;
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int mb = (wm * MF + mf) * 32 + 4 * lh;
        char* const obase = ot + mb * OP + n * E::ES;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          v[r] = acc[mf][nf][r] + bv;
          if constexpr (GEO == 1) {   // CLIP MLP: quick_gelu(x) = x * sigmoid(1.702 x) fused behind fc1 (segk_linear)
            if (a.act) v[r] = v[r] / (1.f + __expf(-1.702f * v[r]));
          }
          if (!FULL) {   // pixels past the image edge are never stored and must not enter the statistics
            const int m = mb + (r & 3) + 8 * (r >> 2);
            v[r] = ((uy0 + (m >> TWL) < H) && (ux0 + (m & (TW - 1)) < W)) ? v[r] : 0.f;
          }
        }
        stage_frag<T>(v, obase, OP, s1[nf], s2[nf]);
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
      float2* dst = (float2*)a.stats + (size_t)umt * a.Ntot + un0 + tid;
      *dst = make_float2(t1, t2);
    }
    constexpr int CPR = BN * E::ES / 16;   // 16-byte chunks per pixel row of the tile
    constexpr int NST = BM * CPR / NTHR;
    static_assert(BM * CPR % NTHR == 0 && (CPR & (CPR - 1)) == 0, "store loop is exact");
    const int cc = tid & (CPR - 1);
    const int n = un0 + cc * E::VEC;
    // destination of this thread's channel slice at the tile origin, and the per-pixel element stride
    T* dbase;
    size_t pstride, rstride;       // elements per pixel step in x / per row step in y
    if (a.shuffle) {               // ConvTranspose2d(k=2,s=2): N = (a*2+c)*Cout + co -> pixel (2y+a, 2x+c)
      const int tq = n / a.CO1, co = n - tq * a.CO1;   // per thread: a channel tile may span several taps
      dbase = (T*)a.out + (((size_t)(ub * 2 * H + 2 * uy0 + (tq >> 1))) * (2 * W) + 2 * ux0 + (tq & 1)) * a.CO1 + co;
      pstride = 2 * (size_t)a.CO1;
      rstride = 4 * (size_t)W * a.CO1;
    } else {
      const size_t pix0 = ((size_t)(ub * H + uy0)) * W + ux0;
      if (n < a.CO1) { dbase = (T*)a.out + pix0 * a.CO1 + n; pstride = a.CO1; }
      else { dbase = (T*)a.out2 + pix0 * a.CO2 + (n - a.CO1); pstride = a.CO2; }
      rstride = (size_t)W * pstride;
    }
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int m = (tid + i * NTHR) / CPR;
      const int ty = m >> TWL, tx = m & (TW - 1);
      const uint4 v = *(const uint4*)(ot + m * OP + cc * 16);
      if (FULL || (uy0 + ty < H && ux0 + tx < W)) *(uint4*)(dbase + ty * rstride + tx * pstride) = v;
    }
  };
  auto epilogue = [&]() {
    if ((uy0 + TH <= H) && (ux0 + TW <= W)) epilogue_t(std::true_type{});
    else epilogue_t(std::false_type{});
  };

  // ---- first unit: stage chunk 0 / step 0
  decode(u, umt, ub, uy0, ux0, un0);
  zero_acc();
  load_patch(0, ub, uy0, ux0);
  load_w(0, un0);
  store_patch(0);
  store_w(0);
  __syncthreads();

  for (;;) {
    const int un = u + GW;
    const bool has_next = un < u_end;
    int nmt = 0, nb = 0, ny0 = 0, nx0 = 0, nn0 = 0;
    if (has_next) decode(un, nmt, nb, ny0, nx0, nn0);

    int kc = 0, tg = 0;
    for (int s = 0; s < nsteps; ++s) {
      const bool last_step = (s + 1 == nsteps);
      const bool more_chunks = (kc + 1 < nchunks);
      // issue-early: next weight step (wrapping into the next unit) and, at a chunk's first step, the next
      // patch chunk (the next unit's chunk 0 during the last chunk)
      if (!last_step) load_w(s + 1, un0);
      else if (has_next) load_w(0, nn0);
      if (tg == 0) {
        if (more_chunks) load_patch(kc + 1, ub, uy0, ux0);
        else if (has_next) load_patch(0, nb, ny0, nx0);
      }
      const char* pb = patch0 + (PBUF == 2 ? (kc & 1) * PB : 0) + (GEO == 0 ? tg * ROWP : 0);
      const char* wb = wbuf0 + (s & 1) * WB;
      mma_step(pb, wb);
      // write-late: the data issued above (or a step / chunk earlier) lands in the other LDS buffers
      if (!last_step) store_w((s + 1) & 1);
      if (tg == SPC - 1 && more_chunks && PBUF == 2) store_patch(((kc + 1) & 1) * PB);
      __syncthreads();
      if (tg == SPC - 1 && more_chunks && PBUF == 1) {
        store_patch(0);
        __syncthreads();
      }
      if (++tg == SPC) { tg = 0; ++kc; }
    }
    epilogue();
    if (!has_next) break;
    __syncthreads();                 // the epilogue's LDS reads are done: staging LDS may be rewritten
    store_w(0);                      // next unit's step 0 (buffer parity restarts at 0)
    store_patch(0);                  // next unit's chunk 0
    zero_acc();
    u = un; umt = nmt; ub = nb; uy0 = ny0; ux0 = nx0; un0 = nn0;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Producer / consumer 3x3 kernel for the MFMA-bound bf16 layers (channel tiles of 128, K = 9*Cin long).
//
// Why it exists (s_memtime traces and ablation builds of the symmetric 8-wave kernels at 512->512 @ 32x32,
// B = 32): with two MFMA waves per SIMD the matrix pipe arbitrates oldest-first, so the two waves of a SIMD
// run one after the other, and each is an in-order stream in which every staging instruction (global load,
// ds_write_b128, address arithmetic) behind MFMA k delays MFMA k+1 -- ~2070-2900 cycles per step for 1536
// cycles of matrix work, whatever the placement of the staging instructions inside the stream.
//
// Structure here: the 8 waves of a workgroup split into
//   * 4 CONSUMER waves (one per SIMD, raised priority), each owning a 128-pixel x 64-channel tile of the
//     256 x 128 workgroup tile: their stream is nothing but MFMAs and ds_read_b128 fragment reads (6 reads
//     per 8 MFMAs, one sub-step ahead; the first fragments of the next step are read BEFORE the barrier
//     that ends the current one);
//   * 4 PRODUCER waves (the SIMDs' second waves) that do all staging: weight steps through a ring of three
//     LDS buffers, stored two steps ahead of their use from registers fetched two steps before that, and
//     the next chunk's patch (BatchNorm+ReLU prologue and halo zero-fill applied on the way), stored one
//     step ahead.  Their waits (vmcnt, LDS-store queueing) no longer sit in any MFMA stream.
// One s_barrier per step joins the two roles.  The weight ring keeps running across work units (the epilogue
// tile overlays only [P0 | P1 | W2], dead at that point).  LDS: [W0 | W1 | P0 | P1 | W2], 80-byte pitch.
// M16: the consumers multiply with v_mfma_f32_16x16x32_bf16 instead of 32x32x16 (the same LDS bytes and matrix cycles per
// FLOP; on random data the chip holds a ~12 % higher clock on the 16x16 shape: tools/ubench/mfma_tiles.hip 1.79 vs
// 1.60 PFLOP/s at this wave tile).  A 16x16x32 operand is 16 rows x one whole 64-byte chunk, so the LDS image changes:
// patch pixels at a 96-byte pitch (conflict-free for 16 consecutive pixels at any tap shift), weight rows unpadded with
// the 16-byte piece index XOR-ed by [0,3,2,1][(row >> 2) & 3] (conflict-free, rows never shift).
// Diagnostic build only (-DSEGK_PIPE_STAMPS, tools/stamp_build.sh): per-wave cycle sums of the loop's phases, written
// over the statistics buffer by lane 0 of every wave; the shipped library contains no stamp.
#ifdef SEGK_PIPE_STAMPS
#define PIPE_STAMP(i)                                                                     \
  do {                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    unsigned long long t_;                                                                \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
    stamp_sum[i] += t_ - stamp_last;                                                      \
    stamp_last = t_;                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  } while (0)
#define PIPE_STAMP_OUT()                                                                  \
  do {                                                                                    \
    if (a.stats != nullptr && lane == 0) {                                                \
      unsigned long long* o_ = (unsigned long long*)a.stats + ((size_t)blockIdx.x * 8 + wave) * 8; \
      for (int i_ = 0; i_ < 6; ++i_) o_[i_] = stamp_sum[i_];                              \
      o_[6] = stamp_last - stamp_t0;                                                      \
    }                                                                                     \
  } while (0)
#else
#define PIPE_STAMP(i) do {} while (0)
#define PIPE_STAMP_OUT() do {} while (0)
#endif
#ifndef PIPE_ILV
#define PIPE_ILV 1     // 16x16x32 consumers read the next sub-step's fragments between the MFMAs (0: in a group ahead of them, rounds 2-3)
#endif
#ifndef PIPE_RAWBAR
#define PIPE_RAWBAR 0  // 1: the 16x16x32 consumers' step barriers as a raw s_barrier (no lgkmcnt(0) drain): 12.31 vs 12.29-12.30 ms, no gain
#endif
__device__ __forceinline__ void step_barrier() {
#if PIPE_RAWBAR
  __builtin_amdgcn_s_barrier();
#else
  __syncthreads();
#endif
}
#ifndef PIPE_ABL
#define PIPE_ABL 0     // diagnostic ablations (results are wrong): 1 producers skip LDS stores, 2 skip global loads, 4 consumers re-use fragments,
                       // 8 DMA form: no output stores, 16 DMA form: no statistics reduction, 32 DMA form: no epilogue at all
#endif

// DMA (round 4; 16x16x32 consumers, no prologue): the producers move NOTHING through registers.  Weights and patch arrive by
// LDS-DMA (global_load_lds, 1 KiB per wave-instruction, out-of-image halo pixels from a zero page), so a staged KiB costs the
// SIMD one issue instead of a global load plus a ds_write_b128 (tools/ubench/pipe_roles.hip: +38 instead of +154 cycles per
// 1536-cycle step beside the consumer), and the consumers' matrix stream is the only LDS writer-free critical path.  A DMA
// instruction writes 64 consecutive 16-byte slots, so both images are unpadded and conflict freedom comes from WHICH piece a
// lane fetches: weight rows keep the XOR image above (it never had padding); the patch becomes rows of 8-pixel blocks of
// 512 B with piece q of pixel x at (x >> 3) * 512 + (q >> 1) * 256 + ((6 x + q) & 15) * 16 -- the slot map of conv_rs.hip,
// conflict-free for a 16-pixel operand read at ANY x shift; a 16-pixel block starts at x = 0 or 16, so the consumers need
// one per-lane address per tap column (three registers instead of eight) plus immediates.
// The DMA form also has NO EPILOGUE TILE and no boundary barriers: its consumers multiply channels x pixels (A = weights, B =
// patch; the weight rows a lane reads are permuted so that accumulator rows 4 lq + j of a pair of 16-channel blocks are 8
// consecutive channels of one pixel) and store 16 bytes per pixel and block pair straight from the accumulators; BatchNorm
// sums are reduced over the 16 pixel lanes with DPP adds and written as one partial row per (unit, consumer pixel half).
// Producers and consumers then run through unit boundaries like through any other step: the next unit's first patch chunk
// and weight steps land under this unit's last steps, and the 128 two-byte LDS stores + three barriers + 512-thread store
// pass of the staged epilogue (8 k cycles per unit: 30 % of a Cin = 128 unit) become ~100 vector instructions of the
// consumers alone.
__device__ __attribute__((aligned(256))) unsigned char g_pipe_zero_page[256];
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;
template <int CTRL> __device__ __forceinline__ float pipe_dpp_add(float v) {
  const int x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
  return v + __int_as_float(x);
}
// sum over the 16 lanes of a DPP row, result in every lane: xor 1, xor 2 (quad permutes), half mirror, mirror -- a fixed tree
__device__ __forceinline__ float pipe_row16_sum(float v) {
  v = pipe_dpp_add<0xB1>(v);
  v = pipe_dpp_add<0x4E>(v);
  v = pipe_dpp_add<0x141>(v);
  v = pipe_dpp_add<0x140>(v);
  return v;
}

template <int TWL, bool PRO, int BN, bool M16, bool DMA = false>
__global__ __launch_bounds__(512, 2) void conv3x3_pipe_kernel(const ConvArgs a) {
  using T = bf16_t;
  using E = ET<T>;
  static_assert(!DMA || (M16 && !PRO), "the LDS-DMA form serves the 16x16x32 consumers without a prologue");
  constexpr int PPIX = DMA ? 64 : (M16 ? 96 : PIXB);   // LDS pitch of a patch pixel's chunk
  constexpr int WPIX = M16 ? 64 : PIXB;            // LDS pitch of a weight row
  // workgroup tile 256 px x 128 ch (consumers 2 x 2) or, for 64-channel layers, 512 px x 64 ch (consumers 4 x 1):
  // the same bytes per step and the same 128 x 64 consumer tile either way
  static_assert(BN == 128 || BN == 64, "channel tile is 128 or 64");
  constexpr int BM = 32768 / BN, NTHR = 512, NPT = 256;       // NPT: producer threads
  constexpr int WN = BN / 64, WM = 4 / WN, MF = 4, NF = 2;    // consumer waves: WM x WN, 128 px x 64 ch each
  constexpr int TW = 1 << TWL, TH = BM >> TWL, PW = TW + 2, PH = TH + 2;
  constexpr int RPX = (PW + 7) & ~7;               // DMA image: pixel slots per patch row (whole 8-pixel blocks)
  constexpr int ROWP = DMA ? RPX * 64 : ((PW * PPIX + 255) & ~255);
  constexpr int PB = PH * ROWP;                    // one patch chunk
  constexpr int WB = 3 * BN * WPIX;                // one step of weights: 3 taps x 128 rows
  constexpr int NP = PH * PW * 4, NPL = (NP + NPT - 1) / NPT;
  constexpr int NWL = 3 * BN * 4 / NPT;            // 16-byte weight pieces per producer thread and step
  constexpr int POFF = 2 * WB, W2OFF = POFF + 2 * PB, MAINB = W2OFF + WB;
  constexpr int NSUB = 6;                          // 3 taps x two k-halves of the 64-byte chunk
  static_assert(3 * BN * 4 % NPT == 0, "weight pieces divide evenly");
  auto wring = [](int r) { return r < 2 ? r * WB : W2OFF; };

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W;

  // ---- work units: every XCD owns a contiguous range, walked by its workgroups with stride GW
  const int NT = a.Ntot / BN;
  const int tpi = a.tiles_x * a.tiles_y;
  const int U = a.B * tpi * NT;
  const int xcd = blockIdx.x & 7, upx = (U + 7) >> 3;
  const int GW = gridDim.x >> 3;
  int u = xcd * upx + (blockIdx.x >> 3);
  const int u_end = min(U, (xcd + 1) * upx);
  if (u >= u_end) return;
  const int nchA = a.CA / E::CH;
  const int nchunks = (a.CA + a.CB) / E::CH;       // >= 2 (checked by the launcher)
  const int nsteps = nchunks * 3;
  auto decode = [&](int uu, int& mt, int& b, int& y0, int& x0, int& n0) {
    mt = uu / NT;
    n0 = (uu - mt * NT) * BN;
    b = mt / tpi;
    const int trem = mt - b * tpi;
    const int tyi = trem / a.tiles_x;
    y0 = tyi * TH;
    x0 = (trem - tyi * a.tiles_x) * TW;
  };
  int ub, uy0, ux0, un0, umt;
  decode(u, umt, ub, uy0, ux0, un0);
#ifdef SEGK_PIPE_STAMPS
  unsigned long long stamp_sum[6] = {0, 0, 0, 0, 0, 0}, stamp_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
  const unsigned long long stamp_t0 = stamp_last;
#endif

  // ---- the part of the epilogue all 512 threads run: BN-statistics reduction over the consumer rows and
  // the 16-byte coalesced NHWC stores of the staged tile
  constexpr int OP = BN * E::ES + 16;
  constexpr int LDSEND = MAINB;                    // DMA: one trash KiB behind the ring (no epilogue tile), then the
  constexpr int STOFF = LDSEND + 1024;             // statistics exchange: [WM][WN][64 ch][2] floats + one flag per consumer wave
  static_assert(DMA ? (LDSEND + 1024 <= 160 * 1024) : (BM * OP + WM * BN * 8 <= MAINB - POFF),
                "epilogue tile fits behind the live ring slots");
  char* const ot = smem + POFF;
  float* const red = (float*)(ot + BM * OP);
  auto store_tile = [&](auto FULLc) {
    constexpr bool FULL = decltype(FULLc)::value;
#ifndef SEGK_PIPE_STAMPS
    if (a.stats != nullptr && tid < BN) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) {  // fixed order: bit-stable
        t1 += red[(w * BN + tid) * 2 + 0];
        t2 += red[(w * BN + tid) * 2 + 1];
      }
      float2* dst = (float2*)a.stats + (size_t)umt * a.Ntot + un0 + tid;
      *dst = make_float2(t1, t2);
    }
#endif
    constexpr int CPR = BN * E::ES / 16;   // 16-byte chunks per pixel row of the tile
    constexpr int NST = BM * CPR / NTHR;
    const int cc = tid & (CPR - 1);
    const int n = un0 + cc * E::VEC;
    T* dbase;
    size_t pstride;
    const size_t pix0 = ((size_t)(ub * H + uy0)) * W + ux0;
    if (n < a.CO1) { dbase = (T*)a.out + pix0 * a.CO1 + n; pstride = a.CO1; }
    else { dbase = (T*)a.out2 + pix0 * a.CO2 + (n - a.CO1); pstride = a.CO2; }
    const size_t rstride = (size_t)W * pstride;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int m = (tid + i * NTHR) / CPR;
      const int ty = m >> TWL, tx = m & (TW - 1);
      const uint4 v = *(const uint4*)(ot + m * OP + cc * 16);
      if (FULL || (uy0 + ty < H && ux0 + tx < W)) *(uint4*)(dbase + ty * rstride + tx * pstride) = v;
    }
  };

  if constexpr (DMA) {
  if (wave >= 4) {
    // =========================================== PRODUCERS (LDS-DMA) ===========================================
    const int pw = wave - 4;                          // producer wave: pieces pw, pw + 4, ...
    constexpr int NPIECE = PH * RPX / 16;             // patch pieces (16 pixel slots = 1 KiB) per chunk
    static_assert(PH * RPX % 16 == 0, "whole pieces");
    constexpr int NPL = (NPIECE + 3) / 4;             // per wave (a wave without a last piece issues it into the trash KiB)
    constexpr int NPL0 = (NPL + 1) / 2;               // issued in the chunk's first step; the rest in its second
    constexpr int G16 = BN / 16;                      // 16-row weight groups per tap
    constexpr int NWL = 3 * G16 / 4;                  // weight pieces per wave and step (6 or 3)
    static_assert(3 * G16 % 4 == 0, "weight pieces divide evenly");
    // lane -> (pixel of the 16-slot piece, 16-byte piece q) by inverting the slot map; -> weight row / piece of a 16-row group
    const int lb = lane >> 5, lh = (lane >> 4) & 1, ls = lane & 15;
    const int p8 = (3 * ((ls >> 1) - lh)) & 7, pq = 2 * lh + (ls & 1);
    const unsigned wl_off = (unsigned)((lane >> 2) * 64 + (((lane & 3) ^ ((0x1230 >> (4 * ((lane >> 4) & 3))) & 3)) << 4));
    const char* const zp = (const char*)g_pipe_zero_page + ls * 16;
    int prel[NPL];                                    // (patch row << 8) | patch column of the lane's pixel; row 32767: never valid
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int j = pw + 4 * i;
      const int b8 = 2 * j + lb;
      const int py = b8 / (RPX / 8), px = (b8 - py * (RPX / 8)) * 8 + p8;
      prel[i] = (j < NPIECE && px < PW) ? ((py << 8) | px) : (0x7fff << 8);
    }
    unsigned pvalid = 0;
    int plin[NPL];
    auto patch_unit = [&](int b, int y0, int x0) {
      pvalid = 0;
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int gy = y0 + (prel[i] >> 8) - 1, gx = x0 + (prel[i] & 255) - 1;
        const bool ok = (gy >= 0) & (gy < H) & (gx >= 0) & (gx < W);
        plin[i] = ok ? (b * H + gy) * W + gx : 0;
        pvalid |= (ok ? 1u : 0u) << i;
      }
    };
    // pieces [I0, I1) of chunk kc into patch buffer `buf`
    auto issue_p = [&](int kc, int buf, auto I0c, auto I1c) {
      constexpr int I0 = decltype(I0c)::value, I1 = decltype(I1c)::value;
      const T* base;
      int C;
      if (kc < nchA) { base = (const T*)a.srcA + kc * E::CH; C = a.CA; }
      else { base = (const T*)a.srcB + (kc - nchA) * E::CH; C = a.CB; }
      const char* const cb = (const char*)(base + pq * E::VEC);
      const unsigned cbytes = (unsigned)C * 2u;
#pragma unroll
      for (int i = I0; i < I1; ++i) {
        const int j = pw + 4 * i;                       // wave-uniform
        char* const dst = smem + ((j < NPIECE) ? POFF + buf * PB + j * 1024 : LDSEND);
        const char* const src = ((pvalid >> i) & 1) ? cb + (size_t)((unsigned)plin[i] * cbytes) : zp;
        __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)dst, 16, 0, 0);
      }
    };
    // weight step `step` of the unit whose channel tile starts at n0 -> ring slot
    auto issue_w = [&](int n0, int step, int ring) {
      const char* const wb = (const char*)a.w + ((size_t)(step * 3) * a.Ntot + n0) * 64 + wl_off;
      char* const dst = smem + wring(ring);
#pragma unroll
      for (int i = 0; i < NWL; ++i) {
        const int j = pw + 4 * i, t = j / G16, g = j - t * G16;
        __builtin_amdgcn_global_load_lds((glb_vp)(wb + ((size_t)t * a.Ntot * 64 + g * 1024)), (lds_vp)(dst + j * 1024), 16, 0, 0);
      }
    };
    using I_0 = std::integral_constant<int, 0>;
    using I_H = std::integral_constant<int, NPL0>;
    using I_N = std::integral_constant<int, NPL>;

    issue_w(un0, 0, 0);
    issue_w(un0, 1, 1);
    patch_unit(ub, uy0, ux0);
    issue_p(0, 0, I_0{}, I_N{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // B0
    PIPE_STAMP(5);
    for (;;) {
      const int un = u + GW;
      const bool has_next = un < u_end;
      int nmt = 0, nb = 0, ny0 = 0, nx0 = 0, nn0 = 0;
      if (has_next) decode(un, nmt, nb, ny0, nx0, nn0);
      for (int kc = 0; kc < nchunks; ++kc) {
        const int s0 = 3 * kc;
        const bool last = (kc + 1 == nchunks);
        // ---- step TG0: weights of step s0 + 2 (slot 2); first half of the next chunk's patch (the next unit's chunk 0 into
        // P0 behind the last chunk: nchunks is even, so the last chunk lives in P1)
        issue_w(un0, s0 + 2, 2);
        if (!last) {
          issue_p(kc + 1, (kc + 1) & 1, I_0{}, I_H{});
          PIPE_STAMP(0);
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPL0) : "memory");     // the weights (older) have landed
        } else if (has_next) {
          patch_unit(nb, ny0, nx0);
          issue_p(0, 0, I_0{}, I_H{});
          PIPE_STAMP(0);
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPL0) : "memory");
        } else {
          PIPE_STAMP(0);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PIPE_STAMP(2);
        // raw s_barrier: __syncthreads() carries a fence that hipcc lowers to s_waitcnt vmcnt(0) -- it would drain the patch
        // pieces this step leaves in flight.  The DMA data a barrier publishes is covered by the counted wait above it.
        __builtin_amdgcn_s_barrier();
        PIPE_STAMP(1);
        // ---- step TG1: weights of step s0 + 3 (slot 0: the next chunk's or the next unit's first step); rest of the patch
        if (!last) issue_w(un0, s0 + 3, 0);
        else if (has_next) issue_w(nn0, 0, 0);
        if (!last) issue_p(kc + 1, (kc + 1) & 1, I_H{}, I_N{});
        else if (has_next) issue_p(0, 0, I_H{}, I_N{});
        PIPE_STAMP(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PIPE_STAMP(2);
        __builtin_amdgcn_s_barrier();
        PIPE_STAMP(1);
        // ---- step TG2: weights of step s0 + 4 (slot 1)
        if (!last) issue_w(un0, s0 + 4, 1);
        else if (has_next) issue_w(nn0, 1, 1);
        PIPE_STAMP(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PIPE_STAMP(2);
        __builtin_amdgcn_s_barrier();
        PIPE_STAMP(1);
      }
      // no boundary: the consumers store their results from registers, and the ring slots the next unit's first steps write
      // were released by the step barriers above
      if (!has_next) break;
      u = un; umt = nmt; ub = nb; uy0 = ny0; ux0 = nx0; un0 = nn0;
    }
    PIPE_STAMP_OUT();
    return;
  }
  } else {
  if (wave >= 4) {
    // =========================================== PRODUCERS ===========================================
    // wave priorities stay at their default: raising the producers (s_setprio 3) or the consumers (2) was measured
    // 3-5 % slower than leaving the oldest-first arbitration alone (512->512: 132 / 132 / 128 us)
    const int ptid = tid - NPT;
    const int trash = MAINB + ptid * 16;
    const int pc = ptid & 3;        // 16-byte slot inside the 64-byte chunk (NPT % 4 == 0: same for every piece)
    int plds[NPL], prel[NPL];
    unsigned pint = 0;              // bit i: piece i is an interior (non-halo) pixel of the tile
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int q = ptid + i * NPT;
      const int pix = q >> 2;
      const int py = pix / PW, px = pix - py * PW;
      plds[i] = (q < NP) ? POFF + py * ROWP + px * PPIX + pc * 16 : trash;
      prel[i] = (q < NP) ? ((py << 8) | px) : (0x7fff << 8);  // row 32767: outside every image, never valid
      pint |= ((q < NP && py >= 1 && py <= TH && px >= 1 && px <= TW) ? 1u : 0u) << i;
    }
    unsigned wsrc[NWL];
    int wlds[NWL];
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int q = ptid + i * NPT;
      const int t = q / (BN * 4), r = q - t * (BN * 4);
      wsrc[i] = t * a.Ntot * 64 + r * 16;              // byte offset from the step's weight base
      const int wrow = r >> 2, wpc = M16 ? ((r & 3) ^ ((0x1230 >> (4 * ((wrow >> 2) & 3))) & 3)) : (r & 3);   // [0,3,2,1]
      wlds[i] = t * (BN * WPIX) + wrow * WPIX + wpc * 16;
    }

    // ---- patch: per-unit pixel indices and validity, per-chunk source; one register set
    u32x4 preg[NPL];
    unsigned pvalid = 0;
    int plin[NPL];
    float psc[E::VEC], psh[E::VEC];
    int pl_aoff = -1;               // channel offset of the fetched chunk inside act_out (or -1)
    bool pl_first = false;          // the fetched patch belongs to a unit of channel tile 0 (writes act_out)
    auto patch_unit = [&](int b, int y0, int x0, int n0) {
      pl_first = (n0 == 0);
      pvalid = 0;
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int gy = y0 + (prel[i] >> 8) - 1, gx = x0 + (prel[i] & 255) - 1;
        const bool ok = (gy >= 0) & (gy < H) & (gx >= 0) & (gx < W);
        const int cy = ok ? gy : y0, cx = ok ? gx : x0;       // clamp to the tile origin: always a valid pixel
        plin[i] = (b * H + cy) * W + cx;
        pvalid |= (ok ? 1u : 0u) << i;
      }
    };
    auto patch_load = [&](int kc) {
      const T* base;
      int C, coff;
      if (kc < nchA) { base = (const T*)a.srcA; C = a.CA; coff = kc * E::CH; }
      else { base = (const T*)a.srcB; C = a.CB; coff = (kc - nchA) * E::CH; }
      base += coff + pc * E::VEC;
      pl_aoff = (kc < nchA) ? coff + pc * E::VEC : -1;      // act_out mirrors srcA only
      if (PIPE_ABL & 2) return;
#pragma unroll
      for (int i = 0; i < NPL; ++i) preg[i] = *(const u32x4*)(base + (size_t)(unsigned)(plin[i] * C));
      if (PRO) {
#pragma unroll
        for (int j = 0; j < E::VEC; ++j) {
          psc[j] = a.scale[coff + pc * E::VEC + j];
          psh[j] = a.shift[coff + pc * E::VEC + j];
        }
      }
    };
    auto patch_store = [&](int pboff) {
      if (PIPE_ABL & 1) return;
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        u32x4 v = preg[i];
        if (PRO) {  // BatchNorm(scale, shift) + ReLU of the producer layer, applied on load
          float f[E::VEC];
          unpack16<T>(make_uint4(v.x, v.y, v.z, v.w), f);
#pragma unroll
          for (int j = 0; j < E::VEC; ++j) f[j] = fmaxf(fmaf(f[j], psc[j], psh[j]), 0.f);
          const uint4 t = pack16<T>(f);
          v = (u32x4){t.x, t.y, t.z, t.w};
        }
        const bool ok = (pvalid >> i) & 1;   // out-of-image halo pixels are exact zeros (after the transform)
        v = ok ? v : (u32x4){0u, 0u, 0u, 0u};
        const int off = (plds[i] < MAINB) ? plds[i] + pboff : trash;
        *(u32x4*)(smem + off) = v;
        if (PRO) {   // side output: the transformed activation itself, interior pixels, once per pixel and chunk
          if (a.act_out && pl_first && pl_aoff >= 0 && ok && ((pint >> i) & 1))
            *(u32x4*)((T*)a.act_out + (size_t)(unsigned)(plin[i] * a.CA) + pl_aoff) = v;
        }
      }
    };

    // ---- weights: step x lives in register set x % 3 from its fetch (during step x-4) to its store
    // (during step x-2) and in ring slot x % 3 from then until step x has been computed
    u32x4 wreg[3][NWL];
    int cu_step = 0, cu_n0 = un0;   // fetch cursor: the next weight step to fetch, (unit, step) walking ahead
    bool cu_ok = true;              // false past the last unit
    int nn0 = 0;                    // channel origin of the next unit (set per unit below)
    bool has_next = false;
    auto fetch_w = [&](u32x4 (&R)[NWL]) {
      if (cu_ok) {
        const char* wb = (const char*)a.w + ((size_t)(cu_step * 3) * a.Ntot + cu_n0) * 64;
#pragma unroll
        for (int i = 0; i < NWL; ++i)
          if (!(PIPE_ABL & 2)) R[i] = *(const u32x4*)(wb + wsrc[i]);
        if (++cu_step == nsteps) { cu_step = 0; cu_n0 = nn0; cu_ok = has_next; }
      }
    };
    auto store_w = [&](int ring, const u32x4 (&R)[NWL]) {
      if (PIPE_ABL & 1) return;
#pragma unroll
      for (int i = 0; i < NWL; ++i) *(u32x4*)(smem + wring(ring) + wlds[i]) = R[i];
    };

    // first unit: weight steps 0, 1 and patch chunk 0 into LDS; steps 2, 3 and patch chunk 1 in registers
    {
      const int un = u + GW;
      has_next = un < u_end;
      int t0, t1, t2, t3;
      if (has_next) decode(un, t0, t1, t2, t3, nn0);
    }
    patch_unit(ub, uy0, ux0, un0);
    patch_load(0);
    fetch_w(wreg[0]);
    fetch_w(wreg[1]);
    store_w(0, wreg[0]);
    store_w(1, wreg[1]);
    patch_store(0);
    fetch_w(wreg[2]);
    fetch_w(wreg[0]);
    patch_load(1);
    __syncthreads();                                   // B0
    PIPE_STAMP(5);
    for (;;) {
      const int un = u + GW;
      has_next = un < u_end;
      int nmt = 0, nb = 0, ny0 = 0, nx0 = 0;
      if (has_next) decode(un, nmt, nb, ny0, nx0, nn0);
      for (int kc = 0; kc < nchunks; ++kc) {
        // step TG0: ring slot 2 <- weights of step s+2; fetch step s+4
        store_w(2, wreg[2]);
        fetch_w(wreg[1]);
        PIPE_STAMP(0);
        __syncthreads();
        PIPE_STAMP(1);
        // step TG1: the next chunk's patch, ring slot 0
        if (kc + 1 < nchunks) patch_store(((kc + 1) & 1) * PB);
        store_w(0, wreg[0]);
        fetch_w(wreg[2]);
        PIPE_STAMP(0);
        __syncthreads();
        PIPE_STAMP(1);
        // step TG2: ring slot 1; fetch the patch two chunks ahead (the next unit's chunk 0 from the
        // second-to-last chunk; the last chunk fetches nothing: its registers are stored after the epilogue)
        store_w(1, wreg[1]);
        fetch_w(wreg[0]);
        if (kc + 2 < nchunks) patch_load(kc + 2);
        else if (kc + 2 == nchunks && has_next) { patch_unit(nb, ny0, nx0, nn0); patch_load(0); }
        PIPE_STAMP(0);
        __syncthreads();
        PIPE_STAMP(1);
      }
      __syncthreads();                                 // E1: the consumers have staged the output tile
      PIPE_STAMP(2);
      if ((uy0 + TH <= H) && (ux0 + TW <= W)) store_tile(std::true_type{});
      else store_tile(std::false_type{});
      PIPE_STAMP(3);
      if (!has_next) break;
      __syncthreads();                                 // E2: tile consumed, [P0 | P1 | W2] may be rewritten
      patch_store(0);                                  // next unit's chunk 0
      patch_load(1);                                   // and its chunk 1, stored during its step 1
      u = un; umt = nmt; ub = nb; uy0 = ny0; ux0 = nx0; un0 = nn0;
      __syncthreads();                                 // E3
      PIPE_STAMP(4);
    }
    PIPE_STAMP_OUT();
    return;
  }
  }

  // ============================================= CONSUMERS =============================================
  if constexpr (!M16) {
  const int wm = wave / WN, wn = wave - wm * WN;
  const int lr = lane & 31, lh = lane >> 5;
  int laneA[MF], laneB[NF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    const int m = (wm * MF + mf) * 32 + lr;
    laneA[mf] = POFF + (m >> TWL) * ROWP + (m & (TW - 1)) * PIXB + lh * 16;
  }
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) laneB[nf] = ((wn * NF + nf) * 32 + lr) * PIXB + lh * 16;

  f32x16 acc[MF][NF];
  auto zero_acc = [&]() {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f;
  };
  uint4 fa[2][MF], fb[2][NF];   // fragment double buffer; [0] is primed before a step begins
  auto rd = [&](int prow, int wb, int i, uint4 (&A)[MF], uint4 (&Bf)[NF]) {
    const int t = i >> 1, kk = i & 1;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) A[mf] = *(const uint4*)(smem + prow + laneA[mf] + t * PIXB + kk * 32);
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) Bf[nf] = *(const uint4*)(smem + wb + laneB[nf] + t * (BN * PIXB) + kk * 32);
  };
  // one step (kernel row TG of chunk kc): 6 sub-steps of 8 MFMAs, fragments read one sub-step (256 matrix
  // cycles) ahead; the last sub-step reads the next step's first fragments
  auto step = [&](auto TGc, int kc) {
    constexpr int TG = decltype(TGc)::value;
    const int prow = (kc & 1) * PB + TG * ROWP;
    const int prow_next = (TG < 2) ? prow + ROWP : ((kc + 1) & 1) * PB;
    constexpr int wb = TG < 2 ? TG * WB : W2OFF, wb_next = (TG + 1) % 3 < 2 ? ((TG + 1) % 3) * WB : W2OFF;
#pragma unroll
    for (int i = 0; i < NSUB; ++i) {
      if (i + 1 < NSUB) rd(prow, wb, i + 1, fa[(i + 1) & 1], fb[(i + 1) & 1]);
      else rd(prow_next, wb_next, 0, fa[0], fb[0]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) Mma<T>::run(fa[i & 1][mf], fb[i & 1][nf], acc[mf][nf]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto stage_tile = [&](bool full) {   // bias, BN partial sums from the fp32 accumulators, tile -> LDS
    if (a.bias) {
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const float bv = a.bias[un0 + (wn * NF + nf) * 32 + lr];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mf][nf][r] += bv;
      }
    }
    if (!full) {   // rare: pixels past the image edge must not enter the statistics (they are never stored)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (wm * MF + mf) * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
          const bool in = (uy0 + (m >> TWL) < H) && (ux0 + (m & (TW - 1)) < W);
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) acc[mf][nf][r] = in ? acc[mf][nf][r] : 0.f;
        }
    }
    const bool do_stats = (a.stats != nullptr);
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      float s1 = 0.f, s2 = 0.f;
      const int n = (wn * NF + nf) * 32 + lr;
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int mb = (wm * MF + mf) * 32 + 4 * lh;
        char* const obase = ot + mb * OP + n * E::ES;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[mf][nf][r];
        stage_frag<T>(v, obase, OP, s1, s2);
      }
      if (do_stats) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (lh == 0) {
          red[(wm * BN + n) * 2 + 0] = s1;
          red[(wm * BN + n) * 2 + 1] = s2;
        }
      }
    }
  };

  zero_acc();
  __syncthreads();                                     // B0
  rd(0, 0, 0, fa[0], fb[0]);
  for (;;) {
    const int un = u + GW;
    const bool has_next = un < u_end;
    int nmt = 0, nb = 0, ny0 = 0, nx0 = 0, nn0 = 0;
    if (has_next) decode(un, nmt, nb, ny0, nx0, nn0);
    for (int kc = 0; kc < nchunks; ++kc) {
      step(std::integral_constant<int, 0>{}, kc);
      __syncthreads();
      step(std::integral_constant<int, 1>{}, kc);
      __syncthreads();
      step(std::integral_constant<int, 2>{}, kc);
      __syncthreads();
    }
    const bool full = (uy0 + TH <= H) && (ux0 + TW <= W);
    stage_tile(full);
    __syncthreads();                                   // E1
    if (full) store_tile(std::true_type{});
    else store_tile(std::false_type{});
    if (!has_next) break;
    zero_acc();
    __syncthreads();                                   // E2
    u = un; umt = nmt; ub = nb; uy0 = ny0; ux0 = nx0; un0 = nn0;
    __syncthreads();                                   // E3: the next unit's chunk 0 is in P0
    rd(0, 0, 0, fa[0], fb[0]);
  }
  } else {
  // ---- 16x16x32 form: the 128 x 64 wave tile is 8 x 4 blocks of 16 x 16; per tap (k = 32 = one whole chunk) 8 patch
  // and 4 weight fragments feed 32 MFMAs.  Sub-step = one tap x one half of the pixel blocks (4 A + 4 B -> 16 MFMAs, 256
  // matrix cycles, as before); fragments are read one sub-step ahead, the weight fragments of a tap once per two.
  const int wm = wave / WN, wn = wave - wm * WN;
  const int lc = lane & 15, lq = lane >> 4;
  constexpr int MB = 8, NB = 4;
  int laneA[DMA ? 3 : MB], laneB[NB];
  if constexpr (DMA) {
    // one address per tap column t: pixel x = (block start: 0 or 16) + t + lc of the wave's first tile row; pixel block mb
    // adds the immediate mboff(mb)
#pragma unroll
    for (int t = 0; t < 3; ++t)
      laneA[t] = POFF + ((wm * MB * 16) >> TWL) * ROWP + ((t + lc) >> 3) * 512 + (lq >> 1) * 256 + ((6 * (t + lc) + lq) & 15) * 16;
  } else {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int m = (wm * MB + mb) * 16 + lc;
      laneA[mb] = POFF + (m >> TWL) * ROWP + (m & (TW - 1)) * PPIX + lq * 16;
    }
  }
  // byte address of the lane's 16 bytes of pixel block mb at tap column t, relative to the patch row base `prow`
  auto pa = [&](int prow, int mb, int t) -> const char* {
    if constexpr (DMA) return smem + prow + laneA[t] + (((mb * 16) >> TWL) * ROWP + (((mb * 16) & (TW - 1)) >> 3) * 512);
    else return smem + prow + laneA[mb] + t * PPIX;
  };
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    // DMA form: the lane's row of block nb is channel 8 (lc >> 2) + 4 (nb & 1) + (lc & 3) of the block pair's 32
    const int row = DMA ? (wn * NB + (nb & ~1)) * 16 + 8 * (lc >> 2) + 4 * (nb & 1) + (lc & 3) : (wn * NB + nb) * 16 + lc;
    laneB[nb] = row * WPIX + ((lq ^ ((0x1230 >> (4 * ((row >> 2) & 3))) & 3)) << 4);
  }

  f32x4 acc[MB][NB];
  auto zero_acc = [&]() {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  uint4 fa[2][4], fb[3][NB];    // patch halves ping-pong per sub-step; weight fragments per tap ([0] primed before a step)
  auto rdA = [&](int prow, int t, int hh, uint4 (&A)[4]) {
    if (PIPE_ABL & 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(A[j].x), "+v"(A[j].y), "+v"(A[j].z), "+v"(A[j].w));
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) A[j] = *(const uint4*)pa(prow, 4 * hh + j, t);
  };
  auto rdB = [&](int wb, int t, uint4 (&Bf)[NB]) {
    if (PIPE_ABL & 4) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) asm volatile("" : "+v"(Bf[nb].x), "+v"(Bf[nb].y), "+v"(Bf[nb].z), "+v"(Bf[nb].w));
      return;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) Bf[nb] = *(const uint4*)(smem + wb + laneB[nb] + t * (BN * WPIX));
  };
  // staged form: pixels x channels (A = patch); DMA form: channels x pixels (A = weights), see the epilogue
  auto mma16 = [&](const uint4& pfrag, const uint4& wfrag, const f32x4& c) {
    if constexpr (DMA) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wfrag), __builtin_bit_cast(bf16x8, pfrag), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, pfrag), __builtin_bit_cast(bf16x8, wfrag), c, 0, 0, 0);
  };
  auto step = [&](auto TGc, int kc) {
    constexpr int TG = decltype(TGc)::value;
    const int prow = (kc & 1) * PB + TG * ROWP;
    const int prow_next = (TG < 2) ? prow + ROWP : ((kc + 1) & 1) * PB;
    constexpr int wb = TG < 2 ? TG * WB : W2OFF, wb_next = (TG + 1) % 3 < 2 ? ((TG + 1) % 3) * WB : W2OFF;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int t = i >> 1, hh = i & 1;
#if PIPE_ILV
      // Round 4: the next sub-step's fragments are read BETWEEN this sub-step's MFMAs, one ds_read_b128 behind every
      // second MFMA (weight fragments of a new tap first: the next sub-step's first MFMAs need all four).  A consumer is
      // alone on its SIMD's matrix pipe: while it issues a group of 4-8 reads back to back the pipe runs dry (tools/ubench/
      // pipe_roles.hip: 1968 -> 1652 cycles per 1536-cycle step in the bare loop, 2006 -> 1806 beside staging partners).
      {
        const int ni = (i + 1) % 6, nt_ = ni >> 1, nh = ni & 1;
        const int rp = (i + 1 < 6) ? prow : prow_next;
        const int rw = (i + 1 < 6) ? wb : wb_next;
        const bool needB = (nh == 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            acc[4 * hh + j][nb] = mma16(fa[i & 1][j], fb[t][nb], acc[4 * hh + j][nb]);
            const int m = j * NB + nb;
            if ((m & 1) && !(PIPE_ABL & 4)) {
              const int r = m >> 1;   // 0 .. 7
              if (needB) {
                if (r < 4) fb[nt_][r] = *(const uint4*)(smem + rw + laneB[r] + nt_ * (BN * WPIX));
                else fa[(i + 1) & 1][r - 4] = *(const uint4*)pa(rp, 4 * nh + r - 4, nt_);
              } else if (r < 4) {
                fa[(i + 1) & 1][r] = *(const uint4*)pa(rp, 4 * nh + r, nt_);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        continue;
      }
#endif
      if (i + 1 < 6) {
        rdA(prow, (i + 1) >> 1, (i + 1) & 1, fa[(i + 1) & 1]);
        if (((i + 1) & 1) == 0) rdB(wb, (i + 1) >> 1, fb[(i + 1) >> 1]);
      } else {
        rdA(prow_next, 0, 0, fa[0]);
        rdB(wb_next, 0, fb[0]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[4 * hh + j][nb] = mma16(fa[i & 1][j], fb[t][nb], acc[4 * hh + j][nb]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto stage_tile = [&](auto FULLc) {   // bias, BN partial sums from the fp32 accumulators, tile -> LDS
    constexpr bool full = decltype(FULLc)::value;      // compile time: a run-time test here is a branch per value
    const bool do_stats = (a.stats != nullptr);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int n = (wn * NB + nb) * 16 + lc;
      const float bv = a.bias ? a.bias[un0 + n] : 0.f;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int m0 = (wm * MB + mb) * 16 + 4 * lq;        // the lane holds pixels m0 .. m0 + 3 of channel n
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = acc[mb][nb][j] + bv;
          if (!full) {   // rare: pixels past the image edge must not enter the statistics (they are never stored)
            const int m = m0 + j;
            v[j] = ((uy0 + (m >> TWL) < H) && (ux0 + (m & (TW - 1)) < W)) ? v[j] : 0.f;
          }
          s1 += v[j];
          s2 = fmaf(v[j], v[j], s2);
        }
        char* const ob = ot + m0 * OP + n * E::ES;
        uint32_t p01, p23;
        p01 = cvt_pk_bf16(v[0], v[1]);
        p23 = cvt_pk_bf16(v[2], v[3]);
        *(uint16_t*)(ob) = (uint16_t)(p01 & 0xffffu);
        *(uint16_t*)(ob + OP) = (uint16_t)(p01 >> 16);
        *(uint16_t*)(ob + 2 * OP) = (uint16_t)(p23 & 0xffffu);
        *(uint16_t*)(ob + 3 * OP) = (uint16_t)(p23 >> 16);
      }
      if (do_stats) {
        s1 += __shfl_xor(s1, 16);
        s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (lq == 0) {
          red[(wm * BN + n) * 2 + 0] = s1;
          red[(wm * BN + n) * 2 + 1] = s2;
        }
      }
    }
  };

  int stat_seq = 0;                                          // DMA form: units this wave has written statistics for
  if constexpr (DMA) {
    if (lane == 0) ((volatile int*)(smem + STOFF + WM * WN * 64 * 8))[wave] = 0;   // flags (consumer waves = wm * WN + wn), before B0
  }
  // DMA form: the unit's results straight from the accumulators.  acc[mb][2k][j] / acc[mb][2k + 1][j] are channels
  // 32 k + 8 lq + j / + 4 + j (of the wave's 64) of pixel 16 mb + lc (of the wave's 128): 8 consecutive channels = one 16-byte
  // store per pixel block and channel group, 64 contiguous bytes per pixel from the four lq lanes.  Both groups of a pixel block
  // are stored BACK TO BACK, so the two 64-byte halves of a pixel's 128-byte line reach L2 within one store pair: with the
  // groups eight stores apart (the first form) the HBM write traffic of these launches was 20-36 % above the output size
  // (half-written lines leaving L2 twice: 183 -> 138 MB at 128->128@128, profiles/r04_pmc_conv_layers.txt).  No bias here --
  // every conv in front of a BatchNorm and every data gradient; a biased conv takes the staged form (launch_pipe): sixteen bias
  // registers beside both groups' statistics spill.
  auto direct_out = [&](auto FULLc) {
    constexpr bool full = decltype(FULLc)::value;
    static_assert(!DMA || NB == 4, "two 32-channel groups per consumer wave");
    const bool do_stats = (a.stats != nullptr);
    const int m0 = wm * MB * 16 + lc;                        // the lane's pixel of block 0
    const int ty0 = m0 >> TWL, tx0 = m0 & (TW - 1);
    const size_t pix0 = ((size_t)(ub * H + uy0 + ty0)) * W + ux0 + tx0;
    T* dpx[2];
    int pstr[2];
    float s1[2][8], s2[2][8];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int nbase = un0 + wn * 64 + 32 * k;              // wave-uniform; a 32-channel group never straddles CO1
      T* dch;
      if (nbase < a.CO1) { dch = (T*)a.out + nbase + 8 * lq; pstr[k] = a.CO1; }
      else { dch = (T*)a.out2 + (nbase - a.CO1) + 8 * lq; pstr[k] = a.CO2; }
      dpx[k] = dch + pix0 * pstr[k];
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[k][e] = 0.f; s2[k][e] = 0.f; }
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int dty = (mb * 16) >> TWL, dtx = (mb * 16) & (TW - 1);     // compile-time per mb
      // rare (!full): pixels past the image edge are neither stored nor counted
      const bool in = full || ((uy0 + ty0 + dty < H) && (ux0 + tx0 + dtx < W));
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] = acc[mb][2 * k + (e >> 2)][e & 3];
          if (!full) v[e] = in ? v[e] : 0.f;
          s1[k][e] += v[e];
          s2[k][e] = fmaf(v[e], v[e], s2[k][e]);
        }
        const uint4 o = make_uint4(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7]));
        if (in && !(PIPE_ABL & 8)) *(uint4*)(dpx[k] + (size_t)(dty * W + dtx) * pstr[k]) = o;
        if (PIPE_ABL & 8) asm volatile("" ::"v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w));
      }
    }
    if (do_stats && !(PIPE_ABL & 16)) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[k][e] = pipe_row16_sum(s1[k][e]); s2[k][e] = pipe_row16_sum(s2[k][e]); }
        if (lc == 0) {   // the wave's sums over its 128 pixels -> its slot of the exchange area
          float4* const dst = (float4*)(smem + STOFF) + ((wm * WN + wn) * 64 + 32 * k + 8 * lq) / 2;
#pragma unroll
          for (int e = 0; e < 8; e += 2) dst[e / 2] = make_float4(s1[k][e], s2[k][e], s1[k][e + 1], s2[k][e + 1]);
        }
      }
    }
    if (do_stats && !(PIPE_ABL & 16)) {
      // ONE statistics row per unit: the consumer waves of a channel half (same wn, WM pixel parts) meet in LDS -- parts 1 ..
      // WM - 1 publish (data, then a sequence flag), part 0 waits for their flags and adds the WM values per channel in part
      // order (bit-stable) -- 512 contiguous bytes per wave instead of WM rows for segk_bn_finalize to walk (a first form with
      // WM rows per unit made the BatchNorm finalize launches 0.17 -> 0.28 ms per step).  No wave waits on a wave that waits:
      // the publishers never wait, and every wave passes at least one workgroup barrier between two units, so a slot is
      // consumed long before it is rewritten.
      volatile int* const flags = (volatile int*)(smem + STOFF + WM * WN * 64 * 8);
      ++stat_seq;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the slot is written
      if (wm > 0) {
        if (lane == 0) flags[wm * WN + wn] = stat_seq;
      } else {
#pragma unroll
        for (int w = 1; w < WM; ++w)
          while (flags[w * WN + wn] != stat_seq) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");                           // the slots are read after the flags, not before
        const float2* const src = (const float2*)(smem + STOFF) + wn * 64 + lane;
        float2 t = src[0];
#pragma unroll
        for (int w = 1; w < WM; ++w) {
          const float2 q = src[w * WN * 64];
          t.x += q.x;
          t.y += q.y;
        }
        ((float2*)a.stats)[(size_t)umt * a.Ntot + un0 + wn * 64 + lane] = t;
      }
    }
  };

  zero_acc();
  __syncthreads();                                     // B0
  PIPE_STAMP(5);
  rdA(0, 0, 0, fa[0]);
  rdB(0, 0, fb[0]);
  for (;;) {
    const int un = u + GW;
    const bool has_next = un < u_end;
    int nmt = 0, nb = 0, ny0 = 0, nx0 = 0, nn0 = 0;
    if (has_next) decode(un, nmt, nb, ny0, nx0, nn0);
    // step barriers: a raw s_barrier.  __syncthreads() adds s_waitcnt lgkmcnt(0), which drains the fragment pre-reads of the next
    // step that the last sub-step left in flight; every read of the slot this barrier releases has already been consumed by an
    // MFMA (so it has returned), and the pre-reads target a slot that was published one barrier earlier.
    for (int kc = 0; kc < nchunks; ++kc) {
      step(std::integral_constant<int, 0>{}, kc);
      PIPE_STAMP(0);
      step_barrier();
      PIPE_STAMP(1);
      step(std::integral_constant<int, 1>{}, kc);
      PIPE_STAMP(0);
      step_barrier();
      PIPE_STAMP(1);
      step(std::integral_constant<int, 2>{}, kc);
      PIPE_STAMP(0);
      step_barrier();
      PIPE_STAMP(1);
    }
    const bool full = (uy0 + TH <= H) && (ux0 + TW <= W);
    if constexpr (DMA) {
      if (PIPE_ABL & 32) { asm volatile("" ::"v"(acc[0][0][0])); }
      else if (full) direct_out(std::true_type{});
      else direct_out(std::false_type{});
      PIPE_STAMP(2);
      if (!has_next) break;
      zero_acc();
      u = un; umt = nmt; ub = nb; uy0 = ny0; ux0 = nx0; un0 = nn0;
      rdA(0, 0, 0, fa[0]);                             // (P0 / W0 of the next unit landed under this unit's last steps)
      rdB(0, 0, fb[0]);
      PIPE_STAMP(4);
      continue;
    }
    if (full) stage_tile(std::true_type{});
    else stage_tile(std::false_type{});
    PIPE_STAMP(2);
    __syncthreads();                                   // E1
    if (full) store_tile(std::true_type{});
    else store_tile(std::false_type{});
    PIPE_STAMP(3);
    if (!has_next) break;
    zero_acc();
    __syncthreads();                                   // E2
    u = un; umt = nmt; ub = nb; uy0 = ny0; ux0 = nx0; un0 = nn0;
    __syncthreads();                                   // E3: the next unit's chunk 0 is in P0
    PIPE_STAMP(4);
    rdA(0, 0, 0, fa[0]);
    rdB(0, 0, fb[0]);
  }
  PIPE_STAMP_OUT();
  }
}

// Weight-stationary variant for the narrow, high-resolution layers (Cin <= 2 chunks, 64 output channels per
// tile: the 3->64 stem, the 64->64 convs of down1/up4 and the 64->128 concat gradient): these are HBM-bound
// (K = 9*Cin is tiny), so the structure is built around keeping the memory pipes busy instead of the
// matrix cores.  All 9*Cin*64 weights of the workgroup's channel tile are loaded into LDS ONCE; a
// persistent workgroup (8 waves, 256-pixel tiles) then streams pixel tiles: the next tile's patch is
// fetched into registers before the current tile's barrier-free MFMA loop (36 k-steps), so its HBM
// latency hides under the matrix work, and the only barriers left are the four around the epilogue.
template <typename T, int TWL, bool PRO>
__global__ __launch_bounds__(512, 2) void conv_ws_kernel(const ConvArgs a) {
  using E = ET<T>;
  constexpr int WM = 4, WN = 2, MF = 2, NF = 1, NTHR = 512, BM = 256, BN = 64, MAXCH = 2;
  constexpr int TW = 1 << TWL, TH = BM >> TWL, PW = TW + 2, PH = TH + 2;
  constexpr int ROWP = (PW * PIXB + 255) & ~255;
  constexpr int PB = PH * ROWP;                    // one chunk of the patch
  constexpr int WTAP = BN * PIXB;                  // one tap of one chunk of the weights
  constexpr int NP = PH * PW * 4, NPL = (NP + NTHR - 1) / NTHR;
  constexpr int WOFF = MAXCH * PB;                 // weights live behind the patch chunks
  constexpr int MAINB = WOFF + MAXCH * 9 * WTAP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  const int lr = lane & 31, lh = lane >> 5;
  const int H = a.H, W = a.W;
  const int nchunks = (a.CA + a.CB) / E::CH;       // 1 or 2 (checked by the launcher)
  const int nchA = a.CA / E::CH;

  // units: the channel tile is fixed per workgroup (its weights stay resident); pixel tiles are walked
  // with stride GW inside the XCD's contiguous range
  const int NT = a.Ntot / BN;
  const int tpi = a.tiles_x * a.tiles_y, MT = a.B * tpi;
  const int xcd = blockIdx.x & 7, wgi = blockIdx.x >> 3, GWX = gridDim.x >> 3;
  const int nt = wgi % NT, GW = GWX / NT;
  const int mpx = (MT + 7) >> 3;
  int mt = xcd * mpx + wgi / NT;
  const int mt_end = min(MT, (xcd + 1) * mpx);
  if (mt >= mt_end) return;
  const int n0 = nt * BN;

  const int trash = MAINB + tid * 16;
  const int pc = tid & 3;
  int plds[NPL], prel[NPL];
  unsigned pint = 0;                               // bit i: piece i is an interior (non-halo) pixel of the tile
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int q = tid + i * NTHR;
    const int pix = q >> 2;
    const int py = pix / PW, px = pix - py * PW;
    plds[i] = (q < NP) ? py * ROWP + px * PIXB + pc * 16 : trash;
    prel[i] = (q < NP) ? ((py << 8) | px) : (0x7fff << 8);
    pint |= ((q < NP && py >= 1 && py <= TH && px >= 1 && px <= TW) ? 1u : 0u) << i;
  }
  int laneA[MF], laneB;
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    const int m = (wm * MF + mf) * 32 + lr;
    laneA[mf] = (m >> TWL) * ROWP + (m & (TW - 1)) * PIXB + lh * 16;
  }
  laneB = WOFF + (wn * 32 + lr) * PIXB + lh * 16;

  // ---- resident weights: [chunk][tap][64 rows][80 B]
  {
    const int total = nchunks * 9 * BN * 4;
    for (int q = tid; q < total; q += NTHR) {
      const int r = q & (BN * 4 - 1), ct = q / (BN * 4);     // ct = chunk*9 + tap
      const u32x4 v = *(const u32x4*)((const char*)a.w + ((size_t)ct * a.Ntot + n0) * 64 + r * 16);
      *(u32x4*)(smem + WOFF + ct * WTAP + (r >> 2) * PIXB + (r & 3) * 16) = v;
    }
  }

  u32x4 preg[MAXCH][NPL];
  unsigned pvalid = 0;
  float psc[MAXCH][E::VEC], psh[MAXCH][E::VEC];
  if (PRO) {
#pragma unroll
    for (int kc = 0; kc < MAXCH; ++kc)
#pragma unroll
      for (int j = 0; j < E::VEC; ++j) {
        const int c = (kc < nchunks ? kc : 0) * E::CH + pc * E::VEC + j;
        psc[kc][j] = a.scale[c];
        psh[kc][j] = a.shift[c];
      }
  }
  int plin[NPL];                                   // pixel index of each fetched piece (for the act_out side output)
  auto load_patch = [&](int b, int y0, int x0) {
    pvalid = 0;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int gy = y0 + (prel[i] >> 8) - 1, gx = x0 + (prel[i] & 255) - 1;
      const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
      const int cy = ok ? gy : y0, cx = ok ? gx : x0;
      const size_t lin = (size_t)(b * H + cy) * W + cx;
      plin[i] = (int)lin;
      pvalid |= (ok ? 1u : 0u) << i;
#pragma unroll
      for (int kc = 0; kc < MAXCH; ++kc) {
        const int k = kc < nchunks ? kc : 0;                  // unused second chunk: harmless duplicate load
        const T* src = (k < nchA) ? (const T*)a.srcA + lin * a.CA + k * E::CH
                                  : (const T*)a.srcB + lin * a.CB + (k - nchA) * E::CH;
        preg[kc][i] = *(const u32x4*)(src + pc * E::VEC);
      }
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int kc = 0; kc < MAXCH; ++kc)
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        u32x4 v = preg[kc][i];
        if (PRO) {
          float f[E::VEC];
          unpack16<T>(make_uint4(v.x, v.y, v.z, v.w), f);
#pragma unroll
          for (int j = 0; j < E::VEC; ++j) f[j] = fmaxf(fmaf(f[j], psc[kc][j], psh[kc][j]), 0.f);
          const uint4 t = pack16<T>(f);
          v = (u32x4){t.x, t.y, t.z, t.w};
        }
        const bool ok = (pvalid >> i) & 1;
        v = ok ? v : (u32x4){0u, 0u, 0u, 0u};
        const int off = plds[i] + ((plds[i] < MAINB) ? kc * PB : 0);
        *(u32x4*)(smem + off) = v;
        if (PRO) {   // side output: the transformed activation itself, interior pixels, once per pixel and chunk
          if (a.act_out && nt == 0 && kc < nchunks && ok && ((pint >> i) & 1))
            *(u32x4*)((T*)a.act_out + (size_t)(unsigned)plin[i] * a.CA + kc * E::CH + pc * E::VEC) = v;
        }
      }
  };

  f32x16 acc[MF];
  int ub, uy0, ux0;
  auto decode = [&](int m, int& b, int& y0, int& x0) {
    b = m / tpi;
    const int trem = m - b * tpi;
    const int tyi = trem / a.tiles_x;
    y0 = tyi * TH;
    x0 = (trem - tyi * a.tiles_x) * TW;
  };

  constexpr int OP = BN * E::ES + 16;
  auto epilogue_t = [&](auto FULLc) {
    constexpr bool FULL = decltype(FULLc)::value;
    char* const ot = smem;
    float* const red = (float*)(smem + BM * OP);
    const bool do_stats = (a.stats != nullptr);
    float s1 = 0.f, s2 = 0.f;
    const int n = wn * 32 + lr;
    const float bv = a.bias ? a.bias[n0 + n] : 0.f;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int mb = (wm * MF + mf) * 32 + 4 * lh;
      char* const obase = ot + mb * OP + n * E::ES;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        v[r] = acc[mf][r] + bv;
        if (!FULL) {
          const int m = mb + (r & 3) + 8 * (r >> 2);
          v[r] = ((uy0 + (m >> TWL) < H) && (ux0 + (m & (TW - 1)) < W)) ? v[r] : 0.f;
        }
      }
      stage_frag<T>(v, obase, OP, s1, s2);
    }
    if (do_stats) {
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (lh == 0) {
        red[(wm * BN + n) * 2 + 0] = s1;
        red[(wm * BN + n) * 2 + 1] = s2;
      }
    }
    __syncthreads();
    if (do_stats && tid < BN) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        t1 += red[(w * BN + tid) * 2 + 0];
        t2 += red[(w * BN + tid) * 2 + 1];
      }
      *((float2*)a.stats + (size_t)mt * a.Ntot + n0 + tid) = make_float2(t1, t2);
    }
    constexpr int CPR = BN * E::ES / 16, NST = BM * CPR / NTHR;
    const int cc = tid & (CPR - 1);
    const int nn = n0 + cc * E::VEC;
    const size_t pix0 = ((size_t)(ub * H + uy0)) * W + ux0;
    T* dbase;
    size_t pstride;
    if (nn < a.CO1) { dbase = (T*)a.out + pix0 * a.CO1 + nn; pstride = a.CO1; }
    else { dbase = (T*)a.out2 + pix0 * a.CO2 + (nn - a.CO1); pstride = a.CO2; }
    const size_t rstride = (size_t)W * pstride;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int m = (tid + i * NTHR) / CPR;
      const int ty = m >> TWL, tx = m & (TW - 1);
      const uint4 v = *(const uint4*)(ot + m * OP + cc * 16);
      if (FULL || (uy0 + ty < H && ux0 + tx < W)) *(uint4*)(dbase + ty * rstride + tx * pstride) = v;
    }
  };

  decode(mt, ub, uy0, ux0);
  load_patch(ub, uy0, ux0);
  store_patch();
  __syncthreads();
  for (;;) {
    const int mnext = mt + GW;
    const bool has_next = mnext < mt_end;
    int nb = ub, ny0 = uy0, nx0 = ux0;
    if (has_next) {
      decode(mnext, nb, ny0, nx0);
      load_patch(nb, ny0, nx0);            // flies under the MFMA loop below
    }
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mf][r] = 0.f;
    for (int kc = 0; kc < nchunks; ++kc) {
      // 18 sub-steps (9 taps x 2 k-steps); fragment reads run one sub-step ahead of the MFMAs
      const char* pbase = smem + kc * PB;
      const char* wbase = smem + kc * 9 * WTAP + laneB;
      uint4 fa[2][MF], fb[2];
      auto rd = [&](int i, uint4 (&A)[MF], uint4& Bf) {
        const int tap = i >> 1, kk = i & 1;
        const char* pb = pbase + (tap / 3) * ROWP + (tap % 3) * PIXB + kk * 32;
        Bf = *(const uint4*)(wbase + tap * WTAP + kk * 32);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) A[mf] = *(const uint4*)(pb + laneA[mf]);
      };
      rd(0, fa[0], fb[0]);
#pragma unroll
      for (int i = 0; i < 18; ++i) {
        if (i + 1 < 18) rd(i + 1, fa[(i + 1) & 1], fb[(i + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch reads ahead of this sub-step's MFMAs
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) Mma<T>::run(fa[i & 1][mf], fb[i & 1], acc[mf]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();                       // patch fully consumed: its LDS becomes the output staging tile
    if ((uy0 + TH <= H) && (ux0 + TW <= W)) epilogue_t(std::true_type{});
    else epilogue_t(std::false_type{});
    if (!has_next) break;
    __syncthreads();
    store_patch();
    mt = mnext; ub = nb; uy0 = ny0; ux0 = nx0;
    __syncthreads();
  }
}

template <typename T, int TWL, bool PRO>
int launch_ws(ConvArgs a, hipStream_t st) {
  using E = ET<T>;
  constexpr int BM = 256, BN = 64, NTHR = 512;
  constexpr int TW = 1 << TWL, TH = BM >> TWL, PW = TW + 2, PH = TH + 2;
  constexpr int ROWP = (PW * PIXB + 255) & ~255;
  constexpr size_t lds = 2 * (size_t)PH * ROWP + 2 * 9 * (size_t)BN * PIXB + NTHR * 16;
  static_assert(lds <= 160 * 1024, "conv_ws: LDS exceeds 160 KiB");
  static_assert(BM * (BN * E::ES + 16) + 4 * BN * 8 <= 2 * PH * ROWP, "output staging tile fits in the patch region");
  a.twl = TWL;
  a.tiles_x = cdiv(a.W, TW);
  a.tiles_y = cdiv(a.H, TH);
  const int NT = a.Ntot / BN;
  const int MT = a.B * a.tiles_x * a.tiles_y;
  int gw = num_cus() / 8;                                  // one workgroup per CU
  gw -= gw % NT;
  const int need = ((MT + 7) / 8) * NT;
  if (gw > need) gw = need;
  if (gw < NT) gw = NT;
  a.persistent = 1;
  auto kern = conv_ws_kernel<T, TWL, PRO>;
  static bool attr_set[SEGK_MAX_DEVICES] = {};     // per device: the attribute is device state
  const int dev = segk_device_index();
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      SEGK_FAIL(-3, "conv_ws: cannot raise dynamic LDS limit");
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3(8 * gw), dim3(NTHR), lds, st, a);
  SEGK_CHECK_LAUNCH("conv_ws");
  return 0;
}

template <int TWL, bool PRO, int BN, bool M16, bool DMA = false>
int launch_pipe_m(ConvArgs a, hipStream_t st) {
  constexpr int BM = 32768 / BN, NTHR = 512;
  constexpr int TW = 1 << TWL, TH = BM >> TWL, PW = TW + 2, PH = TH + 2;
  constexpr int PPIX = DMA ? 64 : (M16 ? 96 : PIXB), WPIX = M16 ? 64 : PIXB;
  constexpr int ROWP = DMA ? ((PW + 7) & ~7) * 64 : ((PW * PPIX + 255) & ~255);
  constexpr size_t ring = 3 * (size_t)(3 * BN * WPIX) + 2 * (size_t)PH * ROWP;
  // + a trash KiB and the statistics exchange (4 waves x 64 channels x 2 floats, 4 flags) / the producers' trash slots
  constexpr size_t lds = DMA ? ring + 1024 + 4 * 64 * 8 + 64 : ring + (NTHR / 2) * 16;
  static_assert(lds <= 160 * 1024, "conv3x3_pipe: LDS exceeds 160 KiB");
  a.twl = TWL;
  a.tiles_x = cdiv(a.W, TW);
  a.tiles_y = cdiv(a.H, TH);
  const int units = a.B * a.tiles_x * a.tiles_y * (a.Ntot / BN);
  const int per_xcd = (units + 7) / 8;
  int gw = num_cus() / 8;                                  // one 8-wave workgroup per CU
  if (gw > per_xcd) gw = per_xcd;
  a.persistent = 1;
  auto kern = conv3x3_pipe_kernel<TWL, PRO, BN, M16, DMA>;
  static bool attr_set[SEGK_MAX_DEVICES] = {};
  const int dev = segk_device_index();
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      SEGK_FAIL(-3, "conv3x3_pipe: cannot raise dynamic LDS limit");
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3(8 * gw), dim3(NTHR), lds, st, a);
  SEGK_CHECK_LAUNCH("conv3x3_pipe");
  return 0;
}

template <int TWL, bool PRO, int BN>
int launch_pipe(ConvArgs a, hipStream_t st) {
  // 16x16x32 consumers on every layer of this kernel.  Round 2 kept the 32x32x16 form for Cin = 128 (same-box kbench: 128->128@128
  // 164 vs 160 us); on the whole step the 16x16 form is 0.5 % faster on all three A/B pairs of round 3 (12.19 / 12.23 / 12.18 vs
  // 12.22 / 12.28 / 12.28 ms), its producers' stores lose 6 % of their LDS cycles to bank conflicts instead of 22 %, and its
  // build has no spilled registers (the 32x32 instances carry 45-60 outside the K loop).  The 32x32x16 form stays compiled
  // for the A/B switch.
  static const char* const force = getenv("SEGK_PIPE_MFMA");  // A/B switch for tools/kbench.py / tools/ab_bench.sh: "16" or "32"
  const bool m16 = force ? (force[0] == '1') : true;
  if constexpr (!PRO) {
    // LDS-DMA producers (round 4): layers without a BatchNorm prologue and without a bias (the register epilogue carries none)
    // whose chunk count is even (the patch ring's parity across the unit boundary) and whose sources stay below 4 GiB (32-bit
    // byte offsets per DMA lane)
    static const char* const nodma = getenv("SEGK_PIPE_DMA");  // "0": the register-staged producers of rounds 1-3 (A/B runs)
    const int nchunks = (a.CA + a.CB) / 32;
    const long long px = (long long)a.B * a.H * a.W;
    int cmax = a.CA > a.CB ? a.CA : a.CB;
    cmax = cmax > a.CO1 ? cmax : a.CO1;
    cmax = cmax > a.CO2 ? cmax : a.CO2;
    if (m16 && !(nodma && nodma[0] == '0') && nchunks % 2 == 0 && px * cmax * 2 < 4294967296LL && a.bias == nullptr)
      return launch_pipe_m<TWL, PRO, BN, true, true>(a, st);
  }
  return m16 ? launch_pipe_m<TWL, PRO, BN, true>(a, st) : launch_pipe_m<TWL, PRO, BN, false>(a, st);
}

template <typename T, int GEO, int TWL, int WM, int WN, int MF, int NF, int PBUF, bool PRO>
int launch_pro(ConvArgs a, hipStream_t st) {
  using E = ET<T>;
  constexpr int BM = WM * MF * 32, BN = WN * NF * 32, NTHR = WM * WN * 64;
  constexpr int HALO = (GEO == 0) ? 1 : 0;
  constexpr int TPS = (GEO == 0) ? 3 : 1;
  constexpr int TW = 1 << TWL, TH = BM >> TWL;
  constexpr int PW = TW + 2 * HALO, PH = TH + 2 * HALO;
  constexpr int ROWP = (PW * PIXB + 255) & ~255;
  a.twl = TWL;
  a.tiles_x = cdiv(a.W, TW);
  a.tiles_y = cdiv(a.H, TH);
  constexpr size_t main_b = PBUF * (size_t)PH * ROWP + 2 * (size_t)TPS * BN * PIXB + NTHR * 16;   // + trash slots
  constexpr size_t epi_b = BM * (size_t)(BN * E::ES + 16) + (size_t)WM * BN * 8;
  constexpr size_t lds = main_b > epi_b ? main_b : epi_b;
  static_assert(lds <= 160 * 1024, "conv_igemm: LDS exceeds 160 KiB");
  const int units = a.B * a.tiles_x * a.tiles_y * (a.Ntot / BN);
  const int per_xcd = (units + 7) / 8;
  // persistent grid: as many workgroups per CU as LDS allows (the register budget allows 2 x 8 waves)
  int wg_per_cu = (int)((160 * 1024) / lds);
  const int wg_cap = (NTHR == 512) ? 1 : 2;
  if (wg_per_cu > wg_cap) wg_per_cu = wg_cap;
  int gw = (num_cus() / 8) * wg_per_cu;
  if (gw > per_xcd) gw = per_xcd;
  a.persistent = 1;
  auto kern = conv_igemm_kernel<T, GEO, TWL, WM, WN, MF, NF, PBUF, PRO>;
  static bool attr_set[SEGK_MAX_DEVICES] = {};  // idempotent; racing setters write the same value
  const int dev = segk_device_index();
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      SEGK_FAIL(-3, "conv_igemm: cannot raise dynamic LDS limit");
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3(8 * gw), dim3(NTHR), lds, st, a);
  SEGK_CHECK_LAUNCH("conv_igemm");
  return 0;
}

template <typename T, int GEO, int TWL, int WM, int WN, int MF, int NF, int PBUF>
int launch_cfg(const ConvArgs& a, hipStream_t st) {
  if constexpr (GEO == 0) {
    if (a.scale) return launch_pro<T, GEO, TWL, WM, WN, MF, NF, PBUF, true>(a, st);
  }
  return launch_pro<T, GEO, TWL, WM, WN, MF, NF, PBUF, false>(a, st);
}

template <typename T, int GEO>
int launch_geo(const ConvArgs& a, hipStream_t st) {
  // BN must divide N (with the pixel-shuffle store a tile may span taps: each thread derives its own tap)
  const int unit = a.Ntot;
  const bool wide = a.W > 16;                      // 8x32 tiles unless the image is at most 16 wide
  if constexpr (GEO == 0 && sizeof(T) == 2) {
    // narrow high-resolution layers, images wider than 16 pixels: register-stationary streaming kernel (conv_rs.hip)
    if (segk_conv_use_rs(a.CA + a.CB, a.Ntot, SEGK_DT_BF16, a.W)) {
      if (!a.bias) return segk_conv_rs_launch(a, st);
      // a biased layer falls through to the weight-stationary kernel, whose statistics rows are per tile while
      // segk_conv_tiles() sized the buffer for conv_rs: the combination is refused instead of overrunning it
      SEGK_REQUIRE(!a.stats, "conv3x3: bias together with BatchNorm statistics is not served for this layer shape");
    }
  }
  if constexpr (GEO == 0 && sizeof(T) == 2) {
    // narrow high-resolution layers: weight-stationary streaming kernel (64-channel tiles)
    if (segk_conv_use_ws(a.CA + a.CB, a.Ntot, sizeof(T) == 2 ? SEGK_DT_BF16 : SEGK_DT_F32)) {
      if (a.scale) return wide ? launch_ws<T, 5, true>(a, st) : launch_ws<T, 4, true>(a, st);
      return wide ? launch_ws<T, 5, false>(a, st) : launch_ws<T, 4, false>(a, st);
    }
  }
  if constexpr (GEO == 0 && sizeof(T) == 2) {
    const int pk = segk_conv_use_pipe(a.CA + a.CB, a.Ntot, SEGK_DT_BF16);   // MFMA-bound bf16 layers: producer/consumer kernel
    if (pk == 128) {
      if (a.scale) return wide ? launch_pipe<5, true, 128>(a, st) : launch_pipe<4, true, 128>(a, st);
      return wide ? launch_pipe<5, false, 128>(a, st) : launch_pipe<4, false, 128>(a, st);
    }
    if (pk == 64) {
      if (a.scale) return wide ? launch_pipe<5, true, 64>(a, st) : launch_pipe<4, true, 64>(a, st);
      return wide ? launch_pipe<5, false, 64>(a, st) : launch_pipe<4, false, 64>(a, st);
    }
  }
  if constexpr (GEO == 1 && sizeof(T) == 2) {
    // long-K GEMMs (ViT projections, ConvTranspose up-sampling and its data gradient): producer/consumer kernel
    const int mode = a.shuffle ? 1 : (a.unshuf ? 2 : 0);
    const int nchA = a.CA / 32, nchunks = a.unshuf ? 4 * nchA : nchA;
    const long M = (long)a.B * a.H * a.W;
    if (!a.srcB && !a.out2 && !a.stats && segk_gemm_pipe_ok(M, nchunks, nchA, a.Ntot, a.CO1, mode)) {
      GemmArgs g{};
      g.A = a.srcA; g.w = (const char*)a.w; g.bias = a.bias; g.out = a.out;
      g.M = M; g.N = a.Ntot; g.nchunks = nchunks; g.nchA = nchA; g.lda = a.CA; g.H = a.H; g.W = a.W; g.Cout = a.CO1;
      g.act = a.act;
      return segk_gemm_pipe_launch(g, mode, st);
    }
  }
  if constexpr (GEO == 1 && sizeof(T) == 2) {
    // 1x1 / ConvTranspose GEMMs have a short K (Cin) and are bound by their output epilogue: 128-pixel tiles on
    // 4-wave workgroups, two per CU, so one workgroup's epilogue overlaps the other's loads and MFMAs
    if (unit % 128 == 0) return launch_cfg<T, GEO, 4, 2, 2, 2, 2, 2>(a, st);
    if (unit % 64 == 0) return launch_cfg<T, GEO, 4, 2, 2, 2, 1, 2>(a, st);
  }
  if (unit % 128 == 0)                             // 256 px x 128 ch, 8 waves
    return wide ? launch_cfg<T, GEO, 5, 4, 2, 2, 2, 2>(a, st) : launch_cfg<T, GEO, 4, 4, 2, 2, 2, 2>(a, st);
  if constexpr (GEO == 0) {
    if (unit % 64 == 0) return launch_cfg<T, GEO, 4, 2, 2, 2, 1, 1>(a, st);        // 128 px x 64 ch, 4 waves
    return launch_cfg<T, GEO, 4, 4, 1, 1, 1, 1>(a, st);                            // 128 px x 32 ch, 4 waves
  } else {
    if (unit % 64 == 0)
      return wide ? launch_cfg<T, GEO, 5, 4, 2, 2, 1, 2>(a, st) : launch_cfg<T, GEO, 4, 4, 2, 2, 1, 2>(a, st);
    return wide ? launch_cfg<T, GEO, 5, 8, 1, 1, 1, 2>(a, st) : launch_cfg<T, GEO, 4, 8, 1, 1, 1, 2>(a, st);
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
  SEGK_REQUIRE(!(a.scale && (a.CB || geo != 0)), "conv_igemm: BN prologue needs the 3x3 geometry and one source");
  SEGK_REQUIRE((a.scale == nullptr) == (a.shift == nullptr), "conv_igemm: scale/shift must come together");
  SEGK_REQUIRE(!a.act_out || (a.scale && geo == 0 && segk_conv_writes_act(a.CA + a.CB, a.Ntot, dtype)),
               "conv_igemm: act_out needs the BN prologue and a layer served by the producer/consumer or weight-stationary kernel");
  SEGK_REQUIRE(a.Ntot > 0 && a.Ntot % 32 == 0, "conv_igemm: N=%d must be a multiple of 32", a.Ntot);
  SEGK_REQUIRE(a.CO1 > 0 && a.CO1 % 32 == 0 && a.CO2 >= 0 && a.CO2 % 32 == 0, "conv_igemm: bad output channels");
  if (a.shuffle) SEGK_REQUIRE(a.Ntot == 4 * a.CO1 && !a.out2 && geo == 1, "conv_igemm: pixel-shuffle needs N=4*Cout");
  else SEGK_REQUIRE(a.Ntot == a.CO1 + a.CO2 && (a.CO2 == 0) == (a.out2 == nullptr), "conv_igemm: N != CO1+CO2");
  SEGK_REQUIRE((long long)a.B * a.H * a.W * 4 < 2147483647LL, "conv_igemm: pixel index overflows int32");
  if (dtype == SEGK_DT_BF16) return geo == 0 ? launch_geo<bf16_t, 0>(a, st) : launch_geo<bf16_t, 1>(a, st);
  return geo == 0 ? launch_geo<float, 0>(a, st) : launch_geo<float, 1>(a, st);
}
