// Producer/consumer bf16 GEMM on CDNA4 matrix cores (gfx950) for the 1x1 geometry with a long K:
//   plain      out[M][N] = A[M][K] . W (+ bias, optional quick_gelu)      -- the CLIP ViT nn.Linear / patch projection
//                                                                            GEMMs (segk_linear) and Conv2d 1x1
//                                                                            (clip/clipunet.py:84,122)
//   shuffle    ConvTranspose2d(k=2,s=2) forward (unet/unet.py:59, clip/clipunet.py:83): N = 4*Cout, the tile is
//              stored pixel-shuffled into [B,2H,2W,Cout]
//   unshuffle  its data gradient: K = 4*Cout gathered from the 2x2 output pixels of every input pixel
// against the packed weights [K/32][N][32] of segk_pack_conv_weight / segk_pack_convt_weight.
//
// The generic streaming kernel (conv_igemm.hip) joins all waves at a barrier every 32 K-elements (8 MFMAs per wave);
// with K = 768..3072 that leaves the matrix pipe idle most of the time (212 TFLOP/s on the ViT GEMMs, 386 on the
// ConvTranspose GEMMs).  Here a 512-thread workgroup owns a 256 x 128 output tile; 4 CONSUMER waves (one per SIMD,
// 128 x 64 each: 8 accumulator fragments) issue only ds_read_b128 + MFMA 32x32x16 -- 32 MFMAs between barriers -- and
// 4 PRODUCER waves move the next 64-element K-stage global -> VGPR -> LDS one stage ahead (two LDS stage slots,
// 80-byte row pitch per 64-byte chunk: the conflict-free fragment layout of the convolution kernels).  Persistent
// over work units, contiguous unit ranges per XCD; consecutive units share their A rows through that XCD's L2.
// The epilogue stages the fp32 accumulators through LDS (bias / activation applied) and stores 16-byte coalesced rows.
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"
#include "segk_internal.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int G_PIX = 80;                 // LDS pitch of one row's 64-byte K-chunk
constexpr int G_BM = 256, G_BN = 128;
constexpr int G_ACH = G_BM * G_PIX;       // one chunk of the A tile
constexpr int G_BCH = G_BN * G_PIX;       // one chunk of the weight tile
constexpr int G_STAGE = 2 * G_ACH + 2 * G_BCH;   // a stage = two chunks (64 K-elements) of both
constexpr int G_MAINB = 2 * G_STAGE;
constexpr int G_OP = G_BN * 2 + 16;       // epilogue tile row pitch
static_assert(G_BM * G_OP <= G_MAINB, "epilogue tile overlays the stage slots");

template <int MODE>   // 0 plain, 1 pixel-shuffle store, 2 un-shuffle gather
__global__ __launch_bounds__(512) void gemm_pipe_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long M = g.M;
  const int N = g.N;
  const int NT = N / G_BN;
  const int MT = (int)((M + G_BM - 1) / G_BM);
  const int KS = g.ksplit > 1 ? g.ksplit : 1;   // split-K (plain mode): unit = (row tile, K split, channel tile)
  const int U = MT * NT * KS;
  const int xcd = blockIdx.x & 7, upx = (U + 7) >> 3;
  const int GW = gridDim.x >> 3;
  int u = xcd * upx + (blockIdx.x >> 3);
  const int u_end = min(U, (xcd + 1) * upx);
  if (u >= u_end) return;
  const int nst = (g.nchunks >> 1) / KS;   // stages per unit (>= 1)
  const int HW = g.H * g.W;

  // the part of the epilogue all 512 threads run: 16-byte coalesced stores of the staged 256 x 128 tile
  auto store_tile = [&](int uu) {
    const long m0 = (long)(uu / (NT * KS)) * G_BM;
    const int n0 = (uu % NT) * G_BN;
    bf16_t* const obase = (bf16_t*)g.out + (size_t)((uu / NT) % KS) * g.split_stride;   // this split's partial output
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + i * 512;
      const int m = idx >> 4, cc = idx & 15;
      const long gm = m0 + m;
      if (gm < M) {
        const uint4 v = *(const uint4*)(smem + m * G_OP + cc * 16);
        const int n = n0 + cc * 8;
        bf16_t* dst;
        if (MODE == 1) {   // pixel-shuffle store: channel n of the GEMM is (tap, co) of output pixel (2y + tap/2, 2x + tap%2)
          const int tap = n / g.Cout, co = n - tap * g.Cout;
          const long b = gm / HW;
          const int r = (int)(gm - b * HW), y = r / g.W, x = r - y * g.W;
          dst = (bf16_t*)g.out + ((((b * 2 * g.H + 2 * y + (tap >> 1)) * 2 * g.W) + 2 * x + (tap & 1)) * (long)g.Cout + co);
        } else {
          dst = obase + gm * (long)N + n;
        }
        *(uint4*)dst = v;
      }
    }
  };

  if (wave >= 4) {
    // =============================================== PRODUCERS ===============================================
    const int ptid = tid - 256;
    const int within = ptid & 7, pcs = within & 3, cch = within >> 2;
    const int prow = ptid >> 3;            // 0..31
    const int a_lds = cch * G_ACH + prow * G_PIX + pcs * 16;
    const int b_lds = 2 * G_ACH + cch * G_BCH + prow * G_PIX + pcs * 16;
    u32x4 ra[8], rb[4];
    long arow[8];                          // element offset of each of this thread's A rows (chunk 0, tap 0)
    auto unit_rows = [&](int uu) {
      const long m0 = (long)(uu / (NT * KS)) * G_BM;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        long m = m0 + prow + 32 * i;
        m = m < M ? m : M - 1;             // rows past the end read a valid row; they are never stored
        if (MODE == 2) {
          const long b = m / HW;
          const int r = (int)(m - b * HW), y = r / g.W, x = r - y * g.W;
          arow[i] = (((b * 2 * g.H + 2 * y) * 2 * g.W) + 2 * x) * (long)g.lda;
        } else {
          arow[i] = m * (long)g.lda;
        }
      }
    };
    auto load = [&](int uu, int s) {
      const int kc = 2 * (((uu / NT) % KS) * nst + s) + cch;
      long koff;
      if (MODE == 2) {
        const int tap = kc / g.nchA, cc = kc - tap * g.nchA;
        koff = ((long)(tap >> 1) * 2 * g.W + (tap & 1)) * g.lda + cc * 32 + pcs * 8;
      } else {
        koff = (long)kc * 32 + pcs * 8;
      }
      const bf16_t* const ab = (const bf16_t*)g.A + koff;
#pragma unroll
      for (int i = 0; i < 8; ++i) ra[i] = *(const u32x4*)(ab + arow[i]);
      const int n0 = (uu % NT) * G_BN;
      const char* const wb = g.w + ((size_t)kc * N + n0 + prow) * 64 + pcs * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i) rb[i] = *(const u32x4*)(wb + (size_t)i * 32 * 64);
    };
    auto store = [&](int slot) {
      char* const sb = smem + slot * G_STAGE;
#pragma unroll
      for (int i = 0; i < 8; ++i) *(u32x4*)(sb + a_lds + i * 32 * G_PIX) = ra[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) *(u32x4*)(sb + b_lds + i * 32 * G_PIX) = rb[i];
    };
    // prologue of the first unit
    unit_rows(u);
    load(u, 0);
    store(0);
    bool has_next = (u + GW) < u_end;
    if (nst > 1) load(u, 1);
    else if (has_next) { unit_rows(u + GW); load(u + GW, 0); }
    __syncthreads();                                   // B0
    for (;;) {
      const int un = u + GW;
      has_next = un < u_end;
      for (int s = 0; s < nst; ++s) {
        if (s + 1 < nst) {
          store((s + 1) & 1);
          if (s + 2 < nst) load(u, s + 2);
          else if (has_next) { unit_rows(un); load(un, 0); }
        }
        __syncthreads();
      }
      __syncthreads();                                 // E1: consumers staged the output tile
      store_tile(u);
      if (!has_next) break;
      __syncthreads();                                 // E2: tile consumed, the stage slots may be rewritten
      store(0);                                        // next unit's stage 0 (held in registers since the last steps)
      if (nst > 1) load(un, 1);
      else if (un + GW < u_end) { unit_rows(un + GW); load(un + GW, 0); }
      u = un;
      __syncthreads();                                 // E3
    }
    return;
  }

  // ================================================= CONSUMERS =================================================
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  int laneA[4], laneB[2];
#pragma unroll
  for (int mf = 0; mf < 4; ++mf) laneA[mf] = ((wm * 4 + mf) * 32 + lr) * G_PIX + lh * 16;
#pragma unroll
  for (int nf = 0; nf < 2; ++nf) laneB[nf] = 2 * G_ACH + ((wn * 2 + nf) * 32 + lr) * G_PIX + lh * 16;
  f32x16 acc[4][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f;
  };
  uint4 fa[2][4], fb[2][2];
  auto rd = [&](const char* sb, int i, uint4 (&A)[4], uint4 (&Bf)[2]) {
    const int c = i >> 1, kk = i & 1;
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) A[mf] = *(const uint4*)(sb + laneA[mf] + c * G_ACH + kk * 32);
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) Bf[nf] = *(const uint4*)(sb + laneB[nf] + c * G_BCH + kk * 32);
  };
  zero_acc();
  __syncthreads();                                     // B0
  for (;;) {
    const int un = u + GW;
    const bool has_next = un < u_end;
    for (int s = 0; s < nst; ++s) {
      const char* const sb = smem + (s & 1) * G_STAGE;
      rd(sb, 0, fa[0], fb[0]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i + 1 < 4) rd(sb, i + 1, fa[(i + 1) & 1], fb[(i + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
          for (int nf = 0; nf < 2; ++nf)
            acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i & 1][mf]),
                                                                 __builtin_bit_cast(bf16x8, fb[i & 1][nf]), acc[mf][nf], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
    // ---- epilogue: bias / activation on the fp32 accumulators, tile -> LDS (the stage slots are dead here)
    {
      const int n0 = (u % NT) * G_BN;
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        const int n = (wn * 2 + nf) * 32 + lr;
        const float bv = (g.bias && (u / NT) % KS == 0) ? g.bias[n0 + n] : 0.f;   // split-K: the bias rides on split 0
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
          const int mb = (wm * 4 + mf) * 32 + 4 * lh;
          char* const obase = smem + mb * G_OP + n * 2;
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            v[r] = acc[mf][nf][r] + bv;
            if (g.act) v[r] = v[r] / (1.f + __expf(-1.702f * v[r]));      // quick_gelu (CLIPMLP)
          }
          float s1 = 0.f, s2 = 0.f;
          stage_frag<bf16_t>(v, obase, G_OP, s1, s2);
        }
      }
    }
    zero_acc();
    __syncthreads();                                   // E1
    store_tile(u);
    if (!has_next) break;
    __syncthreads();                                   // E2
    u = un;
    __syncthreads();                                   // E3
  }
}


// ------------------------------------------------------------------------------------------------------------------------
// Round 4: the same GEMM with LDS-DMA producers, 16x16x32 consumers and results straight from the accumulators -- the form
// the 3x3 producer/consumer convolution took this round (conv_igemm.hip, DMA form), for the same reasons, plus two that are
// specific to a plain GEMM:
//   * it is STAGING-bound: a 256 x 128 tile needs 48 KiB per 64 K-elements = 1024 matrix cycles, i.e. 12 global loads and 12
//     ds_write_b128 per producer wave and stage in the register-staged kernel above (the convolution re-uses a patch for nine
//     taps and needs a third of that).  By LDS-DMA it is 12 instructions, none of which waits for data;
//   * the consumers above start every stage by reading its first fragments (nothing is pre-read across the barrier: two stage
//     slots) and read in groups.  Here the ring is five or six slots of one 32-element chunk each ((BM + 128) x 64 B, unpadded rows with
//     the 16-byte piece XOR-ed by [0,3,2,1][(row >> 2) & 3]: conflict-free for plain and for permuted 16-row operand reads):
//     chunk q + 2 has landed one barrier before it is needed, so the next chunk's fragments are read between this chunk's MFMAs;
//   * BM = 320 (wave tile 160 x 64): M = 3152 rows (B = 16 x 197 tokens) are 10 row tiles instead of 13, so the CLIP MLP's fc1
//     (N = 3072: 24 column tiles) is 240 units = ONE round on 256 CUs instead of 312 = two.
// One barrier per chunk.  MFMA orientation: channels x rows (A = weights with the rows a lane reads
// permuted so that accumulator rows 4 lq + j of a block pair are 8 consecutive output channels): 16-byte stores from
// registers, bias / quick_gelu on the way, no epilogue tile, no boundary barriers -- the ring runs through unit boundaries.
typedef __attribute__((address_space(3))) void* g_lds_vp;
typedef const __attribute__((address_space(1))) void* g_glb_vp;
typedef __attribute__((ext_vector_type(4))) float g_f32x4;

template <int MODE, int BM>   // 0 plain, 1 pixel-shuffle store, 2 un-shuffle gather
__global__ __launch_bounds__(512, 2) void gemm_dma_kernel(const GemmArgs g) {
  constexpr int BN = 128;
  constexpr int SLOT = (BM + BN) * 64;          // one chunk: A rows then weight rows
  constexpr int MBW = BM / 32;                  // 16-row blocks per consumer wave (wave tile BM/2 x 64)
  constexpr int G = MBW / 2;                    // blocks per sub-step
  constexpr int NA = BM / 64;                   // A pieces (16 rows) per producer wave and chunk
  constexpr int NPC = NA + 2;                   // pieces per producer wave and chunk (two weight pieces)
  // Ring: one barrier per chunk.  During chunk q the consumers read slot q and pre-read the first fragments of chunk q + 1;
  // chunks <= q + 2 have landed at the barrier that ends it, and the youngest SLACK = NSLOT - 3 chunks stay in flight across
  // it (BM = 256: six slots, three chunks = ~1 us of matrix work in flight; BM = 320: five slots, two).  A first form with
  // two-chunk stages left zero / one chunk in flight and ran at the DMA latency, not at the matrix rate (fc1 28.8 us).
  constexpr int GD_NSLOT = (6 * SLOT <= 160 * 1024) ? 6 : 5;
  constexpr int SLACK = GD_NSLOT - 3;
  static_assert(BM % 64 == 0 && GD_NSLOT * SLOT <= 160 * 1024, "tile / ring geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long M = g.M;
  const int N = g.N;
  const int NT = N / BN;
  const int MT = (int)((M + BM - 1) / BM);
  const int KS = g.ksplit > 1 ? g.ksplit : 1;
  const int U = MT * NT * KS;
  const int xcd = blockIdx.x & 7, upx = (U + 7) >> 3;
  const int GW = gridDim.x >> 3;
  int u = xcd * upx + (blockIdx.x >> 3);
  const int u_end = min(U, (xcd + 1) * upx);
  if (u >= u_end) return;
  const int nck = g.nchunks / KS;               // chunks per unit (even, >= 2)
  const int HW = g.H * g.W;

  if (wave >= 4) {
    // ================================================ PRODUCERS ================================================
    const int pw = wave - 4;
    const int lrow = lane >> 2;
    const unsigned lpc = (unsigned)(((lane & 3) ^ ((0x1230 >> (4 * ((lane >> 4) & 3))) & 3)) << 4);   // global piece of the lane's LDS slot
    const unsigned wl_off = (unsigned)(lrow * 64) + lpc;
    unsigned arow[NA];                           // byte offset of the lane's row of each A piece (unit of the fetch cursor)
    int cu = u, cc = 0;                          // fetch cursor: unit, chunk within it
    int cn0 = 0, ck0 = 0;                        // that unit's first weight row and first chunk
    bool cok = true;
    auto cursor_unit = [&]() {
      const long m0 = (long)(cu / (NT * KS)) * BM;
      cn0 = (cu % NT) * BN;
      ck0 = ((cu / NT) % KS) * nck;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        long m = m0 + (pw + 4 * i) * 16 + lrow;
        m = m < M ? m : M - 1;                   // rows past the end read a valid row; they are never stored
        if (MODE == 2) {
          const long b = m / HW;
          const int r = (int)(m - b * HW), y = r / g.W, x = r - y * g.W;
          arow[i] = (unsigned)(((((b * 2 * g.H + 2 * y) * 2 * g.W) + 2 * x) * (long)g.lda) * 2);
        } else {
          arow[i] = (unsigned)((m * (long)g.lda) * 2);
        }
      }
    };
    // the next chunk of the stream -> ring slot `slot`; false past the last unit
    auto issue_chunk = [&](int slot) {
      if (!cok) return false;
      const int kc = ck0 + cc;
      long koff;                                 // elements
      if (MODE == 2) {
        const int tap = kc / g.nchA, c2 = kc - tap * g.nchA;
        koff = ((long)(tap >> 1) * 2 * g.W + (tap & 1)) * g.lda + c2 * 32;
      } else {
        koff = (long)kc * 32;
      }
      const char* const ab = (const char*)g.A + koff * 2 + lpc;
      char* const sb = smem + slot * SLOT;
#pragma unroll
      for (int i = 0; i < NA; ++i)
        __builtin_amdgcn_global_load_lds((g_glb_vp)(ab + arow[i]), (g_lds_vp)(sb + (pw + 4 * i) * 1024), 16, 0, 0);
      const char* const wb = g.w + ((size_t)kc * N + cn0) * 64 + wl_off;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_global_load_lds((g_glb_vp)(wb + (pw + 4 * i) * 1024), (g_lds_vp)(sb + BM * 64 + (pw + 4 * i) * 1024), 16, 0, 0);
      if (++cc == nck) {
        cc = 0;
        cu += GW;
        cok = cu < u_end;
        if (cok) cursor_unit();
      }
      return true;
    };
    cursor_unit();
    int q = 0;                                    // stream chunks issued so far
    bool all = true;
#pragma unroll
    for (int i = 0; i < 2 + SLACK; ++i) { all = issue_chunk(q % GD_NSLOT) && all; ++q; }
    if (all) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SLACK * NPC) : "memory");     // chunks 0 and 1 have landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                               // B0
    for (;;) {
      const int un = u + GW;
      const bool has_next = un < u_end;
      for (int c = 0; c < nck; ++c) {
        // the next chunk of the stream into the slot the previous stage released; all but the youngest SLACK chunks have
        // landed at the barrier
        const bool a1 = issue_chunk(q % GD_NSLOT);
        ++q;
        if (a1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SLACK * NPC) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      if (!has_next) break;
      u = un;
    }
    return;
  }

  // ================================================= CONSUMERS =================================================
  const int wm = wave >> 1, wn = wave & 1;
  const int lc = lane & 15, lq = lane >> 4;
  // activation rows: block mb of the wave = rows (wm * MBW + mb) * 16 + lc of the tile: one per-lane address + immediates
  const int laneX = (wm * MBW * 16 + lc) * 64 + ((lq ^ ((0x1230 >> (4 * ((lc >> 2) & 3))) & 3)) << 4);
  int laneW[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int row = (wn * 4 + (nb & ~1)) * 16 + 8 * (lc >> 2) + 4 * (nb & 1) + (lc & 3);   // channel 8 (lc >> 2) + 4 (nb & 1) + (lc & 3) of the pair's 32
    laneW[nb] = BM * 64 + row * 64 + ((lq ^ ((0x1230 >> (4 * ((row >> 2) & 3))) & 3)) << 4);
  }
  g_f32x4 acc[MBW][4];
  auto zero_acc = [&]() {
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = (g_f32x4){0.f, 0.f, 0.f, 0.f};
  };
  uint4 fx[2][G], fw[2][4];
  auto rdX = [&](int sb, int grp, int j) { return *(const uint4*)(smem + sb + laneX + (grp * G + j) * 1024); };
  auto rdW = [&](int sb, int nb) { return *(const uint4*)(smem + sb + laneW[nb]); };
  // one chunk (K = 32): two sub-steps of G x 4 MFMAs; the fragments of the next sub-step / next chunk are read between them
  // (one ds_read_b128 behind every second MFMA).  par = parity of the chunk in the stream (fw buffer)
  auto chunk = [&](int sb, int sb_next, auto PARc) {
    constexpr int par = decltype(PARc)::value;
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
#pragma unroll
      for (int j = 0; j < G; ++j)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          acc[grp * G + j][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fw[par][nb]),
                                                                         __builtin_bit_cast(bf16x8, fx[grp][j]), acc[grp * G + j][nb], 0, 0, 0);
          const int m = j * 4 + nb;
          if (m & 1) {
            const int r = m >> 1;
            if (grp == 0) {
              if (r < G) fx[1][r] = rdX(sb, 1, r);                       // this chunk's second half
            } else {
              if (r < 4) fw[par ^ 1][r] = rdW(sb_next, r);               // next chunk: weights first, then its first half
              else if (r - 4 < G) fx[0][r - 4] = rdX(sb_next, 0, r - 4);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
    }
  };
  // the unit's results straight from the accumulators: acc[mb][2k][j] / acc[mb][2k + 1][j] = channels 32 k + 8 lq + j / + 4 + j
  // (of the wave's 64) of row 16 mb + lc (of the wave's BM / 2)
  auto direct_out = [&](int uu) {
    const long m0 = (long)(uu / (NT * KS)) * BM + wm * (BM / 2) + lc;
    const int n0 = (uu % NT) * BN + wn * 64;
    const int split = (uu / NT) % KS;
    bf16_t* const obase = (bf16_t*)g.out + (size_t)split * g.split_stride;
    // both 32-channel groups of a row block back to back: the two 64-byte halves of a 128-byte output line reach L2 within one
    // store pair (eight stores apart they left L2 half-written: +20-36 % HBM write bytes in the conv kernel's same epilogue)
    float bv[2][8];
    int tapk[2] = {0, 0}, cok[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int n8 = n0 + 32 * k + 8 * lq;
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[k][e] = (g.bias && split == 0) ? g.bias[n8 + e] : 0.f;   // split-K: the bias rides on split 0
      cok[k] = n8;
      if (MODE == 1) { tapk[k] = n8 / g.Cout; cok[k] = n8 - tapk[k] * g.Cout; }
    }
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) {
      const long gm = m0 + mb * 16;
      long pbase = 0;
      if (MODE == 1) {   // pixel-shuffle store: channel n of the GEMM is (tap, co) of output pixel (2y + tap/2, 2x + tap%2)
        const long b = gm / HW;
        const int r = (int)(gm - b * HW), y = r / g.W, x = r - y * g.W;
        pbase = ((b * 2 * g.H + 2 * y) * 2 * g.W) + 2 * x;
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] = acc[mb][2 * k + (e >> 2)][e & 3] + bv[k][e];
          if (g.act) v[e] = v[e] / (1.f + __expf(-1.702f * v[e]));       // quick_gelu (CLIPMLP)
        }
        const uint4 o = make_uint4(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7]));
        if (gm < M) {
          bf16_t* dst;
          if (MODE == 1) dst = (bf16_t*)g.out + ((pbase + (long)(tapk[k] >> 1) * 2 * g.W + (tapk[k] & 1)) * (long)g.Cout + cok[k]);
          else dst = obase + gm * (long)N + cok[k];
          *(uint4*)dst = o;
        }
      }
    }
  };

  zero_acc();
  __syncthreads();                                   // B0
  int q = 0;
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) fw[0][nb] = rdW(0, nb);
#pragma unroll
  for (int j = 0; j < G; ++j) fx[0][j] = rdX(0, 0, j);
  for (;;) {
    const int un = u + GW;
    const bool has_next = un < u_end;
    for (int c = 0; c < nck; c += 2) {
      const int s0 = (q % GD_NSLOT) * SLOT, s1 = ((q + 1) % GD_NSLOT) * SLOT, s2 = ((q + 2) % GD_NSLOT) * SLOT;
      chunk(s0, s1, std::integral_constant<int, 0>{});
      __syncthreads();
      chunk(s1, s2, std::integral_constant<int, 1>{});
      q += 2;
      __syncthreads();
    }
    direct_out(u);
    if (!has_next) break;
    zero_acc();
    u = un;
  }
}

static int g_num_cus() { return segk_num_cus(); }

template <int MODE, int BM>
int launch_dma(const GemmArgs& g, hipStream_t st) {
  auto kern = gemm_dma_kernel<MODE, BM>;
  static bool attr_set[SEGK_MAX_DEVICES] = {};     // per device: the attribute is device state
  const int dev_ = segk_device_index();
  if (!attr_set[dev_]) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      SEGK_FAIL(-3, "gemm_dma: cannot raise dynamic LDS limit");
    attr_set[dev_] = true;
  }
  const long U = ((g.M + BM - 1) / BM) * (g.N / 128) * (g.ksplit > 1 ? g.ksplit : 1);
  const int per_xcd = (int)((U + 7) / 8);
  int gw = g_num_cus() / 8;
  if (gw > per_xcd) gw = per_xcd;
  constexpr int SLOT = (BM + 128) * 64, NSLOT = (6 * SLOT <= 160 * 1024) ? 6 : 5;
  hipLaunchKernelGGL(kern, dim3(8 * gw), dim3(512), NSLOT * SLOT, st, g);
  SEGK_CHECK_LAUNCH("gemm_dma");
  return 0;
}

// rounds of `units` work units on the chip's CUs, in units of one 256-row tile's time
static double gemm_rounds(long M, int N, int ks, int bm) {
  const long units = ((M + bm - 1) / bm) * (long)(N / 128) * ks;
  const long cus = g_num_cus();
  return (double)((units + cus - 1) / cus) * bm / 256.0;
}

template <int MODE>
int launch_mode(const GemmArgs& g, hipStream_t st) {
  // LDS-DMA form (round 4) unless SEGK_GEMM_DMA=0 (A/B runs) or a source is too large for 32-bit byte offsets per DMA lane;
  // 320-row tiles where they need fewer rounds on the chip than 256-row tiles (M = 3152: fc1 240 units instead of 312)
  static const char* const nodma = getenv("SEGK_GEMM_DMA");
  const long arows = MODE == 2 ? 4 * g.M : g.M;
  if (!(nodma && nodma[0] == '0') && arows * (long)g.lda * 2 < 4294967296L && (g.nchunks / (g.ksplit > 1 ? g.ksplit : 1)) % 2 == 0) {
    const int ks = g.ksplit > 1 ? g.ksplit : 1;
    static const char* const bmf = getenv("SEGK_GEMM_BM");     // "256" / "320": force a tile height (A/B runs)
    const bool use320 = bmf ? (bmf[0] == '3') : gemm_rounds(g.M, g.N, ks, 320) < gemm_rounds(g.M, g.N, ks, 256);
    return use320 ? launch_dma<MODE, 320>(g, st) : launch_dma<MODE, 256>(g, st);
  }
  auto kern = gemm_pipe_kernel<MODE>;
  static bool attr_set[SEGK_MAX_DEVICES] = {};     // per device: the attribute is device state
  const int dev_ = segk_device_index();
  if (!attr_set[dev_]) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      SEGK_FAIL(-3, "gemm_pipe: cannot raise dynamic LDS limit");
    attr_set[dev_] = true;
  }
  const long U = ((g.M + G_BM - 1) / G_BM) * (g.N / G_BN) * (g.ksplit > 1 ? g.ksplit : 1);
  const int per_xcd = (int)((U + 7) / 8);
  int gw = g_num_cus() / 8;
  if (gw > per_xcd) gw = per_xcd;
  hipLaunchKernelGGL(kern, dim3(8 * gw), dim3(512), G_MAINB, st, g);
  SEGK_CHECK_LAUNCH("gemm_pipe");
  return 0;
}

}  // namespace

// does the producer/consumer GEMM serve this 1x1-geometry call?  (bf16; whole 64-element K stages that do not
// straddle ConvTranspose taps; 128-wide channel tiles; enough rows to fill a 256-row tile)
// Short-K problems are bound by the unit boundary (epilogue + first-stage latency with one workgroup per CU) and stay
// on the generic kernel (two 4-wave workgroups per CU): threshold in 32-element chunks, per mode; the environment
// variable SEGK_GEMM_PIPE_MIN_CHUNKS overrides all three (kernel A/B measurements: tools/kbench.py convt).
static int min_chunks(int mode) {
  static int env = -2;
  if (env == -2) {
    const char* e = getenv("SEGK_GEMM_PIPE_MIN_CHUNKS");
    env = e ? atoi(e) : -1;
  }
  if (env >= 0) return env;
  // same-box A/B at the U-Net's ConvTranspose shapes (B = 32): pixel-shuffle store K = 1024 / 512 / 256 / 128:
  // 55 / 71 / 93 / 151 us here against 66 / 74 / 83 / 139 us generic; un-shuffle gather K = 2048 / 1024 / 512 / 256:
  // 51 / 54 / 64 / 100 against 69 / 70 / 72 / 110
  return mode == 1 ? 16 : 8;
}

int segk_gemm_pipe_ok(long M, int nchunks, int nchA, int N, int cout_shuffle, int mode) {
  if (nchunks < 2 || nchunks < min_chunks(mode) || (nchunks & 1) || N % G_BN != 0 || M < 128) return 0;
  if (mode == 2 && (nchA & 1)) return 0;
  if (mode == 1 && (cout_shuffle % 8 != 0)) return 0;
  return 1;
}

int segk_gemm_pipe_launch(const GemmArgs& g, int mode, hipStream_t st) {
  SEGK_REQUIRE(g.A && g.w && g.out && g.M > 0, "gemm_pipe: null pointer / empty problem");
  SEGK_REQUIRE(segk_gemm_pipe_ok(g.M, g.nchunks, g.nchA, g.N, g.Cout, mode), "gemm_pipe: unsupported shape");
  SEGK_REQUIRE(g.M * 4 < 2147483647LL * 64, "gemm_pipe: row count too large");
  SEGK_REQUIRE(g.ksplit <= 1 || (mode == 0 && !g.act && (g.nchunks >> 1) % g.ksplit == 0 && g.split_stride >= g.M * (long)g.N),
               "gemm_pipe: split-K needs the plain mode, no activation and a K that splits into whole 64-element stages");
  if (mode == 1) return launch_mode<1>(g, st);
  if (mode == 2) return launch_mode<2>(g, st);
  return launch_mode<0>(g, st);
}
