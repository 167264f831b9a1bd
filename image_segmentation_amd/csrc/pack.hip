// Layout conversion kernels: NCHW fp32 <-> NHWC (channel-padded, fp32/bf16) activations, and
// reference-layout (OIHW / IOHW fp32) parameters <-> the K-chunked MFMA weight layout
// [K/CH][taps][N][CH] used by conv_igemm.hip and the [N][taps][K] fp32 gradient slabs of wgrad.hip.
// All tiny, bandwidth-trivial; one thread per destination element.
#include "common.hpp"
#include "segk_internal.h"
#include "../../include/segk.h"

namespace {

// one thread per (pixel, 16-byte channel group): plane reads are coalesced across consecutive pixels, the store is
// one 16-byte piece of the pixel's NHWC row
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int C, int H, int W,
                                    int Cp) {
  constexpr int VEC = ET<T>::VEC;
  const int CV = Cp / VEC;
  const long hw = (long)H * W;
  const long total = (long)B * hw * CV;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int g = (int)(i % CV);
    const long p = i / CV;
    const int b = (int)(p / hw);
    const long r = p - (long)b * hw;
    float f[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = g * VEC + j;
      f[j] = c < C ? src[((long)b * C + c) * hw + r] : 0.f;
    }
    *(uint4*)(dst + p * Cp + g * VEC) = pack16<T>(f);
  }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int B, int C, int H, int W,
                                    int Cp) {
  const long total = (long)B * C * H * W;
  const long hw = (long)H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i % hw;
    const long bc = i / hw;
    const int c = (int)(bc % C), b = (int)(bc / C);
    dst[i] = to_float<T>(src[((long)b * hw + r) * Cp + c]);
  }
}

// map a padded dual-source channel index to the logical reference channel (or -1 for padding)
__device__ __forceinline__ int dual_map(int kp, int CA, int CAp, int CB, int CBp) {
  if (kp < CAp) return kp < CA ? kp : -1;
  const int k = kp - CAp;
  return k < CB ? CA + k : -1;
}

// Conv2d weight [Cout][Cin][taps] (OIHW) -> packed.
//  mode 0 (forward):   dst[kc][tap][n][j] = W[n][ci(kc*CH+j)][tap]                   K = input channels
//  mode 1 (data grad): dst[kc][tap][n][j] = W[co = kc*CH+j][ci(n)][taps-1-tap]        K = output channels
template <typename T>
__device__ __forceinline__ T conv_packed_value(const float* __restrict__ w, long i, int Cout, int CA, int CB, int Coutp, int CAp,
                                              int CBp, int taps, int mode) {
  constexpr int CH = ET<T>::CH;
  const int Cin = CA + CB, Cinp = CAp + CBp;
  const int Np = mode == 0 ? Coutp : Cinp;
  const int j = (int)(i % CH);
  long r = i / CH;
  const int n = (int)(r % Np); r /= Np;
  const int tap = (int)(r % taps);
  const int kc = (int)(r / taps);
  const int k = kc * CH + j;
  float v = 0.f;
  if (mode == 0) {
    const int ci = dual_map(k, CA, CAp, CB, CBp);
    if (ci >= 0 && n < Cout) v = w[((long)n * Cin + ci) * taps + tap];
  } else {
    const int ci = dual_map(n, CA, CAp, CB, CBp);
    if (ci >= 0 && k < Cout) v = w[((long)k * Cin + ci) * taps + (taps - 1 - tap)];
  }
  return from_float<T>(v);
}
template <typename T>
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, T* __restrict__ dst, int Cout, int CA, int CB,
                                        int Coutp, int CAp, int CBp, int taps, int mode) {
  const int Cinp = CAp + CBp;
  const int Kp = mode == 0 ? Cinp : Coutp, Np = mode == 0 ? Coutp : Cinp;
  const long total = (long)Kp * taps * Np;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256)
    dst[i] = conv_packed_value<T>(w, i, Cout, CA, CB, Coutp, CAp, CBp, taps, mode);
}

// Both packed layouts of a 3x3 weight in ONE pass (they are re-made after every optimizer step: 35 + 8 packs per U-Net
// step otherwise).  A block owns a 32 (output channels) x 32 (padded input channels) x 9 tile: coalesced fp32 reads
// into LDS, then per tap one contiguous 32 x 32 block of the forward layout and one of the data-gradient layout.
template <typename T>
__device__ __forceinline__ void pack_conv3x3_both_block(const float* __restrict__ w, T* __restrict__ df, T* __restrict__ dd,
                                                        int Cout, int CA, int CB, int Coutp, int CAp, int CBp, int bx, int by,
                                                        float (&tile)[32][32 * 9 + 1]) {
  using E = ET<T>;
  constexpr int CH = E::CH, VEC = E::VEC, VPR = 32 / VEC;     // 16-byte vectors per 32-element row
  const int Cin = CA + CB, Cinp = CAp + CBp;
  const int kp0 = bx * 32, co0 = by * 32;
  const int ci0 = dual_map(kp0, CA, CAp, CB, CBp), ci31 = dual_map(kp0 + 31, CA, CAp, CB, CBp);
  if (ci0 >= 0 && ci31 == ci0 + 31 && co0 + 32 <= Cout && (Cin & 3) == 0 && (ci0 & 3) == 0) {
    // whole tile inside the parameter: every output channel's 32 x 9 floats are contiguous and 16-byte aligned
    for (int idx = threadIdx.x; idx < 32 * 72; idx += 256) {
      const int j = idx / 72, q = idx - j * 72;
      const float4 v = *(const float4*)(w + ((long)(co0 + j) * Cin + ci0) * 9 + q * 4);
      float* t = &tile[j][q * 4];
      t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    }
  } else {
    for (int idx = threadIdx.x; idx < 32 * 288; idx += 256) {
      const int j = idx / 288, r = idx - j * 288, kk = r / 9;
      const int ci = dual_map(kp0 + kk, CA, CAp, CB, CBp);
      float v = 0.f;
      if (ci >= 0 && co0 + j < Cout) v = w[((long)(co0 + j) * Cin + ci) * 9 + (r - kk * 9)];
      tile[j][r] = v;
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 9 * 32 * VPR; idx += 256) {
    const int tap = idx / (32 * VPR), r = idx - tap * (32 * VPR);
    float f[VEC];
    {   // forward: K = input channels (kp), N = output channels; VEC consecutive kp of one output channel
      const int j = r / VPR, kk = (r - j * VPR) * VEC, kp = kp0 + kk;
#pragma unroll
      for (int e = 0; e < VEC; ++e) f[e] = tile[j][(kk + e) * 9 + tap];
      *(uint4*)(df + (((long)(kp / CH) * 9 + tap) * Coutp + co0 + j) * CH + kp % CH) = pack16<T>(f);
    }
    if (dd) {   // data gradient: K = output channels, N = input channels, taps flipped; VEC consecutive co of one kp
      const int kk = r / VPR, j = (r - kk * VPR) * VEC, co = co0 + j;
#pragma unroll
      for (int e = 0; e < VEC; ++e) f[e] = tile[j + e][kk * 9 + (8 - tap)];
      *(uint4*)(dd + (((long)(co / CH) * 9 + tap) * Cinp + kp0 + kk) * CH + co % CH) = pack16<T>(f);
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void pack_conv3x3_both_kernel(const float* __restrict__ w, T* __restrict__ df,
                                                                T* __restrict__ dd, int Cout, int CA, int CB, int Coutp,
                                                                int CAp, int CBp) {
  __shared__ float tile[32][32 * 9 + 1];
  pack_conv3x3_both_block<T>(w, df, dd, Cout, CA, CB, Coutp, CAp, CBp, blockIdx.x, blockIdx.y, tile);
}
// ConvTranspose2d(k=2,s=2) weight [Cin][Cout][2][2] -> packed.
//  mode 0 (forward GEMM, N = q*Coutp + co, K = ci):         dst[kc][0][n][j] = W[ci=kc*CH+j][co][q]
//  mode 1 (data grad, K = q*Coutp + co via un-shuffle):     dst[kc][0][n=ci][j] = W[ci][co][q], kc = q*(Coutp/CH)+cc
template <typename T>
__device__ __forceinline__ T convt_packed_value(const float* __restrict__ w, long i, int Cin, int Cout, int Cinp, int Coutp,
                                               int mode);
template <typename T>
__global__ void pack_convt_weight_kernel(const float* __restrict__ w, T* __restrict__ dst, int Cin, int Cout, int Cinp,
                                         int Coutp, int mode) {
  const long total = (long)Cinp * 4 * Coutp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256)
    dst[i] = convt_packed_value<T>(w, i, Cin, Cout, Cinp, Coutp, mode);
}

template <typename T>
__device__ __forceinline__ T convt_packed_value(const float* __restrict__ w, long i, int Cin, int Cout, int Cinp, int Coutp,
                                               int mode) {
  constexpr int CH = ET<T>::CH;
  const int Np = mode == 0 ? 4 * Coutp : Cinp;
  const int j = (int)(i % CH);
  long r = i / CH;
  const int n = (int)(r % Np);
  const int kc = (int)(r / Np);
  int ci, co, q;
  if (mode == 0) { ci = kc * CH + j; q = n / Coutp; co = n - q * Coutp; }
  else { const int ncc = Coutp / CH; q = kc / ncc; co = (kc - q * ncc) * CH + j; ci = n; }
  float v = 0.f;
  if (ci < Cin && co < Cout) v = w[((long)ci * Cout + co) * 4 + q];
  return from_float<T>(v);
}

// Every packed copy a model refreshes after an optimizer step in ONE launch (segk_pack_multi): `table` lists the tensors, a
// block finds its tensor by its first block index (entries sorted, <= 64 of them) and runs that tensor's routine.
constexpr int PACK_CONVT_CHUNK = 2048;     // elements of a ConvTranspose weight per block
template <typename T>
__global__ __launch_bounds__(256) void pack_multi_kernel(const SegkPackEntry* __restrict__ table, int n) {
  __shared__ float tile[32][32 * 9 + 1];
  int e = 0;
  for (int i = 1; i < n; ++i) e = ((int)blockIdx.x >= table[i].block0) ? i : e;     // scalar loop (uniform)
  const SegkPackEntry t = table[e];
  const int lid = blockIdx.x - t.block0;
  if (t.kind == 0) {
    const int gx = (t.CAp + t.CBp) / 32;
    pack_conv3x3_both_block<T>(t.w, (T*)t.dst_fwd, (T*)t.dst_dgrad, t.Cout, t.CA, t.CB, t.Coutp, t.CAp, t.CBp, lid % gx, lid / gx,
                               tile);
  } else if (t.kind == 1) {
    const long total = (long)t.CAp * 4 * t.Coutp;
    const long i0 = (long)lid * PACK_CONVT_CHUNK;
    for (long i = i0 + threadIdx.x; i < i0 + PACK_CONVT_CHUNK && i < total; i += 256) {
      ((T*)t.dst_fwd)[i] = convt_packed_value<T>(t.w, i, t.CA, t.Cout, t.CAp, t.Coutp, 0);
      if (t.dst_dgrad) ((T*)t.dst_dgrad)[i] = convt_packed_value<T>(t.w, i, t.CA, t.Cout, t.CAp, t.Coutp, 1);
    }
  } else if (t.kind == 3) {                        // Conv2d 1x1 weight [Cout][CA]: forward (and data-gradient) layout, taps = 1
    const long total = (long)t.CAp * t.Coutp;
    const long i0 = (long)lid * PACK_CONVT_CHUNK;
    for (long i = i0 + threadIdx.x; i < i0 + PACK_CONVT_CHUNK && i < total; i += 256) {
      ((T*)t.dst_fwd)[i] = conv_packed_value<T>(t.w, i, t.Cout, t.CA, 0, t.Coutp, t.CAp, 0, 1, 0);
      if (t.dst_dgrad) ((T*)t.dst_dgrad)[i] = conv_packed_value<T>(t.w, i, t.Cout, t.CA, 0, t.Coutp, t.CAp, 0, 1, 1);
    }
  } else {                                         // kind 2: bias [Cout] -> fp32 [reps][Coutp], reps = CA (0 means 4)
    float* d = (float*)t.dst_fwd;
    const int reps = t.CA > 0 ? t.CA : 4;
    for (int i = threadIdx.x; i < reps * t.Coutp; i += 256) {
      const int c = i % t.Coutp;
      d[i] = c < t.Cout ? t.w[c] : 0.f;
    }
  }
}

// Sum S gradient slabs [S][Np][taps][Kp] (fp32) into the dense reference-layout gradient [N][K][taps]
// (OIHW for Conv2d, IOHW for ConvTranspose2d).  One block per output row n: the slab rows are read with
// coalesced float4 loads (k fastest), transposed through LDS, and the row is written as one contiguous run.
// One output row n of the gradient: sum the S slabs in slab order (bit-stable), transpose [tap][k] -> [k][tap] through LDS
__device__ __forceinline__ void wgrad_reduce_row(const float* __restrict__ slabs, int S, float* __restrict__ grad, int n, int CA,
                                                 int CB, int Np, int CAp, int CBp, int taps, float* row) {
  const int Kp = CAp + CBp, K = CA + CB;
  const long slab = (long)Np * taps * Kp;
  const float* base = slabs + (long)n * taps * Kp;
  const int nvec = taps * Kp / 4;                // Kp % 32 == 0
  for (int v = threadIdx.x; v < nvec; v += 256) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // eight slabs in flight (unconditional loads of a clamped slab index; left as a plain loop the compiler issues one load
    // and waits for it: S serial round trips per vector); added in slab order: bit-stable
    for (int t0 = 0; t0 < S; t0 += 8) {
      float4 x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = *(const float4*)(base + (long)(t0 + u < S ? t0 + u : S - 1) * slab + (long)v * 4);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (t0 + u < S) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
    }
    const int e = v * 4, tap = e / Kp, kp = e - tap * Kp;
    const float vals[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = kp + j;
      const int k = q < CAp ? (q < CA ? q : -1) : (q - CAp < CB ? CA + q - CAp : -1);
      if (k >= 0) row[k * taps + tap] = vals[j];
    }
  }
  __syncthreads();
  float* dst = grad + (long)n * K * taps;
  for (int i = threadIdx.x; i < K * taps; i += 256) dst[i] = row[i];
}
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, int S,
                                                           float* __restrict__ grad, int N, int CA, int CB, int Np,
                                                           int CAp, int CBp, int taps) {
  extern __shared__ float row[];                 // [K][taps] in output order
  wgrad_reduce_row(slabs, S, grad, blockIdx.x, CA, CB, Np, CAp, CBp, taps, row);
}

// The same sum for MANY thin slabs (narrow layers: split-K factors of 64 .. 512 over a gradient of a few hundred KB), in
// one launch: a block owns 16 float4 vectors of one output row and walks the slabs in 16 interleaved streams
// (thread = vector lane x slab lane), then adds the 16 stream sums in lane order (fixed order: bit-stable) and scatters the
// 64 values to their [k][tap] places.
__device__ __forceinline__ void wgrad_reduce_wide(const float* __restrict__ slabs, int S, float* __restrict__ grad, int n, int vy,
                                                  int CA, int CB, int Np, int CAp, int CBp, int taps, float4 (&red)[16][16]) {
  const int Kp = CAp + CBp, K = CA + CB;
  const int vl = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int nvec = taps * Kp / 4;
  const int v = vy * 16 + vl;
  const long slab = (long)Np * taps * Kp;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (v < nvec) {
    const float* base = slabs + (long)n * taps * Kp + (long)v * 4;
    // eight of the lane's slabs in flight (see wgrad_reduce_row); added in the old order: bit-stable
    for (int t0 = sl; t0 < S; t0 += 128) {
      float4 x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = *(const float4*)(base + (long)(t0 + 16 * u < S ? t0 + 16 * u : S - 1) * slab);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (t0 + 16 * u < S) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
    }
  }
  red[sl][vl] = acc;
  __syncthreads();
  if (sl != 0 || v >= nvec) return;
#pragma unroll
  for (int r = 1; r < 16; ++r) {
    const float4 x = red[r][vl];
    acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
  }
  const int e = v * 4, tap = e / Kp, kp = e - tap * Kp;
  const float vals[4] = {acc.x, acc.y, acc.z, acc.w};
  float* dst = grad + (long)n * K * taps;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = kp + j;
    const int k = q < CAp ? (q < CA ? q : -1) : (q - CAp < CB ? CA + q - CAp : -1);
    if (k >= 0) dst[k * taps + tap] = vals[j];
  }
}
__global__ __launch_bounds__(256) void wgrad_reduce_wide_kernel(const float* __restrict__ slabs, int S,
                                                                float* __restrict__ grad, int N, int CA, int CB, int Np,
                                                                int CAp, int CBp, int taps) {
  __shared__ float4 red[16][16];
  wgrad_reduce_wide(slabs, S, grad, blockIdx.x, blockIdx.y, CA, CB, Np, CAp, CBp, taps, red);
}

// Column sums of BatchNorm-style partial rows: out[c] = sum over rows of part[row][col0 + c][0] (stride 2 floats per
// channel): the ConvTranspose bias gradient taken from the concat data-gradient's per-tile channel sums.  One block per 8
// channels: 32 row lanes with eight rows in flight each (the walk is latency-bound), fp64, fixed order.
__device__ __forceinline__ void colsum_block(const float* __restrict__ part, int rows, int Ctot, int col0, int C,
                                             float* __restrict__ out, int cb, double (&sh)[32][8]) {
  const int cx = threadIdx.x & 7, ry = threadIdx.x >> 3;
  const int c = cb * 8 + cx;
  double a8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (c < C)
    for (int r = ry; r < rows; r += 256) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {   // unconditional loads of a clamped row (a conditional load compiles to branch + wait)
        const int rr = r + 32 * u < rows ? r + 32 * u : rows - 1;
        v[u] = part[((size_t)rr * Ctot + col0 + c) * 2];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) a8[u] += (r + 32 * u < rows) ? (double)v[u] : 0.0;
    }
  double s = 0.0;
#pragma unroll
  for (int u = 0; u < 8; ++u) s += a8[u];
  sh[ry][cx] = s;
  __syncthreads();
  if (ry != 0 || c >= C) return;
  s = 0.0;
  for (int r = 0; r < 32; ++r) s += sh[r][cx];
  out[c] = (float)s;
}

// Up to four of the reductions above in ONE launch (the weight gradients of a DoubleConv block, the ConvTranspose weight
// gradient of an Up block and its bias gradient): the jobs travel by value in the kernel arguments; a block finds its job
// by its first block index.  Results are bit-identical to the single-job launches.
struct ReduceJob {
  const float* src;     // slabs | partial rows
  float* dst;           // gradient | column sums
  int S, N, CA, CB, Np, CAp, CBp, taps;     // kind 2: S = rows, N = total channels of a row, CA = first column, CB = columns
  int kind;             // 0: row form, 1: slab-parallel form, 2: column sums
  int block0, ny;       // first block of the job; kind 1: blocks per output row
  int pad_;
};
struct ReduceJobs {
  ReduceJob j[4];
  int n;
};
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const ReduceJobs p) {
  extern __shared__ float row[];
  __shared__ float4 red[16][16];
  int ji = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < p.n && (int)blockIdx.x >= p.j[i].block0) ji = i;
  const ReduceJob& jb = p.j[ji];
  const int local = blockIdx.x - jb.block0;
  if (jb.kind == 0) {
    wgrad_reduce_row(jb.src, jb.S, jb.dst, local, jb.CA, jb.CB, jb.Np, jb.CAp, jb.CBp, jb.taps, row);
  } else if (jb.kind == 1) {
    wgrad_reduce_wide(jb.src, jb.S, jb.dst, local / jb.ny, local % jb.ny, jb.CA, jb.CB, jb.Np, jb.CAp, jb.CBp, jb.taps, red);
  } else {
    colsum_block(jb.src, jb.S, jb.N, jb.CA, jb.CB, jb.dst, local, *(double(*)[32][8])red);
  }
}

static int grid_for(long total) {
  long g = (total + 255) / 256;
  return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

}  // namespace

int segk_nchw_to_nhwc_impl(const float* src, void* dst, int B, int C, int H, int W, int Cp, int dtype, hipStream_t st) {
  SEGK_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && Cp >= C && Cp % 32 == 0, "nchw_to_nhwc: bad arguments");
  const long total = (long)B * H * W * (Cp / (dtype == SEGK_DT_BF16 ? 8 : 4));
  if (dtype == SEGK_DT_BF16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, src, (bf16_t*)dst, B, C, H, W, Cp);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, src, (float*)dst, B, C, H, W, Cp);
  SEGK_CHECK_LAUNCH("nchw_to_nhwc");
  return 0;
}

int segk_nhwc_to_nchw_impl(const void* src, float* dst, int B, int C, int H, int W, int Cp, int dtype, hipStream_t st) {
  SEGK_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && Cp >= C, "nhwc_to_nchw: bad arguments");
  const long total = (long)B * H * W * C;
  if (dtype == SEGK_DT_BF16)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, (const bf16_t*)src, dst, B, C, H, W, Cp);
  else
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)src, dst, B, C, H, W, Cp);
  SEGK_CHECK_LAUNCH("nhwc_to_nchw");
  return 0;
}

int segk_pack_conv_weight_impl(const float* w, void* dst, int Cout, int CA, int CB, int Coutp, int CAp, int CBp,
                               int taps, int mode, int dtype, hipStream_t st) {
  const int CH = dtype == SEGK_DT_BF16 ? 32 : 16;
  SEGK_REQUIRE(w && dst && Cout > 0 && CA > 0 && CB >= 0 && (taps == 9 || taps == 1) && (mode == 0 || mode == 1),
               "pack_conv_weight: bad arguments");
  SEGK_REQUIRE(Coutp >= Cout && CAp >= CA && CBp >= CB && Coutp % 32 == 0 && CAp % CH == 0 && CBp % CH == 0 &&
                   (CBp == 0) == (CB == 0),
               "pack_conv_weight: bad padding");
  const long total = (long)(CAp + CBp) * taps * Coutp;
  if (dtype == SEGK_DT_BF16)
    hipLaunchKernelGGL(pack_conv_weight_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, w, (bf16_t*)dst, Cout,
                       CA, CB, Coutp, CAp, CBp, taps, mode);
  else
    hipLaunchKernelGGL(pack_conv_weight_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, w, (float*)dst, Cout, CA,
                       CB, Coutp, CAp, CBp, taps, mode);
  SEGK_CHECK_LAUNCH("pack_conv_weight");
  return 0;
}

int segk_pack_conv3x3_both_impl(const float* w, void* dst_fwd, void* dst_dgrad, int Cout, int CA, int CB, int Coutp, int CAp,
                                 int CBp, int dtype, hipStream_t st) {
  SEGK_REQUIRE(w && dst_fwd && Cout > 0 && CA > 0 && CB >= 0, "pack_conv3x3_both: bad arguments");
  SEGK_REQUIRE(Coutp >= Cout && CAp >= CA && CBp >= CB && Coutp % 32 == 0 && CAp % 32 == 0 && CBp % 32 == 0 &&
                   (CBp == 0) == (CB == 0),
               "pack_conv3x3_both: bad padding");
  const dim3 grid((CAp + CBp) / 32, Coutp / 32);
  if (dtype == SEGK_DT_BF16)
    hipLaunchKernelGGL(pack_conv3x3_both_kernel<bf16_t>, grid, dim3(256), 0, st, w, (bf16_t*)dst_fwd, (bf16_t*)dst_dgrad, Cout,
                       CA, CB, Coutp, CAp, CBp);
  else
    hipLaunchKernelGGL(pack_conv3x3_both_kernel<float>, grid, dim3(256), 0, st, w, (float*)dst_fwd, (float*)dst_dgrad, Cout, CA,
                       CB, Coutp, CAp, CBp);
  SEGK_CHECK_LAUNCH("pack_conv3x3_both");
  return 0;
}

int segk_pack_multi_impl(const void* table, int n, int total_blocks, int dtype, hipStream_t st) {
  SEGK_REQUIRE(table && n > 0 && n <= 64 && total_blocks > 0, "pack_multi: bad arguments");
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "pack_multi: bad dtype %d", dtype);
  if (dtype == SEGK_DT_BF16)
    hipLaunchKernelGGL(pack_multi_kernel<bf16_t>, dim3(total_blocks), dim3(256), 0, st, (const SegkPackEntry*)table, n);
  else
    hipLaunchKernelGGL(pack_multi_kernel<float>, dim3(total_blocks), dim3(256), 0, st, (const SegkPackEntry*)table, n);
  SEGK_CHECK_LAUNCH("pack_multi");
  return 0;
}
int segk_pack_convt_chunk_impl() { return PACK_CONVT_CHUNK; }

int segk_pack_convt_weight_impl(const float* w, void* dst, int Cin, int Cout, int Cinp, int Coutp, int mode, int dtype,
                                hipStream_t st) {
  SEGK_REQUIRE(w && dst && Cin > 0 && Cout > 0 && Cinp >= Cin && Coutp >= Cout && Cinp % 32 == 0 && Coutp % 32 == 0 &&
                   (mode == 0 || mode == 1),
               "pack_convt_weight: bad arguments");
  const long total = (long)Cinp * 4 * Coutp;
  if (dtype == SEGK_DT_BF16)
    hipLaunchKernelGGL(pack_convt_weight_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, w, (bf16_t*)dst, Cin,
                       Cout, Cinp, Coutp, mode);
  else
    hipLaunchKernelGGL(pack_convt_weight_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, w, (float*)dst, Cin,
                       Cout, Cinp, Coutp, mode);
  SEGK_CHECK_LAUNCH("pack_convt_weight");
  return 0;
}

// jobs: host array of n <= 4 segk_reduce_job (include/segk.h); see wgrad_reduce_multi_kernel
int segk_wgrad_reduce_multi_impl(const segk_reduce_job* jobs, int n, hipStream_t st) {
  SEGK_REQUIRE(jobs && n >= 1 && n <= 4, "wgrad_reduce_multi: 1..4 jobs");
  ReduceJobs p{};
  p.n = n;
  int blocks = 0;
  size_t lds = 0;
  for (int i = 0; i < n; ++i) {
    const segk_reduce_job& q = jobs[i];
    ReduceJob& r = p.j[i];
    r.src = q.src; r.dst = q.dst; r.block0 = blocks; r.ny = 1;
    SEGK_REQUIRE(q.src && q.dst, "wgrad_reduce_multi: job %d: null pointer", i);
    if (q.kind == 0) {        // weight-gradient slabs
      SEGK_REQUIRE(q.S > 0 && q.N > 0 && q.CA > 0 && q.CB >= 0 && q.Np >= q.N && q.CAp >= q.CA && q.CBp >= q.CB && q.taps > 0 &&
                       (q.CAp + q.CBp) % 32 == 0,
                   "wgrad_reduce_multi: job %d: bad arguments", i);
      r.S = q.S; r.N = q.N; r.CA = q.CA; r.CB = q.CB; r.Np = q.Np; r.CAp = q.CAp; r.CBp = q.CBp; r.taps = q.taps;
      if (q.S > 16) {
        r.kind = 1;
        r.ny = (q.taps * (q.CAp + q.CBp) / 4 + 15) / 16;
        blocks += q.N * r.ny;
      } else {
        r.kind = 0;
        const size_t need = (size_t)(q.CA + q.CB) * q.taps * sizeof(float);
        SEGK_REQUIRE(need <= 64 * 1024, "wgrad_reduce_multi: a gradient row of %zu bytes exceeds the LDS staging limit", need);
        if (need > lds) lds = need;
        blocks += q.N;
      }
    } else if (q.kind == 1) { // column sums of partial rows
      SEGK_REQUIRE(q.S > 0 && q.N > 0 && q.CA >= 0 && q.CB > 0 && q.CA + q.CB <= q.N, "wgrad_reduce_multi: job %d: bad column range", i);
      r.kind = 2; r.S = q.S; r.N = q.N; r.CA = q.CA; r.CB = q.CB;
      blocks += (q.CB + 7) / 8;
    } else {
      SEGK_FAIL(-2, "wgrad_reduce_multi: job %d: bad kind %d", i, q.kind);
    }
  }
  if (lds > 48 * 1024) {
    static bool attr_set[SEGK_MAX_DEVICES] = {};
    const int dev = segk_device_index();
    if (!attr_set[dev]) {
      if (hipFuncSetAttribute((const void*)wgrad_reduce_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) !=
          hipSuccess)
        SEGK_FAIL(-3, "wgrad_reduce_multi: cannot raise dynamic LDS limit");
      attr_set[dev] = true;
    }
  }
  hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(blocks), dim3(256), lds, st, p);
  SEGK_CHECK_LAUNCH("wgrad_reduce_multi");
  return 0;
}

int segk_wgrad_reduce_impl(const float* slabs, int S, float* grad, int N, int CA, int CB, int Np, int CAp, int CBp,
                           int taps, hipStream_t st) {
  SEGK_REQUIRE(slabs && grad && S > 0 && N > 0 && CA > 0 && CB >= 0 && Np >= N && CAp >= CA && CBp >= CB && taps > 0,
               "wgrad_reduce: bad arguments");
  if (S > 16) {      // many thin slabs (narrow layers): slab-parallel reduction, one launch
    const int nvec = taps * (CAp + CBp) / 4;
    hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3(N, (nvec + 15) / 16), dim3(256), 0, st, slabs, S, grad, N, CA, CB, Np, CAp,
                       CBp, taps);
    SEGK_CHECK_LAUNCH("wgrad_reduce_wide");
    return 0;
  }
  const size_t lds = (size_t)(CA + CB) * taps * sizeof(float);
  SEGK_REQUIRE(lds <= 64 * 1024, "wgrad_reduce: a gradient row of %zu bytes exceeds the LDS staging limit", lds);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(N), dim3(256), lds, st, slabs, S, grad, N, CA, CB, Np, CAp, CBp, taps);
  SEGK_CHECK_LAUNCH("wgrad_reduce");
  return 0;
}
