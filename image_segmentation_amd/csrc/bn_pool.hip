// BatchNorm2d (train/eval) finalisation, BN+ReLU apply, BN backward reductions, MaxPool2d(2,2)
// forward/backward on NHWC tensors.  HBM-bound streaming kernels: every thread owns one 16-byte
// channel vector (fixed channels -> per-channel constants live in registers) and walks pixels, so a
// row of threads reads/writes whole contiguous pixel rows (coalesced 16 B per lane).
//
// Reference semantics: torch.nn.BatchNorm2d as instantiated at unet/unet.py:17,20 and
// clip/clipunet.py:88,91 (eps 1e-5, momentum 0.1, biased variance to normalise, unbiased variance into
// running_var), nn.ReLU, nn.MaxPool2d(2,2) at unet.py:40 (backward routes to the first maximum in
// row-major window order).
#include <atomic>
#include "common.hpp"
#include "segk_internal.h"
#include "ticket.hpp"
#include "../../include/segk.h"

namespace {
__device__ unsigned g_tickets[TICKET_SLOTS * TICKET_GROUPS];     // zeroed per launch by the host (ticket.hpp)

// block = CVB channel-vectors x ROWS pixel lanes (CVB*ROWS <= 256); grid.y covers channel blocks.
struct Lanes {
  int cx, ry, cv;
  bool active;
};
__device__ __forceinline__ Lanes lanes(int cvb, int cvec) {
  Lanes l;
  l.cx = threadIdx.x % cvb;
  l.ry = threadIdx.x / cvb;
  l.cv = blockIdx.y * cvb + l.cx;
  l.active = l.cv < cvec;
  return l;
}

// ---------------------------------------------------------------------------------------------
constexpr int NCH = 32;   // first-stage chunks of the per-tile statistics reduction
// stage A: per-tile partials [MT][C][2] float -> [NCH][C][2] double (fixed order inside a chunk: bit-stable)
__device__ __forceinline__ void stats_chunk(const float* __restrict__ part, int MT, int C, double* __restrict__ out,
                                            double (&sh)[8][32][2]) {
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  const int chunk = (MT + NCH - 1) / NCH;
  const int m0 = blockIdx.y * chunk, m1 = min(MT, m0 + chunk);
  double s1 = 0.0, s2 = 0.0;
  for (int m = m0 + ry; m < m1; m += 8) {
    const float2 v = ((const float2*)part)[(size_t)m * C + c];
    s1 += (double)v.x;
    s2 += (double)v.y;
  }
  sh[ry][cx][0] = s1;
  sh[ry][cx][1] = s2;
  __syncthreads();
  if (ry == 0) {
    s1 = 0.0; s2 = 0.0;
    for (int r = 0; r < 8; ++r) { s1 += sh[r][cx][0]; s2 += sh[r][cx][1]; }
    out[((size_t)blockIdx.y * C + c) * 2 + 0] = s1;
    out[((size_t)blockIdx.y * C + c) * 2 + 1] = s2;
  }
}

// stage B: [MT][C][2] double partials -> scale/shift (+ running stats) for the 32 channels of blockIdx.x
__device__ __forceinline__ void finalize_channels(const double* __restrict__ part, int MT, int C, int C_real,
                                                  double count, const float* conv_bias,
                                                  const float* gamma, const float* beta, float* rmean,
                                                  float* rvar, float momentum, float eps, int training,
                                                  float* scale, float* shift, float* mean_out,
                                                  float* rstd_out, double (&sh)[8][32][2]) {
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  double s1 = 0.0, s2 = 0.0;
  if (training && c < C)
    for (int m = ry; m < MT; m += 8) {
      s1 += part[((size_t)m * C + c) * 2 + 0];
      s2 += part[((size_t)m * C + c) * 2 + 1];
    }
  sh[ry][cx][0] = s1;
  sh[ry][cx][1] = s2;
  __syncthreads();
  if (ry != 0 || c >= C) return;
  if (c >= C_real) {  // padded channel: stays identically zero
    scale[c] = 0.f; shift[c] = 0.f;
    if (mean_out) { mean_out[c] = 0.f; rstd_out[c] = 0.f; }
    return;
  }
  const float g = gamma[c], be = beta[c];
  const float cb = conv_bias ? conv_bias[c] : 0.f;
  if (training) {
    s1 = 0.0; s2 = 0.0;
    for (int r = 0; r < 8; ++r) { s1 += sh[r][cx][0]; s2 += sh[r][cx][1]; }
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = g * rstd;
    scale[c] = sc;
    shift[c] = be - (float)mean * sc;      // applied to the bias-free conv output: the bias cancels
    mean_out[c] = (float)mean;
    rstd_out[c] = rstd;
    if (rmean) {
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * ((float)mean + cb);
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
  } else {
    const float rstd = 1.f / sqrtf(rvar[c] + eps);
    const float sc = g * rstd;
    scale[c] = sc;
    shift[c] = be + (cb - rmean[c]) * sc;
    if (mean_out) { mean_out[c] = rmean[c] - cb; rstd_out[c] = rstd; }
  }
}

// eval mode (no statistics to reduce): stage B alone
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ part, int MT, int C, int C_real,
                                                          double count, const float* conv_bias,
                                                          const float* gamma, const float* beta, float* rmean,
                                                          float* rvar, float momentum, float eps, int training,
                                                          float* scale, float* shift, float* mean_out,
                                                          float* rstd_out) {
  __shared__ double sh[8][32][2];
  finalize_channels(part, MT, C, C_real, count, conv_bias, gamma, beta, rmean, rvar, momentum, eps, training, scale, shift,
                    mean_out, rstd_out, sh);
}

// training mode, few partial rows (MT <= 1024: every layer but the 128-channel ones at 128x128 and above): one block of
// 32 row lanes x 32 channels walks all rows itself, sixteen rows in flight per lane, and finishes -- no second stage at all
__global__ __launch_bounds__(1024) void bn_stats_finalize_small_kernel(const float* __restrict__ part, int MT, int C,
                                                                       int C_real, double count, const float* conv_bias,
                                                                       const float* gamma, const float* beta, float* rmean,
                                                                       float* rvar, float momentum, float eps, float* scale,
                                                                       float* shift, float* mean_out, float* rstd_out) {
  __shared__ double sh[32][32][2];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  // sixteen rows in flight per lane and ONE accumulator pair (the registers hold loads, not partial sums: 1024 threads leave
  // 128 per lane): MT <= 512 is a single memory round trip.  Fixed order of additions: bit-stable
  double s1 = 0.0, s2 = 0.0;
#pragma unroll 1
  for (int m = ry; m < MT; m += 512) {
    float2 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {   // unconditional loads of a clamped row (a conditional load compiles to branch + wait)
      const int r = m + 32 * u < MT ? m + 32 * u : MT - 1;
      v[u] = ((const float2*)part)[(unsigned)r * (unsigned)C + (unsigned)c];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const bool ok = m + 32 * u < MT;
      s1 += ok ? (double)v[u].x : 0.0;
      s2 += ok ? (double)v[u].y : 0.0;
    }
  }
  sh[ry][cx][0] = s1;
  sh[ry][cx][1] = s2;
  __syncthreads();
  if (ry != 0) return;
  s1 = 0.0; s2 = 0.0;
#pragma unroll 8
  for (int r = 0; r < 32; ++r) { s1 += sh[r][cx][0]; s2 += sh[r][cx][1]; }   // (32 reads in flight would spill)
  if (c >= C_real) {  // padded channel: stays identically zero
    scale[c] = 0.f; shift[c] = 0.f; mean_out[c] = 0.f; rstd_out[c] = 0.f;
    return;
  }
  const float g = gamma[c], be = beta[c];
  const float cb = conv_bias ? conv_bias[c] : 0.f;
  const double mean = s1 / count;
  double var = s2 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = g * rstd;
  scale[c] = sc;
  shift[c] = be - (float)mean * sc;      // applied to the bias-free conv output: the bias cancels
  mean_out[c] = (float)mean;
  rstd_out[c] = rstd;
  if (rmean) {
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * ((float)mean + cb);
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
  }
}

// training mode, ONE launch: grid (C/32, NCH) blocks reduce their chunk of the per-tile partials (stage A); the block that
// arrives last for a channel group finishes it (stage B over the NCH chunk sums, in chunk order: bit-stable)
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float* __restrict__ part, int MT, int C, int C_real,
                                                                double count, const float* conv_bias,
                                                                const float* gamma, const float* beta, float* rmean,
                                                                float* rvar, float momentum, float eps,
                                                                float* scale, float* shift, float* mean_out,
                                                                float* rstd_out, double* __restrict__ scratch,
                                                                unsigned* __restrict__ tickets) {
  __shared__ double sh[8][32][2];
  __shared__ int last;
  stats_chunk(part, MT, C, scratch, sh);
  if (!last_arriver(tickets + blockIdx.x, gridDim.y, &last)) return;
  finalize_channels(scratch, (int)gridDim.y, C, C_real, count, conv_bias, gamma, beta, rmean, rvar, momentum, eps, 1, scale,
                    shift, mean_out, rstd_out, sh);
}

// ---------------------------------------------------------------------------------------------
// y = relu(z*scale + shift)
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_apply_kernel(const T* __restrict__ z, T* __restrict__ y,
                                                            const float* scale, const float* shift, long P, int C,
                                                            int cvb, int rows) {
  using E = ET<T>;
  const Lanes l = lanes(cvb, C / E::VEC);
  if (!l.active || l.ry >= rows) return;
  float sc[E::VEC], sh[E::VEC];
#pragma unroll
  for (int j = 0; j < E::VEC; ++j) { sc[j] = scale[l.cv * E::VEC + j]; sh[j] = shift[l.cv * E::VEC + j]; }
  auto one = [&](const uint4 raw, const size_t off) {
    float f[E::VEC];
    unpack16<T>(raw, f);
#pragma unroll
    for (int j = 0; j < E::VEC; ++j) f[j] = fmaxf(fmaf(f[j], sc[j], sh[j]), 0.f);
    *(uint4*)(y + off) = pack16<T>(f);
  };
  // one pixel per thread per pass: like the backward apply pass (see there) this read + write stream does not gain from
  // more loads in flight per thread
  for (long p = (long)blockIdx.x * rows + l.ry; p < P; p += (long)gridDim.x * rows) {
    const size_t off = (size_t)p * C + (size_t)l.cv * E::VEC;
    one(*(const uint4*)(z + off), off);
  }
}

// ---------------------------------------------------------------------------------------------
// BN backward pass 1: g = dy * [z*scale+shift > 0];  partial sums of g and g*xhat per channel.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ z,
                                                            const float* scale, const float* shift,
                                                            const float* mean, const float* rstd, long P, int C,
                                                            int cvb, int rows, float* part) {
  using E = ET<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = (float*)smem;  // [rows][cvb][2*VEC]
  const Lanes l = lanes(cvb, C / E::VEC);
  float sg[E::VEC], sgx[E::VEC];
#pragma unroll
  for (int j = 0; j < E::VEC; ++j) { sg[j] = 0.f; sgx[j] = 0.f; }
  if (l.active && l.ry < rows) {
    float sc[E::VEC], sh[E::VEC], mu[E::VEC], rs[E::VEC];
#pragma unroll
    for (int j = 0; j < E::VEC; ++j) {
      const int c = l.cv * E::VEC + j;
      sc[j] = scale[c]; sh[j] = shift[c]; mu[j] = mean[c]; rs[j] = rstd[c];
    }
    auto one = [&](const uint4 rz, const uint4 rd) {
      float fz[E::VEC], fd[E::VEC];
      unpack16<T>(rz, fz);
      unpack16<T>(rd, fd);
#pragma unroll
      for (int j = 0; j < E::VEC; ++j) {
        const float g = (fmaf(fz[j], sc[j], sh[j]) > 0.f) ? fd[j] : 0.f;
        sg[j] += g;
        sgx[j] = fmaf(g, (fz[j] - mu[j]) * rs[j], sgx[j]);
      }
    };
    // four pixels (eight loads) per thread in flight: the grid is only two blocks per CU (short finalize), and left to itself
    // the compiler keeps one pixel in flight.  Pixels are consumed in the old order: bit-identical sums
    const long step = (long)gridDim.x * rows;
    const size_t cofs = (size_t)l.cv * E::VEC;
    long p = (long)blockIdx.x * rows + l.ry;
    for (; p + 3 * step < P; p += 4 * step) {
      uint4 rz[4], rd[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t off = (size_t)(p + u * step) * C + cofs;
        rz[u] = *(const uint4*)(z + off);
        rd[u] = *(const uint4*)(dy + off);
      }
      __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks the loads back to two in flight)
#pragma unroll
      for (int u = 0; u < 4; ++u) one(rz[u], rd[u]);
    }
    for (; p < P; p += step) {
      const size_t off = (size_t)p * C + cofs;
      one(*(const uint4*)(z + off), *(const uint4*)(dy + off));
    }
  }
  if (l.ry < rows) {
    float* r = red + ((size_t)l.ry * cvb + l.cx) * (2 * E::VEC);
#pragma unroll
    for (int j = 0; j < E::VEC; ++j) { r[j] = sg[j]; r[E::VEC + j] = sgx[j]; }
  }
  __syncthreads();
  if (l.ry == 0 && l.active) {
#pragma unroll
    for (int j = 0; j < E::VEC; ++j) {
      float a = 0.f, b = 0.f;
      for (int r = 0; r < rows; ++r) {  // fixed order
        const float* q = red + ((size_t)r * cvb + l.cx) * (2 * E::VEC);
        a += q[j];
        b += q[E::VEC + j];
      }
      float2* dst = (float2*)part + (size_t)blockIdx.x * C + l.cv * E::VEC + j;
      *dst = make_float2(a, b);
    }
  }
}

// partials [NB][C][2] -> dgamma, dbeta, coef = (sum_g/N, sum_gx/N)
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ part, int NB, int C,
                                                               int C_real, double count, float* dgamma, float* dbeta,
                                                               float* coef) {
  // 32 channels x 32 row-lanes per block: the NB block partials are walked in 32 interleaved streams
  __shared__ double sh[32][32][2];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    // sixteen rows in flight per lane, one accumulator pair: NB <= 512 (segk_bn_bwd_blocks) is ONE memory round trip.
    // Fixed order of additions: bit-stable
    for (int m = ry; m < NB; m += 512) {
      float2 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {   // unconditional loads of a clamped row (a conditional load compiles to branch + wait)
        const int r = m + 32 * u < NB ? m + 32 * u : NB - 1;
        v[u] = ((const float2*)part)[(unsigned)r * (unsigned)C + (unsigned)c];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const bool ok = m + 32 * u < NB;
        s1 += ok ? (double)v[u].x : 0.0;
        s2 += ok ? (double)v[u].y : 0.0;
      }
    }
  }
  sh[ry][cx][0] = s1;
  sh[ry][cx][1] = s2;
  __syncthreads();
  if (ry != 0 || c >= C) return;
  s1 = 0.0; s2 = 0.0;
  for (int r = 0; r < 32; ++r) { s1 += sh[r][cx][0]; s2 += sh[r][cx][1]; }   // fixed order
  if (c < C_real) { dbeta[c] = (float)s1; dgamma[c] = (float)s2; }
  coef[2 * c] = (float)(s1 / count);
  coef[2 * c + 1] = (float)(s2 / count);
}

// BN backward pass 2 (in place allowed): dz = scale * (g - c1 - xhat*c2)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ z,
                                                           T* __restrict__ dz, const float* scale,
                                                           const float* shift, const float* mean, const float* rstd,
                                                           const float* coef, long P, int C, int cvb, int rows) {
  using E = ET<T>;
  const Lanes l = lanes(cvb, C / E::VEC);
  if (!l.active || l.ry >= rows) return;
  float sc[E::VEC], sh[E::VEC], mu[E::VEC], rs[E::VEC], c1[E::VEC], c2[E::VEC];
#pragma unroll
  for (int j = 0; j < E::VEC; ++j) {
    const int c = l.cv * E::VEC + j;
    sc[j] = scale[c]; sh[j] = shift[c]; mu[j] = mean[c]; rs[j] = rstd[c];
    c1[j] = coef[2 * c]; c2[j] = coef[2 * c + 1];
  }
  auto one = [&](const uint4 rz, const uint4 rd, const size_t off) {
    float fz[E::VEC], fd[E::VEC];
    unpack16<T>(rz, fz);
    unpack16<T>(rd, fd);
#pragma unroll
    for (int j = 0; j < E::VEC; ++j) {
      const float g = (fmaf(fz[j], sc[j], sh[j]) > 0.f) ? fd[j] : 0.f;
      const float xh = (fz[j] - mu[j]) * rs[j];
      fd[j] = sc[j] * (g - c1[j] - xh * c2[j]);
    }
    *(uint4*)(dz + off) = pack16<T>(fd);
  };
  // APF pixels (2 APF loads) per thread in flight, all issued before the first store (dz may be dy or z: in place allowed,
  // a thread only ever reads the elements it writes).  Measured at B = 32 (reduce + finalize + apply, us, same box):
  // APF 1 / 2 / 4 = 250 / 274 / 278 at 64 ch x 256^2, 122 / 125 / 132 at 128 x 128^2 -- a pass that WRITES as much as it reads
  // streams best with one pixel per thread and the occupancy that leaves; the read-only reduce pass gains from four
#ifndef SEGK_BN_APPLY_FLIGHT
#define SEGK_BN_APPLY_FLIGHT 1
#endif
  constexpr int APF = SEGK_BN_APPLY_FLIGHT;
  const long step = (long)gridDim.x * rows;
  const size_t cofs = (size_t)l.cv * E::VEC;
  long p = (long)blockIdx.x * rows + l.ry;
  for (; p + (APF - 1) * step < P; p += APF * step) {
    size_t o[APF];
    uint4 rz[APF], rd[APF];
#pragma unroll
    for (int u = 0; u < APF; ++u) {
      o[u] = (size_t)(p + u * step) * C + cofs;
      rz[u] = *(const uint4*)(z + o[u]);
      rd[u] = *(const uint4*)(dy + o[u]);
    }
#pragma unroll
    for (int u = 0; u < APF; ++u) one(rz[u], rd[u], o[u]);
  }
  for (; p < P; p += step) {
    const size_t o0 = (size_t)p * C + cofs;
    one(*(const uint4*)(z + o0), *(const uint4*)(dy + o0), o0);
  }
}

// y = relu(z*scale + shift) AND its MaxPool2d(2,2) in one pass (a DoubleConv block whose output feeds a Down block):
// one thread per 2x2 window x channel vector; the pooled value is the maximum of the ROUNDED y values, i.e. exactly what
// maxpool_fwd_kernel reads back, so fused and unfused paths agree bit for bit.
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_apply_pool_kernel(const T* __restrict__ z, T* __restrict__ y,
                                                                 T* __restrict__ pooled, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, int B, int H, int W, int C) {
  using E = ET<T>;
  const int Ho = H >> 1, Wo = W >> 1, CV = C / E::VEC;
  const int Hc = (H + 1) >> 1, Wc = (W + 1) >> 1;   // also visit the odd border (no pooled output there)
  const long total = (long)B * Hc * Wc * CV;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {
    const Idx4 ix = split4(i, (unsigned)CV, (unsigned)Wc, (unsigned)Hc);
    const int cv = ix.cv, xo = ix.x, yo = ix.y, b = ix.b;
    float sc[E::VEC], sh[E::VEC], mx[E::VEC];
#pragma unroll
    for (int j = 0; j < E::VEC; ++j) { sc[j] = scale[cv * E::VEC + j]; sh[j] = shift[cv * E::VEC + j]; mx[j] = 0.f; }
    // the four loads of the window are issued together, unconditionally, from coordinates clamped into the image (a load
    // under a per-lane condition compiles to branch + load + wait: four serial round trips per window)
    size_t offs[4];
    uint4 raw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = 2 * yo + (k >> 1), xx = 2 * xo + (k & 1);
      offs[k] = (((size_t)(b * H + (yy < H ? yy : H - 1))) * W + (xx < W ? xx : W - 1)) * C + cv * E::VEC;
      raw[k] = *(const uint4*)(z + offs[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = 2 * yo + (k >> 1), xx = 2 * xo + (k & 1);
      if (yy >= H || xx >= W) continue;
      const size_t off = offs[k];
      float f[E::VEC];
      unpack16<T>(raw[k], f);
#pragma unroll
      for (int j = 0; j < E::VEC; ++j) f[j] = fmaxf(fmaf(f[j], sc[j], sh[j]), 0.f);
      const uint4 pk = pack16<T>(f);
      *(uint4*)(y + off) = pk;
      unpack16<T>(pk, f);                           // the rounded values the pooling would read back
#pragma unroll
      for (int j = 0; j < E::VEC; ++j) mx[j] = fmaxf(mx[j], f[j]);   // y >= 0: 0 is a neutral start
    }
    if (yo < Ho && xo < Wo) *(uint4*)(pooled + (((size_t)(b * Ho + yo)) * Wo + xo) * C + cv * E::VEC) = pack16<T>(mx);
  }
}

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H,
                                                          int W, int C) {
  using E = ET<T>;
  const int Ho = H >> 1, Wo = W >> 1, CV = C / E::VEC;
  const long total = (long)B * Ho * Wo * CV;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {
    const Idx4 ix = split4(i, (unsigned)CV, (unsigned)Wo, (unsigned)Ho);
    const int cv = ix.cv, xo = ix.x, yo = ix.y, b = ix.b;
    const T* src = x + (((size_t)(b * H + 2 * yo)) * W + 2 * xo) * C + cv * E::VEC;
    float f0[E::VEC], f1[E::VEC], f2[E::VEC], f3[E::VEC];
    unpack16<T>(*(const uint4*)src, f0);
    unpack16<T>(*(const uint4*)(src + C), f1);
    unpack16<T>(*(const uint4*)(src + (size_t)W * C), f2);
    unpack16<T>(*(const uint4*)(src + (size_t)W * C + C), f3);
#pragma unroll
    for (int j = 0; j < E::VEC; ++j) f0[j] = fmaxf(fmaxf(f0[j], f1[j]), fmaxf(f2[j], f3[j]));
    *(uint4*)(y + (((size_t)(b * Ho + yo)) * Wo + xo) * C + cv * E::VEC) = pack16<T>(f0);
  }
}

// dx (+)= route(dy) ; one thread per 2x2 window x channel vector (windows tile the even part of the image).
// STAT: x is the output y = relu(bn(z)) of a DoubleConv block and dx its complete gradient (skip gradient + routed
// pool gradient): the kernel also accumulates that BatchNorm's backward reductions sum(g), sum(g * xhat) with
// g = dx where y > 0 -- xhat is recovered from y itself where the ReLU is active, xhat = (y - shift) * rstd / scale - mean * rstd,
// and pixels with y == 0 contribute nothing -- so the block's bn_bwd_reduce pass (a read of dx and z) disappears.
// Needs a power-of-two number of channel vectors (a thread then keeps its channels); channels with scale == 0 or
// |scale| < |shift| / 16 take xhat from z instead (see from_z below; without z they would get xhat = -mean * rstd).
template <typename T, bool STAT>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                          T* __restrict__ dx, int B, int H, int W, int C,
                                                          int accumulate, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, float* __restrict__ part,
                                                          const T* __restrict__ z) {
  using E = ET<T>;
  const int Ho = H >> 1, Wo = W >> 1, CV = C / E::VEC;
  const int Hc = (H + 1) >> 1, Wc = (W + 1) >> 1;  // also visit the odd border (gradient zero there)
  const long total = (long)B * Hc * Wc * CV;
  float xa[E::VEC], xb[E::VEC], sg[E::VEC], sgx[E::VEC];
  // xhat cannot be recovered from y where scale == 0 (y = relu(shift) is constant), and only badly where |scale| << |shift|
  // (the rounding of y is amplified by 1 / scale).  A thread whose channels include such a one reads z (when the caller
  // passed it) and takes xhat = (z - mean) * rstd like the stand-alone reduction, in a second pass of its own after the
  // main loop; every other thread never touches z.
  bool from_z = false;
  if (STAT) {
    const int cv0 = threadIdx.x % CV;              // fixed for the thread: 256 and the grid stride are multiples of CV
#pragma unroll
    for (int j = 0; j < E::VEC; ++j) {
      const int c = cv0 * E::VEC + j;
      const float sc = scale[c], rs = rstd[c], sh = shift[c];
      from_z = from_z || (z != nullptr && (sc == 0.f || fabsf(sc) * 16.f < fabsf(sh)));
      xa[j] = sc != 0.f ? rs / sc : 0.f;
      xb[j] = -sh * xa[j] - mean[c] * rs;
      sg[j] = 0.f; sgx[j] = 0.f;
    }
  }
#ifdef SEGK_POOL_NO_Z
  from_z = false;                                  // diagnostic build: the kernel without its rare path (A/B of its cost)
#endif
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {
    const Idx4 ix = split4(i, (unsigned)CV, (unsigned)Wc, (unsigned)Hc);
    const int cv = ix.cv, xo = ix.x, yo = ix.y, b = ix.b;
    const bool inwin = (yo < Ho) && (xo < Wo);
    // every load of the window is issued before any is used, unconditionally, from coordinates clamped into the image (a
    // load under a per-lane condition compiles to branch + load + wait: nine serial memory round trips per window)
    const int yoc = yo < Ho ? yo : Ho - 1, xoc = xo < Wo ? xo : Wo - 1;
    const uint4 rg = *(const uint4*)(dy + (((size_t)(b * Ho + yoc)) * Wo + xoc) * C + cv * E::VEC);
    uint4 rv[4], ro[4];
    size_t off[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = 2 * yo + (k >> 1), xx = 2 * xo + (k & 1);
      off[k] = (((size_t)(b * H + (yy < H ? yy : H - 1))) * W + (xx < W ? xx : W - 1)) * C + cv * E::VEC;
      rv[k] = *(const uint4*)(x + off[k]);
    }
    if (accumulate) {
#pragma unroll
      for (int k = 0; k < 4; ++k) ro[k] = *(const uint4*)(dx + off[k]);
    }
    float g[E::VEC];
    float v[4][E::VEC];
    unpack16<T>(rg, g);
#pragma unroll
    for (int k = 0; k < 4; ++k) unpack16<T>(rv[k], v[k]);
    int sel[E::VEC];
#pragma unroll
    for (int j = 0; j < E::VEC; ++j) {
      sel[j] = 0;
      float m = v[0][j];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (v[k][j] > m) { m = v[k][j]; sel[j] = k; }  // strict '>' keeps the FIRST maximum
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = 2 * yo + (k >> 1), xx = 2 * xo + (k & 1);
      if (yy >= H || xx >= W) continue;
      float o[E::VEC];
      if (accumulate) unpack16<T>(ro[k], o);
#pragma unroll
      for (int j = 0; j < E::VEC; ++j) {
        const float r = (inwin && sel[j] == k) ? g[j] : 0.f;
        o[j] = accumulate ? o[j] + r : r;
        if (STAT) {
          const float gg = v[k][j] > 0.f ? o[j] : 0.f;
          sg[j] += gg;
          sgx[j] = fmaf(gg, fmaf(v[k][j], xa[j], xb[j]), sgx[j]);
        }
      }
      *(uint4*)(dx + off[k]) = pack16<T>(o);
    }
  }
  if (STAT) {
    __shared__ float red[256][2 * E::VEC + 1];
    if (from_z) {
      // Rare second pass of this thread over ITS OWN items (a separate loop: the main loop above keeps its registers):
      // sum(g * xhat) again with xhat = (z - mean) * rstd, from the dx values the thread itself wrote (program order makes
      // them visible to it), y for the ReLU mask, and z
#pragma unroll
      for (int j = 0; j < E::VEC; ++j) sgx[j] = 0.f;
      for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {
        const Idx4 ix = split4(i, (unsigned)CV, (unsigned)Wc, (unsigned)Hc);
        const int cv = ix.cv, xo = ix.x, yo = ix.y, b = ix.b;
        for (int k = 0; k < 4; ++k) {
          const int yy = 2 * yo + (k >> 1), xx = 2 * xo + (k & 1);
          if (yy >= H || xx >= W) continue;
          const size_t off = (((size_t)(b * H + yy)) * W + xx) * C + cv * E::VEC;
          float fy[E::VEC], fd[E::VEC], fz[E::VEC];
          unpack16<T>(*(const uint4*)(x + off), fy);
          unpack16<T>(*(const uint4*)(dx + off), fd);
          unpack16<T>(*(const uint4*)(z + off), fz);
#pragma unroll
          for (int j = 0; j < E::VEC; ++j) {
            const int c = cv * E::VEC + j;
            const float gg = fy[j] > 0.f ? fd[j] : 0.f;
            sgx[j] = fmaf(gg, (fz[j] - mean[c]) * rstd[c], sgx[j]);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < E::VEC; ++j) { red[threadIdx.x][j] = sg[j]; red[threadIdx.x][E::VEC + j] = sgx[j]; }
    __syncthreads();
    for (int q = threadIdx.x; q < CV * E::VEC; q += 256) {
      const int cv = q / E::VEC, j = q - cv * E::VEC;
      float a = 0.f, bsum = 0.f;
      for (int t = cv; t < 256; t += CV) {          // fixed order
        a += red[t][j];
        bsum += red[t][E::VEC + j];
      }
      ((float2*)part)[(size_t)blockIdx.x * C + q] = make_float2(a, bsum);
    }
  }
}

// per-channel sum over pixels (bias gradient of ConvTranspose2d / 1x1 Conv2d): partials [gridDim.x][C]
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_kernel(const T* __restrict__ x, long P, int C, int cvb, int rows,
                                                          float* part) {
  using E = ET<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = (float*)smem;  // [rows][cvb][VEC]
  const Lanes l = lanes(cvb, C / E::VEC);
  float s[E::VEC];
#pragma unroll
  for (int j = 0; j < E::VEC; ++j) s[j] = 0.f;
  if (l.active && l.ry < rows) {
    auto one = [&](const uint4 raw) {
      float f[E::VEC];
      unpack16<T>(raw, f);
#pragma unroll
      for (int j = 0; j < E::VEC; ++j) s[j] += f[j];
    };
    // a read-only pass on two blocks per CU: four pixels per thread in flight, consumed in the old order (bit-identical sums)
    const long step = (long)gridDim.x * rows;
    const size_t cofs = (size_t)l.cv * E::VEC;
    long p = (long)blockIdx.x * rows + l.ry;
    for (; p + 3 * step < P; p += 4 * step) {
      uint4 r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) r[u] = *(const uint4*)(x + (size_t)(p + u * step) * C + cofs);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) one(r[u]);
    }
    for (; p < P; p += step) one(*(const uint4*)(x + (size_t)p * C + cofs));
  }
  if (l.ry < rows) {
    float* r = red + ((size_t)l.ry * cvb + l.cx) * E::VEC;
#pragma unroll
    for (int j = 0; j < E::VEC; ++j) r[j] = s[j];
  }
  __syncthreads();
  if (l.ry == 0 && l.active)
#pragma unroll
    for (int j = 0; j < E::VEC; ++j) {
      float a = 0.f;
      for (int r = 0; r < rows; ++r) a += red[((size_t)r * cvb + l.cx) * E::VEC + j];
      part[(size_t)blockIdx.x * C + l.cv * E::VEC + j] = a;
    }
}
__global__ __launch_bounds__(1024) void channel_sum_finalize_kernel(const float* __restrict__ part, int NB, int C,
                                                                    int C_real, float* out) {
  // 32 channels x 32 row lanes, sixteen rows in flight per lane (unconditional loads of a clamped row): NB <= 512
  // (segk_bn_bwd_blocks) is one memory round trip.  Fixed order of additions: bit-stable
  __shared__ double sh[32][32];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  double s = 0.0;
  if (c < C) {
    for (int m = ry; m < NB; m += 512) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int r = m + 32 * u < NB ? m + 32 * u : NB - 1;
        v[u] = part[(unsigned)r * (unsigned)C + (unsigned)c];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) s += (m + 32 * u < NB) ? (double)v[u] : 0.0;
    }
  }
  sh[ry][cx] = s;
  __syncthreads();
  if (ry != 0 || c >= C_real) return;
  s = 0.0;
#pragma unroll 8
  for (int r = 0; r < 32; ++r) s += sh[r][cx];
  out[c] = (float)s;
}

static inline void lane_geometry(int C, int vec, int* cvb, int* rows, int* gy) {
  const int cvec = C / vec;
  *cvb = cvec < 128 ? cvec : 128;
  *rows = 256 / *cvb;
  *gy = cdiv(cvec, *cvb);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
unsigned* segk_ticket_slot(int groups, hipStream_t st) {
  static std::atomic<unsigned> n{0};
  static unsigned* base[SEGK_MAX_DEVICES] = {};      // per device: the symbol lives in that device's module image
  const int dev = segk_device_index();
  if (groups < 1 || groups > TICKET_GROUPS) return nullptr;
  if (!base[dev] && hipGetSymbolAddress((void**)&base[dev], HIP_SYMBOL(g_tickets)) != hipSuccess) return nullptr;
  unsigned* const p = base[dev] + (size_t)(n.fetch_add(1, std::memory_order_relaxed) % TICKET_SLOTS) * TICKET_GROUPS;
  // zero what this launch will count in, in stream order right before it (graph-capture safe: a memset node)
  if (hipMemsetAsync(p, 0, (size_t)groups * sizeof(unsigned), st) != hipSuccess) return nullptr;
  return p;
}

// diagnostics / tests: overwrite every ticket counter of the current device with the low 32 bits of `pattern` (what an
// aborted launch, or a stray store, would leave behind); every launch that draws a ticket afterwards must still elect
// exactly one finisher
int segk_debug_poison_tickets_impl(unsigned long long pattern, hipStream_t st) {
  unsigned* base = nullptr;
  SEGK_REQUIRE(hipGetSymbolAddress((void**)&base, HIP_SYMBOL(g_tickets)) == hipSuccess, "poison_tickets: no ticket array");
  static unsigned host[TICKET_SLOTS * TICKET_GROUPS];
  for (int i = 0; i < TICKET_SLOTS * TICKET_GROUPS; ++i) host[i] = (unsigned)pattern;
  SEGK_REQUIRE(hipMemcpyAsync(base, host, sizeof(host), hipMemcpyHostToDevice, st) == hipSuccess, "poison_tickets: copy failed");
  SEGK_REQUIRE(hipStreamSynchronize(st) == hipSuccess, "poison_tickets: sync failed");
  return 0;
}

int segk_bn_finalize_impl(const float* part, int MT, int C, int C_real, double count, const float* conv_bias,
                          const float* gamma, const float* beta, float* rmean, float* rvar, float momentum, float eps,
                          int training, float* scale, float* shift, float* mean, float* rstd, hipStream_t st) {
  SEGK_REQUIRE(C > 0 && C % 32 == 0 && C_real > 0 && C_real <= C, "bn_finalize: bad channels C=%d real=%d", C, C_real);
  SEGK_REQUIRE(gamma && beta && scale && shift, "bn_finalize: null pointer");
  if (training) SEGK_REQUIRE(part && MT > 0 && count > 0 && mean && rstd, "bn_finalize: training needs partials");
  else SEGK_REQUIRE(rmean && rvar, "bn_finalize: eval needs running statistics");
  // the stats buffer carries NCH*C*2 doubles of scratch behind the [MT][C][2] float partials
  double* scratch = part ? (double*)(const_cast<float*>(part) + (size_t)MT * C * 2) : nullptr;
  if (training && MT <= 1024) {
    hipLaunchKernelGGL(bn_stats_finalize_small_kernel, dim3(C / 32), dim3(1024), 0, st, part, MT, C, C_real, count, conv_bias,
                       gamma, beta, rmean, rvar, momentum, eps, scale, shift, mean, rstd);
    SEGK_CHECK_LAUNCH("bn_stats_finalize_small");
    return 0;
  }
  if (training) {
    SEGK_REQUIRE(C / 32 <= TICKET_GROUPS, "bn_finalize: at most %d channels", 32 * TICKET_GROUPS);
    unsigned* const tickets = segk_ticket_slot(C / 32, st);
    SEGK_REQUIRE(tickets != nullptr, "bn_finalize: no ticket array");
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(C / 32, NCH), dim3(256), 0, st, part, MT, C, C_real, count, conv_bias,
                       gamma, beta, rmean, rvar, momentum, eps, scale, shift, mean, rstd, scratch, tickets);
    SEGK_CHECK_LAUNCH("bn_stats_finalize");
    return 0;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C / 32), dim3(256), 0, st, scratch, NCH, C, C_real, count, conv_bias,
                     gamma, beta, rmean, rvar, momentum, eps, training, scale, shift, mean, rstd);
  SEGK_CHECK_LAUNCH("bn_finalize");
  return 0;
}
int segk_bn_stats_floats(int tiles, int Cp) {
  if (tiles <= 0 || Cp <= 0) return 0;
  const long long n = (long long)tiles * Cp * 2 + (long long)NCH * Cp * 4;
  return n > 0x7fffffffLL ? 0 : (int)n;     // sizes that do not fit an int are not served (the launch entries refuse them)
}

template <typename T>
static int bn_relu_apply_t(const void* z, void* y, const float* scale, const float* shift, long P, int C, hipStream_t st) {
  int cvb, rows, gy;
  lane_geometry(C, ET<T>::VEC, &cvb, &rows, &gy);
  long gx = (P + rows - 1) / rows;
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(bn_relu_apply_kernel<T>, dim3((int)gx, gy), dim3(256), 0, st, (const T*)z, (T*)y, scale, shift, P,
                     C, cvb, rows);
  SEGK_CHECK_LAUNCH("bn_relu_apply");
  return 0;
}
int segk_bn_relu_apply_impl(const void* z, void* y, const float* scale, const float* shift, long P, int C, int dtype,
                            hipStream_t st) {
  SEGK_REQUIRE(z && y && scale && shift && P > 0 && C > 0 && C % 32 == 0, "bn_relu_apply: bad arguments");
  return dtype == SEGK_DT_BF16 ? bn_relu_apply_t<bf16_t>(z, y, scale, shift, P, C, st)
                               : bn_relu_apply_t<float>(z, y, scale, shift, P, C, st);
}

int segk_bn_relu_apply_pool_impl(const void* z, void* y, void* pooled, const float* scale, const float* shift, int B, int H,
                                  int W, int C, int dtype, hipStream_t st) {
  SEGK_REQUIRE(z && y && pooled && scale && shift && B > 0 && H >= 2 && W >= 2 && C > 0 && C % 32 == 0,
               "bn_relu_apply_pool: bad arguments");
  const int vec = dtype == SEGK_DT_BF16 ? 8 : 4;
  long total = (long)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / vec);
  SEGK_REQUIRE(total < (1L << 31), "bn_relu_apply_pool: more than 2^31 windows x channel vectors");
  long g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  if (dtype == SEGK_DT_BF16)
    hipLaunchKernelGGL(bn_relu_apply_pool_kernel<bf16_t>, dim3((int)g), dim3(256), 0, st, (const bf16_t*)z, (bf16_t*)y,
                       (bf16_t*)pooled, scale, shift, B, H, W, C);
  else
    hipLaunchKernelGGL(bn_relu_apply_pool_kernel<float>, dim3((int)g), dim3(256), 0, st, (const float*)z, (float*)y,
                       (float*)pooled, scale, shift, B, H, W, C);
  SEGK_CHECK_LAUNCH("bn_relu_apply_pool");
  return 0;
}

int segk_bn_bwd_blocks(long P, int C, int dtype) {
  if (P <= 0 || C <= 0 || C % 32 != 0) return 0;      // nonsense input: no blocks (the launch entries refuse it)
  const int vec = dtype == SEGK_DT_BF16 ? 8 : 4;
  int cvb, rows, gy;
  lane_geometry(C, vec, &cvb, &rows, &gy);
  long gx = (P + rows - 1) / rows;
  if (gx > 512) gx = 512;      // enough blocks to fill the chip; keeps the finalize pass short
  return (int)gx;
}

template <typename T>
static int bn_bwd_t(const void* dy, const void* z, void* dz, const float* scale, const float* shift, const float* mean,
                    const float* rstd, long P, int C, int C_real, float* part, float* dgamma, float* dbeta,
                    float* coef, hipStream_t st) {
  using E = ET<T>;
  int cvb, rows, gy;
  lane_geometry(C, E::VEC, &cvb, &rows, &gy);
  const int gx = segk_bn_bwd_blocks(P, C, sizeof(T) == 2 ? SEGK_DT_BF16 : SEGK_DT_F32);
  const size_t lds = (size_t)rows * cvb * 2 * E::VEC * sizeof(float);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel<T>, dim3(gx, gy), dim3(256), lds, st, (const T*)dy, (const T*)z, scale, shift,
                     mean, rstd, P, C, cvb, rows, part);
  SEGK_CHECK_LAUNCH("bn_bwd_reduce");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C / 32), dim3(1024), 0, st, part, gx, C, C_real, (double)P, dgamma,
                     dbeta, coef);
  SEGK_CHECK_LAUNCH("bn_bwd_finalize");
  long ga = (P + rows - 1) / rows;
  if (ga > 4096) ga = 4096;
  hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3((int)ga, gy), dim3(256), 0, st, (const T*)dy, (const T*)z, (T*)dz,
                     scale, shift, mean, rstd, coef, P, C, cvb, rows);
  SEGK_CHECK_LAUNCH("bn_bwd_apply");
  return 0;
}
// finalize + apply from partials another kernel already produced (nb rows of [C][2])
template <typename T>
static int bn_bwd_from_part_t(const void* dy, const void* z, void* dz, const float* scale, const float* shift, const float* mean,
                              const float* rstd, long P, int C, int C_real, const float* part, int nb, float* dgamma,
                              float* dbeta, float* coef, hipStream_t st) {
  using E = ET<T>;
  int cvb, rows, gy;
  lane_geometry(C, E::VEC, &cvb, &rows, &gy);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C / 32), dim3(1024), 0, st, part, nb, C, C_real, (double)P, dgamma,
                     dbeta, coef);
  SEGK_CHECK_LAUNCH("bn_bwd_finalize");
  long ga = (P + rows - 1) / rows;
  if (ga > 4096) ga = 4096;
  hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3((int)ga, gy), dim3(256), 0, st, (const T*)dy, (const T*)z, (T*)dz,
                     scale, shift, mean, rstd, coef, P, C, cvb, rows);
  SEGK_CHECK_LAUNCH("bn_bwd_apply");
  return 0;
}
int segk_bn_bwd_from_part_impl(const void* dy, const void* z, void* dz, const float* scale, const float* shift,
                               const float* mean, const float* rstd, long P, int C, int C_real, const float* part, int nb,
                               float* dgamma, float* dbeta, float* coef, int dtype, hipStream_t st) {
  SEGK_REQUIRE(dy && z && dz && scale && shift && mean && rstd && part && dgamma && dbeta && coef && nb > 0,
               "bn_bwd_from_part: bad arguments");
  SEGK_REQUIRE(P > 0 && C > 0 && C % 32 == 0 && C_real > 0 && C_real <= C, "bn_bwd_from_part: bad shape");
  return dtype == SEGK_DT_BF16
             ? bn_bwd_from_part_t<bf16_t>(dy, z, dz, scale, shift, mean, rstd, P, C, C_real, part, nb, dgamma, dbeta, coef, st)
             : bn_bwd_from_part_t<float>(dy, z, dz, scale, shift, mean, rstd, P, C, C_real, part, nb, dgamma, dbeta, coef, st);
}

int segk_bn_bwd_impl(const void* dy, const void* z, void* dz, const float* scale, const float* shift, const float* mean,
                     const float* rstd, long P, int C, int C_real, float* part, float* dgamma, float* dbeta, float* coef,
                     int dtype, hipStream_t st) {
  SEGK_REQUIRE(dy && z && dz && scale && shift && mean && rstd && part && dgamma && dbeta && coef,
               "bn_bwd: null pointer");
  SEGK_REQUIRE(P > 0 && C > 0 && C % 32 == 0 && C_real > 0 && C_real <= C, "bn_bwd: bad shape");
  return dtype == SEGK_DT_BF16
             ? bn_bwd_t<bf16_t>(dy, z, dz, scale, shift, mean, rstd, P, C, C_real, part, dgamma, dbeta, coef, st)
             : bn_bwd_t<float>(dy, z, dz, scale, shift, mean, rstd, P, C, C_real, part, dgamma, dbeta, coef, st);
}

int segk_maxpool_fwd_impl(const void* x, void* y, int B, int H, int W, int C, int dtype, hipStream_t st) {
  SEGK_REQUIRE(x && y && B > 0 && H >= 2 && W >= 2 && C > 0 && C % 32 == 0, "maxpool_fwd: bad arguments");
  const int vec = dtype == SEGK_DT_BF16 ? 8 : 4;
  long total = (long)B * (H / 2) * (W / 2) * (C / vec);
  SEGK_REQUIRE(total < (1L << 31), "maxpool_fwd: more than 2^31 windows x channel vectors");
  long g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  if (dtype == SEGK_DT_BF16)
    hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3((int)g), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, B, H, W, C);
  else
    hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3((int)g), dim3(256), 0, st, (const float*)x, (float*)y, B, H, W, C);
  SEGK_CHECK_LAUNCH("maxpool_fwd");
  return 0;
}

int segk_maxpool_bwd_impl(const void* x, const void* dy, void* dx, int B, int H, int W, int C, int accumulate, int dtype,
                          hipStream_t st) {
  SEGK_REQUIRE(x && dy && dx && B > 0 && H >= 2 && W >= 2 && C > 0 && C % 32 == 0, "maxpool_bwd: bad arguments");
  const int vec = dtype == SEGK_DT_BF16 ? 8 : 4;
  long total = (long)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / vec);
  SEGK_REQUIRE(total < (1L << 31), "maxpool_bwd: more than 2^31 windows x channel vectors");
  long g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  if (dtype == SEGK_DT_BF16)
    hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t, false>), dim3((int)g), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy,
                       (bf16_t*)dx, B, H, W, C, accumulate, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  else
    hipLaunchKernelGGL((maxpool_bwd_kernel<float, false>), dim3((int)g), dim3(256), 0, st, (const float*)x, (const float*)dy,
                       (float*)dx, B, H, W, C, accumulate, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  SEGK_CHECK_LAUNCH("maxpool_bwd");
  return 0;
}

// blocks (= partial rows) of the fused pooling-backward + BatchNorm-reduce kernel, or 0 when the shape is not served
int segk_maxpool_bwd_stat_blocks(int B, int H, int W, int C, int dtype) {
  const int vec = dtype == SEGK_DT_BF16 ? 8 : 4;
  if (B <= 0 || H < 2 || W < 2 || C <= 0 || C % 32 != 0) return 0;
  const int cv = C / vec;
  if ((cv & (cv - 1)) != 0 || cv > 256) return 0;
  long total = (long)B * ((H + 1) / 2) * ((W + 1) / 2) * cv;
  if (total >= (1L << 31)) return 0;                  // the kernels index windows x channel vectors with 32 bits
  long g = (total + 255) / 256;
  return (int)(g > 1024 ? 1024 : g);
}

int segk_maxpool_bwd_bnstat_impl(const void* x, const void* dy, void* dx, int B, int H, int W, int C, int accumulate,
                                 const float* scale, const float* shift, const float* mean, const float* rstd, float* part,
                                 const void* z, int dtype, hipStream_t st) {
  SEGK_REQUIRE(x && dy && dx && scale && shift && mean && rstd && part, "maxpool_bwd_bnstat: null pointer");
  const int g = segk_maxpool_bwd_stat_blocks(B, H, W, C, dtype);
  SEGK_REQUIRE(g > 0, "maxpool_bwd_bnstat: shape not served (channel vectors must be a power of two)");
  if (dtype == SEGK_DT_BF16)
    hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t, true>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy,
                       (bf16_t*)dx, B, H, W, C, accumulate, scale, shift, mean, rstd, part, (const bf16_t*)z);
  else
    hipLaunchKernelGGL((maxpool_bwd_kernel<float, true>), dim3(g), dim3(256), 0, st, (const float*)x, (const float*)dy,
                       (float*)dx, B, H, W, C, accumulate, scale, shift, mean, rstd, part, (const float*)z);
  SEGK_CHECK_LAUNCH("maxpool_bwd_bnstat");
  return 0;
}

int segk_channel_sum_impl(const void* x, long P, int C, int C_real, float* part, float* out, int dtype, hipStream_t st) {
  SEGK_REQUIRE(x && part && out && P > 0 && C > 0 && C % 32 == 0 && C_real > 0 && C_real <= C, "channel_sum: bad arguments");
  const int vec = dtype == SEGK_DT_BF16 ? 8 : 4;
  int cvb, rows, gy;
  lane_geometry(C, vec, &cvb, &rows, &gy);
  const int gx = segk_bn_bwd_blocks(P, C, dtype);
  const size_t lds = (size_t)rows * cvb * vec * sizeof(float);
  if (dtype == SEGK_DT_BF16)
    hipLaunchKernelGGL(channel_sum_kernel<bf16_t>, dim3(gx, gy), dim3(256), lds, st, (const bf16_t*)x, P, C, cvb, rows, part);
  else
    hipLaunchKernelGGL(channel_sum_kernel<float>, dim3(gx, gy), dim3(256), lds, st, (const float*)x, P, C, cvb, rows, part);
  SEGK_CHECK_LAUNCH("channel_sum");
  hipLaunchKernelGGL(channel_sum_finalize_kernel, dim3(C / 32), dim3(1024), 0, st, part, gx, C, C_real, out);
  SEGK_CHECK_LAUNCH("channel_sum_finalize");
  return 0;
}
