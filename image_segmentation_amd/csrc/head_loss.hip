// Output head (1x1 Conv2d to a handful of classes), per-pixel CrossEntropy + soft-Dice loss, and the
// argmax/confusion-count metric kernel.  All HBM-bound; logits are fp32 NCHW exactly like the reference
// returns them (unet/unet.py:91,105; clip/clipunet.py:181,187).
//
// Loss semantics (one fused pass over logits + labels):
//   CE   : nn.CrossEntropyLoss(mean, optional weight / ignore_index)  -- utils/training.py:47,
//          utils/weighted_loss.py:132-138,163:  sum_p w[y_p] * nll_p / sum_p w[y_p] over non-ignored pixels
//   Dice : utils/weighted_loss.py:31-98: p = softmax; per class I = sum p*onehot, Sp = sum p, Sg = sum onehot
//          over N,H,W; dc = (2I+s)/clip(Sp+Sg+s, 1e-8); (weighted) mean over non-ignored classes; loss = -mean
//   combined = dice_weight * dice + ce_weight * ce   (weighted_loss.py:165)
// Metric: utils/MetricsHistory.py:65-75 (argmax -> first maximum, per-class TP/FP/FN/TN).
#include <type_traits>
#include "common.hpp"
#include "segk_internal.h"
#include "ticket.hpp"
#include "../../include/segk.h"

namespace {
constexpr int MAXC = 8;    // classes supported by the fused kernels
constexpr int HT = 256;    // pixels per head tile (one per thread)
constexpr int HCH = 64;    // channels staged per pass

// pixel index -> (image, offset inside the image) with ONE 32-bit division: a 64-bit one is a ~130-instruction loop, per
// pixel more than the rest of a loss kernel's arithmetic.  Every launcher below requires P < 2^31.
__device__ __forceinline__ void split_hw(long p, long HW, long& b, long& r) {
  const unsigned bb = (unsigned)p / (unsigned)HW;
  b = (long)bb;
  r = p - (long)bb * HW;
}

// ------------------------------------------------------------------------------------------------
// logits[b][k][y][x] = bias[k] + sum_c y[p][c] * Wt[k][c]
// NC = compiled class count (>= ncls): the per-class loops are unrolled to it, not to MAXC
template <typename T, int NC>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ y, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ logits,
                                                       long P, long HW, int Cp, int C, int ncls,
                                                       const float* __restrict__ zsc, const float* __restrict__ zsh) {
  // zsc != NULL: the input is the pre-activation z of the last DoubleConv block and y = relu(z * zsc + zsh), rounded to T
  // as segk_bn_relu_apply would have stored it, is formed on the way into LDS (the block's output is never written)
  using E = ET<T>;
  constexpr int PITCH = HCH * E::ES + 16;
  extern __shared__ __attribute__((aligned(16))) char tile[];   // HT * PITCH bytes
  __shared__ float ws[MAXC * HCH + 2 * HCH];
  float* const zs = ws + MAXC * HCH;                            // [2][HCH] scale, shift of this pass
  const int tid = threadIdx.x;
  for (long p0 = (long)blockIdx.x * HT; p0 < P; p0 += (long)gridDim.x * HT) {
    float acc[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) acc[k] = (k < ncls) ? bias[k] : 0.f;
    for (int c0 = 0; c0 < Cp; c0 += HCH) {
      __syncthreads();
      constexpr int VPR = HCH / E::VEC;  // 16-byte vectors per pixel row of this pass
      if (zsc != nullptr) {
        if (tid < 2 * HCH) {
          const int c = c0 + (tid & (HCH - 1));
          zs[tid] = c < Cp ? (tid < HCH ? zsc[c] : zsh[c]) : 0.f;
        }
        __syncthreads();
      }
      // all of the thread's vectors of this pass are fetched before the first is used, unconditionally, from a pixel /
      // channel index clamped into the tensor (left as a loop of conditional loads the compiler keeps ONE in flight)
      constexpr int NQ = HT * VPR / 256;
      uint4 raw[NQ];
#pragma unroll
      for (int u = 0; u < NQ; ++u) {
        const int q = tid + 256 * u, px = q / VPR, v = q - px * VPR;
        const long pp = p0 + px < P ? p0 + px : P - 1;
        const int cc = c0 + v * E::VEC < Cp ? c0 + v * E::VEC : 0;
        raw[u] = *(const uint4*)(y + (size_t)pp * Cp + cc);
      }
#pragma unroll
      for (int u = 0; u < NQ; ++u) {
        const int q = tid + 256 * u, px = q / VPR, v = q - px * VPR;
        uint4 d = make_uint4(0, 0, 0, 0);
        if (p0 + px < P && c0 + v * E::VEC < Cp) {
          d = raw[u];
          if (zsc != nullptr) {
            float f[E::VEC];
            unpack16<T>(d, f);
#pragma unroll
            for (int j = 0; j < E::VEC; ++j) f[j] = fmaxf(fmaf(f[j], zs[v * E::VEC + j], zs[HCH + v * E::VEC + j]), 0.f);
            d = pack16<T>(f);
          }
        }
        *(uint4*)(tile + px * PITCH + v * 16) = d;
      }
      for (int q = tid; q < MAXC * HCH; q += 256) {
        const int k = q / HCH, c = c0 + (q - k * HCH);
        ws[q] = (k < ncls && c < C) ? w[(size_t)k * C + c] : 0.f;
      }
      __syncthreads();
#pragma unroll 4
      for (int v = 0; v < VPR; ++v) {
        float f[E::VEC];
        unpack16<T>(*(const uint4*)(tile + tid * PITCH + v * 16), f);
#pragma unroll
        for (int j = 0; j < E::VEC; ++j)
#pragma unroll
          for (int k = 0; k < NC; ++k) acc[k] = fmaf(f[j], ws[k * HCH + v * E::VEC + j], acc[k]);
      }
    }
    const long p = p0 + tid;
    if (p < P) {
      long b, r; split_hw(p, HW, b, r);
#pragma unroll
      for (int k = 0; k < NC; ++k)
        if (k < ncls) logits[(b * ncls + k) * HW + r] = acc[k];
    }
  }
}

// dy[p][c] = sum_k dl[p][k] * W[k][c];  partial dW[k][c] = sum_p dl[p][k]*y[p][c], db[k] = sum_p dl[p][k].
// Pure streaming: a thread owns one 16-byte channel vector (its weights W[k][c..c+VEC) and its dW partials
// live in registers) and walks pixels; a row of threads covers whole contiguous pixel rows of y / dy.
// ZIN: `y` holds the pre-activation z of the last DoubleConv block: y = relu(z * scale + shift) is re-formed per value
// (rounded to T like the stored tensor would be) and the BatchNorm reductions take xhat = (z - mean) * rstd from z itself.
// Vector width of the head's backward pass.  bf16 threads own FOUR channels (an 8-byte vector) instead of the eight of a
// 16-byte one: the per-thread weight / gradient / BatchNorm arrays halve (176 -> ~100 registers), twice the waves fit a CU, and
// the pass -- whose arithmetic (~73 us) and traffic (~100 us) added up at 8 waves per CU -- overlaps them.
#ifndef HEAD_BWD_VEC
#define HEAD_BWD_VEC 4
#endif
template <typename T> struct HeadVec;
template <> struct HeadVec<float> {
  static constexpr int HV = 4;
  using Raw = uint4;
  static __device__ __forceinline__ void unpack(const Raw& v, float* f) { unpack16<float>(v, f); }
  static __device__ __forceinline__ Raw pack(const float* f) { return pack16<float>(f); }
};
template <> struct HeadVec<bf16_t> {
  static constexpr int HV = HEAD_BWD_VEC;
  static_assert(HV == 4 || HV == 8, "head backward: 4 or 8 bf16 channels per thread");
  using Raw = std::conditional_t<HV == 8, uint4, uint2>;
  static __device__ __forceinline__ void unpack(const uint4& v, float* f) { unpack16<bf16_t>(v, f); }
  static __device__ __forceinline__ void unpack(const uint2& v, float* f) {
    f[0] = bf2f(v.x & 0xffffu); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = bf2f(v.y & 0xffffu); f[3] = __uint_as_float(v.y & 0xffff0000u);
  }
  static __device__ __forceinline__ uint4 pack_n(const float* f, std::integral_constant<int, 8>) { return pack16<bf16_t>(f); }
  static __device__ __forceinline__ uint2 pack_n(const float* f, std::integral_constant<int, 4>) {
    return make_uint2(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]));
  }
  static __device__ __forceinline__ Raw pack(const float* f) { return pack_n(f, std::integral_constant<int, HV>{}); }
};

template <typename T, int NC, bool ZIN>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dlog, const T* __restrict__ y,
                                                       const float* __restrict__ w, T* __restrict__ dy,
                                                       float* __restrict__ part, long P, long HW, int Cp, int C,
                                                       int ncls, int cvb, int rows, const float* __restrict__ bn_scale,
                                                       const float* __restrict__ bn_shift, const float* __restrict__ bn_mean,
                                                       const float* __restrict__ bn_rstd, float* __restrict__ bnpart) {
  constexpr int HV = HeadVec<T>::HV;                   // channels per thread (bf16: 4 = an 8-byte vector)
  using Raw = typename HeadVec<T>::Raw;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = (float*)smem_raw;                       // [rows][cvb][MAXC*VEC + MAXC]
  // bnpart != NULL: y is the output relu(bn(z)) of the last DoubleConv block and dy its complete gradient: also accumulate
  // that BatchNorm's backward reductions sum(g), sum(g*xhat) (xhat recovered from y where the ReLU is active), like
  // maxpool_bwd_kernel<STAT> does for the Down blocks
  const bool stat = bnpart != nullptr;
  float xa[HV], xb[HV], sg[HV], sgx[HV];
#pragma unroll
  for (int j = 0; j < HV; ++j) { xa[j] = 0.f; xb[j] = 0.f; sg[j] = 0.f; sgx[j] = 0.f; }
  constexpr int RW = MAXC * HV + MAXC;
  const int cx = threadIdx.x % cvb, ry = threadIdx.x / cvb;
  const int cv = blockIdx.y * cvb + cx;
  const bool active = cv < Cp / HV && ry < rows;
  float wr[NC][HV], aw[NC][HV], ab[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    ab[k] = 0.f;
#pragma unroll
    for (int j = 0; j < HV; ++j) {
      const int c = cv * HV + j;
      wr[k][j] = (active && k < ncls && c < C) ? w[(size_t)k * C + c] : 0.f;
      aw[k][j] = 0.f;
    }
  }
  float zc[ZIN ? HV : 1], zh[ZIN ? HV : 1];     // ZIN: scale, shift (xa = rstd, xb = -mean * rstd)
  if (active && (stat || ZIN)) {
#pragma unroll
    for (int j = 0; j < HV; ++j) {
      const int c = cv * HV + j;
      const float sc = bn_scale[c], rs = bn_rstd[c];
      if constexpr (ZIN) {
        zc[j] = sc; zh[j] = bn_shift[c];
        xa[j] = rs; xb[j] = -bn_mean[c] * rs;
      } else {
        xa[j] = sc != 0.f ? rs / sc : 0.f;
        xb[j] = -bn_shift[c] * xa[j] - bn_mean[c] * rs;
      }
    }
  }
  if (active) {
    // one pixel: dy, the dW / db partials and (stat) the BatchNorm reductions from the raw input vector `raw`
    auto pixel = [&](long p, const float (&dl)[NC], const Raw raw) {
      float f[HV], z[HV], o[HV];
      HeadVec<T>::unpack(raw, f);
      if constexpr (ZIN) {
#pragma unroll
        for (int j = 0; j < HV; ++j) { z[j] = f[j]; f[j] = fmaxf(fmaf(f[j], zc[j], zh[j]), 0.f); }
        HeadVec<T>::unpack(HeadVec<T>::pack(f), f);               // the value segk_bn_relu_apply would have stored
      }
#pragma unroll
      for (int j = 0; j < HV; ++j) o[j] = 0.f;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        ab[k] += dl[k];
#pragma unroll
        for (int j = 0; j < HV; ++j) {
          o[j] = fmaf(dl[k], wr[k][j], o[j]);
          aw[k][j] = fmaf(dl[k], f[j], aw[k][j]);
        }
      }
      *(Raw*)(dy + (size_t)p * Cp + cv * HV) = HeadVec<T>::pack(o);
      if (stat) {
#pragma unroll
        for (int j = 0; j < HV; ++j) {
          const float gg = f[j] > 0.f ? o[j] : 0.f;
          sg[j] += gg;
          sgx[j] = fmaf(gg, fmaf(ZIN ? z[j] : f[j], xa[j], xb[j]), sgx[j]);
        }
      }
    };
    // (image, offset) of a pixel stream is carried along instead of divided out per pixel: no division in the loop
    auto load_dl = [&](long b, long r, float (&dl)[NC]) {
#pragma unroll
      for (int k = 0; k < NC; ++k) {   // unconditional loads (class index clamped), selected afterwards
        const float v = dlog[(b * ncls + (k < ncls ? k : ncls - 1)) * HW + r];
        dl[k] = (k < ncls) ? v : 0.f;
      }
    };
    auto advance = [&](long& b, long& r, long d) {
      r += d;
      while (r >= HW) { r -= HW; ++b; }
    };
    // PIF pixels per iteration: all their loads are in flight before the first is used (the loop is latency-bound otherwise);
    // every pixel stream carries its own (image, offset)
#ifndef HEAD_BWD_PIF
#define HEAD_BWD_PIF 2
#endif
    constexpr int PIF = HEAD_BWD_PIF;
    const long step = (long)gridDim.x * rows;
    long p = (long)blockIdx.x * rows + ry;
    long bS[PIF], rS[PIF];
    bS[0] = 0; rS[0] = 0;
    if (p < P) split_hw(p, HW, bS[0], rS[0]);
#pragma unroll
    for (int u = 1; u < PIF; ++u) { bS[u] = bS[u - 1]; rS[u] = rS[u - 1]; advance(bS[u], rS[u], step); }
    for (; p + (PIF - 1) * step < P; p += PIF * step) {
      float dl[PIF][NC];
      Raw rw[PIF];
#pragma unroll
      for (int u = 0; u < PIF; ++u) rw[u] = *(const Raw*)(y + (size_t)(p + u * step) * Cp + cv * HV);
#pragma unroll
      for (int u = 0; u < PIF; ++u) load_dl(bS[u], rS[u], dl[u]);
#pragma unroll
      for (int u = 0; u < PIF; ++u) pixel(p + u * step, dl[u], rw[u]);
#pragma unroll
      for (int u = 0; u < PIF; ++u) advance(bS[u], rS[u], PIF * step);
    }
    for (; p < P; p += step) {                      // tail: one pixel at a time, stream 0 walks on by one step
      float dl0[NC];
      const Raw r0 = *(const Raw*)(y + (size_t)p * Cp + cv * HV);
      load_dl(bS[0], rS[0], dl0);
      pixel(p, dl0, r0);
      advance(bS[0], rS[0], step);
    }
  }
  if (ry < rows) {
    float* q = red + ((size_t)ry * cvb + cx) * RW;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
#pragma unroll
      for (int j = 0; j < HV; ++j) q[k * HV + j] = aw[k][j];
      q[MAXC * HV + k] = ab[k];
    }
  }
  __syncthreads();
  if (ry == 0 && cv < Cp / HV) {
    // part layout: [block][MAXC][Cp + 1]  (last column = bias gradient, written by channel vector 0)
    float* dst = part + (size_t)blockIdx.x * MAXC * (Cp + 1);
#pragma unroll
    for (int k = 0; k < NC; ++k) {
#pragma unroll
      for (int j = 0; j < HV; ++j) {
        float s = 0.f;
        for (int r2 = 0; r2 < rows; ++r2) s += red[((size_t)r2 * cvb + cx) * RW + k * HV + j];   // fixed order
        dst[k * (Cp + 1) + cv * HV + j] = s;
      }
      if (cv == 0) {
        float s = 0.f;
        for (int r2 = 0; r2 < rows; ++r2) s += red[((size_t)r2 * cvb + cx) * RW + MAXC * HV + k];
        dst[k * (Cp + 1) + Cp] = s;
      }
    }
  }
  if (stat) {
    __syncthreads();
    if (ry < rows) {
      float* q = red + ((size_t)ry * cvb + cx) * RW;
#pragma unroll
      for (int j = 0; j < HV; ++j) { q[j] = sg[j]; q[HV + j] = sgx[j]; }
    }
    __syncthreads();
    if (ry == 0 && cv < Cp / HV) {
#pragma unroll
      for (int j = 0; j < HV; ++j) {
        float a = 0.f, b2 = 0.f;
        for (int r2 = 0; r2 < rows; ++r2) {   // fixed order
          const float* q = red + ((size_t)r2 * cvb + cx) * RW;
          a += q[j];
          b2 += q[HV + j];
        }
        ((float2*)bnpart)[(size_t)blockIdx.x * Cp + cv * HV + j] = make_float2(a, b2);
      }
    }
  }
}

// one 64-lane wave per output element: lanes stride the NB block partials, fixed-order tree reduce
__global__ __launch_bounds__(256) void head_bwd_finalize_kernel(const float* __restrict__ part, int NB, int Cp, int C,
                                                                int ncls, float* __restrict__ dw,
                                                                float* __restrict__ db) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= ncls * (C + 1)) return;
  const int k = i / (C + 1), c = i - k * (C + 1);
  const int col = (c == C) ? Cp : c;
  double s = 0.0;
  for (int b = lane; b < NB; b += 64) s += (double)part[(size_t)b * MAXC * (Cp + 1) + k * (Cp + 1) + col];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane != 0) return;
  if (c == C) db[k] = (float)s;
  else dw[(size_t)k * C + c] = (float)s;
}

// ------------------------------------------------------------------------------------------------
// loss partials per block: [0]=sum w*nll, [1]=sum w, then I[MAXC], Sp[MAXC], Sg[MAXC]
constexpr int LP = 2 + 3 * MAXC;

__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// state: [0]=loss [1]=ce [2]=dice(-mean dc) [3]=ce_den, [4..4+MAXC) dc, [.. ) den_raw(Sp+Sg+smooth), [..) a_k
constexpr int LS = 4 + 3 * MAXC;
// runs in ONE block of LT threads: the block of loss_fwd_kernel that finished last (ticket.hpp).  NB <= LROWS partial rows:
// every thread takes one column of LROWS / (LT / 32) rows, all loads in flight at once (ONE memory round trip: the rows were
// written by other XCDs and come from memory); additions in one fixed order: bit-stable
constexpr int LT = 1024, LROWS = 256;
__device__ __forceinline__ void loss_finalize_block(const float* __restrict__ part, int NB, int C, const float* cw,
                                                    int ignore_index, float smooth, float dice_weight, float ce_weight,
                                                    float* __restrict__ state, float* __restrict__ loss_out) {
  __shared__ double tot[LP];
  __shared__ double sh[LT / 32][32];
  static_assert(LP <= 32, "one column lane per partial");
  constexpr int RG = LT / 32, FL = LROWS / RG;
  const int t = threadIdx.x, cx = t & 31, ry = t >> 5;
  double acc = 0.0;
  if (cx < LP) {
    float v[FL];
#pragma unroll
    for (int u = 0; u < FL; ++u) {   // unconditional loads of a clamped row (a conditional load compiles to branch + wait)
      const int r = ry + RG * u < NB ? ry + RG * u : NB - 1;
      v[u] = part[(unsigned)r * LP + cx];
    }
#pragma unroll
    for (int u = 0; u < FL; ++u) acc += (ry + RG * u < NB) ? (double)v[u] : 0.0;
  }
  sh[ry][cx] = acc;
  __syncthreads();
  if (t < LP) {
    double s = 0.0;
    for (int r = 0; r < RG; ++r) s += sh[r][t];   // fixed order
    tot[t] = s;
  }
  __syncthreads();
  if (t != 0) return;
  const double ce = tot[1] > 0.0 ? tot[0] / tot[1] : NAN;   // all pixels ignored -> NaN like torch
  double wsum = 0.0, dsum = 0.0;
  int nvalid = 0;
  for (int k = 0; k < C; ++k) {
    const double den_raw = tot[2 + MAXC + k] + tot[2 + 2 * MAXC + k] + (double)smooth;
    const double den = den_raw < 1e-8 ? 1e-8 : den_raw;
    const double dc = (2.0 * tot[2 + k] + (double)smooth) / den;
    state[4 + k] = (float)dc;
    state[4 + MAXC + k] = (float)den_raw;
    const bool valid = !(ignore_index >= 0 && ignore_index < C && k == ignore_index);
    if (valid) {
      const double wk = cw ? (double)cw[k] : 1.0;
      wsum += wk; dsum += dc * wk; ++nvalid;
    }
  }
  if (cw && wsum < 1e-8) wsum = 1e-8;
  const double dice = -(dsum / wsum);
  for (int k = 0; k < C; ++k) {
    const bool valid = !(ignore_index >= 0 && ignore_index < C && k == ignore_index);
    state[4 + 2 * MAXC + k] = valid ? (float)((cw ? (double)cw[k] : 1.0) / wsum) : 0.f;
  }
  state[0] = (float)((double)dice_weight * dice + (double)ce_weight * ce);
  if (loss_out) *loss_out = state[0];
  state[1] = (float)ce;
  state[2] = (float)dice;
  state[3] = (float)tot[1];
  (void)nvalid;
}

// PROB: the input already holds class probabilities (prompt model, prompt_based/prompt.py:33-56): Dice on the values
// themselves (weighted_loss.py:206-209 with apply_softmax=False) and NLLLoss on nll_nonlin(x) = log(x + eps)
// (nll_log, prompt.ipynb's stable_log) or on x itself (weighted_loss.py:338-340)
// NC = compiled class count (>= C): the per-class work and the accumulators are sized to it, not to MAXC.  Two pixels per
// thread are in flight per pass (the walk is latency-bound otherwise); additions stay in one fixed order: bit-stable.
template <bool PROB, int NC>
__global__ __launch_bounds__(LT) void loss_fwd_kernel(const float* __restrict__ logits,
                                                       const long long* __restrict__ labels,
                                                       const float* __restrict__ cw, long P, long HW, int C,
                                                       int ignore_index, float* __restrict__ part, int nll_log, float eps,
                                                       float smooth, float dice_weight, float ce_weight,
                                                       float* __restrict__ state, float* __restrict__ loss_out,
                                                       unsigned* __restrict__ ticket) {
  __shared__ float sh[LT / 64][LP];
  __shared__ int last;
  float a0 = 0.f, a1 = 0.f, aI[NC], aP[NC], aG[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) { aI[k] = 0.f; aP[k] = 0.f; aG[k] = 0.f; }
  auto fetch = [&](long p, float (&raw)[NC], long long& y) {
    long b, r; split_hw(p, HW, b, r);
#pragma unroll
    for (int k = 0; k < NC; ++k) {   // unconditional loads (class index clamped), selected afterwards
      const float v = logits[(b * C + (k < C ? k : C - 1)) * HW + r];
      raw[k] = (k < C) ? v : -INFINITY;
    }
    y = labels[p];
  };
  auto pixel = [&](const float (&raw)[NC], const long long y) {
    float l[NC], m = -INFINITY;
#pragma unroll
    for (int k = 0; k < NC; ++k) m = fmaxf(m, raw[k]);
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      if constexpr (PROB) l[k] = (k < C) ? raw[k] : 0.f;
      else l[k] = (k < C) ? expf(raw[k] - m) : 0.f;
      se += l[k];
    }
    const float inv = PROB ? 1.f : 1.f / se, lse = PROB ? 0.f : logf(se);
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const float pk = l[k] * inv;
      const float oh = (y == k) ? 1.f : 0.f;
      aI[k] += pk * oh;
      aP[k] += pk;
      aG[k] += oh;
      if (y == k && k < C && y != ignore_index) {
        const float wy = cw ? cw[k] : 1.f;
        if constexpr (PROB) a0 += wy * (nll_log ? -logf(raw[k] + eps) : -raw[k]);
        else a0 += wy * (lse - (raw[k] - m));   // -log softmax = log(sum exp) - (logit - max)
        a1 += wy;
      }
    }
  };
  // four pixels per thread in flight per pass (the walk is latency-bound otherwise); one fixed order of additions
  const long step = (long)gridDim.x * LT;
  long p = (long)blockIdx.x * LT + threadIdx.x;
  for (; p + 3 * step < P; p += 4 * step) {
    float r0[NC], r1[NC], r2[NC], r3[NC];
    long long y0, y1, y2, y3;
    fetch(p, r0, y0);
    fetch(p + step, r1, y1);
    fetch(p + 2 * step, r2, y2);
    fetch(p + 3 * step, r3, y3);
    pixel(r0, y0);
    pixel(r1, y1);
    pixel(r2, y2);
    pixel(r3, y3);
  }
  for (; p < P; p += step) {
    float r0[NC];
    long long y0;
    fetch(p, r0, y0);
    pixel(r0, y0);
  }
  // wave sums by butterfly, the wave rows through LDS, one row of LP partials per block (unused class slots = 0)
  float v[LP];
#pragma unroll
  for (int i = 0; i < LP; ++i) v[i] = 0.f;
  v[0] = a0; v[1] = a1;
#pragma unroll
  for (int k = 0; k < NC; ++k) { v[2 + k] = aI[k]; v[2 + MAXC + k] = aP[k]; v[2 + 2 * MAXC + k] = aG[k]; }
#pragma unroll
  for (int i = 0; i < LP; ++i) {
    if (i >= 2 && ((i - 2) % MAXC) >= NC) continue;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[i] += __shfl_xor(v[i], o);
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int i = 0; i < LP; ++i) sh[threadIdx.x >> 6][i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x < LP) {
    const int i = threadIdx.x;
    float s2 = 0.f;
#pragma unroll
    for (int w = 0; w < LT / 64; ++w) s2 += sh[w][i];   // fixed order
    part[(size_t)blockIdx.x * LP + i] = s2;
  }
  // the block that finishes last turns the partial rows into the loss (no second launch, nobody waits)
  if (last_arriver(ticket, gridDim.x, &last))
    loss_finalize_block(part, (int)gridDim.x, C, cw, ignore_index, smooth, dice_weight, ce_weight, state, loss_out);
}


template <bool PROB, int NC>
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ logits,
                                                       const long long* __restrict__ labels,
                                                       const float* __restrict__ cw, const float* __restrict__ state,
                                                       const float* __restrict__ gout, long P, long HW, int C,
                                                       int ignore_index, float dice_weight, float ce_weight,
                                                       float* __restrict__ dlogits, int nll_log, float eps) {
  const float go = gout[0];
  const float ce_den = state[3];
  float G0[NC], G1[NC];   // dL_dice/dp_k = G0 + onehot*G1
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    G0[k] = 0.f; G1[k] = 0.f;
    if (k < C) {
      const float dc = state[4 + k], den_raw = state[4 + MAXC + k], ak = state[4 + 2 * MAXC + k];
      if (den_raw < 1e-8f) { G1[k] = -ak * 2.f / 1e-8f; }           // clip active: denominator constant
      else { G0[k] = ak * dc / den_raw; G1[k] = -ak * 2.f / den_raw; }
    }
  }
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
    long b, r; split_hw(p, HW, b, r);
    float l[NC], m = -INFINITY;
#pragma unroll
    for (int k = 0; k < NC; ++k) {   // unconditional loads (class index clamped), selected afterwards
      const float v = logits[(b * C + (k < C ? k : C - 1)) * HW + r];
      l[k] = (k < C) ? v : -INFINITY;
      m = fmaxf(m, l[k]);
    }
    const long long y = labels[p];
    const bool ce_valid = (y >= 0 && y < C && y != ignore_index);
    float wy = 0.f;
    if (ce_valid) wy = (cw ? cw[y] : 1.f) / ce_den;
    if constexpr (PROB) {   // d/dx of  dice(x) + nll(log(x + eps) | x)
#pragma unroll
      for (int k = 0; k < NC; ++k)
        if (k < C) {
          const bool hit = (y == k);
          const float dd = G0[k] + (hit ? G1[k] : 0.f);
          const float dn = hit ? (nll_log ? -wy / (l[k] + eps) : -wy) : 0.f;
          dlogits[(b * C + k) * HW + r] = go * (dice_weight * dd + ce_weight * dn);
        }
      continue;
    }
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) { l[k] = (k < C) ? expf(l[k] - m) : 0.f; se += l[k]; }
    const float inv = 1.f / se;
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      l[k] *= inv;
      dot += l[k] * (G0[k] + ((y == k) ? G1[k] : 0.f));
    }
#pragma unroll
    for (int k = 0; k < NC; ++k)
      if (k < C) {
        const float gk = G0[k] + ((y == k) ? G1[k] : 0.f);
        const float dd = l[k] * (gk - dot);
        const float dce = wy * (l[k] - ((y == k) ? 1.f : 0.f));
        dlogits[(b * C + k) * HW + r] = go * (dice_weight * dd + ce_weight * dce);
      }
  }
}

// confusion matrix M[pred][label] (uint64) over [N][C][HW] logits; argmax = FIRST maximum (torch.argmax)
__global__ __launch_bounds__(256) void confusion_kernel(const float* __restrict__ logits,
                                                        const long long* __restrict__ labels, long P, long HW, int C,
                                                        unsigned long long* __restrict__ M) {
  __shared__ unsigned int hist[MAXC * MAXC];
  if (threadIdx.x < MAXC * MAXC) hist[threadIdx.x] = 0;
  __syncthreads();
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
    long b, r; split_hw(p, HW, b, r);
    int best = 0;
    float bv = logits[(b * C) * HW + r];
    for (int k = 1; k < C; ++k) {
      const float v = logits[(b * C + k) * HW + r];
      if (v > bv || (v != v && bv == bv)) { bv = v; best = k; }   // NaN counts as maximal, like torch
    }
    const long long y = labels[p];
    if (y >= 0 && y < C) atomicAdd(&hist[best * MAXC + (int)y], 1u);
  }
  __syncthreads();
  if (threadIdx.x < MAXC * MAXC && hist[threadIdx.x])
    atomicAdd(&M[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
}

}  // namespace

template <typename T> static size_t head_lds() { return (size_t)HT * (HCH * ET<T>::ES + 16); }
static bool raise_lds(const void* f) {
  return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
}

// ------------------------------------------------------------------------------------------------
#ifndef SEGK_HEAD_BLOCKS_CAP
#define SEGK_HEAD_BLOCKS_CAP 1024
#endif
int segk_head_blocks(long P) {
  if (P <= 0) return 0;
  long g = (P + 31) / 32;
  return (int)(g > SEGK_HEAD_BLOCKS_CAP ? SEGK_HEAD_BLOCKS_CAP : g);
}
int segk_head_part_floats(long P, int Cp) {
  if (P <= 0 || Cp <= 0) return 0;
  const long long n = (long long)segk_head_blocks(P) * MAXC * ((long long)Cp + 1);
  return n > 0x7fffffffLL ? 0 : (int)n;
}

int segk_head_fwd_impl(const void* y, const float* w, const float* bias, float* logits, int B, int H, int W, int Cp,
                       int C, int ncls, const float* zsc, const float* zsh, int dtype, hipStream_t st) {
  SEGK_REQUIRE(y && w && bias && logits && B > 0 && H > 0 && W > 0, "head_fwd: bad arguments");
  SEGK_REQUIRE((zsc == nullptr) == (zsh == nullptr), "head_fwd: scale and shift come together");
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "head_fwd: bad dtype %d", dtype);
  SEGK_REQUIRE(ncls >= 1 && ncls <= MAXC, "head_fwd: 1..%d classes supported, got %d", MAXC, ncls);
  SEGK_REQUIRE(Cp % 32 == 0 && C > 0 && C <= Cp, "head_fwd: bad channels");
  const long P = (long)B * H * W, HW = (long)H * W;
  SEGK_REQUIRE(P < (1L << 31), "head / loss kernels index pixels with 32 bits: %ld pixels", P);
  long g = (P + HT - 1) / HT;
  if (g > 4096) g = 4096;
  // kernels are compiled for 2, 3, 4 and MAXC classes: the smallest that holds ncls
  auto launch = [&](auto Tc, auto NCc) {
    using T = decltype(Tc);
    constexpr int NC = decltype(NCc)::value;
    auto kern = head_fwd_kernel<T, NC>;
    if (!raise_lds((const void*)kern)) return false;
    hipLaunchKernelGGL(kern, dim3((int)g), dim3(256), head_lds<T>(), st, (const T*)y, w, bias, logits, P, HW, Cp, C, ncls, zsc,
                       zsh);
    return true;
  };
  auto by_nc = [&](auto Tc) {
    if (ncls <= 2) return launch(Tc, std::integral_constant<int, 2>{});
    if (ncls == 3) return launch(Tc, std::integral_constant<int, 3>{});
    if (ncls == 4) return launch(Tc, std::integral_constant<int, 4>{});
    return launch(Tc, std::integral_constant<int, MAXC>{});
  };
  const bool ok = dtype == SEGK_DT_BF16 ? by_nc(bf16_t{}) : by_nc(float{});
  SEGK_REQUIRE(ok, "head_fwd: cannot raise dynamic LDS limit");
  SEGK_CHECK_LAUNCH("head_fwd");
  return 0;
}

template <typename T>
static int head_bwd_t(const float* dlog, const void* y, const float* w, void* dy, float* part, float* dw, float* db,
                      long P, long HW, int Cp, int C, int ncls, const float* const* bn, float* bnpart, bool zin,
                      hipStream_t st) {
  constexpr int HV = HeadVec<T>::HV;
  const int cvec = Cp / HV;
  const int cvb = cvec < 64 ? cvec : 64;
  const int rows = 256 / cvb, gy = cdiv(cvec, cvb);
  const int nb = segk_head_blocks(P);
  const size_t lds = (size_t)rows * cvb * (MAXC * HV + MAXC) * sizeof(float);
  // gy > 1 (more than 64 channel vectors: above 256 channels): blockIdx.y slices the channels; every slice walks the
  // block's pixels and writes its own columns of the partial rows (tests/test_gpu_kernels.py at 288 and 512 channels)
  auto launch_z = [&](auto NCc, auto ZINc) {
    constexpr int NC = decltype(NCc)::value;
    auto kern = head_bwd_kernel<T, NC, decltype(ZINc)::value>;
    if (!raise_lds((const void*)kern)) return false;
    hipLaunchKernelGGL(kern, dim3(nb, gy), dim3(256), lds, st, dlog, (const T*)y, w, (T*)dy, part, P, HW, Cp, C, ncls, cvb, rows,
                       bn ? bn[0] : nullptr, bn ? bn[1] : nullptr, bn ? bn[2] : nullptr, bn ? bn[3] : nullptr, bnpart);
    return true;
  };
  auto launch = [&](auto NCc) { return zin ? launch_z(NCc, std::true_type{}) : launch_z(NCc, std::false_type{}); };
  const bool ok = ncls <= 2 ? launch(std::integral_constant<int, 2>{})
                : ncls == 3 ? launch(std::integral_constant<int, 3>{})
                : ncls == 4 ? launch(std::integral_constant<int, 4>{})
                            : launch(std::integral_constant<int, MAXC>{});
  SEGK_REQUIRE(ok, "head_bwd: cannot raise dynamic LDS limit");
  SEGK_CHECK_LAUNCH("head_bwd");
  hipLaunchKernelGGL(head_bwd_finalize_kernel, dim3(cdiv(ncls * (C + 1), 4)), dim3(256), 0, st, part, nb, Cp, C, ncls, dw, db);
  SEGK_CHECK_LAUNCH("head_bwd_finalize");
  return 0;
}

int segk_head_bwd_impl(const float* dlog, const void* y, const float* w, void* dy, float* part, float* dw, float* db,
                       int B, int H, int W, int Cp, int C, int ncls, const float* bn_scale, const float* bn_shift,
                       const float* bn_mean, const float* bn_rstd, float* bnpart, int zin, int dtype, hipStream_t st) {
  SEGK_REQUIRE(dlog && y && w && dy && part && dw && db && B > 0 && H > 0 && W > 0, "head_bwd: bad arguments");
  SEGK_REQUIRE(ncls >= 1 && ncls <= MAXC && Cp % 32 == 0 && C > 0 && C <= Cp, "head_bwd: bad channels/classes");
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "head_bwd: bad dtype %d", dtype);
  SEGK_REQUIRE(!(bnpart || zin) || (bn_scale && bn_shift && bn_mean && bn_rstd),
               "head_bwd: BatchNorm reductions / a pre-activation input need scale, shift, mean and rstd");
  const long P = (long)B * H * W, HW = (long)H * W;
  SEGK_REQUIRE(P < (1L << 31), "head / loss kernels index pixels with 32 bits: %ld pixels", P);
  const float* bn[4] = {bn_scale, bn_shift, bn_mean, bn_rstd};
  const float* const* bnp = (bnpart || zin) ? bn : nullptr;
  return dtype == SEGK_DT_BF16 ? head_bwd_t<bf16_t>(dlog, y, w, dy, part, dw, db, P, HW, Cp, C, ncls, bnp, bnpart, zin != 0, st)
                               : head_bwd_t<float>(dlog, y, w, dy, part, dw, db, P, HW, Cp, C, ncls, bnp, bnpart, zin != 0, st);
}
int segk_head_blocks_q(long P) { return segk_head_blocks(P); }

// ---- prompt model remix (prompt_based/prompt.py:33-56), 4 CLIP classes x 1 mask channel, fp32 NCHW ----
//   p = softmax(clip_logits), m = sigmoid(mask_logit)
//   final[0] = 1 - m;  final[1] = m*p0 + m*p3;  final[2] = m*p1;  final[3] = m*p2
__global__ __launch_bounds__(256) void prompt_mix_fwd_kernel(const float* __restrict__ clip, const float* __restrict__ mask,
                                                             float* __restrict__ out, long P, long HW) {
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
    long b, r; split_hw(p, HW, b, r);
    float l[4], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) { l[k] = clip[(b * 4 + k) * HW + r]; mx = fmaxf(mx, l[k]); }
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { l[k] = expf(l[k] - mx); se += l[k]; }
    const float inv = 1.f / se;
    const float m = 1.f / (1.f + expf(-mask[p]));
    float sel[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) sel[k] = m * (l[k] * inv);
    out[(b * 4 + 0) * HW + r] = 1.0f - m;
    out[(b * 4 + 1) * HW + r] = sel[0] + sel[3];
    out[(b * 4 + 2) * HW + r] = sel[1];
    out[(b * 4 + 3) * HW + r] = sel[2];
  }
}
// gradient w.r.t. the mask logit (the CLIP branch is frozen, prompt.py:30-31)
__global__ __launch_bounds__(256) void prompt_mix_bwd_kernel(const float* __restrict__ clip, const float* __restrict__ mask,
                                                             const float* __restrict__ dout, float* __restrict__ dmask,
                                                             long P, long HW) {
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
    long b, r; split_hw(p, HW, b, r);
    float l[4], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) { l[k] = clip[(b * 4 + k) * HW + r]; mx = fmaxf(mx, l[k]); }
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { l[k] = expf(l[k] - mx); se += l[k]; }
    const float inv = 1.f / se;
    const float m = 1.f / (1.f + expf(-mask[p]));
    const float d0 = dout[(b * 4 + 0) * HW + r], d1 = dout[(b * 4 + 1) * HW + r];
    const float d2 = dout[(b * 4 + 2) * HW + r], d3 = dout[(b * 4 + 3) * HW + r];
    const float dm = -d0 + d1 * ((l[0] + l[3]) * inv) + d2 * (l[1] * inv) + d3 * (l[2] * inv);
    dmask[p] = dm * m * (1.f - m);
  }
}

int segk_prompt_mix_impl(const float* clip, const float* mask, const float* dout, float* out, int N, long HW, hipStream_t st) {
  SEGK_REQUIRE(clip && mask && out && N > 0 && HW > 0, "prompt_mix: bad arguments");
  const long P = (long)N * HW;
  SEGK_REQUIRE(P < (1L << 31), "head / loss kernels index pixels with 32 bits: %ld pixels", P);
  long g = (P + 255) / 256;
  if (g > 4096) g = 4096;
  if (dout) hipLaunchKernelGGL(prompt_mix_bwd_kernel, dim3((int)g), dim3(256), 0, st, clip, mask, dout, out, P, HW);
  else hipLaunchKernelGGL(prompt_mix_fwd_kernel, dim3((int)g), dim3(256), 0, st, clip, mask, out, P, HW);
  SEGK_CHECK_LAUNCH("prompt_mix");
  return 0;
}

int segk_loss_blocks(long P) {
  if (P <= 0) return 0;
  long g = (P + 4 * LT - 1) / (4 * LT);           // four pixels per thread per pass, at most LROWS partial rows
  return (int)(g > LROWS ? LROWS : g < 1 ? 1 : g);
}
int segk_loss_part_floats(long P) { return segk_loss_blocks(P) * LP; }
int segk_loss_state_floats(void) { return LS; }

int segk_loss_fwd_impl(const float* logits, const long long* labels, const float* cw, int N, int C, long HW,
                       int ignore_index, float smooth, float dice_weight, float ce_weight, float* part, float* state,
                       float* loss_out, int prob, int nll_log, float eps, hipStream_t st) {
  SEGK_REQUIRE(logits && labels && part && state && N > 0 && HW > 0, "loss_fwd: bad arguments");
  SEGK_REQUIRE(C >= 1 && C <= MAXC, "loss_fwd: 1..%d classes supported, got %d", MAXC, C);
  const long P = (long)N * HW;
  SEGK_REQUIRE(P < (1L << 31), "head / loss kernels index pixels with 32 bits: %ld pixels", P);
  const int nb = segk_loss_blocks(P);
  unsigned* const ticket = segk_ticket_slot(1, st);
  SEGK_REQUIRE(ticket != nullptr, "loss_fwd: no ticket array");
  auto launch = [&](auto PROBc, auto NCc) {
    constexpr bool PR = decltype(PROBc)::value;
    constexpr int NC = decltype(NCc)::value;
    hipLaunchKernelGGL((loss_fwd_kernel<PR, NC>), dim3(nb), dim3(LT), 0, st, logits, labels, cw, P, HW, C, ignore_index, part,
                       PR ? nll_log : 0, PR ? eps : 0.f, smooth, dice_weight, ce_weight, state, loss_out, ticket);
  };
  auto by_nc = [&](auto PROBc) {
    if (C <= 2) launch(PROBc, std::integral_constant<int, 2>{});
    else if (C <= 4) launch(PROBc, std::integral_constant<int, 4>{});
    else launch(PROBc, std::integral_constant<int, MAXC>{});
  };
  if (prob) by_nc(std::true_type{});
  else by_nc(std::false_type{});
  SEGK_CHECK_LAUNCH("loss_fwd");
  return 0;
}

int segk_loss_bwd_impl(const float* logits, const long long* labels, const float* cw, const float* state,
                       const float* gout, int N, int C, long HW, int ignore_index, float dice_weight, float ce_weight,
                       float* dlogits, int prob, int nll_log, float eps, hipStream_t st) {
  SEGK_REQUIRE(logits && labels && state && gout && dlogits && N > 0 && HW > 0 && C >= 1 && C <= MAXC, "loss_bwd: bad arguments");
  const long P = (long)N * HW;
  SEGK_REQUIRE(P < (1L << 31), "head / loss kernels index pixels with 32 bits: %ld pixels", P);
  long g = (P + 255) / 256;
  if (g > 4096) g = 4096;
  auto launch = [&](auto PROBc, auto NCc) {
    constexpr bool PR = decltype(PROBc)::value;
    constexpr int NC = decltype(NCc)::value;
    hipLaunchKernelGGL((loss_bwd_kernel<PR, NC>), dim3((int)g), dim3(256), 0, st, logits, labels, cw, state, gout, P, HW, C,
                       ignore_index, dice_weight, ce_weight, dlogits, PR ? nll_log : 0, PR ? eps : 0.f);
  };
  auto by_nc = [&](auto PROBc) {
    if (C <= 2) launch(PROBc, std::integral_constant<int, 2>{});
    else if (C <= 4) launch(PROBc, std::integral_constant<int, 4>{});
    else launch(PROBc, std::integral_constant<int, MAXC>{});
  };
  if (prob) by_nc(std::true_type{});
  else by_nc(std::false_type{});
  SEGK_CHECK_LAUNCH("loss_bwd");
  return 0;
}

int segk_confusion_impl(const float* logits, const long long* labels, int N, int C, long HW, unsigned long long* M,
                        hipStream_t st) {
  SEGK_REQUIRE(logits && labels && M && N > 0 && HW > 0 && C >= 1 && C <= MAXC, "confusion: bad arguments");
  const long P = (long)N * HW;
  SEGK_REQUIRE(P < (1L << 31), "head / loss kernels index pixels with 32 bits: %ld pixels", P);
  long g = (P + 255) / 256;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(confusion_kernel, dim3((int)g), dim3(256), 0, st, logits, labels, P, HW, C, M);
  SEGK_CHECK_LAUNCH("confusion");
  return 0;
}
