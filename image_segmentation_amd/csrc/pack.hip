// Layout conversion kernels: NCHW fp32 <-> NHWC (channel-padded, fp32/bf16) activations, and
// reference-layout (OIHW / IOHW fp32) parameters <-> the K-chunked MFMA weight layout
// [K/CH][taps][N][CH] used by conv_igemm.hip and the [N][taps][K] fp32 gradient slabs of wgrad.hip.
// All tiny, bandwidth-trivial; one thread per destination element.
#include "common.hpp"
#include "segk_internal.h"

namespace {

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int C, int H, int W,
                                    int Cp) {
  const long total = (long)B * H * W * Cp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % Cp);
    const long p = i / Cp;
    const long hw = (long)H * W;
    const int b = (int)(p / hw);
    const long r = p - (long)b * hw;
    dst[i] = from_float<T>(c < C ? src[((long)b * C + c) * hw + r] : 0.f);
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
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, T* __restrict__ dst, int Cout, int CA, int CB,
                                        int Coutp, int CAp, int CBp, int taps, int mode) {
  constexpr int CH = ET<T>::CH;
  const int Cin = CA + CB, Cinp = CAp + CBp;
  const int Kp = mode == 0 ? Cinp : Coutp, Np = mode == 0 ? Coutp : Cinp;
  const long total = (long)Kp * taps * Np;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
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
    dst[i] = from_float<T>(v);
  }
}

// ConvTranspose2d(k=2,s=2) weight [Cin][Cout][2][2] -> packed.
//  mode 0 (forward GEMM, N = q*Coutp + co, K = ci):         dst[kc][0][n][j] = W[ci=kc*CH+j][co][q]
//  mode 1 (data grad, K = q*Coutp + co via un-shuffle):     dst[kc][0][n=ci][j] = W[ci][co][q], kc = q*(Coutp/CH)+cc
template <typename T>
__global__ void pack_convt_weight_kernel(const float* __restrict__ w, T* __restrict__ dst, int Cin, int Cout, int Cinp,
                                         int Coutp, int mode) {
  constexpr int CH = ET<T>::CH;
  const long total = (long)Cinp * 4 * Coutp;
  const int Np = mode == 0 ? 4 * Coutp : Cinp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int j = (int)(i % CH);
    long r = i / CH;
    const int n = (int)(r % Np);
    const int kc = (int)(r / Np);
    int ci, co, q;
    if (mode == 0) { ci = kc * CH + j; q = n / Coutp; co = n - q * Coutp; }
    else { const int ncc = Coutp / CH; q = kc / ncc; co = (kc - q * ncc) * CH + j; ci = n; }
    float v = 0.f;
    if (ci < Cin && co < Cout) v = w[((long)ci * Cout + co) * 4 + q];
    dst[i] = from_float<T>(v);
  }
}

// Sum S gradient slabs [S][Np][taps][Kp] (fp32) and scatter into the reference layout.
//  kind 0: Conv2d OIHW        grad[n=co][ci(k)][tap]
//  kind 1: ConvTranspose IOHW grad[n=ci][k=co][tap]   (no dual map)
__global__ void wgrad_reduce_kernel(const float* __restrict__ slabs, int S, float* __restrict__ grad, int N, int CA,
                                    int CB, int Np, int CAp, int CBp, int taps, int kind) {
  // one thread per OUTPUT element (coalesced stores); the strided slab reads are absorbed by L2
  const int Kp = CAp + CBp, K = CA + CB;
  const long slab = (long)Np * taps * Kp;
  const long total = (long)N * K * taps;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int tap = (int)(i % taps);
    long r = i / taps;
    const int k = (int)(r % K);
    const int n = (int)(r / K);
    const int kp = k < CA ? k : CAp + (k - CA);
    const long src = ((long)n * taps + tap) * Kp + kp;
    float s = 0.f;
    for (int t = 0; t < S; ++t) s += slabs[(long)t * slab + src];  // fixed order: bit-stable
    grad[i] = s;
    (void)kind;
  }
}

static int grid_for(long total) {
  long g = (total + 255) / 256;
  return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

}  // namespace

int segk_nchw_to_nhwc_impl(const float* src, void* dst, int B, int C, int H, int W, int Cp, int dtype, hipStream_t st) {
  SEGK_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && Cp >= C && Cp % 32 == 0, "nchw_to_nhwc: bad arguments");
  const long total = (long)B * H * W * Cp;
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

int segk_wgrad_reduce_impl(const float* slabs, int S, float* grad, int N, int CA, int CB, int Np, int CAp, int CBp,
                           int taps, hipStream_t st) {
  SEGK_REQUIRE(slabs && grad && S > 0 && N > 0 && CA > 0 && CB >= 0 && Np >= N && CAp >= CA && CBp >= CB && taps > 0,
               "wgrad_reduce: bad arguments");
  const long total = (long)N * (CA + CB) * taps;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for(total)), dim3(256), 0, st, slabs, S, grad, N, CA, CB, Np, CAp,
                     CBp, taps, 0);
  SEGK_CHECK_LAUNCH("wgrad_reduce");
  return 0;
}
