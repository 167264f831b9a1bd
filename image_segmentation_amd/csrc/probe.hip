// Diagnostic entry (not on the product path): the shader clock the chip holds under a dense bf16 MFMA load, so that a
// bench line can carry a number that explains box-to-box spread (MI355X_MICROARCH.md, DVFS give-back: devices differ by
// up to 12 % in the clock they hold under matrix load).  One wave per SIMD runs `iters` rounds of 16 independent
// v_mfma_f32_32x32x16_bf16 (shape 0) or the same matrix work as 32 v_mfma_f32_16x16x32_bf16 (shape 1: the chip can hold
// another clock on the other shape) on pseudo-random register operands and brackets the loop with s_memtime (shader cycles) and
// s_memrealtime (constant 100 MHz): clock = d(memtime) / d(memrealtime) x 100 MHz.  Nothing but the two differences
// leaves the kernel.
#include "common.hpp"
#include "segk_internal.h"
#include "../../include/segk.h"

namespace {
__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
template <bool M16>
__global__ __launch_bounds__(256) void clock_probe_kernel(unsigned long long* __restrict__ out, int iters) {
  const unsigned gid = blockIdx.x * 256u + threadIdx.x;
  union { bf16x8 v; unsigned u[4]; } a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // two bf16 values in [-2, 2) per dword: random sign and mantissa, exponent 0x3f / 0x3e / 0x3d
    const unsigned ha = hash32(gid * 8u + i), hb = hash32(gid * 8u + 4 + i);
    a.u[i] = (ha & 0x80ff80ffu) | 0x3f003e00u;
    b.u[i] = (hb & 0x80ff80ffu) | 0x3e003f00u;
  }
  f32x16 acc[4];
  f32x4 q[16];                                            // shape 1: 16 independent 16x16 accumulators
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = (float)(k + 1);            // distinct chains: nothing for the compiler to merge
#pragma unroll
  for (int k = 0; k < 16; ++k) q[k] = (f32x4){(float)(k + 1), 0.f, 0.f, 0.f};
  // opaque to the optimiser: 16 (4) independent accumulation chains stay 16 (4) chains
#pragma unroll
  for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(q[k]));
#pragma unroll
  for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(acc[k]));
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (M16) {          // the same matrix work per round as 16 MFMAs 32x32x16: 32 MFMAs 16x16x32 (16 cycles each)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 16; ++k) q[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, q[k], 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc[k], 0, 0, 0);
    }
    // keep the products bounded (the accumulators would otherwise run to infinity and the multiplier to a fixed point)
    if ((it & 63) == 63) {
      if constexpr (M16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) q[k] *= 0.001f;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[k][r] *= 0.001f;
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float keep = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) keep += acc[k][0] + acc[k][15];
#pragma unroll
  for (int k = 0; k < 16; ++k) keep += q[k][0] + q[k][3];
  asm volatile("" ::"v"(keep));
  if ((threadIdx.x & 63) == 0) {
    const unsigned w = blockIdx.x * 4u + (threadIdx.x >> 6);
    out[2 * w] = c1 - c0;
    out[2 * w + 1] = r1 - r0;
  }
}
}  // namespace

int segk_clock_probe_impl(unsigned long long* out, int blocks, int iters, int shape, hipStream_t st) {
  SEGK_REQUIRE(out && blocks > 0 && blocks <= 4096 && iters > 0 && (shape == 0 || shape == 1), "clock_probe: bad arguments");
  if (shape == 1) hipLaunchKernelGGL(clock_probe_kernel<true>, dim3(blocks), dim3(256), 0, st, out, iters);
  else hipLaunchKernelGGL(clock_probe_kernel<false>, dim3(blocks), dim3(256), 0, st, out, iters);
  SEGK_CHECK_LAUNCH("clock_probe");
  return 0;
}
