// Bilinear resize of NHWC tensors, align_corners=False -- the skip-feature up-sampling of the CLIP decoder
// (reference clip/clipunet.py:99-100: F.interpolate(skip, size=x.shape[2:], mode='bilinear',
// align_corners=False); always taken: 14x14 -> 28/56/112/224).  Source index follows ATen's
// area_pixel_compute_source_index: src = scale*(dst+0.5)-0.5, clamped at 0; x1 = min(x0+1, in-1).
// Forward: one thread per output pixel x 16-byte channel vector.  Backward: a GATHER (one thread per INPUT
// pixel x channel vector walks the output pixels that reference it) -- deterministic, no atomics.
#include "common.hpp"
#include "segk_internal.h"
#include "../../include/segk.h"

namespace {

__device__ __forceinline__ void src_index(int o, float scale, int in_size, int& i0, int& i1, float& lam) {
  float s = scale * ((float)o + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  lam = s - (float)i0;
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int IH,
                                                           int IW, int OH, int OW, int C) {
  using E = ET<T>;
  const int CV = C / E::VEC;
  const float sh = (float)IH / (float)OH, sw = (float)IW / (float)OW;
  const long total = (long)B * OH * OW * CV;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    long p = i / CV;
    const int ox = (int)(p % OW); p /= OW;
    const int oy = (int)(p % OH);
    const int b = (int)(p / OH);
    int y0, y1, x0, x1;
    float ly, lx;
    src_index(oy, sh, IH, y0, y1, ly);
    src_index(ox, sw, IW, x0, x1, lx);
    const T* base = x + (size_t)b * IH * IW * C + cv * E::VEC;
    float f00[E::VEC], f01[E::VEC], f10[E::VEC], f11[E::VEC];
    unpack16<T>(*(const uint4*)(base + ((size_t)y0 * IW + x0) * C), f00);
    unpack16<T>(*(const uint4*)(base + ((size_t)y0 * IW + x1) * C), f01);
    unpack16<T>(*(const uint4*)(base + ((size_t)y1 * IW + x0) * C), f10);
    unpack16<T>(*(const uint4*)(base + ((size_t)y1 * IW + x1) * C), f11);
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
#pragma unroll
    for (int j = 0; j < E::VEC; ++j) f00[j] = w00 * f00[j] + w01 * f01[j] + w10 * f10[j] + w11 * f11[j];
    *(uint4*)(y + (((size_t)b * OH + oy) * OW + ox) * C + cv * E::VEC) = pack16<T>(f00);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int IH,
                                                           int IW, int OH, int OW, int C) {
  using E = ET<T>;
  const int CV = C / E::VEC;
  const float sh = (float)IH / (float)OH, sw = (float)IW / (float)OW;
  const float rh = (float)OH / (float)IH, rw = (float)OW / (float)IW;
  const long total = (long)B * IH * IW * CV;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    long p = i / CV;
    const int ix = (int)(p % IW); p /= IW;
    const int iy = (int)(p % IH);
    const int b = (int)(p / IH);
    // conservative window of output pixels whose taps can touch (iy, ix); each is re-tested exactly
    int oy_lo = (int)floorf(((float)iy - 1.f) * rh) - 1, oy_hi = (int)ceilf(((float)iy + 2.f) * rh) + 1;
    int ox_lo = (int)floorf(((float)ix - 1.f) * rw) - 1, ox_hi = (int)ceilf(((float)ix + 2.f) * rw) + 1;
    oy_lo = max(oy_lo, 0); ox_lo = max(ox_lo, 0);
    oy_hi = min(oy_hi, OH - 1); ox_hi = min(ox_hi, OW - 1);
    float acc[E::VEC];
#pragma unroll
    for (int j = 0; j < E::VEC; ++j) acc[j] = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1; float ly;
      src_index(oy, sh, IH, y0, y1, ly);
      const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1; float lx;
        src_index(ox, sw, IW, x0, x1, lx);
        const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
        if (wx == 0.f) continue;
        float g[E::VEC];
        unpack16<T>(*(const uint4*)(dy + (((size_t)b * OH + oy) * OW + ox) * C + cv * E::VEC), g);
        const float w = wy * wx;
#pragma unroll
        for (int j = 0; j < E::VEC; ++j) acc[j] = fmaf(w, g[j], acc[j]);
      }
    }
    *(uint4*)(dx + (((size_t)b * IH + iy) * IW + ix) * C + cv * E::VEC) = pack16<T>(acc);
  }
}

}  // namespace

extern "C" int segk_bilinear_fwd(const void* x, void* y, int B, int IH, int IW, int OH, int OW, int Cp, int dtype,
                                 segk_stream_t s) {
  SEGK_REQUIRE(x && y && B > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0 && Cp > 0 && Cp % 32 == 0, "bilinear_fwd: bad arguments");
  const int vec = dtype == SEGK_DT_BF16 ? 8 : 4;
  long g = ((long)B * OH * OW * (Cp / vec) + 255) / 256;
  if (g > 8192) g = 8192;
  hipStream_t st = (hipStream_t)s;
  if (dtype == SEGK_DT_BF16)
    hipLaunchKernelGGL(bilinear_fwd_kernel<bf16_t>, dim3((int)g), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, B, IH, IW, OH, OW, Cp);
  else
    hipLaunchKernelGGL(bilinear_fwd_kernel<float>, dim3((int)g), dim3(256), 0, st, (const float*)x, (float*)y, B, IH, IW, OH, OW, Cp);
  SEGK_CHECK_LAUNCH("bilinear_fwd");
  return 0;
}

extern "C" int segk_bilinear_bwd(const void* dy, void* dx, int B, int IH, int IW, int OH, int OW, int Cp, int dtype,
                                 segk_stream_t s) {
  SEGK_REQUIRE(dy && dx && B > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0 && Cp > 0 && Cp % 32 == 0, "bilinear_bwd: bad arguments");
  const int vec = dtype == SEGK_DT_BF16 ? 8 : 4;
  long g = ((long)B * IH * IW * (Cp / vec) + 255) / 256;
  if (g > 8192) g = 8192;
  hipStream_t st = (hipStream_t)s;
  if (dtype == SEGK_DT_BF16)
    hipLaunchKernelGGL(bilinear_bwd_kernel<bf16_t>, dim3((int)g), dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dx, B, IH, IW, OH, OW, Cp);
  else
    hipLaunchKernelGGL(bilinear_bwd_kernel<float>, dim3((int)g), dim3(256), 0, st, (const float*)dy, (float*)dx, B, IH, IW, OH, OW, Cp);
  SEGK_CHECK_LAUNCH("bilinear_bwd");
  return 0;
}
