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

// Separable backward for large up-sampling factors (the CLIP skips go 14x14 -> up to 224x224: a 2-D gather walks
// ~(3r)^2 candidate output pixels per input pixel, r = 16).  Pass 1 reduces along x into fp32 [B,OH,IW,C], pass 2 along
// y: each pass walks ~3r candidates, the output gradient is read about twice, and the result is deterministic.
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_x_kernel(const T* __restrict__ dy, float* __restrict__ tmp, int B, int IW,
                                                             int OH, int OW, int C) {
  using E = ET<T>;
  const int CV = C / E::VEC;
  const float sw = (float)IW / (float)OW, rw = (float)OW / (float)IW;
  const long total = (long)B * OH * IW * CV;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    long p = i / CV;
    const int ix = (int)(p % IW);
    const long row = p / IW;                       // b * OH + oy
    int lo = (int)floorf(((float)ix - 1.f) * rw) - 1, hi = (int)ceilf(((float)ix + 2.f) * rw) + 1;
    lo = max(lo, 0); hi = min(hi, OW - 1);
    float acc[E::VEC];
#pragma unroll
    for (int j = 0; j < E::VEC; ++j) acc[j] = 0.f;
    for (int ox = lo; ox <= hi; ++ox) {
      int x0, x1; float lx;
      src_index(ox, sw, IW, x0, x1, lx);
      const float w = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
      if (w == 0.f) continue;
      float g[E::VEC];
      unpack16<T>(*(const uint4*)(dy + ((size_t)row * OW + ox) * C + cv * E::VEC), g);
#pragma unroll
      for (int j = 0; j < E::VEC; ++j) acc[j] = fmaf(w, g[j], acc[j]);
    }
    float* dst = tmp + ((size_t)row * IW + ix) * C + cv * E::VEC;
#pragma unroll
    for (int j = 0; j < E::VEC; ++j) dst[j] = acc[j];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_y_kernel(const float* __restrict__ tmp, T* __restrict__ dx, int B, int IH,
                                                             int IW, int OH, int C) {
  using E = ET<T>;
  const int CV = C / E::VEC;
  const float sh = (float)IH / (float)OH, rh = (float)OH / (float)IH;
  const long total = (long)B * IH * IW * CV;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    long p = i / CV;
    const int ix = (int)(p % IW); p /= IW;
    const int iy = (int)(p % IH);
    const int b = (int)(p / IH);
    int lo = (int)floorf(((float)iy - 1.f) * rh) - 1, hi = (int)ceilf(((float)iy + 2.f) * rh) + 1;
    lo = max(lo, 0); hi = min(hi, OH - 1);
    float acc[E::VEC];
#pragma unroll
    for (int j = 0; j < E::VEC; ++j) acc[j] = 0.f;
    for (int oy = lo; oy <= hi; ++oy) {
      int y0, y1; float ly;
      src_index(oy, sh, IH, y0, y1, ly);
      const float w = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
      if (w == 0.f) continue;
      const float* src = tmp + (((size_t)b * OH + oy) * IW + ix) * C + cv * E::VEC;
#pragma unroll
      for (int j = 0; j < E::VEC; ++j) acc[j] = fmaf(w, src[j], acc[j]);
    }
    *(uint4*)(dx + (((size_t)b * IH + iy) * IW + ix) * C + cv * E::VEC) = pack16<T>(acc);
  }
}

}  // namespace

extern "C" int segk_bilinear_fwd(const void* x, void* y, int B, int IH, int IW, int OH, int OW, int Cp, int dtype,
                                 segk_stream_t s) {
  SEGK_REQUIRE(x && y && B > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0 && Cp > 0 && Cp % 32 == 0, "bilinear_fwd: bad arguments");
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "bilinear_fwd: bad dtype %d", dtype);
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

extern "C" int segk_bilinear_bwd(const void* dy, void* dx, float* scratch, int B, int IH, int IW, int OH, int OW, int Cp,
                                 int dtype, segk_stream_t s) {
  SEGK_REQUIRE(dy && dx && B > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0 && Cp > 0 && Cp % 32 == 0, "bilinear_bwd: bad arguments");
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "bilinear_bwd: bad dtype %d", dtype);
  const int vec = dtype == SEGK_DT_BF16 ? 8 : 4;
  if (scratch) {   // separable two-pass form: scratch holds B*OH*IW*Cp floats
    hipStream_t st2 = (hipStream_t)s;
    long g1 = ((long)B * OH * IW * (Cp / vec) + 255) / 256, g2 = ((long)B * IH * IW * (Cp / vec) + 255) / 256;
    if (g1 > 16384) g1 = 16384;
    if (g2 > 16384) g2 = 16384;
    if (dtype == SEGK_DT_BF16) {
      hipLaunchKernelGGL(bilinear_bwd_x_kernel<bf16_t>, dim3((int)g1), dim3(256), 0, st2, (const bf16_t*)dy, scratch, B, IW, OH, OW, Cp);
      hipLaunchKernelGGL(bilinear_bwd_y_kernel<bf16_t>, dim3((int)g2), dim3(256), 0, st2, scratch, (bf16_t*)dx, B, IH, IW, OH, Cp);
    } else {
      hipLaunchKernelGGL(bilinear_bwd_x_kernel<float>, dim3((int)g1), dim3(256), 0, st2, (const float*)dy, scratch, B, IW, OH, OW, Cp);
      hipLaunchKernelGGL(bilinear_bwd_y_kernel<float>, dim3((int)g2), dim3(256), 0, st2, scratch, (float*)dx, B, IH, IW, OH, Cp);
    }
    SEGK_CHECK_LAUNCH("bilinear_bwd");
    return 0;
  }
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

// ---- eval-time pre/post-processing on device (reference utils/utils.py:13-115) -----------------------------
// resize_pad: one image [C,H,W] -> its slot [C,T,T] of the network batch: aspect-preserving resize to (nh,nw) and
// zero padding (utils.py:13-49).  The reference resizes with torchvision's TF.resize, whose tensor branch is
// F.interpolate(mode, align_corners=False, antialias=True): ATen's separable anti-aliased triangle filter
// (UpSampleKernel.cpp, _compute_indices_min_size_weights_aa): scale = in/out, support = max(scale,1),
// center = scale*(o+0.5), taps [int(center-support+0.5), int(center+support+0.5)) clipped to the image,
// w = 1-|(j-center+0.5)/max(scale,1)| normalised to sum 1; plain bilinear when up-scaling.  Labels use
// mode "nearest": src = min(floor(o*scale), in-1).
// crop_resize: slot [C,T,T] -> crop the (nh,nw) window -> [C,oh,ow] with F.interpolate bilinear
// (align_corners=False, no anti-aliasing) or nearest (utils.py:51-75).
namespace {

struct AA {
  int lo, n;
  float center, inv, total;
};
__device__ __forceinline__ AA aa_taps(int o, float scale, int in_size) {
  AA a;
  const float support = scale >= 1.f ? scale : 1.f;
  a.inv = scale >= 1.f ? 1.f / scale : 1.f;
  a.center = scale * ((float)o + 0.5f);
  int lo = (int)(a.center - support + 0.5f);
  a.lo = lo < 0 ? 0 : lo;
  int hi = (int)(a.center + support + 0.5f);
  hi = hi > in_size ? in_size : hi;
  a.n = hi - a.lo;
  float t = 0.f;
  for (int j = 0; j < a.n; ++j) {
    const float x = fabsf(((float)(j + a.lo) - a.center + 0.5f) * a.inv);
    t += x < 1.f ? 1.f - x : 0.f;
  }
  a.total = t;
  return a;
}
__device__ __forceinline__ float aa_w(const AA& a, int j) {
  const float x = fabsf(((float)(j + a.lo) - a.center + 0.5f) * a.inv);
  const float w = x < 1.f ? 1.f - x : 0.f;
  return a.total != 0.f ? w / a.total : w;
}

template <typename V>
__global__ __launch_bounds__(256) void resize_pad_kernel(const V* __restrict__ img, V* __restrict__ out, int C, int H, int W,
                                                         int nh, int nw, int T, int pt, int pl, int mode) {
  const long total = (long)C * T * T;
  const float sh = (float)H / (float)nh, sw = (float)W / (float)nw;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int tx = (int)(i % T);
    const int ty = (int)((i / T) % T);
    const int c = (int)(i / ((long)T * T));
    const int oy = ty - pt, ox = tx - pl;
    V v = (V)0;
    if (oy >= 0 && oy < nh && ox >= 0 && ox < nw) {
      const V* src = img + (size_t)c * H * W;
      if (mode == 1) {
        int sy = (int)floorf((float)oy * sh), sx = (int)floorf((float)ox * sw);
        sy = sy > H - 1 ? H - 1 : sy;
        sx = sx > W - 1 ? W - 1 : sx;
        v = src[(size_t)sy * W + sx];
      } else if (mode == 2) {   // plain two-tap bilinear, align_corners=False (torchvision's tensor resize before 0.17)
        int y0, y1, x0, x1;
        float ly, lx;
        src_index(oy, sh, H, y0, y1, ly);
        src_index(ox, sw, W, x0, x1, lx);
        const float a = (float)src[(size_t)y0 * W + x0], b = (float)src[(size_t)y0 * W + x1];
        const float d = (float)src[(size_t)y1 * W + x0], e = (float)src[(size_t)y1 * W + x1];
        v = (V)((1.f - ly) * ((1.f - lx) * a + lx * b) + ly * ((1.f - lx) * d + lx * e));
      } else {
        const AA ay = aa_taps(oy, sh, H), ax = aa_taps(ox, sw, W);
        float acc = 0.f;
        for (int jy = 0; jy < ay.n; ++jy) {
          const float wy = aa_w(ay, jy);
          float row = 0.f;
          for (int jx = 0; jx < ax.n; ++jx) row = fmaf(aa_w(ax, jx), (float)src[(size_t)(ay.lo + jy) * W + ax.lo + jx], row);
          acc = fmaf(wy, row, acc);
        }
        v = (V)acc;
      }
    }
    out[i] = v;
  }
}

__global__ __launch_bounds__(256) void crop_resize_kernel(const float* __restrict__ slot, float* __restrict__ out, int C, int T,
                                                          int pt, int pl, int nh, int nw, int oh, int ow, int mode) {
  const long total = (long)C * oh * ow;
  const float sh = (float)nh / (float)oh, sw = (float)nw / (float)ow;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ox = (int)(i % ow);
    const int oy = (int)((i / ow) % oh);
    const int c = (int)(i / ((long)oh * ow));
    const float* src = slot + ((size_t)c * T + pt) * T + pl;       // window origin; row pitch T
    float v;
    if (mode == 1) {
      int sy = (int)floorf((float)oy * sh), sx = (int)floorf((float)ox * sw);
      sy = sy > nh - 1 ? nh - 1 : sy;
      sx = sx > nw - 1 ? nw - 1 : sx;
      v = src[(size_t)sy * T + sx];
    } else {
      int y0, y1, x0, x1;
      float ly, lx;
      src_index(oy, sh, nh, y0, y1, ly);
      src_index(ox, sw, nw, x0, x1, lx);
      const float a = src[(size_t)y0 * T + x0], b = src[(size_t)y0 * T + x1];
      const float d = src[(size_t)y1 * T + x0], e = src[(size_t)y1 * T + x1];
      // ATen: (1-ly)*((1-lx)*a + lx*b) + ly*((1-lx)*d + lx*e)
      v = (1.f - ly) * ((1.f - lx) * a + lx * b) + ly * ((1.f - lx) * d + lx * e);
    }
    out[i] = v;
  }
}

}  // namespace

extern "C" int segk_resize_pad(const void* img, void* out, int C, int H, int W, int nh, int nw, int T, int pad_top,
                               int pad_left, int mode, int elem, segk_stream_t s) {
  SEGK_REQUIRE(img && out && C > 0 && H > 0 && W > 0 && nh > 0 && nw > 0 && T > 0, "resize_pad: bad shape");
  SEGK_REQUIRE(pad_top >= 0 && pad_left >= 0 && pad_top + nh <= T && pad_left + nw <= T, "resize_pad: window outside the target");
  SEGK_REQUIRE(mode >= 0 && mode <= 2 && (elem == 0 || elem == 1), "resize_pad: bad mode/element type");
  SEGK_REQUIRE(!(elem == 1 && mode != 1), "resize_pad: integer images resize with mode nearest only");
  long g = ((long)C * T * T + 255) / 256;
  if (g > 16384) g = 16384;
  hipStream_t st = (hipStream_t)s;
  if (elem == 1)
    hipLaunchKernelGGL(resize_pad_kernel<long long>, dim3((int)g), dim3(256), 0, st, (const long long*)img, (long long*)out, C, H, W,
                       nh, nw, T, pad_top, pad_left, mode);
  else
    hipLaunchKernelGGL(resize_pad_kernel<float>, dim3((int)g), dim3(256), 0, st, (const float*)img, (float*)out, C, H, W, nh, nw, T,
                       pad_top, pad_left, mode);
  SEGK_CHECK_LAUNCH("resize_pad");
  return 0;
}

extern "C" int segk_crop_resize(const float* slot, float* out, int C, int T, int pad_top, int pad_left, int nh, int nw, int oh,
                                int ow, int mode, segk_stream_t s) {
  SEGK_REQUIRE(slot && out && C > 0 && T > 0 && nh > 0 && nw > 0 && oh > 0 && ow > 0, "crop_resize: bad shape");
  SEGK_REQUIRE(pad_top >= 0 && pad_left >= 0 && pad_top + nh <= T && pad_left + nw <= T, "crop_resize: window outside the slot");
  SEGK_REQUIRE(mode == 0 || mode == 1, "crop_resize: bad mode");
  long g = ((long)C * oh * ow + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(crop_resize_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)s, slot, out, C, T, pad_top, pad_left, nh, nw, oh,
                     ow, mode);
  SEGK_CHECK_LAUNCH("crop_resize");
  return 0;
}
