// Shared device/host helpers for the segk kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define SEGK_DT_F32 0
#define SEGK_DT_BF16 1

// ---- error plumbing (thread-local message, negative return codes; no exceptions cross the ABI)
extern thread_local char g_segk_err[512];
#define SEGK_FAIL(code, ...)                                   \
  do {                                                         \
    snprintf(g_segk_err, sizeof(g_segk_err), __VA_ARGS__);     \
    return (code);                                             \
  } while (0)
#define SEGK_REQUIRE(cond, ...)                                \
  do {                                                         \
    if (!(cond)) SEGK_FAIL(-2, __VA_ARGS__);                   \
  } while (0)
#define SEGK_CHECK_LAUNCH(name)                                                        \
  do {                                                                                 \
    hipError_t e_ = hipGetLastError();                                                 \
    if (e_ != hipSuccess) SEGK_FAIL(-3, "%s: launch failed: %s", name, hipGetErrorString(e_)); \
  } while (0)

// ---- element traits: one 16-byte LDS/global vector holds VEC elements
template <typename T> struct ET;
template <> struct ET<float> {
  static constexpr int VEC = 4;     // elements per 16 B
  static constexpr int ES = 4;      // element size
  static constexpr int CH = 16;     // channels per 64-byte K-chunk
};
template <> struct ET<bf16_t> {
  static constexpr int VEC = 8;
  static constexpr int ES = 2;
  static constexpr int CH = 32;
};

__device__ __forceinline__ float bf2f(uint32_t lo16) { return __uint_as_float(lo16 << 16); }

// unpack a 16-byte vector of T into floats
template <typename T> __device__ __forceinline__ void unpack16(const uint4& v, float* f);
template <> __device__ __forceinline__ void unpack16<float>(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y);
  f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const uint4& v, float* f) {
  f[0] = bf2f(v.x & 0xffffu); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = bf2f(v.y & 0xffffu); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = bf2f(v.z & 0xffffu); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = bf2f(v.w & 0xffffu); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
// Two floats -> one packed bf16 pair (round to nearest even, NaN-preserving): ONE v_cvt_pk_bf16_f32, emitted by the compiler
// from a vector conversion.  Never write this instruction as inline asm on accumulator values: the compiler inserts the
// wait states an MFMA result needs before a VALU read only for instructions it knows -- an asm statement placed right behind
// the MFMA reads the register before the matrix pipe has written it (round 3: the ConvTranspose streaming kernel produced
// garbage as soon as the 64-bit index arithmetic that used to sit between the two was shortened).
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v){lo, hi}, bf16x2v));
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) { return cvt_pk_bf16(lo, hi); }
template <typename T> __device__ __forceinline__ uint4 pack16(const float* f);
template <> __device__ __forceinline__ uint4 pack16<float>(const float* f) {
  return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
template <> __device__ __forceinline__ uint4 pack16<bf16_t>(const float* f) {
  return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]),
                    pack_bf16x2(f[6], f[7]));
}
template <typename T> __device__ __forceinline__ T from_float(float f);
template <> __device__ __forceinline__ float from_float<float>(float f) { return f; }
template <> __device__ __forceinline__ bf16_t from_float<bf16_t>(float f) { return (bf16_t)f; }
template <typename T> __device__ __forceinline__ float to_float(T v);
template <> __device__ __forceinline__ float to_float<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_float<bf16_t>(bf16_t v) { return (float)v; }

// Conv epilogue helper: the 16 fp32 results one lane holds of a 32x32 accumulator fragment (one output channel,
// pixel rows (r & 3) + 8 (r >> 2) of the fragment) -> running per-channel (sum, sum of squares) for
// training-mode BatchNorm, and stores into the [pixel][channel] LDS staging tile (row pitch OP bytes).
// Packed fp32 math (v_pk_add_f32 / v_pk_fma_f32) and paired bf16 conversion halve the VALU count; the
// pairwise summation order is fixed, so the statistics stay bit-stable run to run.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ void stage_frag(const float (&v)[16], char* obase, int OP, float& s1, float& s2) {
  f32x2_t p1 = {0.f, 0.f}, p2 = {0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const int dm = (r & 3) + 8 * (r >> 2);          // r even: rows dm and dm + 1
    const f32x2_t x = {v[r], v[r + 1]};
    p1 += x;
    p2 = __builtin_elementwise_fma(x, x, p2);
    if constexpr (sizeof(T) == 2) {
      const uint32_t pk = cvt_pk_bf16(x.x, x.y);   // ONE conversion for the pair
      *(uint16_t*)(obase + dm * OP) = (uint16_t)(pk & 0xffffu);
      *(uint16_t*)(obase + (dm + 1) * OP) = (uint16_t)(pk >> 16);
    } else {
      *(float*)(obase + dm * OP) = x.x;
      *(float*)(obase + (dm + 1) * OP) = x.y;
    }
  }
  s1 += p1.x + p1.y;
  s2 += p2.x + p2.y;
}

// XCD-aware block remap: blocks b and b+8 share an XCD (observed round-robin dispatch; speed only).
// Gives each XCD a contiguous range of logical ids so neighbouring tiles share one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// Item index -> (channel vector, x, y, image) in 32-bit arithmetic.  A 64-bit division by a run-time value is a ~130-
// instruction loop on this ISA, and a streaming kernel that decomposes a `long` index per item (five div/mod) spends more
// VALU time on that than on its data; every launcher that uses this checks that its item count is below 2^31.
struct Idx4 { int cv, x, y, b; };
__device__ __forceinline__ Idx4 split4(unsigned i, unsigned CV, unsigned Wc, unsigned Hc) {
  const unsigned p = i / CV, q = p / Wc, b = q / Hc;
  Idx4 r;
  r.cv = (int)(i - p * CV); r.x = (int)(p - q * Wc); r.y = (int)(q - b * Hc); r.b = (int)b;
  return r;
}
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
