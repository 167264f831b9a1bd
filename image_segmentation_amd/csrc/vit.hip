// CLIP vision transformer (frozen feature extractor of the CLIP-UNet) -- the pieces around its GEMMs.
//
// Reference: clip/clipunet.py:25-46 runs transformers.CLIPVisionModel (third party, not vendored) with
// output_hidden_states=True and uses hidden states 3,5,7,9 and the last one.  The published CLIP ViT algorithm
// (transformers/models/clip/modeling_clip.py: CLIPVisionEmbeddings, CLIPEncoderLayer, CLIPAttention, CLIPMLP):
//   e      = [class_embedding ; Conv2d(3, D, ps, stride ps, bias=False)(x) as tokens] + position_embedding
//   h_0    = LayerNorm_pre(e)
//   h_l+1  = r + fc2(quick_gelu(fc1(LN2(r)))),  r = h_l + out_proj(softmax(q k^T / sqrt(hd)) v),  q,k,v = proj(LN1(h_l))
// All dense contractions (patch projection, q/k/v/out projections, fc1, fc2) run as MFMA GEMMs through
// segk_linear (conv_igemm.hip, 1x1 geometry); this file holds the HBM/latency-bound rest:
//   vit_patchify     NCHW fp32 image -> patch rows [B*G*G][Kp] (k = c*ps*ps + i*ps + j, the flattening of the
//                    patch_embedding weight), so the patch projection is a GEMM
//   vit_embed_ln     class token + patch tokens + position embedding, pre-LayerNorm -> fp32 residual stream
//   add_layernorm    h += delta (a GEMM output); out = LayerNorm(h) in the compute dtype (one wave per token)
//   attention        bf16, head_dim 64: flash-style on the matrix cores (attention_mfma_kernel below); fp32 parity mode
//                    and other head sizes: per (image, head, 64-query block) softmax(q k^T * scale) v in exact fp32
//                    arithmetic on the VALU with K/V of the head resident in LDS, 4 waves splitting the keys and
//                    merging their online-softmax partials through LDS
//   vit_tokens_to_grid   drop CLS, residual stream -> NHWC feature grid [B,G,G,D] (clipunet.py:48-63)
// The residual stream stays fp32 in both compute modes (it is 2.4 MB per image batch of 16); GEMM operands are
// the compute dtype.
#include <stdlib.h>
#include "common.hpp"
#include "../../include/segk.h"

namespace {

constexpr int LN_MAXPER = 32;   // channels per lane: D <= 2048

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void vit_patchify_kernel(const float* __restrict__ x, T* __restrict__ rows, int B, int C,
                                                           int H, int W, int ps, int G, int Kp) {
  const long total = (long)B * G * G * Kp;
  const int K = C * ps * ps;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int k = (int)(i % Kp);
    long r = i / Kp;
    const int gx = (int)(r % G); r /= G;
    const int gy = (int)(r % G);
    const int b = (int)(r / G);
    float v = 0.f;
    if (k < K) {
      const int c = k / (ps * ps), ij = k % (ps * ps), ii = ij / ps, jj = ij % ps;
      v = x[(((size_t)b * C + c) * H + gy * ps + ii) * W + gx * ps + jj];
    }
    rows[i] = from_float<T>(v);
  }
}

// row statistics and normalisation of LN_MAXPER-per-lane register rows (two-pass variance like ATen's CPU kernel)
__device__ __forceinline__ void ln_rows(float (&v)[LN_MAXPER], int per, int D, float eps, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXPER; ++i)
    if (i < per) s += v[i];
  mean = wave_sum(s) / (float)D;
  float q = 0.f;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < LN_MAXPER; ++i)
    if (i < per && lane + 64 * i < D) {
      const float d = v[i] - mean;
      q = fmaf(d, d, q);
    }
  rstd = rsqrtf(wave_sum(q) / (float)D + eps);
}

template <typename T>
__global__ __launch_bounds__(256) void vit_embed_ln_kernel(const T* __restrict__ proj, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps,
                                                           float* __restrict__ h, int B, int Tn, int D, int Dp) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)B * Tn) return;
  const int t = (int)(row % Tn), b = (int)(row / Tn);
  const int per = (D + 63) / 64;
  float v[LN_MAXPER];
#pragma unroll
  for (int i = 0; i < LN_MAXPER; ++i) {
    v[i] = 0.f;
    const int c = lane + 64 * i;
    if (i < per && c < D) {
      const float e = (t == 0) ? cls[c] : to_float<T>(proj[((size_t)b * (Tn - 1) + (t - 1)) * Dp + c]);
      v[i] = e + pos[(size_t)t * D + c];
    }
  }
  float mean, rstd;
  ln_rows(v, per, D, eps, mean, rstd);
#pragma unroll
  for (int i = 0; i < LN_MAXPER; ++i) {
    const int c = lane + 64 * i;
    if (i < per && c < D) h[(size_t)row * D + c] = (v[i] - mean) * rstd * gamma[c] + beta[c];
  }
}

// 16-byte accesses: a lane owns float4 number lane + 64*i of the row (D % 4 == 0), LN4 = 8 float4 per lane covers D <= 2048
constexpr int LN4 = LN_MAXPER / 4;
template <typename T> __device__ __forceinline__ void load4(const T* p, float (&f)[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&f)[4]) {
  const float4 v = *(const float4*)p;
  f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float (&f)[4]) {
  const uint2 v = *(const uint2*)p;
  f[0] = bf2f(v.x & 0xffffu); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = bf2f(v.y & 0xffffu); f[3] = __uint_as_float(v.y & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float (&f)[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&f)[4]) {
  *(float4*)p = make_float4(f[0], f[1], f[2], f[3]);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float (&f)[4]) {
  *(uint2*)p = make_uint2(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]));
}

template <typename T>
__global__ __launch_bounds__(256) void add_layernorm_kernel(float* __restrict__ h, const T* __restrict__ delta,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps, T* __restrict__ out, long M, int D, int Dp,
                                                            int nparts, long part_stride) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nq = D >> 2;                       // float4 per row
  float v[LN4][4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN4; ++i) {
    const int q = lane + 64 * i;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[i][e] = 0.f;
    if (q < nq) {
      load4<float>(h + (size_t)row * D + q * 4, v[i]);
      if (delta) {
        for (int pp = 0; pp < nparts; ++pp) {   // split-K partial products of the GEMM before, fixed order
          float d[4];
          load4<T>(delta + (size_t)pp * part_stride + (size_t)row * Dp + q * 4, d);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[i][e] += d[e];
        }
        store4<float>(h + (size_t)row * D + q * 4, v[i]);
      }
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
  }
  if (!out) return;
  const float mean = wave_sum(s) / (float)D;
  float qs = 0.f;
#pragma unroll
  for (int i = 0; i < LN4; ++i)
    if (lane + 64 * i < nq) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[i][e] - mean;
        qs = fmaf(d, d, qs);
      }
    }
  const float rstd = rsqrtf(wave_sum(qs) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < LN4; ++i) {
    const int q = lane + 64 * i;
    if (q < nq) {
      float g[4], b[4], o[4];
      load4<float>(gamma + q * 4, g);
      load4<float>(beta + q * 4, b);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
      store4<T>(out + (size_t)row * Dp + q * 4, o);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void vit_grid_kernel(const float* __restrict__ h, T* __restrict__ out, int B, int Tn, int D,
                                                       int Dp) {
  const long total = (long)B * (Tn - 1) * Dp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % Dp);
    const long r = i / Dp;
    const int p = (int)(r % (Tn - 1)), b = (int)(r / (Tn - 1));
    out[i] = from_float<T>(c < D ? h[((size_t)b * Tn + 1 + p) * D + c] : 0.f);
  }
}

// ---- attention ---------------------------------------------------------------------------------------------
// grid (ceil(Tn/64), B*heads), 256 threads.  lane = one query of the 64-query block; the four waves take the key
// groups (4 keys) g = wave, wave+4, ... and keep online-softmax partials (m, l, acc[HD]); K and V of the head are
// staged once in LDS in the storage dtype and read as wave-uniform (broadcast) 16-byte vectors.
template <typename T, int HD>
__global__ __launch_bounds__(256) void attention_kernel(const T* __restrict__ qkv, T* __restrict__ ctx, int Tn, int heads,
                                                        int ldq, int ldo, float scale) {
  using E = ET<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int VPR = HD / E::VEC;                // 16-byte vectors per K/V row
  const int Tp = (Tn + 3) & ~3;
  const int bh = blockIdx.y, b = bh / heads, hh = bh % heads;
  const int D = heads * HD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint4* const Ks = (uint4*)smem;
  uint4* const Vs = Ks + (size_t)Tp * VPR;
  const T* const base = qkv + (size_t)b * Tn * ldq + hh * HD;
  for (int i = threadIdx.x; i < Tp * VPR; i += 256) {
    const int j = i / VPR, c = i % VPR;
    uint4 kv = make_uint4(0, 0, 0, 0), vv = kv;
    if (j < Tn) {
      kv = *(const uint4*)(base + (size_t)j * ldq + D + c * E::VEC);
      vv = *(const uint4*)(base + (size_t)j * ldq + 2 * D + c * E::VEC);
    }
    Ks[i] = kv;
    Vs[i] = vv;
  }
  const int qi = blockIdx.x * 64 + lane;
  const int qc = qi < Tn ? qi : Tn - 1;
  float q[HD], acc[HD];
#pragma unroll
  for (int c = 0; c < VPR; ++c) {
    float f[E::VEC];
    unpack16<T>(*(const uint4*)(base + (size_t)qc * ldq + c * E::VEC), f);
#pragma unroll
    for (int e = 0; e < E::VEC; ++e) q[c * E::VEC + e] = f[e] * scale;
  }
#pragma unroll
  for (int d = 0; d < HD; ++d) acc[d] = 0.f;
  float m = -INFINITY, l = 0.f;
  __syncthreads();
  for (int g = wave; g * 4 < Tn; g += 4) {
    const int j0 = g * 4;
    float s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int c = 0; c < VPR; ++c) {
        float f[E::VEC];
        unpack16<T>(Ks[(size_t)(j0 + i) * VPR + c], f);
#pragma unroll
        for (int e = 0; e < E::VEC; e += 2) {
          a0 = fmaf(q[c * E::VEC + e], f[e], a0);
          a1 = fmaf(q[c * E::VEC + e + 1], f[e + 1], a1);
        }
      }
      s[i] = (j0 + i < Tn) ? a0 + a1 : -INFINITY;
    }
    const float mn = fmaxf(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), m);   // finite: key j0 is always valid
    const float corr = __expf(m - mn);
    float p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = __expf(s[i] - mn);
    l = l * corr + ((p[0] + p[1]) + (p[2] + p[3]));
    m = mn;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] *= corr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int c = 0; c < VPR; ++c) {
        float f[E::VEC];
        unpack16<T>(Vs[(size_t)(j0 + i) * VPR + c], f);
#pragma unroll
        for (int e = 0; e < E::VEC; ++e) acc[c * E::VEC + e] = fmaf(p[i], f[e], acc[c * E::VEC + e]);
      }
    }
  }
  __syncthreads();                                  // K/V are dead: the partials overlay them
  float* const part = (float*)smem;                 // [wave][HD + 2][64]
#pragma unroll
  for (int d = 0; d < HD; ++d) part[(wave * (HD + 2) + d) * 64 + lane] = acc[d];
  part[(wave * (HD + 2) + HD) * 64 + lane] = m;
  part[(wave * (HD + 2) + HD + 1) * 64 + lane] = l;
  __syncthreads();
  float mw[4], fw[4];
  float M = -INFINITY;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    mw[w] = part[(w * (HD + 2) + HD) * 64 + lane];
    M = fmaxf(M, mw[w]);
  }
  float L = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    fw[w] = __expf(mw[w] - M);                      // a wave that saw no key holds m = -inf, l = 0: factor 0
    L = fmaf(part[(w * (HD + 2) + HD + 1) * 64 + lane], fw[w], L);
  }
  const float inv = 1.f / L;
  constexpr int DW = HD / 4;                        // output dims per wave
  float o[DW];
#pragma unroll
  for (int d = 0; d < DW; ++d) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) a = fmaf(part[(w * (HD + 2) + wave * DW + d) * 64 + lane], fw[w], a);
    o[d] = a * inv;
  }
  if (qi < Tn) {
    T* const dst = ctx + ((size_t)b * Tn + qi) * ldo + hh * HD + wave * DW;
#pragma unroll
    for (int c = 0; c < DW / E::VEC; ++c) *(uint4*)(dst + c * E::VEC) = pack16<T>(o + c * E::VEC);
  }
}

// ---- attention on the matrix cores (bf16, head_dim 64) ---------------------------------------------------------
// grid (ceil(T/128), B*heads), 256 threads: each wave owns 32 queries and walks the keys in blocks of 32, flash style.
//   S^T block [32 keys x 32 queries] = K_blk . Q^T      4 x MFMA 32x32x16 (K rows from LDS, the Q fragment in registers)
//   online softmax per query column: the 32x32 accumulator layout gives a lane 16 keys of ONE query, the other half-wave
//   holds the other 16 (one __shfl_xor for the block maximum); exp2 with the scale folded into the exponent
//   O^T [64 x 32 queries] += V^T_blk . P^T              4 x MFMA: the bf16-packed P registers ARE the B fragment (the k
//   slots of a lane are its own 8 keys), the A fragment comes from V staged TRANSPOSED in LDS (two 8-byte reads)
// LDS: K [Tp][144 B] + V^T [64][Tp*2 + 16 B] (Tp = T rounded up to 32; padding keys are zero and masked).
__global__ __launch_bounds__(256) void attention_mfma_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx, int Tn,
                                                             int heads, int ldq, int ldo, float scale) {
  constexpr int HD = 64, KP = 144;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Tp = (Tn + 31) & ~31;
  const int VP = Tp * 2 + 16;
  char* const Ks = smem;
  char* const Vt = smem + (size_t)Tp * KP;
  const int bh = blockIdx.y, b = bh / heads, hh = bh % heads;
  const int D = heads * HD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const bf16_t* const base = qkv + (size_t)b * Tn * ldq + hh * HD;
  // ---- stage K (rows) and V (transposed) of this head
  for (int i = threadIdx.x; i < Tp * 8; i += 256) {
    const int j = i >> 3, c = i & 7;
    uint4 kv = make_uint4(0, 0, 0, 0), vv = kv;
    if (j < Tn) {
      kv = *(const uint4*)(base + (size_t)j * ldq + D + c * 8);
      vv = *(const uint4*)(base + (size_t)j * ldq + 2 * D + c * 8);
    }
    *(uint4*)(Ks + j * KP + c * 16) = kv;
    const uint32_t w[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int e = 0; e < 8; ++e)
      *(uint16_t*)(Vt + (c * 8 + e) * VP + j * 2) = (uint16_t)(w[e >> 1] >> ((e & 1) * 16));
  }
  // ---- this wave's queries: Q fragment (B operand) straight from global memory
  const int q0 = (blockIdx.x * 4 + wave) * 32;
  const int qi = q0 + lr;
  const int qc = qi < Tn ? qi : Tn - 1;
  uint4 qf[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) qf[j] = *(const uint4*)(base + (size_t)qc * ldq + j * 16 + lh * 8);
  __syncthreads();
  if (q0 >= Tn) return;                              // a wave without queries (after the only barrier)
  const float c2 = scale * 1.44269504088896341f;     // exp(x*scale) = exp2(x*c2)
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m = -INFINITY, l = 0.f;
  const int nkb = Tp >> 5;
  for (int kb = 0; kb < nkb; ++kb) {
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
    const char* const kr = Ks + (kb * 32 + lr) * KP + lh * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint4 kf = *(const uint4*)(kr + j * 32);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[j]), sacc, 0, 0, 0);
    }
    const int kbase = kb * 32 + 4 * lh;
    float bm = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kbase + (r & 3) + 8 * (r >> 2);
      sacc[r] = key < Tn ? sacc[r] : -INFINITY;
      bm = fmaxf(bm, sacc[r]);
    }
    bm = fmaxf(bm, __shfl_xor(bm, 32));
    const float mn = fmaxf(m, bm);                   // finite from the first block on (key 0 is always valid)
    const float corr = exp2f((m - mn) * c2);
    m = mn;
    l *= corr;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= corr; o1[r] *= corr; }
    uint32_t pk[8];
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const float p0 = exp2f((sacc[r] - mn) * c2), p1 = exp2f((sacc[r + 1] - mn) * c2);
      ps += p0 + p1;
      pk[r >> 1] = pack_bf16x2(p0, p1);
    }
    l += ps;
    // P^T fragments: k slots of this lane = its keys of registers 0..7 (first MFMA) and 8..15 (second)
    const uint4 pb0 = make_uint4(pk[0], pk[1], pk[2], pk[3]), pb1 = make_uint4(pk[4], pk[5], pk[6], pk[7]);
    // V^T fragments: row d = dblk*32 + lr, keys kbase + {0..3, 8..11} and kbase + {16..19, 24..27}
    const char* const v0 = Vt + lr * VP + kbase * 2;
    const char* const v1 = v0 + 32 * VP;
    uint2 a, c;
    a = *(const uint2*)(v0); c = *(const uint2*)(v0 + 16);
    o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, c.x, c.y)),
                                                 __builtin_bit_cast(bf16x8, pb0), o0, 0, 0, 0);
    a = *(const uint2*)(v0 + 32); c = *(const uint2*)(v0 + 48);
    o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, c.x, c.y)),
                                                 __builtin_bit_cast(bf16x8, pb1), o0, 0, 0, 0);
    a = *(const uint2*)(v1); c = *(const uint2*)(v1 + 16);
    o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, c.x, c.y)),
                                                 __builtin_bit_cast(bf16x8, pb0), o1, 0, 0, 0);
    a = *(const uint2*)(v1 + 32); c = *(const uint2*)(v1 + 48);
    o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, c.x, c.y)),
                                                 __builtin_bit_cast(bf16x8, pb1), o1, 0, 0, 0);
  }
  l += __shfl_xor(l, 32);                            // both half-waves applied the same corrections: plain sum
  const float inv = 1.f / l;
  if (qi < Tn) {
    bf16_t* const dst = ctx + ((size_t)b * Tn + qi) * ldo + hh * HD + 4 * lh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {                    // registers 4g..4g+3 = head dims 4*lh + 8g + {0..3} (+32 for o1)
      *(uint2*)(dst + 8 * g) = make_uint2(pack_bf16x2(o0[4 * g] * inv, o0[4 * g + 1] * inv),
                                          pack_bf16x2(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv));
      *(uint2*)(dst + 32 + 8 * g) = make_uint2(pack_bf16x2(o1[4 * g] * inv, o1[4 * g + 1] * inv),
                                               pack_bf16x2(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv));
    }
  }
}

int launch_attention_mfma(const void* qkv, void* ctx, int B, int Tn, int heads, int ldq, int ldo, float scale, hipStream_t st) {
  const int Tp = (Tn + 31) & ~31;
  const size_t lds = (size_t)Tp * 144 + (size_t)64 * (Tp * 2 + 16);
  SEGK_REQUIRE(lds <= 160 * 1024, "attention: %d tokens do not fit the 160 KiB LDS (%zu bytes)", Tn, lds);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)attention_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      SEGK_FAIL(-3, "attention: cannot raise dynamic LDS limit");
    attr_set = true;
  }
  hipLaunchKernelGGL(attention_mfma_kernel, dim3(cdiv(Tn, 128), B * heads), dim3(256), lds, st, (const bf16_t*)qkv, (bf16_t*)ctx,
                     Tn, heads, ldq, ldo, scale);
  SEGK_CHECK_LAUNCH("attention_mfma");
  return 0;
}

template <typename T, int HD>
int launch_attention(const void* qkv, void* ctx, int B, int Tn, int heads, int ldq, int ldo, float scale, hipStream_t st) {
  const int Tp = (Tn + 3) & ~3;
  size_t lds = (size_t)2 * Tp * HD * sizeof(T);
  const size_t comb = (size_t)4 * (HD + 2) * 64 * 4;
  if (lds < comb) lds = comb;
  SEGK_REQUIRE(lds <= 160 * 1024, "attention: %d tokens x %d head dims do not fit the 160 KiB LDS (%zu bytes)", Tn, HD, lds);
  auto kern = attention_kernel<T, HD>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      SEGK_FAIL(-3, "attention: cannot raise dynamic LDS limit");
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(cdiv(Tn, 64), B * heads), dim3(256), lds, st, (const T*)qkv, (T*)ctx, Tn, heads, ldq, ldo,
                     scale);
  SEGK_CHECK_LAUNCH("attention");
  return 0;
}

}  // namespace

extern "C" int segk_vit_patchify(const float* x, void* rows, int B, int C, int H, int W, int ps, int Kp, int dtype,
                                 segk_stream_t s) {
  SEGK_REQUIRE(x && rows && B > 0 && C > 0 && ps > 0 && H >= ps && W >= ps && H / ps == W / ps, "vit_patchify: bad shape");
  SEGK_REQUIRE(Kp >= C * ps * ps && Kp % 32 == 0, "vit_patchify: Kp=%d must be a multiple of 32 covering %d", Kp, C * ps * ps);
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "vit_patchify: bad dtype");
  const int G = H / ps;
  long g = ((long)B * G * G * Kp + 255) / 256;
  if (g > 16384) g = 16384;
  hipStream_t st = (hipStream_t)s;
  if (dtype == SEGK_DT_BF16)
    hipLaunchKernelGGL(vit_patchify_kernel<bf16_t>, dim3((int)g), dim3(256), 0, st, x, (bf16_t*)rows, B, C, H, W, ps, G, Kp);
  else
    hipLaunchKernelGGL(vit_patchify_kernel<float>, dim3((int)g), dim3(256), 0, st, x, (float*)rows, B, C, H, W, ps, G, Kp);
  SEGK_CHECK_LAUNCH("vit_patchify");
  return 0;
}

extern "C" int segk_vit_embed_ln(const void* proj, const float* cls, const float* pos, const float* gamma, const float* beta,
                                 float eps, float* h, int B, int T, int D, int Dp, int dtype, segk_stream_t s) {
  SEGK_REQUIRE(proj && cls && pos && gamma && beta && h && B > 0 && T > 1, "vit_embed_ln: bad arguments");
  SEGK_REQUIRE(D > 0 && D <= 64 * LN_MAXPER && Dp >= D, "vit_embed_ln: hidden size %d unsupported (max %d)", D, 64 * LN_MAXPER);
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "vit_embed_ln: bad dtype");
  hipStream_t st = (hipStream_t)s;
  const long rows = (long)B * T;
  if (dtype == SEGK_DT_BF16)
    hipLaunchKernelGGL(vit_embed_ln_kernel<bf16_t>, dim3((int)((rows + 3) / 4)), dim3(256), 0, st, (const bf16_t*)proj, cls, pos,
                       gamma, beta, eps, h, B, T, D, Dp);
  else
    hipLaunchKernelGGL(vit_embed_ln_kernel<float>, dim3((int)((rows + 3) / 4)), dim3(256), 0, st, (const float*)proj, cls, pos,
                       gamma, beta, eps, h, B, T, D, Dp);
  SEGK_CHECK_LAUNCH("vit_embed_ln");
  return 0;
}

extern "C" int segk_add_layernorm_parts(float* h, const void* delta, int nparts, long part_stride, const float* gamma,
                                        const float* beta, float eps, void* out, long M, int D, int Dp, int dtype,
                                        segk_stream_t s);
extern "C" int segk_add_layernorm(float* h, const void* delta, const float* gamma, const float* beta, float eps, void* out,
                                  long M, int D, int Dp, int dtype, segk_stream_t s) {
  return segk_add_layernorm_parts(h, delta, 1, 0, gamma, beta, eps, out, M, D, Dp, dtype, s);
}
extern "C" int segk_add_layernorm_parts(float* h, const void* delta, int nparts, long part_stride, const float* gamma,
                                        const float* beta, float eps, void* out, long M, int D, int Dp, int dtype,
                                        segk_stream_t s) {
  SEGK_REQUIRE(h && M > 0 && (delta || out), "add_layernorm: bad arguments");
  SEGK_REQUIRE(nparts >= 1 && (nparts == 1 || part_stride >= M * (long)Dp), "add_layernorm: bad partial-product layout");
  SEGK_REQUIRE(!out || (gamma && beta), "add_layernorm: LayerNorm output needs gamma and beta");
  SEGK_REQUIRE(D > 0 && D <= 64 * LN_MAXPER && Dp >= D && D % 4 == 0 && Dp % 4 == 0,
               "add_layernorm: hidden size %d unsupported (multiple of 4, max %d)", D, 64 * LN_MAXPER);
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "add_layernorm: bad dtype");
  hipStream_t st = (hipStream_t)s;
  const int g = (int)((M + 3) / 4);
  if (dtype == SEGK_DT_BF16)
    hipLaunchKernelGGL(add_layernorm_kernel<bf16_t>, dim3(g), dim3(256), 0, st, h, (const bf16_t*)delta, gamma, beta, eps,
                       (bf16_t*)out, M, D, Dp, nparts, part_stride);
  else
    hipLaunchKernelGGL(add_layernorm_kernel<float>, dim3(g), dim3(256), 0, st, h, (const float*)delta, gamma, beta, eps,
                       (float*)out, M, D, Dp, nparts, part_stride);
  SEGK_CHECK_LAUNCH("add_layernorm");
  return 0;
}

extern "C" int segk_attention(const void* qkv, void* ctx, int B, int T, int heads, int head_dim, int ldq, int ldo, float scale,
                              int dtype, segk_stream_t s) {
  SEGK_REQUIRE(qkv && ctx && B > 0 && T > 0 && heads > 0, "attention: bad arguments");
  SEGK_REQUIRE(head_dim == 64 || head_dim == 32, "attention: head_dim %d unsupported (32 or 64)", head_dim);
  SEGK_REQUIRE(ldq >= 3 * heads * head_dim && ldo >= heads * head_dim && ldq % 8 == 0 && ldo % 8 == 0,
               "attention: bad row pitches %d / %d", ldq, ldo);
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "attention: bad dtype");
  SEGK_REQUIRE((long)B * heads <= 65535, "attention: B*heads exceeds the grid limit");
  hipStream_t st = (hipStream_t)s;
  if (dtype == SEGK_DT_BF16 && head_dim == 64 && !getenv("SEGK_ATTENTION_VALU"))   // matrix-core kernel
    return launch_attention_mfma(qkv, ctx, B, T, heads, ldq, ldo, scale, st);
  if (dtype == SEGK_DT_BF16)
    return head_dim == 64 ? launch_attention<bf16_t, 64>(qkv, ctx, B, T, heads, ldq, ldo, scale, st)
                          : launch_attention<bf16_t, 32>(qkv, ctx, B, T, heads, ldq, ldo, scale, st);
  return head_dim == 64 ? launch_attention<float, 64>(qkv, ctx, B, T, heads, ldq, ldo, scale, st)
                        : launch_attention<float, 32>(qkv, ctx, B, T, heads, ldq, ldo, scale, st);
}

extern "C" int segk_vit_tokens_to_grid(const float* h, void* out, int B, int T, int D, int Dp, int dtype, segk_stream_t s) {
  SEGK_REQUIRE(h && out && B > 0 && T > 1 && D > 0 && Dp >= D && Dp % 32 == 0, "vit_tokens_to_grid: bad arguments");
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "vit_tokens_to_grid: bad dtype");
  long g = ((long)B * (T - 1) * Dp + 255) / 256;
  if (g > 16384) g = 16384;
  hipStream_t st = (hipStream_t)s;
  if (dtype == SEGK_DT_BF16)
    hipLaunchKernelGGL(vit_grid_kernel<bf16_t>, dim3((int)g), dim3(256), 0, st, h, (bf16_t*)out, B, T, D, Dp);
  else
    hipLaunchKernelGGL(vit_grid_kernel<float>, dim3((int)g), dim3(256), 0, st, h, (float*)out, B, T, D, Dp);
  SEGK_CHECK_LAUNCH("vit_tokens_to_grid");
  return 0;
}
