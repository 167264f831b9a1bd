// Micro-benchmark 3 (round 4): what does STAGING cost the matrix pipe, by who issues it and how?
// One step = K 96 of a 3x3 conv's implicit GEMM (3 taps x one 32-channel chunk), 256 workgroups, LDS filled with random bf16.
//   V0: 4 consumer waves (16x16x32, wave tile 128 px x 64 ch, 96 MFMAs + 36 ds_read_b128 per step) + 4 idle partner waves
//   V1: V0 + partners stage 8 KiB per wave and step the way conv3x3_pipe_kernel does: 8 global_load_dwordx4 + 8 ds_write_b128
//   V2: V0 + partners stage the same 8 KiB per wave and step by LDS-DMA: 8 global_load_lds_dwordx4, counted vmcnt
//   V3: 4 waves only (one per SIMD, 512 registers), wave tile 128 x 128 (192 MFMAs + 48 reads per step), no staging
//   V4: V3 + the multiplying waves issue the DMA themselves (10 pieces per wave and step, one per 16 MFMAs)
//   V5: V2 with 4 pieces per wave and step (what a 512 px x 128 ch... half the staging) -- slope check
// Reported: TFLOP/s of the matrix work; the difference to V0 / V3 is what the staging costs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <type_traits>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

template <int OFF> __device__ __forceinline__ void lds_rd128(uint4& d, int addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}

__device__ __forceinline__ void fill_lds(char* smem, int tid, int nthr) {
  for (int i = tid; i < 150 * 1024 / 16; i += nthr) {
    unsigned z = (unsigned)i * 2654435761u + 12345u; z ^= z >> 15; z *= 2246822519u; z ^= z >> 13;
    auto rb = [&](unsigned r) { return (r & 0x80ffu) | 0x3f00u | ((r >> 3) & 0x0080u); };
    ((uint4*)smem)[i] = make_uint4(rb(z) | (rb(z >> 7) << 16), rb(z * 3u) | (rb(z * 5u) << 16), rb(z * 7u) | (rb(z * 11u) << 16), rb(z * 13u) | (rb(z * 17u) << 16));
  }
}

// ---------------- producer / consumer form (512 threads) ----------------
template <int MODE, int NPIECE, bool ILV = false>
__global__ __launch_bounds__(512, 2) void k_pc(float* out, const char* __restrict__ gsrc, int nsteps, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  fill_lds(smem, tid, 512);
  __syncthreads();
  if (wave >= 4) {
    // producers: the ring slots written are 96 KiB .. 150 KiB (never read by the consumers)
    const int pw = wave - 4;
    const char* g = gsrc + (size_t)blockIdx.x * (256 * 1024) + pw * 32768 + lane * 16;
    u32x4 R[NPIECE] = {};
    for (int s = 0; s < nsteps; ++s) {
      const int goff = (s & 3) * 8192;
      const int ring = 96 * 1024 + (s % 3) * 16384 + pw * 4096;   // 4 KiB per wave and slot region (pieces overlap: timing only)
      if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) *(u32x4*)(smem + ring + (i & 3) * 1024 + lane * 16) = R[i];
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) R[i] = *(const u32x4*)(g + goff + i * 1024);
      } else if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i)
          __builtin_amdgcn_global_load_lds((glb_vp)(g + goff + i * 1024), (lds_vp)(smem + ring + (i & 3) * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
      }
      __syncthreads();
    }
    if (MODE == 1) {
      unsigned r = 0;
      for (int i = 0; i < NPIECE; ++i) r += R[i].x;
      if (r == 0x12345678u) out[tid] = 1.f;
    }
    return;
  }
  const int lc = lane & 15, lq = lane >> 4;
  const int rowA = lc * 96 + lq * 16;
  f32x4 acc[8][4] = {};
  uint4 fa[2][4], fb[3][4];
  auto rdA = [&](int base, int t, int hh, uint4 (&A)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) A[j] = *(const uint4*)(smem + base + rowA + (4 * hh + j) * 1536 + t * 96);
  };
  auto rdB = [&](int base, int t, uint4 (&B)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) B[j] = *(const uint4*)(smem + base + 49152 + rowA + j * 1536 + t * 6144);
  };
  rdA(0, 0, 0, fa[0]);
  rdB(0, 0, fb[0]);
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int s = 0; s < nsteps; ++s) {
    const int base = (s & 1) * 20480, nbase = ((s + 1) & 1) * 20480;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int t = i >> 1, hh = i & 1;
      if constexpr (ILV) {
        // the next sub-step's fragments are read BETWEEN this sub-step's MFMAs: one ds_read_b128 after every second MFMA
        const int ni = (i + 1) % 6, nt_ = ni >> 1, nh = ni & 1;
        const int rb = (i + 1 < 6) ? base : nbase;
        const bool needB = (nh == 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) {
            acc[4 * hh + j][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i & 1][j]),
                                                                          __builtin_bit_cast(bf16x8, fb[t][nb]), acc[4 * hh + j][nb], 0, 0, 0);
            const int m = j * 4 + nb;
            if (m & 1) {
              const int r = m >> 1;     // 0..7
              // weight fragments first (all four are needed by the next sub-step's first MFMAs), then the patch fragments
              if (needB) {
                if (r < 4) fb[nt_][r] = *(const uint4*)(smem + rb + 49152 + rowA + r * 1536 + nt_ * 6144);
                else fa[(i + 1) & 1][r - 4] = *(const uint4*)(smem + rb + rowA + (4 * nh + r - 4) * 1536 + nt_ * 96);
              } else if (r < 4) fa[(i + 1) & 1][r] = *(const uint4*)(smem + rb + rowA + (4 * nh + r) * 1536 + nt_ * 96);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        continue;
      }
      if (i + 1 < 6) {
        rdA(base, (i + 1) >> 1, (i + 1) & 1, fa[(i + 1) & 1]);
        if (((i + 1) & 1) == 0) rdB(base, (i + 1) >> 1, fb[(i + 1) >> 1]);
      } else {
        rdA(nbase, 0, 0, fa[0]);
        rdB(nbase, 0, fb[0]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
          acc[4 * hh + j][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i & 1][j]),
                                                                        __builtin_bit_cast(bf16x8, fb[t][nb]), acc[4 * hh + j][nb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) { stamps[blockIdx.x * 2] = c1 - c0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
  float r = 0;
  for (int a = 0; a < 8; ++a) for (int b = 0; b < 4; ++b) for (int j = 0; j < 4; ++j) r += acc[a][b][j];
  out[blockIdx.x * 512 + tid] = r;
}

// ---------------- one wave per SIMD, 128 x 128 wave tile (256 threads) ----------------
template <int NDMA>
__global__ __launch_bounds__(256, 1) void k_big(float* out, const char* __restrict__ gsrc, int nsteps, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  fill_lds(smem, tid, 256);
  __syncthreads();
  const int lc = lane & 15, lq = lane >> 4;
  const int rowA = lc * 96 + lq * 16;
  const char* g = gsrc + (size_t)blockIdx.x * (256 * 1024) + wave * 65536 + lane * 16;
  f32x4 acc[8][8] = {};
  uint4 fa[2][8], fb[2][8];
  auto rd = [&](int base, int t, uint4 (&A)[8], uint4 (&B)[8]) {
    const int a0 = base + rowA + t * 96, b0 = base + 40960 + rowA + t * 12288;
    lds_rd128<0 * 1536>(A[0], a0); lds_rd128<1 * 1536>(A[1], a0); lds_rd128<2 * 1536>(A[2], a0); lds_rd128<3 * 1536>(A[3], a0);
    lds_rd128<4 * 1536>(A[4], a0); lds_rd128<5 * 1536>(A[5], a0); lds_rd128<6 * 1536>(A[6], a0); lds_rd128<7 * 1536>(A[7], a0);
    lds_rd128<0 * 1536>(B[0], b0); lds_rd128<1 * 1536>(B[1], b0); lds_rd128<2 * 1536>(B[2], b0); lds_rd128<3 * 1536>(B[3], b0);
    lds_rd128<4 * 1536>(B[4], b0); lds_rd128<5 * 1536>(B[5], b0); lds_rd128<6 * 1536>(B[6], b0); lds_rd128<7 * 1536>(B[7], b0);
  };
  rd(0, 0, fa[0], fb[0]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int s = 0; s < nsteps; s += 2) {
    // two steps (6 taps) per iteration so that the fragment double buffer alternates without copies
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int st = s + u / 3, t = u % 3;
      const int base = (st & 1) * 1024, nbase = ((st + 1) & 1) * 1024;
      const int goff = (st & 3) * 16384;
      const int ring = 100 * 1024 + (st % 3) * 16384 + wave * 4096;
      if (t < 2) rd(base, t + 1, fa[(u + 1) & 1], fb[(u + 1) & 1]);
      else rd(nbase, 0, fa[(u + 1) & 1], fb[(u + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // one DMA piece per 16 MFMAs until the step's pieces are out
        if (t * 4 + q < NDMA) {
          __builtin_amdgcn_global_load_lds((glb_vp)(g + goff + (t * 4 + q) * 1024), (lds_vp)(smem + ring + ((t * 4 + q) & 3) * 1024), 16, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int nb = 0; nb < 8; ++nb)
            acc[2 * q + j][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[u & 1][2 * q + j]),
                                                                         __builtin_bit_cast(bf16x8, fb[u & 1][nb]), acc[2 * q + j][nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (t == 2) {
        if (NDMA > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA > 0 ? NDMA : 0) : "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) { stamps[blockIdx.x * 2] = c1 - c0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
  float r = 0;
  for (int a = 0; a < 8; ++a) for (int b = 0; b < 8; ++b) for (int j = 0; j < 4; ++j) r += acc[a][b][j];
  out[blockIdx.x * 512 + tid] = r;
}

#include <algorithm>
#include <vector>
template <typename K> void run(const char* name, K kern, int threads, double mfma_per_wave_step, float* d, const char* gsrc) {
  const int nsteps = 2000, grid = 256;
  static unsigned long long* st = nullptr;
  if (!st) hipMalloc(&st, 256 * 16);
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float ms = 0, best = 1e9;
  for (int it = 0; it < 6; ++it) {
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 150 * 1024, 0, d, gsrc, nsteps, st);
    hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    if (it > 1 && ms < best) best = ms;
  }
  hipError_t e = hipGetLastError();
  std::vector<unsigned long long> h(512);
  hipMemcpy(h.data(), st, 512 * 8, hipMemcpyDeviceToHost);
  std::vector<double> cyc, ghz;
  for (int i = 0; i < 256; ++i) { cyc.push_back((double)h[2 * i] / nsteps); ghz.push_back((double)h[2 * i] / ((double)h[2 * i + 1] * 10.0)); }
  std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
  const double fl = (double)grid * 4 * nsteps * mfma_per_wave_step * 16384.0;
  printf("%-72s %7.3f ms %7.1f TF  cyc/step %6.0f (mfma %4.0f)  %.2f GHz %s\n", name, best, fl / best / 1e9, cyc[128], mfma_per_wave_step * 16, ghz[128],
         e == hipSuccess ? "" : hipGetErrorString(e));
}
int main() {
  float* d; hipMalloc(&d, 1024 * 512 * 4);
  char* g; hipMalloc(&g, (size_t)256 * 256 * 1024 + 65536); hipMemset(g, 0x3c, (size_t)256 * 256 * 1024 + 65536);
  for (int rep = 0; rep < 2; ++rep) {
  run("V0 pc: 4 consumers 128x64, partners idle", k_pc<0, 8>, 512, 96, d, g);
  run("V1 pc: partners 8 x (global_load + ds_write_b128) per wave and step", k_pc<1, 8>, 512, 96, d, g);
  run("V2 pc: partners 8 x global_load_lds per wave and step", k_pc<2, 8>, 512, 96, d, g);
  run("V8 pc: V0, fragment reads interleaved with the MFMAs", k_pc<0, 8, true>, 512, 96, d, g);
  run("V9 pc: V1, fragment reads interleaved", k_pc<1, 8, true>, 512, 96, d, g);
  run("V10 pc: V2, fragment reads interleaved", k_pc<2, 8, true>, 512, 96, d, g);
  run("V3 big: 4 waves 128x128, no staging", k_big<0>, 256, 192, d, g);
  run("V4 big: 10 self-issued DMA pieces per wave and step", k_big<10>, 256, 192, d, g);
  run("V7 big: 5 self-issued DMA pieces per wave and step", k_big<5>, 256, 192, d, g);
  }
  return 0;
}
