// Micro-benchmark 2: which wave tiling / MFMA shape sustains the most on random bf16 data?
// All variants: 256 workgroups x 8 waves, fragments read from LDS with ds_read_b128 one sub-step ahead,
// one barrier per 6 sub-steps (one sub-step = K 16 for 32x32x16, K 32 for 16x16x32).
//   V0: 32x32x16, wave tile  64x64  (MF 2, NF 2)      V1: 32x32x16, wave tile 128x64 (MF 4, NF 2)
//   V2: 16x16x32, wave tile  64x64  (4x4 tiles)       V3: 16x16x32, wave tile 128x64 (8x4 tiles)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int V>
__global__ __launch_bounds__(512, 2) void k(float* out, int nsteps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 150 * 1024 / 16; i += 512) {
    unsigned z = (unsigned)i * 2654435761u + 12345u; z ^= z >> 15; z *= 2246822519u; z ^= z >> 13;
    auto rb = [&](unsigned r) { return (r & 0x80ffu) | 0x3f00u | ((r >> 3) & 0x0080u); };
    ((uint4*)smem)[i] = make_uint4(rb(z) | (rb(z >> 7) << 16), rb(z * 3u) | (rb(z * 5u) << 16), rb(z * 7u) | (rb(z * 11u) << 16), rb(z * 13u) | (rb(z * 17u) << 16));
  }
  __syncthreads();
  constexpr bool BIG = (V == 1 || V == 3), M16 = (V >= 2);
  constexpr int NA = M16 ? (BIG ? 8 : 4) : (BIG ? 4 : 2);   // A fragments per sub-step
  constexpr int NB = M16 ? 4 : 2;
  // per-lane row offsets (80 B pitch, conflict-free); fragments just need distinct valid addresses
  const int rowA = M16 ? (lane & 15) * 80 + (lane >> 4) * 16 : (lane & 31) * 80 + (lane >> 5) * 16;
  f32x16 acc32[4][2] = {};
  f32x4 acc16[8][4] = {};
  uint4 fa[2][NA], fb[2][NB];
  for (int s = 0; s < nsteps; ++s) {
    const char* pb = smem + (s & 1) * 30000;
    const char* wb = smem + 70000 + (s & 1) * 30000;
    auto rd = [&](int i, uint4 (&A)[NA], uint4 (&B)[NB]) {
#pragma unroll
      for (int a = 0; a < NA; ++a) A[a] = *(const uint4*)(pb + rowA + a * (M16 ? 1280 : 2560) + (i >> 1) * 80 + (i & 1) * 32);
#pragma unroll
      for (int b = 0; b < NB; ++b) B[b] = *(const uint4*)(wb + rowA + b * (M16 ? 1280 : 2560) + (i >> 1) * 5120 + (i & 1) * 32);
    };
    rd(0, fa[0], fb[0]);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (i + 1 < 6) rd(i + 1, fa[(i + 1) & 1], fb[(i + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if (M16) acc16[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i & 1][a]), __builtin_bit_cast(bf16x8, fb[i & 1][b]), acc16[a][b], 0, 0, 0);
          else acc32[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i & 1][a]), __builtin_bit_cast(bf16x8, fb[i & 1][b]), acc32[a][b], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  float r = 0;
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 2; ++b) for (int j = 0; j < 16; ++j) r += acc32[a][b][j];
  for (int a = 0; a < 8; ++a) for (int b = 0; b < 4; ++b) for (int j = 0; j < 4; ++j) r += acc16[a][b][j];
  out[blockIdx.x * 512 + tid] = r;
}

template <int V> void run(const char* name, float* d) {
  const int nsteps = 2000, grid = 256;
  hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float ms = 0;
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(a);
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(512), 150 * 1024, 0, d, nsteps);
    hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
  }
  constexpr bool BIG = (V == 1 || V == 3), M16 = (V >= 2);
  // flops per wave per sub-step: tile M x N x K*2
  const double fl_sub = (BIG ? 128.0 : 64.0) * 64.0 * (M16 ? 32.0 : 16.0) * 2.0;
  const double fl = (double)grid * 8 * nsteps * 6 * fl_sub;
  printf("%-44s %8.3f ms  %8.1f TFLOP/s\n", name, ms, fl / ms / 1e9);
}
int main() {
  float* d; hipMalloc(&d, 1024 * 512 * 4);
  run<0>("32x32x16  wave 64x64  (4 reads / 4 MFMA)", d);
  run<1>("32x32x16  wave 128x64 (6 reads / 8 MFMA)", d);
  run<2>("16x16x32  wave 64x64  (8 reads / 16 MFMA)", d);
  run<3>("16x16x32  wave 128x64 (12 reads / 32 MFMA)", d);
  return 0;
}
