// Micro-benchmark: what does the conv inner-loop STRUCTURE sustain on MI355X?
// 256 workgroups x 8 waves, each wave a 64x64 tile (2x2 MFMA 32x32x16 bf16), fragments read from LDS
// with ds_read_b128 exactly like conv_igemm (6 sub-steps of 4 reads + 4 MFMAs per step).
//   mode 0: MFMA only (operands stay in registers)          mode 1: + LDS fragment reads (pipelined)
//   mode 2: + one barrier per step                           mode 3: mode 2 with 16x16x32 MFMAs (same tile)
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_lds mfma_lds.hip ; run: ./mfma_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, int nsteps, unsigned seed) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 120 * 1024 / 16; i += 512) {
    unsigned z = (unsigned)i * 2654435761u + seed; z ^= z >> 15; z *= 2246822519u; z ^= z >> 13;
    // random bf16 pairs in [-2,2): sign/mantissa random, exponent 0x3f..0x40
    auto rb = [&](unsigned r) { return (r & 0x80ffu) | 0x3f00u | ((r >> 3) & 0x0080u); };
    unsigned a0 = rb(z) | (rb(z >> 7) << 16), a1 = rb(z * 3u) | (rb(z * 5u) << 16), a2 = rb(z * 7u) | (rb(z * 11u) << 16), a3 = rb(z * 13u) | (rb(z * 17u) << 16);
    ((uint4*)smem)[i] = seed ? make_uint4(a0, a1, a2, a3) : make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  }
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
  int laneA[2], laneB[2];
  for (int mf = 0; mf < 2; ++mf) laneA[mf] = ((wm * 2 + mf) * 32 + lr) * 80 + lh * 16;
  for (int nf = 0; nf < 2; ++nf) laneB[nf] = 57344 + ((wn * 2 + nf) * 32 + lr) * 80 + lh * 16;
  f32x16 acc[2][2] = {};
  f32x4 acc4[4][4] = {};
  uint4 fa[2][2], fb[2][2];
  fa[0][0] = fa[0][1] = fb[0][0] = fb[0][1] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  fa[1][0] = fa[1][1] = fb[1][0] = fb[1][1] = fa[0][0];
  for (int s = 0; s < nsteps; ++s) {
    const char* pb = smem + (s & 1) * 28160;
    const char* wb = smem + (s & 1) * 30720;
    auto rd = [&](int i, uint4 (&A)[2], uint4 (&B)[2]) {
      const int t = i >> 1, kk = i & 1;
      for (int mf = 0; mf < 2; ++mf) A[mf] = *(const uint4*)(pb + laneA[mf] + t * 80 + kk * 32);
      for (int nf = 0; nf < 2; ++nf) B[nf] = *(const uint4*)(wb + t * 10240 + laneB[nf] + kk * 32);
    };
    if (MODE >= 1) rd(0, fa[0], fb[0]);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (MODE >= 1 && i + 1 < 6) rd(i + 1, fa[(i + 1) & 1], fb[(i + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE == 3) {
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
          for (int nf = 0; nf < 4; ++nf)   // 16 MFMAs 16x16x32 cover the same 64x64x(16) work per sub-step? (K=32: double work) -> use half
            if ((mf + nf) & 1 || true) acc4[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i & 1][mf & 1]), __builtin_bit_cast(bf16x8, fb[i & 1][nf & 1]), acc4[mf][nf], 0, 0, 0);
      } else {
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
          for (int nf = 0; nf < 2; ++nf)
            acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i & 1][mf]), __builtin_bit_cast(bf16x8, fb[i & 1][nf]), acc[mf][nf], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE >= 2) __syncthreads();
  }
  float r = 0;
  for (int mf = 0; mf < 2; ++mf) for (int nf = 0; nf < 2; ++nf) for (int j = 0; j < 16; ++j) r += acc[mf][nf][j];
  for (int mf = 0; mf < 4; ++mf) for (int nf = 0; nf < 4; ++nf) for (int j = 0; j < 4; ++j) r += acc4[mf][nf][j];
  out[blockIdx.x * 512 + tid] = r;
}

template <int MODE> void run(const char* name, float* d, int grid, unsigned seed) {
  const int nsteps = 2000;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int it = 0; it < 2; ++it) {
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 120 * 1024, 0, d, nsteps, seed);
    hipEventRecord(b); hipEventSynchronize(b);
  }
  float ms; hipEventElapsedTime(&ms, a, b);
  // flops: per wave per step 24 MFMAs 32x32x16 = 24*32768 (mode 3: 96 x 16x16x32 = 96*16384)
  double fl = (double)grid * 8 * nsteps * (MODE == 3 ? 96.0 * 16384 : 24.0 * 32768);
  printf("%-34s grid %4d seed %u %8.3f ms  %8.1f TFLOP/s\n", name, grid, seed, ms, fl / ms / 1e9);
}
int main() {
  float* d; hipMalloc(&d, 1024 * 512 * 4);
  for (unsigned seed : {0u, 12345u}) {
    run<0>("mfma only", d, 256, seed);
    run<1>("+ pipelined LDS fragment reads", d, 256, seed);
    run<2>("+ barrier per step", d, 256, seed);
    run<3>("16x16x32 MFMAs + reads + barrier", d, 256, seed);
  }
  return 0;
}
