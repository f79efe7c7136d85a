// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate (a) register operands only, (b) with the
// GEMM's LDS fragment reads in the loop.  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NACC>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[224 * 36];
  for (int i = threadIdx.x; i < 224 * 36; i += 256) lds[i] = seed * (i % 7);
  __syncthreads();
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  f32x4 fa = {seed, seed * 2, seed * 3, seed * 4};
  f32x4 fb[NACC];
  for (int j = 0; j < NACC; ++j) fb[j] = fa * (float)(j + 1);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      if (MODE == 1) {
        fa = *reinterpret_cast<const f32x4*>(lds + (wave * 32 + li) * 36 + kb * 8 + 4 * h);
#pragma unroll
        for (int j = 0; j < NACC; ++j)
          fb[j] = *reinterpret_cast<const f32x4*>(lds + (128 + j * 32 + li) * 36 + kb * 8 + 4 * h);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], fb[j][s], acc[j], 0, 0, 0);
    }
  }
  float sum = 0.f;
  for (int j = 0; j < NACC; ++j)
    for (int r = 0; r < 16; ++r) sum += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <int MODE, int NACC>
void run(const char* name, int blocks_per_cu) {
  float* out;
  int nb = 256 * blocks_per_cu;
  hipMalloc(&out, nb * 256 * sizeof(float));
  int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, NACC>), dim3(nb), dim3(256), 0, 0, out, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, NACC>), dim3(nb), dim3(256), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)nb * 4 * iters * 16.0 * NACC * 4096.0;
  printf("%-34s NACC=%d blocks/CU=%d : %8.3f ms  %7.1f TFLOP/s\n", name, NACC, blocks_per_cu, ms, flop / ms / 1e9);
  hipFree(out);
}

int main() {
  run<0, 3>("registers only", 1);
  run<0, 4>("registers only", 1);
  run<0, 3>("registers only", 2);
  run<1, 3>("with ds_read_b128 fragment reads", 1);
  run<1, 4>("with ds_read_b128 fragment reads", 1);
  run<1, 3>("with ds_read_b128 fragment reads", 2);
  return 0;
}
