// Micro-benchmark: issue rate of v_mfma_f32_16x16x32_bf16 against the number of INDEPENDENT accumulators it is dealt over (1 = every
// MFMA accumulates into the result of the one before it, 2, 4, 16) -- what a kernel may order its products by without stalling the
// matrix pipe.  One block of 256 threads (one wave per SIMD), shader-clock ticks (s_memtime) per MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_chain.hip -o tools/micro/mfma_chain && tools/micro/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP error at line %d\n", __LINE__); exit(1); } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ __launch_bounds__(256, 1) void k(const unsigned* __restrict__ rnd, float* out, long long* ticks, int iters) {
  const int t = threadIdx.x;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = __builtin_bit_cast(bf16x8, reinterpret_cast<const uint4*>(rnd)[(t * 8 + i) & 65535]);
    b[i] = __builtin_bit_cast(bf16x8, reinterpret_cast<const uint4*>(rnd)[(t * 8 + 4 + i) & 65535]);
  }
  f32x4 acc[CH];
  for (int j = 0; j < CH; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32 / CH; ++u)
#pragma unroll
      for (int j = 0; j < CH; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(u + j) & 3], b[(u >> 2) & 3], acc[j], 0, 0, 0);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float sum = 0.f;
  for (int j = 0; j < CH; ++j)
    for (int r = 0; r < 4; ++r) sum += acc[j][r];
  out[t] = sum;
  if (t == 0) ticks[0] = t1 - t0;
}

template <int CH>
void run(const unsigned* rnd) {
  float* out;
  long long* ticks;
  CK(hipMalloc(&out, 256 * sizeof(float)));
  CK(hipMalloc(&ticks, sizeof(long long)));
  const int iters = 20000;
  hipLaunchKernelGGL(k<CH>, dim3(1), dim3(256), 0, 0, rnd, out, ticks, 100);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(k<CH>, dim3(1), dim3(256), 0, 0, rnd, out, ticks, iters);
  CK(hipDeviceSynchronize());
  long long h;
  CK(hipMemcpy(&h, ticks, sizeof(long long), hipMemcpyDeviceToHost));
  printf("v_mfma_f32_16x16x32_bf16 over %2d independent accumulators: %6.2f ticks per MFMA\n", CH, (double)h / (32.0 * iters));
  CK(hipFree(out));
  CK(hipFree(ticks));
}

int main() {
  unsigned* rnd;
  CK(hipMalloc(&rnd, 65536 * 16));
  unsigned* h = (unsigned*)malloc(65536 * 16);
  for (int i = 0; i < 65536 * 4; ++i) h[i] = 0x3f803f80u + ((unsigned)rand() & 0x007f007fu);
  CK(hipMemcpy(rnd, h, 65536 * 16, hipMemcpyHostToDevice));
  run<1>(rnd);
  run<2>(rnd);
  run<4>(rnd);
  run<16>(rnd);
  return 0;
}
