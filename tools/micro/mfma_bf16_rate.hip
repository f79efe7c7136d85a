// Micro-benchmark: sustained bf16 MFMA rate of the whole chip (one wave per SIMD, register operands holding random bf16
// data), v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16, with the shader-clock ticks (s_memtime) per MFMA and the
// tick rate against wall time -- the numbers behind DESIGN.md 4.1 ("the chip holds ~1.75 GHz under a dense bf16 MFMA stream").
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_bf16_rate.hip -o tools/micro/mfma_bf16_rate && tools/micro/mfma_bf16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP error at line %d\n", __LINE__); exit(1); } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>  // 0: 32x32x16 (4 accumulators of 16), 1: 16x16x32 (16 accumulators of 4)
__global__ __launch_bounds__(256, 1) void k(const unsigned* __restrict__ rnd, float* out, long long* ticks, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    uint4 ua = reinterpret_cast<const uint4*>(rnd)[(t * 8 + i) & 65535], ub = reinterpret_cast<const uint4*>(rnd)[(t * 8 + 4 + i) & 65535];
    a[i] = __builtin_bit_cast(bf16x8, ua);
    b[i] = __builtin_bit_cast(bf16x8, ub);
  }
  float sum = 0.f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  if constexpr (SHAPE == 0) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(u + j) & 3], b[j], acc[j], 0, 0, 0);
    }
    for (int j = 0; j < 4; ++j)
      for (int r = 0; r < 16; ++r) sum += acc[j][r];
  } else {
    f32x4 acc[16];
    for (int j = 0; j < 16; ++j)
      for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(u + j) & 3], b[j & 3], acc[j], 0, 0, 0);
    }
    for (int j = 0; j < 16; ++j)
      for (int r = 0; r < 4; ++r) sum += acc[j][r];
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[t] = sum;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int SHAPE>
void run(const char* name, const unsigned* rnd, int blocks) {
  float* out;
  long long* ticks;
  CK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(float)));
  CK(hipMalloc(&ticks, blocks * sizeof(long long)));
  const int iters = 20000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, rnd, out, ticks, 100);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, rnd, out, ticks, iters);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> h(blocks);
  CK(hipMemcpy(h.data(), ticks, blocks * sizeof(long long), hipMemcpyDeviceToHost));
  double tk = 0;
  for (auto v : h) tk += (double)v;
  tk /= blocks;
  const double mfma_per_wave = SHAPE == 0 ? 16.0 * iters : 32.0 * iters;
  const double fpi = SHAPE == 0 ? 32768.0 : 16384.0;  // flop per instruction: 2 * 32*32*16, 2 * 16*16*32
  const double flop = (double)blocks * 4 * mfma_per_wave * fpi;
  printf("%-26s blocks %4d: %8.3f ms  %7.1f TFLOP/s  %6.2f ticks per MFMA  tick rate %.3f GHz  -> %.2f ns per 32768 flop per SIMD\n", name, blocks, ms,
         flop / ms / 1e9, tk / mfma_per_wave, tk / (ms * 1e6), ms * 1e6 / mfma_per_wave * (32768.0 / fpi));
  CK(hipFree(out));
  CK(hipFree(ticks));
}

int main() {
  std::vector<unsigned> h(65536 * 4);
  srand(1);
  for (auto& v : h) {  // random bf16 pairs in [-2, 2): sign, exponent 125..127, random mantissa
    auto one = [] { return (unsigned)(((rand() & 1) << 15) | ((125 + rand() % 3) << 7) | (rand() & 127)); };
    v = one() | (one() << 16);
  }
  unsigned* rnd;
  CK(hipMalloc(&rnd, h.size() * 4));
  CK(hipMemcpy(rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  for (int blocks : {1, 256}) {
    run<0>("v_mfma_f32_32x32x16_bf16", rnd, blocks);
    run<1>("v_mfma_f32_16x16x32_bf16", rnd, blocks);
  }
  return 0;
}
