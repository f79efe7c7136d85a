// LDS read bandwidth per CU for the fragment pattern of the split GEMM: ds_read_b128, lane (li = l & 31, h = l >> 5) reads
// 16 bytes at row li * STRIDE + 16 h (+ fragment offsets).  One block per CU, WAVES waves, each wave issues `iters` x 12 reads.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_bw.hip -o /tmp/lds_bw && /tmp/lds_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int STRIDE>
__global__ void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 15360; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = (float)i;
  __syncthreads();
  const unsigned char* base = smem + li * STRIDE + 16 * h + (wave & 1) * 32 * STRIDE;
  f4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const f4 v = *reinterpret_cast<const f4*>(base + (j % 6) * 64 * STRIDE % 40960 + (j / 6) * 32);
      acc += v;
    }
    asm volatile("" ::: "memory");
  }
  if (acc.x == 12345.f) out[0] = acc.y + acc.z + acc.w;
}
template <int STRIDE>
static void run(int waves, const char* name) {
  float* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  hipFuncSetAttribute((const void*)k<STRIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, 61440);
  hipLaunchKernelGGL(k<STRIDE>, dim3(256), dim3(waves * 64), 61440, 0, out, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<STRIDE>, dim3(256), dim3(waves * 64), 61440, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)iters * 12 * 1024 * waves;  // per CU
  printf("%s stride %d B, %d waves/CU: %.1f GB/s per CU = %.1f B/clk at 2.4 GHz\n", name, STRIDE, waves, bytes / ms / 1e6, bytes / ms / 1e6 / 2.4);
}
int main() {
  for (int w : {1, 2, 4, 8}) run<80>(w, "ds_read_b128");
  for (int w : {4, 8}) run<64>(w, "ds_read_b128");
  for (int w : {4, 8}) run<48>(w, "ds_read_b128");
  return 0;
}
