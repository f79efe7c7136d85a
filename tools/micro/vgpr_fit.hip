// Streaming kernels that hold NV float4 loads in flight (NV = 2 .. 14 -> ~16 .. 64 VGPRs): which register counts still fit into a CU
// that a 228-register, 2-waves-per-SIMD GEMM block occupies?  Built as a small shared library; tools/coresident_vgpr.py launches
// them beside the grouped weight gradients.   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/micro/libvgpr_fit.so tools/micro/vgpr_fit.hip
#include <hip/hip_runtime.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NV>
__global__ __launch_bounds__(256) void stream_kernel(const f4* __restrict__ src, f4* __restrict__ dst, long n, long stride) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
    f4 v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = src[i + k * stride];
    // (keep every loaded value live until all loads have been issued)
#pragma unroll
    for (int k = 0; k < NV; ++k) asm volatile("" : "+v"(v[k].x), "+v"(v[k].y), "+v"(v[k].z), "+v"(v[k].w));
    f4 s = v[0];
#pragma unroll
    for (int k = 1; k < NV; ++k) s += v[k];
    dst[i] = s;
  }
}
#define CASE(NV_) case NV_: hipLaunchKernelGGL((stream_kernel<NV_>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f4*)src, (f4*)dst, n, stride); break;
extern "C" int vgpr_fit_launch(int nv, const void* src, void* dst, long n, long stride, int blocks, void* stream) {
  switch (nv) {
    CASE(2) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(14)
    default: return 1;
  }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
