// AdamW update (reference modules/train.py:894-926 builds torch.optim.AdamW groups and
// :621-625 steps them) and the bf16 gradient wire format of mtvaf_amd/parallel.py.  All kernels are HBM-bound
// streams: 16 bytes per lane per access, grid sized to a few waves per SIMD.
//
//   AdamW, per element (decoupled weight decay, bias-corrected, as torch.optim.AdamW):
//     p  -= lr * wd * p
//     m   = m + (1 - b1) * (g - m)            (1 - b rounded once from double, like torch's python scalars)
//     v   = b2 * v + (1 - b2) * g * g
//     p  -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
//   7 fp32 streams per parameter (read p, g, m, v; write p, m, v) = 28 B; an optional bf16 copy of the new
//   parameter (the GEMM operand shadow of the bf16 compute mode) adds 2 B.
#include "common.h"
#include "planes.h"

namespace mtvaf {

struct AdamHyper {
  float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt, grad_scale;
  float omb1, omb2;  // 1 - beta, rounded ONCE from double (as torch does: float(1 - 0.999) != 1.f - float(0.999) by 1.3e-5)
};
static inline AdamHyper make_hyper(float lr, double beta1, double beta2, float eps, float wd, float bc1, float bc2_sqrt,
                                   float grad_scale) {
  return AdamHyper{lr, (float)beta1, (float)beta2, eps, wd, bc1, bc2_sqrt, grad_scale, (float)(1.0 - beta1), (float)(1.0 - beta2)};
}

__device__ __forceinline__ void adamw1(float& p, float g, float& m, float& v, const AdamHyper& h) {
  g *= h.grad_scale;
  p -= h.lr * h.wd * p;
  m += h.omb1 * (g - m);
  v = h.beta2 * v + h.omb2 * g * g;
  const float denom = sqrtf(v) / h.bc2_sqrt + h.eps;
  p -= (h.lr / h.bc1) * (m / denom);
}

__device__ __forceinline__ void adamw_span(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                           float* __restrict__ v, __bf16* __restrict__ ph, long n, long i0, long stride,
                                           const AdamHyper& h) {
  // vector body: 4 floats per lane per iteration when every base is 16-byte aligned (views into packed parameter
  // storage may not be), scalar otherwise and for the tail
  const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) && (!ph || ((uintptr_t)ph & 7) == 0);
  const long n4 = vec ? (n >> 2) : 0;
  for (long i = i0; i < n4; i += stride) {
    // non-temporal streams: 28 bytes per parameter pass through once per step -- keeping them out of L2 / Infinity Cache
    // leaves the GEMM panels of the backward pass this update runs next to (overlap mode) resident
    f32x4 P = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(p) + i);
    const f32x4 G = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i);
    f32x4 M = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m) + i);
    f32x4 V = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v) + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float pj = P[j], mj = M[j], vj = V[j];
      adamw1(pj, G[j], mj, vj, h);
      P[j] = pj; M[j] = mj; V[j] = vj;
    }
    reinterpret_cast<f32x4*>(p)[i] = P;  // (read again by the next forward pass)
    __builtin_nontemporal_store(M, reinterpret_cast<f32x4*>(m) + i);
    __builtin_nontemporal_store(V, reinterpret_cast<f32x4*>(v) + i);
    if (ph) {
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      bf16x4 o = {(__bf16)P.x, (__bf16)P.y, (__bf16)P.z, (__bf16)P.w};
      reinterpret_cast<bf16x4*>(ph)[i] = o;
    }
  }
  for (long i = (n4 << 2) + i0; i < n; i += stride) {
    float P = p[i], M = m[i], V = v[i];
    adamw1(P, g[i], M, V, h);
    p[i] = P; m[i] = M; v[i] = V;
    if (ph) ph[i] = (__bf16)P;
  }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, __bf16* ph, long n,
                                                    AdamHyper h) {
  adamw_span(p, g, m, v, ph, n, (long)blockIdx.x * 256 + threadIdx.x, (long)gridDim.x * 256, h);
}

// The update of a flat parameter buffer that ALSO rewrites the plane images (planes.h) of up to four row-major matrices inside it
// (round 5, pre-split operands: an encoder layer's wqkv / wo / w1 / w2): a float4 of matrix s, at element e of it, is row e / cols,
// columns e % cols .. + 3 of its image -- 6 more bytes written per weight instead of a split pass that reads the 4 again.
struct AdamSegs {
  long b4[4], e4[4];  // the matrix's float4 range in the flat buffer
  int rows[4], cols[4];
  unsigned char* img[4];
  int n;
};
__global__ __launch_bounds__(256) void adamw_planes_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, long n4, AdamHyper h, AdamSegs sg) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 P = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(p) + i);
    const f32x4 G = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i);
    f32x4 M = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m) + i);
    f32x4 V = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v) + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float pj = P[j], mj = M[j], vj = V[j];
      adamw1(pj, G[j], mj, vj, h);
      P[j] = pj; M[j] = mj; V[j] = vj;
    }
    reinterpret_cast<f32x4*>(p)[i] = P;
    __builtin_nontemporal_store(M, reinterpret_cast<f32x4*>(m) + i);
    __builtin_nontemporal_store(V, reinterpret_cast<f32x4*>(v) + i);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < sg.n && i >= sg.b4[s] && i < sg.e4[s]) {
        const long e = (i - sg.b4[s]) * 4;
        const long row = e / sg.cols[s];
        planes_store4(sg.img[s], sg.rows[s], row, (int)(e - row * sg.cols[s]), P);
      }
    }
  }
}

// several tensors of one parameter group in one launch: blockIdx.x -> (tensor, block within tensor) through a prefix table
constexpr int ADAM_MAXT = 48;
struct AdamMulti {
  float* p[ADAM_MAXT];
  const float* g[ADAM_MAXT];
  float* m[ADAM_MAXT];
  float* v[ADAM_MAXT];
  long n[ADAM_MAXT];
  int blk0[ADAM_MAXT + 1];
  int count;
};

__global__ __launch_bounds__(256) void adamw_multi_kernel(AdamMulti t, AdamHyper h) {
  int k = 0;
  const int b = blockIdx.x;
  while (k + 1 < t.count && b >= t.blk0[k + 1]) ++k;
  const int nb = t.blk0[k + 1] - t.blk0[k];
  adamw_span(t.p[k], t.g[k], t.m[k], t.v[k], nullptr, t.n[k], (long)(b - t.blk0[k]) * 256 + threadIdx.x, (long)nb * 256, h);
}

// ---- bf16 gradient wire format ------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// dst[i] = bf16(src[i]) for i < n, 0 for n <= i < npad   (npad % 8 == 0)
__global__ __launch_bounds__(256) void grad_pack_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long n, long npad) {
  const long stride = (long)gridDim.x * 256;
  for (long i8 = (long)blockIdx.x * 256 + threadIdx.x; i8 < (npad >> 3); i8 += stride) {
    const long i = i8 << 3;
    bf16x8 o;
    if (i + 8 <= n) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(src + i), b = *reinterpret_cast<const f32x4*>(src + i + 4);
      o = bf16x8{(__bf16)a.x, (__bf16)a.y, (__bf16)a.z, (__bf16)a.w, (__bf16)b.x, (__bf16)b.y, (__bf16)b.z, (__bf16)b.w};
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (i + j < n) ? (__bf16)src[i + j] : (__bf16)0.f;
    }
    *reinterpret_cast<bf16x8*>(dst + i) = o;
  }
}

// out[c] = bf16(scale * sum_{r < W} float(recv[r * chunk + c]))  -- fp32 accumulation in rank order (deterministic)
__global__ __launch_bounds__(256) void grad_reduce_kernel(const __bf16* __restrict__ recv, __bf16* __restrict__ out, int W,
                                                         long chunk, float scale) {
  const long stride = (long)gridDim.x * 256;
  for (long i8 = (long)blockIdx.x * 256 + threadIdx.x; i8 < (chunk >> 3); i8 += stride) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < W; ++r) {
      const bf16x8 x = *reinterpret_cast<const bf16x8*>(recv + (long)r * chunk + (i8 << 3));
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += (float)x[j];
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (__bf16)(acc[j] * scale);
    *reinterpret_cast<bf16x8*>(out + (i8 << 3)) = o;
  }
}

// dst[i] = float(src[i]) for i < n
__global__ __launch_bounds__(256) void grad_unpack_kernel(const __bf16* __restrict__ src, float* __restrict__ dst, long n) {
  const long stride = (long)gridDim.x * 256;
  for (long i8 = (long)blockIdx.x * 256 + threadIdx.x; (i8 << 3) < n; i8 += stride) {
    const long i = i8 << 3;
    const bf16x8 x = *reinterpret_cast<const bf16x8*>(src + i);
    if (i + 8 <= n) {
      *reinterpret_cast<f32x4*>(dst + i) = f32x4{(float)x[0], (float)x[1], (float)x[2], (float)x[3]};
      *reinterpret_cast<f32x4*>(dst + i + 4) = f32x4{(float)x[4], (float)x[5], (float)x[6], (float)x[7]};
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (i + j < n) dst[i + j] = (float)x[j];
    }
  }
}

static inline int stream_grid(long work_items) {
  long b = (work_items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 256 * 8 ? 256 * 8 : b));  // up to 8 blocks per CU: enough loads in flight for HBM
}

}  // namespace mtvaf

using namespace mtvaf;

extern "C" {

// One flat tensor (an encoder layer's parameter buffer).  bias corrections are passed in: bc1 = 1 - beta1^t,
// bc2_sqrt = sqrt(1 - beta2^t).  p_bf16 (optional) receives the updated parameter rounded to bf16.
// max_blocks > 0 caps the grid: a BACKGROUND update.  At full width (2048 blocks) the update streams at the HBM rate
// (6.3 TB/s) and starves the operand fetches of the MFMA-bound products it runs beside for its 32 us; 128 blocks trickle
// at ~1 TB/s under them instead (measured per step, same box: fp32 bs 32 18.95 -> 18.70 ms, bf16 bs 64 11.78 -> 11.56;
// 64 blocks and fewer take longer than the layer's backward pass they hide behind and lose).  0: full width.
int mtvaf_adamw(float* p, const float* g, float* m, float* v, long n, float lr, double beta1, double beta2, float eps,
                float weight_decay, float bc1, float bc2_sqrt, float grad_scale, void* p_bf16, int max_blocks,
                hipStream_t stream) {
  if (!p || !g || !m || !v || n <= 0) return MTVAF_ERR_ARG;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 3) return MTVAF_ERR_ALIGN;
  const AdamHyper h = make_hyper(lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale);
  int grid = stream_grid((n + 3) / 4);
  if (max_blocks > 0 && max_blocks < grid) grid = max_blocks;
  hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, stream, p, g, m, v,
                     static_cast<__bf16*>(p_bf16), n, h);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

// mtvaf_adamw over a flat buffer (n % 4 == 0, 16-byte aligned) that also rewrites the plane images of nseg <= 4 row-major fp32
// matrices inside it: matrix s starts at element seg_begin[s] (% 4 == 0) of the buffer, is [seg_rows[s]][seg_cols[s]] (cols % 32 == 0)
// and has its tile-blocked image (6 bytes per element) at seg_img[s].  Same update as mtvaf_adamw, bit for bit.
int mtvaf_adamw_planes(float* p, const float* g, float* m, float* v, long n, float lr, double beta1, double beta2, float eps,
                       float weight_decay, float bc1, float bc2_sqrt, float grad_scale, int nseg, const long* seg_begin,
                       const int* seg_rows, const int* seg_cols, void* const* seg_img, int max_blocks, hipStream_t stream) {
  if (!p || !g || !m || !v || n <= 0 || nseg < 0 || nseg > 4 || (nseg && (!seg_begin || !seg_rows || !seg_cols || !seg_img))) return MTVAF_ERR_ARG;
  if ((n & 3) || (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15)) return MTVAF_ERR_ALIGN;
  AdamSegs sg = {};
  sg.n = nseg;
  for (int s = 0; s < nseg; ++s) {
    const long cnt = (long)seg_rows[s] * seg_cols[s];
    if (seg_rows[s] <= 0 || seg_cols[s] <= 0 || seg_cols[s] % 32 || seg_begin[s] < 0 || (seg_begin[s] & 3) || seg_begin[s] + cnt > n || !seg_img[s] ||
        (((uintptr_t)seg_img[s]) & 15))
      return MTVAF_ERR_SHAPE;
    sg.b4[s] = seg_begin[s] / 4; sg.e4[s] = (seg_begin[s] + cnt) / 4;
    sg.rows[s] = seg_rows[s]; sg.cols[s] = seg_cols[s];
    sg.img[s] = static_cast<unsigned char*>(seg_img[s]);
  }
  const AdamHyper h = make_hyper(lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale);
  int grid = stream_grid(n / 4);
  if (max_blocks > 0 && max_blocks < grid) grid = max_blocks;
  hipLaunchKernelGGL(adamw_planes_kernel, dim3(grid), dim3(256), 0, stream, p, g, m, v, n / 4, h, sg);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

// `count` tensors sharing one set of hyper-parameters (a torch parameter group); host arrays of device pointers.
int mtvaf_adamw_multi(int count, float* const* p, const float* const* g, float* const* m, float* const* v, const long* n,
                      float lr, double beta1, double beta2, float eps, float weight_decay, float bc1, float bc2_sqrt,
                      float grad_scale, hipStream_t stream) {
  if (count < 0 || (count && (!p || !g || !m || !v || !n))) return MTVAF_ERR_ARG;
  const AdamHyper h = make_hyper(lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale);
  for (int base = 0; base < count; base += ADAM_MAXT) {
    AdamMulti t;
    t.count = count - base < ADAM_MAXT ? count - base : ADAM_MAXT;
    int blocks = 0;
    for (int k = 0; k < t.count; ++k) {
      const int i = base + k;
      if (!p[i] || !g[i] || !m[i] || !v[i] || n[i] <= 0) return MTVAF_ERR_ARG;
      if (((uintptr_t)p[i] | (uintptr_t)g[i] | (uintptr_t)m[i] | (uintptr_t)v[i]) & 3) return MTVAF_ERR_ALIGN;
      t.p[k] = p[i]; t.g[k] = g[i]; t.m[k] = m[i]; t.v[k] = v[i]; t.n[k] = n[i];
      t.blk0[k] = blocks;
      long b = ((n[i] + 3) / 4 + 255) / 256;
      blocks += (int)(b > 1024 ? 1024 : b);
    }
    t.blk0[t.count] = blocks;
    hipLaunchKernelGGL(adamw_multi_kernel, dim3(blocks), dim3(256), 0, stream, t, h);
    MTVAF_LAUNCH_CHECK();
  }
  return MTVAF_OK;
}

int mtvaf_grad_pack_bf16(const float* src, void* dst, long n, long npad, hipStream_t stream) {
  if (!src || !dst || n <= 0 || npad < n || (npad & 7)) return MTVAF_ERR_ARG;
  if (((uintptr_t)src | (uintptr_t)dst) & 15) return MTVAF_ERR_ALIGN;
  hipLaunchKernelGGL(grad_pack_kernel, dim3(stream_grid(npad / 8)), dim3(256), 0, stream, src, static_cast<__bf16*>(dst), n, npad);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

int mtvaf_grad_reduce_bf16(const void* recv, void* out, int world, long chunk, float scale, hipStream_t stream) {
  if (!recv || !out || world <= 0 || chunk <= 0 || (chunk & 7)) return MTVAF_ERR_ARG;
  if (((uintptr_t)recv | (uintptr_t)out) & 15) return MTVAF_ERR_ALIGN;
  hipLaunchKernelGGL(grad_reduce_kernel, dim3(stream_grid(chunk / 8)), dim3(256), 0, stream, static_cast<const __bf16*>(recv),
                     static_cast<__bf16*>(out), world, chunk, scale);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

int mtvaf_grad_unpack_bf16(const void* src, float* dst, long n, hipStream_t stream) {
  if (!src || !dst || n <= 0) return MTVAF_ERR_ARG;
  if (((uintptr_t)src | (uintptr_t)dst) & 15) return MTVAF_ERR_ALIGN;
  hipLaunchKernelGGL(grad_unpack_kernel, dim3(stream_grid((n + 7) / 8)), dim3(256), 0, stream, static_cast<const __bf16*>(src), dst, n);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

}  // extern "C"
