// Shared device/host helpers for the MTVAF gfx950 kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MTVAF_OK 0
#define MTVAF_ERR_SHAPE (-1)
#define MTVAF_ERR_ALIGN (-2)
#define MTVAF_ERR_ARG (-3)
#define MTVAF_ERR_WORKSPACE (-4)

#define MTVAF_LAUNCH_CHECK()                      \
  do {                                            \
    hipError_t e_ = hipGetLastError();            \
    if (e_ != hipSuccess) return (int)e_;         \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace mtvaf {

constexpr int WAVE = 64;

// Device-side dropout epoch (runtime.hip: mtvaf_rng_set_epoch_ptr).  NULL (default): masks are a pure function of the
// (seed, offset) passed by the host.  Non-NULL: every dropout kernel adds the 64-bit word it points to into its counter
// stream, so a CAPTURED launch (HIP graph replay: kernel arguments are frozen) draws fresh masks on every replay once a
// captured mtvaf_rng_epoch_advance has bumped the word.  Forward and backward kernels of one step read the same value.
const uint64_t* rng_epoch_ptr();
__device__ __forceinline__ uint64_t epoch_offset(uint64_t offset, const uint64_t* __restrict__ ep) {
  return ep ? offset + (*ep << 20) : offset;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG (Salmon et al. 2011).  Dropout masks are a pure function of
// (seed, site/offset, element index) so the backward pass regenerates them instead of storing them.
// ---------------------------------------------------------------------------------------------
struct uint4_ { uint32_t x, y, z, w; };

__device__ __forceinline__ uint4_ philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                             uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  return {c0, c1, c2, c3};
}

// keep-mask for 4 consecutive elements whose first linear index is idx4*4.
// returns 4 bits; element j kept iff bit j set.  p_drop in [0,1).
__device__ __forceinline__ uint32_t dropout_keep4(uint64_t seed, uint64_t offset, uint64_t idx4, float p_drop) {
  uint4_ r = philox4x32((uint32_t)idx4, (uint32_t)(idx4 >> 32), (uint32_t)offset, (uint32_t)(offset >> 32),
                        (uint32_t)seed, (uint32_t)(seed >> 32));
  // keep iff u >= p  with u = r * 2^-32
  uint32_t thr = (uint32_t)fminf(p_drop * 4294967296.0f, 4294967040.0f);
  return (r.x >= thr ? 1u : 0u) | (r.y >= thr ? 2u : 0u) | (r.z >= thr ? 4u : 0u) | (r.w >= thr ? 8u : 0u);
}
// Per-element keep decision for attention probabilities: the three attention kernels hold a
// (query, key) element in different lanes/registers, so a 4-wide Philox call would be 4x wasted in
// two of them.  A 2-round multiply-xorshift hash of (seed, offset, element index) is enough for
// dropout and costs ~8 VALU ops per element.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t attn_dropout_key(uint64_t seed, uint64_t offset) {
  return mix32((uint32_t)seed ^ mix32((uint32_t)(seed >> 32) ^ mix32((uint32_t)offset ^ 0x9E3779B9u)));
}
__device__ __forceinline__ bool attn_dropout_keep(uint32_t key, uint32_t row, uint32_t col, uint32_t thr) {
  // row = (b*NH+h)*S + q  (< 2^32 for every supported shape), col = key index
  uint32_t h = mix32(row * 0x9E3779B1u + key) ^ (col * 0x85EBCA77u);
  return mix32(h) >= thr;
}
// Cheaper split form for the attention kernels (the VALU work per probability decides their speed): the row part
// is mixed once per lane, an element costs one add, one xor, one multiply and the compare.  The top bits of the
// product depend on every bit of (rowhash ^ column term); adjacent-row / adjacent-column correlations of the
// resulting masks are at the sampling-noise level (checked over 16k rows x 164 columns).
constexpr uint32_t ATTN_DROP_C2 = 0x85EBCA77u;
__device__ __forceinline__ uint32_t attn_dropout_rowhash(uint32_t key, uint32_t row) { return mix32(row * 0x9E3779B1u + key); }
__device__ __forceinline__ bool attn_dropout_keep2(uint32_t rowhash, uint32_t col_term, uint32_t thr) {
  // col_term = col * ATTN_DROP_C2
  return (rowhash ^ col_term) * 0x2c1b3c6du >= thr;
}
__device__ __forceinline__ uint32_t attn_epoch_key(uint32_t key, const uint64_t* __restrict__ ep) {
  return ep ? mix32(key ^ ((uint32_t)(*ep) * 0x9E3779B9u + 0x7F4A7C15u)) : key;
}
__device__ __forceinline__ uint32_t dropout_threshold(float p_drop) {
  return (uint32_t)fminf(p_drop * 4294967296.0f, 4294967040.0f);
}

// erf by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. fp32 rounding level for 1 + erf):
// one v_rcp, one v_exp and six FMAs instead of the ~40-instruction libm erff in the GEMM epilogue.
// e2 = exp(-x*x/2) is shared with the GELU derivative.
__device__ __forceinline__ float erf_as(float z, float ez2) {  // z >= 0, ez2 = exp(-z*z)
  const float t = __frcp_rn(1.0f + 0.3275911f * z);
  const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  return 1.0f - poly * ez2;
}
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float e = erf_as(z, __expf(-z * z));
  return 0.5f * x * (1.0f + copysignf(e, x));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float ez2 = __expf(-z * z);
  const float e = erf_as(z, ez2);
  return 0.5f * (1.0f + copysignf(e, x)) + x * 0.39894228040143268f * ez2;
}

// The mixed-precision kernels' GELU (256 x 256 tile: 128 evaluations per lane in the epilogue -- 10 us of VALU time per tile
// with the form above against a 17-us k-loop at K = 768): Phi(x) = 0.5 erfc(-x / sqrt 2) from ONE exponential and no
// reciprocal.  log2 of the lower tail Q(z) = Phi(-z), z = min(|x|, 6), is -1 + z P(z) with P of degree 6 fitted for the
// absolute error of x Phi(x) (tools/fit_gelu.py: 1.9e-7 for GELU, 1.6e-7 for GELU' in fp32 arithmetic -- the level of the
// form above); Phi = Q or 1 - Q by the sign (no cancellation: Q <= 1/2).  Two elements per call: the polynomial compiles to
// v_pk_fma_f32.  Results of this mode are rounded to bf16 right after.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 phi_fast2(f32x2 x) {
  const f32x2 z = {fminf(fabsf(x.x), 6.0f), fminf(fabsf(x.y), 6.0f)};
  f32x2 p = {6.466566447e-06f, 6.466566447e-06f};
  p = p * z + f32x2{-3.617612674e-05f, -3.617612674e-05f};
  p = p * z + f32x2{-4.794954148e-04f, -4.794954148e-04f};
  p = p * z + f32x2{7.477600127e-03f, 7.477600127e-03f};
  p = p * z + f32x2{-5.275298283e-02f, -5.275298283e-02f};
  p = p * z + f32x2{-4.591362476e-01f, -4.591362476e-01f};
  p = p * z + f32x2{-1.151111722e+00f, -1.151111722e+00f};
  const f32x2 s = p * z - f32x2{1.0f, 1.0f};
  const f32x2 q = {__builtin_amdgcn_exp2f(s.x), __builtin_amdgcn_exp2f(s.y)};
  return f32x2{x.x < 0.f ? q.x : 1.0f - q.x, x.y < 0.f ? q.y : 1.0f - q.y};
}
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) { return x * phi_fast2(x); }
__device__ __forceinline__ f32x2 gelu_fast_grad2(f32x2 x) {
  const f32x2 e = x * x * f32x2{-0.72134752f, -0.72134752f};  // -x^2 / 2 in log2 units
  const f32x2 pdf = f32x2{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)} * f32x2{0.39894228f, 0.39894228f};
  return phi_fast2(x) + x * pdf;
}

}  // namespace mtvaf
