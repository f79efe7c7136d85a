// HBM-bound row kernels of the path (one wavefront per token row, float4 per lane, wave shuffles for
// the statistics):
//   K1  embeddings gather + LayerNorm + dropout      models/modeling_bert.py:212-222,
//                                                    models/modeling_roberta.py:105-140, 1706-1719
//   K5/K7 dropout + residual + LayerNorm             models/modeling_bert.py:354-355, 434-435
//   bias / LayerNorm-affine gradient column sums, elementwise dropout (models/bert_model.py:506).
// Dropout masks are regenerated from (seed, offset, element index) in the backward kernels.
#include "common.h"
#include "planes.h"

#include <cstdlib>

namespace mtvaf {

constexpr int MAXC = 4;  // float4 chunks per lane -> H <= 1024
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct RowCtx {
  int lane, nchunk, H;
};

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// MODE 0: z = dropout(x) + res ; y = LN(z)
// MODE 1: z = word[id] + type[tt] + pos[p] ; y = dropout(LN(z))
// MODE 2 (round 5): MODE 0 with x = slab 0 + slab 1 + ... + bias, the unreduced split-K slabs of the dense product in front of
//   it (mtvaf_gemm_f32_slabs): `x` = slab 0, `wword` = the bias, `wpos` = where the reduced x is stored (the backward pass reads
//   it), S = the slab count, `wtype` unused; slab stride = M * H.  The sum runs in the order of the reduction launch it replaces.
// PL: `out16` is not a bf16 copy but the tile-blocked PLANE IMAGE of the output (planes_store4)
template <int MODE, bool PL = false>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                    const int64_t* __restrict__ ids, const int64_t* __restrict__ tts,
                                                    const int32_t* __restrict__ pos_ids, const float* __restrict__ wword,
                                                    const float* __restrict__ wpos, const float* __restrict__ wtype,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float* __restrict__ out, float* __restrict__ mean_o,
                                                    float* __restrict__ rstd_o, int M, int S, int H, float eps,
                                                    float p_drop, uint64_t seed, uint64_t offset,
                                                    __bf16* __restrict__ out16, const uint64_t* __restrict__ epoch) {
  offset = epoch_offset(offset, epoch);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nch = H >> 2;
  const float scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    f32x4 z[MAXC];
    float s = 0.f;
    const float* wr = nullptr; const float* tr = nullptr; const float* pr = nullptr;
    if (MODE == 1) {
      wr = wword + (long)ids[row] * H;
      tr = wtype + (long)tts[row] * H;
      const int p = pos_ids ? pos_ids[row] : (row % S);
      pr = wpos + (long)p * H;
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < nch) {
        if (MODE == 0 || MODE == 2) {
          f32x4 xv = *reinterpret_cast<const f32x4*>(x + (long)row * H + c * 4);
          if (MODE == 2) {
            for (int zs = 1; zs < S; ++zs) xv += *reinterpret_cast<const f32x4*>(x + (long)zs * M * H + (long)row * H + c * 4);
            if (wword) xv += *reinterpret_cast<const f32x4*>(wword + c * 4);
            *reinterpret_cast<f32x4*>(const_cast<float*>(wpos) + (long)row * H + c * 4) = xv;
          }
          f32x4 rv = *reinterpret_cast<const f32x4*>(res + (long)row * H + c * 4);
          if (p_drop > 0.f) {
            const uint32_t k = dropout_keep4(seed, offset, (uint64_t)row * nch + c, p_drop);
            xv.x = (k & 1) ? xv.x * scale : 0.f; xv.y = (k & 2) ? xv.y * scale : 0.f;
            xv.z = (k & 4) ? xv.z * scale : 0.f; xv.w = (k & 8) ? xv.w * scale : 0.f;
          }
          z[i] = xv + rv;
        } else {
          z[i] = *reinterpret_cast<const f32x4*>(wr + c * 4) + *reinterpret_cast<const f32x4*>(tr + c * 4) +
                 *reinterpret_cast<const f32x4*>(pr + c * 4);
        }
        s += z[i].x + z[i].y + z[i].z + z[i].w;
      }
    }
    const float mu = wave_sum(s) / H;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        const f32x4 d = z[i] - mu;
        v += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
      }
    }
    const float rstd = rsqrtf(wave_sum(v) / H + eps);
    if (lane == 0) { mean_o[row] = mu; rstd_o[row] = rstd; }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c * 4);
        f32x4 y = (z[i] - mu) * rstd * g + b;
        if (MODE == 1 && p_drop > 0.f) {
          const uint32_t k = dropout_keep4(seed, offset, (uint64_t)row * nch + c, p_drop);
          y.x = (k & 1) ? y.x * scale : 0.f; y.y = (k & 2) ? y.y * scale : 0.f;
          y.z = (k & 4) ? y.z * scale : 0.f; y.w = (k & 8) ? y.w * scale : 0.f;
        }
        *reinterpret_cast<f32x4*>(out + (long)row * H + c * 4) = y;
        if constexpr (PL) {
          planes_store4(reinterpret_cast<unsigned char*>(out16), M, row, c * 4, y);
        } else if (out16) {  // bf16 copy: the next projection's operand in the mixed-precision mode (no separate cast pass)
          *reinterpret_cast<bf16x4*>(out16 + (long)row * H + c * 4) = bf16x4{(__bf16)y.x, (__bf16)y.y, (__bf16)y.z, (__bf16)y.w};
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward.  partials layout: [gridDim.x][NP][H] with NP = 3 in MODE 0 (dgamma, dbeta, column sums of dx
// = the bias gradient of the dense layer that produced x) and NP = 4 in MODE 1 (dgamma, dbeta, 2 token-type rows).  MODE 0 writes dx (grad wrt the dropout input) and dres (grad wrt the residual);
// MODE 1 writes dz (grad wrt the summed embeddings) to dx.
// ------------------------------------------------------------------------------------------
template <int MODE, bool PL = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                    const float* __restrict__ res, const int64_t* __restrict__ ids,
                                                    const int64_t* __restrict__ tts, const int32_t* __restrict__ pos_ids,
                                                    const float* __restrict__ wword, const float* __restrict__ wpos,
                                                    const float* __restrict__ wtype, const float* __restrict__ gamma,
                                                    const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                    float* __restrict__ dx, float* __restrict__ dres, int dres_acc,
                                                    float* __restrict__ partials, int M, int S, int H, float p_drop,
                                                    uint64_t seed, uint64_t offset, __bf16* __restrict__ dx16,
                                                    const uint64_t* __restrict__ epoch) {
  offset = epoch_offset(offset, epoch);
  constexpr int NP = MODE == 1 ? 4 : 3;
  __shared__ f32x4 red[4][MAXC * 64];  // [wave][H/4 <= 256]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nch = H >> 2;
  const float scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  f32x4 ag[MAXC], ab[MAXC], at0[MAXC], at1[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) ag[i] = ab[i] = at0[i] = at1[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    const float mu = mean_i[row], rstd = rstd_i[row];
    f32x4 xh[MAXC], g[MAXC];
    const float* wr = nullptr; const float* tr = nullptr; const float* pr = nullptr;
    int tt = 0;
    if (MODE == 1) {
      wr = wword + (long)ids[row] * H;
      tt = (int)tts[row];
      tr = wtype + (long)tt * H;
      const int p = pos_ids ? pos_ids[row] : (row % S);
      pr = wpos + (long)p * H;
    }
    float s1 = 0.f, s2 = 0.f;
    uint32_t keep[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      keep[i] = 0xF;
      xh[i] = g[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < nch) {
        if (p_drop > 0.f) keep[i] = dropout_keep4(seed, offset, (uint64_t)row * nch + c, p_drop);
        f32x4 z;
        if (MODE == 0) {
          f32x4 xv = *reinterpret_cast<const f32x4*>(x + (long)row * H + c * 4);
          const f32x4 rv = *reinterpret_cast<const f32x4*>(res + (long)row * H + c * 4);
          if (p_drop > 0.f) {
            xv.x = (keep[i] & 1) ? xv.x * scale : 0.f; xv.y = (keep[i] & 2) ? xv.y * scale : 0.f;
            xv.z = (keep[i] & 4) ? xv.z * scale : 0.f; xv.w = (keep[i] & 8) ? xv.w * scale : 0.f;
          }
          z = xv + rv;
        } else {
          z = *reinterpret_cast<const f32x4*>(wr + c * 4) + *reinterpret_cast<const f32x4*>(tr + c * 4) +
              *reinterpret_cast<const f32x4*>(pr + c * 4);
        }
        xh[i] = (z - mu) * rstd;
        f32x4 dy = *reinterpret_cast<const f32x4*>(dout + (long)row * H + c * 4);
        if (MODE == 0 && wword) {  // dout = (slab 0 + slab 1 + ...) + dout: the unreduced dX product in front (mtvaf_gemm_f32_slabs, accumulate)
          f32x4 sl = *reinterpret_cast<const f32x4*>(wword + (long)row * H + c * 4);
          for (int zs = 1; zs < S; ++zs) sl += *reinterpret_cast<const f32x4*>(wword + (long)zs * M * H + (long)row * H + c * 4);
          dy = sl + dy;
        }
        if (MODE == 1 && p_drop > 0.f) {
          dy.x = (keep[i] & 1) ? dy.x * scale : 0.f; dy.y = (keep[i] & 2) ? dy.y * scale : 0.f;
          dy.z = (keep[i] & 4) ? dy.z * scale : 0.f; dy.w = (keep[i] & 8) ? dy.w * scale : 0.f;
        }
        ag[i] += dy * xh[i];
        ab[i] += dy;
        g[i] = dy * *reinterpret_cast<const f32x4*>(gamma + c * 4);
        s1 += g[i].x + g[i].y + g[i].z + g[i].w;
        s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
      }
    }
    const float m1 = wave_sum(s1) / H, m2 = wave_sum(s2) / H;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        const f32x4 dz = (g[i] - m1 - xh[i] * m2) * rstd;
        if (MODE == 0) {
          f32x4 r = dz;
          float* dr = dres + (long)row * H + c * 4;
          if (dres_acc) r += *reinterpret_cast<const f32x4*>(dr);
          *reinterpret_cast<f32x4*>(dr) = r;
          f32x4 d = dz;
          if (p_drop > 0.f) {
            d.x = (keep[i] & 1) ? d.x * scale : 0.f; d.y = (keep[i] & 2) ? d.y * scale : 0.f;
            d.z = (keep[i] & 4) ? d.z * scale : 0.f; d.w = (keep[i] & 8) ? d.w * scale : 0.f;
          }
          if (dx) *reinterpret_cast<f32x4*>(dx + (long)row * H + c * 4) = d;
          if constexpr (PL) {  // the gradient is only a GEMM operand downstream: its plane image, no fp32 copy needed
            planes_store4(reinterpret_cast<unsigned char*>(dx16), M, row, c * 4, d);
          } else if (dx16) {  // the gradient is only a GEMM operand downstream (dX and dW products of the dense layer)
            *reinterpret_cast<bf16x4*>(dx16 + (long)row * H + c * 4) = bf16x4{(__bf16)d.x, (__bf16)d.y, (__bf16)d.z, (__bf16)d.w};
          }
          at0[i] += d;
        } else {
          *reinterpret_cast<f32x4*>(dx + (long)row * H + c * 4) = dz;
          if (tt == 0) at0[i] += dz; else if (tt == 1) at1[i] += dz;
        }
      }
    }
  }
  // block reduction of the per-wave column partials, in fixed wave order (deterministic)
#pragma unroll
  for (int np = 0; np < NP; ++np) {
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      red[wave][c] = np == 0 ? ag[i] : np == 1 ? ab[i] : np == 2 ? at0[i] : at1[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < nch; c += blockDim.x) {
      const f32x4 t = red[0][c] + red[1][c] + red[2][c] + red[3][c];
      *reinterpret_cast<f32x4*>(partials + ((long)blockIdx.x * NP + np) * H + c * 4) = t;
    }
    __syncthreads();
  }
}

// MODE 0 in a form that fits INTO the CUs a one-round GEMM launch occupies (round 6).  The grouped weight gradients of a layer run
// on the second stream as one block per CU on 216 of 256 CUs (228 registers x 2 waves per SIMD, 144 of 160 KiB of LDS), and the
// LayerNorm backward of the main stream (169 registers, 16 KiB) was confined to the 40 free CUs: 91 us instead of 18
// (profiles/r06_coresident_probe.txt).  Here a ROW is spread over the block (H / 4 threads: one float4 column chunk per lane, 192
// threads at H = 768) instead of over one wave, so a lane carries 4 instead of 16 elements of every row vector and the column
// partials of its own chunk only: <= 48 registers, 64 bytes of LDS (the two row statistics cross the waves through it, one barrier
// per row, double-buffered).  Same arithmetic per element as ln_bwd_kernel<0>; the row sums s1 / s2 and the column partials are
// summed in another (fixed) order.  H % 256 == 0, H <= 1024.  partials: [gridDim.x][3][H] as in ln_bwd_kernel.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
// every access of a row = a buffer descriptor (scalar registers, built from kernel arguments) + ONE 32-bit lane offset + the row's
// byte offset in a scalar register: the 64-bit vector addresses of ten arrays would cost the lane more registers than its data
#define LEAN_RSRC(p, bytes) __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(p)), 0, (int)(bytes), 0x00020000)
__device__ __forceinline__ f32x4 lean_ld(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned sb) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, v, sb, 0));
}
__device__ __forceinline__ void lean_st(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned sb, f32x4 x) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, x), r, v, sb, 0);
}
// sum over the 64 lanes by DPP row operations (no index registers, unlike __shfl_xor's ds_bpermute): -> the sum, wave-uniform
#define LEAN_DPP(x, ctrl, rows) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (ctrl), (rows), 0xF, false))
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += LEAN_DPP(v, 0xB1, 0xF);   // quad_perm [1,0,3,2]
  v += LEAN_DPP(v, 0x4E, 0xF);   // quad_perm [2,3,0,1]
  v += LEAN_DPP(v, 0x141, 0xF);  // row_half_mirror
  v += LEAN_DPP(v, 0x140, 0xF);  // row_mirror: every lane = the sum of its row of 16
  v += LEAN_DPP(v, 0x142, 0xA);  // row_bcast15 into rows 1 and 3 (the other rows add 0)
  v += LEAN_DPP(v, 0x143, 0xC);  // row_bcast31 into rows 2 and 3: lane 63 = the sum
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float uni(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }

// NS: the slab count as a compile-time constant (0, 1, 2: straight-line loads, all of a row's requests in flight together) or -1 (any)
template <bool PL, int NS>
__global__ __launch_bounds__(256) void ln_bwd_lean_kernel(const float* __restrict__ dout, const float* __restrict__ slabs, int nslab,
                                                         const float* __restrict__ x, const float* __restrict__ res,
                                                         const float* __restrict__ gamma, const float* __restrict__ mean_i,
                                                         const float* __restrict__ rstd_i, float* __restrict__ dx,
                                                         float* __restrict__ dres, int dres_acc, float* __restrict__ partials, int M,
                                                         int H, float p_drop, uint64_t seed, uint64_t offset,
                                                         __bf16* __restrict__ dx16, const uint64_t* __restrict__ epoch) {
  offset = epoch_offset(offset, epoch);
  __shared__ float red[2][4][2];
  const int c = threadIdx.x;  // this lane's float4 column chunk; blockDim.x == H / 4
  const int lane = c & 63, wave = c >> 6;
  const int nch = H >> 2;
  const float scale = uni(p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f);
  const float invH = uni(1.f / H);
  const uint32_t thr = __builtin_amdgcn_readfirstlane((uint32_t)fminf(p_drop * 4294967296.0f, 4294967040.0f));
  const unsigned rowb = (unsigned)H * 4u;        // bytes of a row
  const unsigned tb = (unsigned)M * rowb;        // bytes of a [M, H] tensor (the host checks nslab * tb, 1.5 * tb < 2^32)
  const unsigned lo = (unsigned)c * 16u;         // byte offset of the lane's chunk in a row ...
  const unsigned lp = (unsigned)(c >> 3) * 3u * (unsigned)M * 64u + (unsigned)(c & 7) * 8u;  // ... and in the plane image [H / 32][3][M][32]
  const auto r_dout = LEAN_RSRC(dout, tb), r_x = LEAN_RSRC(x, tb), r_res = LEAN_RSRC(res, tb), r_dres = LEAN_RSRC(dres, tb);
  const auto r_sl = LEAN_RSRC(slabs, slabs ? (unsigned)nslab * tb : 0u);
  const auto r_dx = LEAN_RSRC(dx, dx ? tb : 0u);                          // (a zero-record descriptor drops the store)
  const auto r_16 = LEAN_RSRC(dx16, dx16 ? (PL ? tb / 2u * 3u : tb / 2u) : 0u);
  if (c < 16) reinterpret_cast<float*>(red)[c] = 0.f;  // (entries of waves this block does not have read as zero)
  __syncthreads();
  const auto r_gam = LEAN_RSRC(gamma, rowb);
  f32x4 ag = f32x4{0.f, 0.f, 0.f, 0.f}, ab = ag, at0 = ag;
  int par = 0;
  for (int row = blockIdx.x; row < M; row += gridDim.x, par ^= 1) {
    const unsigned rb = (unsigned)row * rowb;
    const float mu = mean_i[row], rstd = rstd_i[row];
    uint32_t keep = 0xF;
    if (p_drop > 0.f) {  // dropout_keep4 with its threshold in a scalar register
      const uint64_t idx4 = (uint64_t)row * nch + c;
      const uint4_ r = philox4x32((uint32_t)idx4, (uint32_t)(idx4 >> 32), (uint32_t)offset, (uint32_t)(offset >> 32), (uint32_t)seed,
                                  (uint32_t)(seed >> 32));
      keep = (r.x >= thr ? 1u : 0u) | (r.y >= thr ? 2u : 0u) | (r.z >= thr ? 4u : 0u) | (r.w >= thr ? 8u : 0u);
    }
    f32x4 dy = lean_ld(r_dout, lo, rb);
    f32x4 xv = lean_ld(r_x, lo, rb);
    const f32x4 rv = lean_ld(r_res, lo, rb);
    const f32x4 gam = lean_ld(r_gam, lo, 0);  // (re-read per row from the cache: four registers less across the loop)
    // dout = (slab 0 + slab 1 + ...) + dout, in the order of the reduction launch this replaces
    if constexpr (NS == 1) {
      dy = lean_ld(r_sl, lo, rb) + dy;
    } else if constexpr (NS == 2) {
      const f32x4 s0 = lean_ld(r_sl, lo, rb), s1_ = lean_ld(r_sl, lo, tb + rb);
      dy = (s0 + s1_) + dy;
    } else if constexpr (NS < 0) {
      f32x4 sl = lean_ld(r_sl, lo, rb);
      for (int zs = 1; zs < nslab; ++zs) sl += lean_ld(r_sl, lo, (unsigned)zs * tb + rb);
      dy = sl + dy;
    }
    f32x4 xh;
    {
      if (p_drop > 0.f) {
        xv.x = (keep & 1) ? xv.x * scale : 0.f; xv.y = (keep & 2) ? xv.y * scale : 0.f;
        xv.z = (keep & 4) ? xv.z * scale : 0.f; xv.w = (keep & 8) ? xv.w * scale : 0.f;
      }
      xh = ((xv + rv) - mu) * rstd;
    }
    ag += dy * xh;
    ab += dy;
    const f32x4 g = dy * gam;
    float s1 = wave_sum_dpp(g.x + g.y + g.z + g.w);
    float s2 = wave_sum_dpp(g.x * xh.x + g.y * xh.y + g.z * xh.z + g.w * xh.w);
    if (lane == 0) { red[par][wave][0] = s1; red[par][wave][1] = s2; }
    __syncthreads();
    s1 = ((red[par][0][0] + red[par][1][0]) + red[par][2][0]) + red[par][3][0];
    s2 = ((red[par][0][1] + red[par][1][1]) + red[par][2][1]) + red[par][3][1];
    const float m1 = s1 * invH, m2 = s2 * invH;
    const f32x4 dz = (g - m1 - xh * m2) * rstd;
    {
      f32x4 r = dz;
      if (dres_acc) r += lean_ld(r_dres, lo, rb);
      lean_st(r_dres, lo, rb, r);
    }
    f32x4 d = dz;
    if (p_drop > 0.f) {
      d.x = (keep & 1) ? d.x * scale : 0.f; d.y = (keep & 2) ? d.y * scale : 0.f;
      d.z = (keep & 4) ? d.z * scale : 0.f; d.w = (keep & 8) ? d.w * scale : 0.f;
    }
    lean_st(r_dx, lo, rb, d);
    if constexpr (PL) {  // planes_store4 with the row's share of the address in scalar registers: bit for bit the same image
      f32x4 v = d;
      asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
      unsigned h0, m0, l0, h1, m1_, l1;
      pl_split(pl_f32x2{v.x, v.y}, h0, m0, l0);
      pl_split(pl_f32x2{v.z, v.w}, h1, m1_, l1);
      const unsigned pr = (unsigned)row * 64u, ps = (unsigned)M * 64u;
      __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{h0, h1}, r_16, lp, pr, 0);
      __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{m0, m1_}, r_16, lp, pr + ps, 0);
      __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{l0, l1}, r_16, lp, pr + 2u * ps, 0);
    } else {
      const bf16x4 q = bf16x4{(__bf16)d.x, (__bf16)d.y, (__bf16)d.z, (__bf16)d.w};
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, q), r_16, lo >> 1, rb >> 1, 0);
    }
    at0 += d;
  }
  float* pp = partials + (size_t)blockIdx.x * 3 * H + c * 4;
  *reinterpret_cast<f32x4*>(pp) = ag;
  *reinterpret_cast<f32x4*>(pp + H) = ab;
  *reinterpret_cast<f32x4*>(pp + 2 * H) = at0;
}

// out[c] = (acc ? out[c] : 0) + sum_r x[r*ld + c].  Block = 32 columns x 8 row groups; the 8 partial sums
// are combined in fixed order through LDS (deterministic).  grid = ceil(cols / 32).
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ x, int rows, int cols, long ld,
                                                          float* __restrict__ out, int accumulate) {
  __shared__ float red[8][32];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < cols) {
    int r = rg;
    for (; r + 24 < rows; r += 32) {
      s0 += x[(long)r * ld + c];
      s1 += x[(long)(r + 8) * ld + c];
      s2 += x[(long)(r + 16) * ld + c];
      s3 += x[(long)(r + 24) * ld + c];
    }
    for (; r < rows; r += 8) s0 += x[(long)r * ld + c];
  }
  red[rg][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && c < cols) {
    float s = red[0][cl];
#pragma unroll
    for (int i = 1; i < 8; ++i) s += red[i][cl];
    if (accumulate) s += out[c];
    out[c] = s;
  }
}

// Same reduction for NP stacked partial rows [rows][NP][H] -> NP separate outputs in ONE launch
// (LayerNorm backward: dgamma, dbeta, dense-bias gradient, token-type rows).  grid = ceil(NP*H / 32).
// RG row groups per block (blockDim = 32 RG): 8 (256 threads) or 32 (round 5: the 512 partial rows of a LayerNorm backward are a
// serial chain of 64 loads per thread with 8 groups -- the launch is latency, not bandwidth: 7 us for 4.7 MB -- and of 16 with 32)
struct OutPtrs { float* p[4]; };
template <int RG>
__global__ __launch_bounds__(32 * RG) void colsum_final_multi_kernel(const float* __restrict__ x, int rows, int H, int NP,
                                                                    OutPtrs outs, int accumulate) {
  __shared__ float red[RG][32];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int cols = NP * H;
  const long ld = cols;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < cols) {
    int r = rg;
    for (; r + 3 * RG < rows; r += 4 * RG) {
      s0 += x[(long)r * ld + c];
      s1 += x[(long)(r + RG) * ld + c];
      s2 += x[(long)(r + 2 * RG) * ld + c];
      s3 += x[(long)(r + 3 * RG) * ld + c];
    }
    for (; r < rows; r += RG) s0 += x[(long)r * ld + c];
  }
  red[rg][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && c < cols) {
    float* out = outs.p[c / H];
    if (!out) return;
    float s = red[0][cl];
#pragma unroll
    for (int i = 1; i < RG; ++i) s += red[i][cl];
    const int cc = c % H;
    if (accumulate) s += out[cc];
    out[cc] = s;
  }
}

// stage 1 of a tall column sum: partial[chunk][c] = sum over this chunk's rows
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int rows, int cols, long ld,
                                                            float* __restrict__ partial, int rows_per_chunk) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int r0 = blockIdx.y * rows_per_chunk;
  const int r1 = min(rows, r0 + rows_per_chunk);
  float s = 0.f;
  if (c < cols)
    for (int r = r0 + rg; r < r1; r += 4) s += x[(long)r * ld + c];
  red[rg][cl] = s;
  __syncthreads();
  if (rg == 0 && c < cols) partial[(long)blockIdx.y * cols + c] = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
}

__global__ void embed_scatter_kernel(const float* __restrict__ dz, const int64_t* __restrict__ ids,
                                     const int32_t* __restrict__ pos_ids, float* __restrict__ dword,
                                     float* __restrict__ dpos, int M, int H, int word_pad, int pos_pad) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    const long id = ids[row];
    const int p = pos_ids ? pos_ids[row] : -1;
    for (int c = lane; c < H; c += 64) {
      const float v = dz[(long)row * H + c];
      if (id != word_pad) atomicAdd(dword + id * H + c, v);
      if (pos_ids && p != pos_pad) atomicAdd(dpos + (long)p * H + c, v);
    }
  }
}

// Deterministic form of the scatter-add (the default; MTVAF_EMBED_ATOMIC=1 keeps the float-atomic kernel above), O(M + V):
//   1. row_nonzero_kernel: flag[r] = the gradient row r is not EXACTLY zero.  Zero rows neither own nor contribute -- adding
//      them changes nothing -- which is what keeps the segments short: the reference's dataset pads RoBERTa ids with 0 = <s>,
//      not with the padding id (modules/dataset.py:414-415), so ~40 % of a batch's rows share ONE id, all with exact-zero
//      gradients (masked rows: DESIGN.md section 4.5b);
//   2. count[id] (integer atomics: the final counts do not depend on order), an exclusive scan over the table's V ids ->
//      offset[id], and a fill pass seg[offset[id] + cursor[id]++] = r: every id's rows sit in one segment, in ARBITRARY order;
//   3. one wave per flagged row: the row that is the MINIMUM of its segment owns the table row; it ranks the segment (each
//      lane counts the entries below its own), walks it in increasing row order four rows at a time and adds the sum to the
//      table with plain stores.  One writer per table row, a fixed summation order, no float atomics.
// KEY selects the table: 0 word ids (int64), 1 position ids (int32, RoBERTa).  (A first version let the first row of an id
// find its partners by scanning the other rows' ids: O(M^2 / 64) dependent L2 round trips, 2 ms at 65 536 rows.)
constexpr int DET_NC = 16;  // columns per lane of the deterministic scatter: H <= 1024
__global__ __launch_bounds__(256) void row_nonzero_kernel(const float* __restrict__ dz, int* __restrict__ flags, int M, int H) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    bool nz = false;
    for (int c = lane * 4; c < H; c += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(dz + (long)row * H + c);
      nz = nz || v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f;
    }
    const bool any = __ballot(nz) != 0ull;
    if (lane == 0) flags[row] = any ? 1 : 0;
  }
}
template <int KEY>
__device__ __forceinline__ long det_key(const int64_t* __restrict__ ids, const int32_t* __restrict__ pos_ids, int r) {
  return KEY == 0 ? (long)ids[r] : (long)pos_ids[r];
}
template <int KEY>
__global__ __launch_bounds__(256) void det_count_kernel(const int64_t* __restrict__ ids, const int32_t* __restrict__ pos_ids,
                                                       const int* __restrict__ flags, int* __restrict__ count, int M, long pad) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= M) return;
  const long id = det_key<KEY>(ids, pos_ids, r);
  if (id != pad && flags[r]) atomicAdd(count + id, 1);
}
// offset[i] = sum of count[0 .. i) for the V table rows: one block of 1024 threads; a thread owns a contiguous share of
// PER = 4 * ceil(V / 4096) entries, fetched as int4 vectors (all loads in flight before the first add: as a plain loop the
// scan took 47 us of dependent L2 round trips), the shares are scanned with wave shuffles.  count / offset are allocated with
// room for 1024 * PER entries; entries beyond V are zero (the fill that precedes the count pass covers them).
__global__ __launch_bounds__(1024) void det_scan_kernel(const int* __restrict__ count, int* __restrict__ offset, int V) {
  __shared__ int wsum[16];
  constexpr int MAXQ = 16;  // int4 vectors per thread: V <= 65536
  const int nq = (V + 4095) / 4096;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int4* src = reinterpret_cast<const int4*>(count) + (long)threadIdx.x * nq;
  int4 v[MAXQ];
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) v[i] = i < nq ? src[i] : int4{0, 0, 0, 0};
  int s = 0;
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
  // inclusive scan of the thread sums inside the wave, then over the 16 waves
  int inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  int run = base + inc - s;  // exclusive prefix of this thread's share
  int4* dst = reinterpret_cast<int4*>(offset) + (long)threadIdx.x * nq;
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) {
    if (i < nq) {
      int4 o;
      o.x = run; run += v[i].x;
      o.y = run; run += v[i].y;
      o.z = run; run += v[i].z;
      o.w = run; run += v[i].w;
      dst[i] = o;
    }
  }
}
template <int KEY>
__global__ __launch_bounds__(256) void det_fill_kernel(const int64_t* __restrict__ ids, const int32_t* __restrict__ pos_ids,
                                                      const int* __restrict__ flags, const int* __restrict__ offset,
                                                      int* __restrict__ cursor, int* __restrict__ seg, int M, long pad) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= M) return;
  const long id = det_key<KEY>(ids, pos_ids, r);
  if (id != pad && flags[r]) seg[offset[id] + atomicAdd(cursor + id, 1)] = r;
}
template <int KEY>
__global__ __launch_bounds__(256) void embed_scatter_det_kernel(const float* __restrict__ dz, const int64_t* __restrict__ ids,
                                                               const int32_t* __restrict__ pos_ids, const int* __restrict__ flags,
                                                               const int* __restrict__ count, const int* __restrict__ offset,
                                                               const int* __restrict__ seg, float* __restrict__ table, int M,
                                                               int H, long pad) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    const long id = det_key<KEY>(ids, pos_ids, row);
    if (id == pad || flags[row] == 0) continue;  // (wave-uniform)
    const int n = count[id];
    const int* sg = seg + offset[id];
    // the owner is the smallest row of the segment (wave-uniform decision)
    int mn = 0x7fffffff;
    for (int i = lane; i < n; i += 64) mn = min(mn, sg[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mn = min(mn, __shfl_xor(mn, o, 64));
    if (mn != row) continue;
    float acc[DET_NC];
#pragma unroll
    for (int i = 0; i < DET_NC; ++i) acc[i] = 0.f;
    auto add_rows = [&](const int (&rr)[4], const bool (&ok)[4]) {
      float v[4][DET_NC];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < DET_NC; ++i) {
          const int c = lane + 64 * i;
          v[u][i] = (c < H) ? dz[(long)rr[u] * H + c] : 0.f;
        }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < DET_NC; ++i) acc[i] += ok[u] ? v[u][i] : 0.f;
    };
    if (n <= 64) {
      // rank the segment in registers: lane i holds entry i, rank = how many entries are smaller (rows are distinct)
      const int mine = lane < n ? sg[lane] : 0x7fffffff;
      int rank = 0;
      for (int j = 0; j < n; ++j) rank += (__shfl(mine, j, 64) < mine) ? 1 : 0;
      for (int k = 0; k < n; k += 4) {
        int rr[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          ok[u] = k + u < n;
          const unsigned long long who = __ballot(lane < n && rank == k + u);
          rr[u] = ok[u] ? __shfl(mine, (int)__builtin_ctzll(who | (1ull << 63)), 64) : row;
        }
        add_rows(rr, ok);
      }
    } else {
      // long segments (hundreds of tokens with one id and non-zero gradients: rare): repeated minimum above the last row
      int last = -1;
      for (int k = 0; k < n; k += 4) {
        int rr[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          ok[u] = k + u < n;
          int nx = 0x7fffffff;
          if (ok[u]) {
            for (int i = lane; i < n; i += 64) { const int v = sg[i]; nx = (v > last && v < nx) ? v : nx; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) nx = min(nx, __shfl_xor(nx, o, 64));
            last = nx;
          }
          rr[u] = ok[u] ? nx : row;
        }
        add_rows(rr, ok);
      }
    }
#pragma unroll
    for (int i = 0; i < DET_NC; ++i) {
      const int c = lane + 64 * i;
      if (c < H) table[id * H + c] += acc[i];  // (zero-filled just before unless the caller accumulates)
    }
  }
}

// BERT positions are arange(S): dpos[s] (+)= sum_b dz[b*S+s]   (deterministic)
__global__ void pos_reduce_kernel(const float* __restrict__ dz, float* __restrict__ dpos, int B, int S, int H,
                                  int accumulate) {
  const int s = blockIdx.x;
  for (int c = threadIdx.x; c < H; c += blockDim.x) {
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += dz[((long)b * S + s) * H + c];
    if (accumulate) a += dpos[(long)s * H + c];
    dpos[(long)s * H + c] = a;
  }
}

__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n4, float p_drop,
                               uint64_t seed, uint64_t offset, const uint64_t* __restrict__ epoch) {
  offset = epoch_offset(offset, epoch);
  const float scale = 1.f / (1.f - p_drop);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
    const uint32_t k = dropout_keep4(seed, offset, (uint64_t)i, p_drop);
    v.x = (k & 1) ? v.x * scale : 0.f; v.y = (k & 2) ? v.y * scale : 0.f;
    v.z = (k & 4) ? v.z * scale : 0.f; v.w = (k & 8) ? v.w * scale : 0.f;
    *reinterpret_cast<f32x4*>(y + i * 4) = v;
  }
}

// RoBERTa position ids: cumsum(ids != pad) * (ids != pad) + pad   (one wave per sequence)
__global__ void roberta_pos_kernel(const int64_t* __restrict__ ids, int32_t* __restrict__ pos, int B, int S, int pad) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  int carry = 0;
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    const int m = (s < S && ids[(long)b * S + s] != pad) ? 1 : 0;
    int v = m;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(v, o, 64);
      if (lane >= o) v += t;
    }
    if (s < S) pos[(long)b * S + s] = (carry + v) * m + pad;
    carry += __shfl(v, 63, 64);
  }
}

// zero fill as a KERNEL (not hipMemsetAsync): a captured 94-MB memset node replayed wrongly on ROCm 7.2 (every fourth
// column of the word-table gradient kept stale data under HIP graph replay), a kernel node replays exactly
__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ p, long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
    reinterpret_cast<f32x4*>(p)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = 0.f;
}
static inline void zero_f32(float* p, long n, hipStream_t st) {
  const long b = ((n >> 2) + 255) / 256;
  hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b))), dim3(256), 0, st, p, n);
}

static inline int row_grid(int M) { return std::max(1, std::min((M + 3) / 4, 1024)); }
// LN backward keeps per-block column partials: fewer, fatter blocks (2 per CU) keep the partial slab small
// blocks of the LayerNorm backward (4 rows per block and pass; each block leaves one row of column partials).
// MTVAF_LN_BWD_BLOCKS overrides the cap (experiment switch; read once: the partial-buffer size follows it).
static inline int row_grid_bwd(int M) {
  static const int cap = [] { const char* e = getenv("MTVAF_LN_BWD_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();
  return std::max(1, std::min((M + 3) / 4, cap));
}

// the lean LayerNorm backward (ln_bwd_lean_kernel) wherever its shape rule holds; MTVAF_LN_LEAN=0: the one-wave-per-row kernel
static inline bool ln_lean(int M, int H, int nslab) {
  const char* e = getenv("MTVAF_LN_LEAN");  // (read per call: the parity test switches between the two kernels in one process)
  const int on = e ? atoi(e) : 1;
  // (32-bit buffer offsets: the slab stack and the plane image must stay below 2 GiB)
  return on && H % 256 == 0 && H <= 1024 && (size_t)std::max(nslab, 2) * M * H * 4 < ((size_t)1 << 31);
}

template <bool PL>
static void ln_bwd_lean(int g, hipStream_t st, const float* dout, const float* slabs, int nslab, const float* x, const float* res,
                        const float* gamma, const float* mean, const float* rstd, float* dx, float* dres, int dres_acc, float* part, int M,
                        int H, float p_drop, uint64_t seed, uint64_t offset, __bf16* dx16) {
  if (!slabs) nslab = 0;
#define MTVAF_LEAN(NS_)                                                                                                             \
  hipLaunchKernelGGL((ln_bwd_lean_kernel<PL, NS_>), dim3(g), dim3(H / 4), 0, st, dout, slabs, nslab, x, res, gamma, mean, rstd, dx, \
                     dres, dres_acc, part, M, H, p_drop, seed, offset, dx16, rng_epoch_ptr())
  if (nslab == 0) MTVAF_LEAN(0);
  else if (nslab == 1) MTVAF_LEAN(1);
  else if (nslab == 2) MTVAF_LEAN(2);
  else MTVAF_LEAN(-1);
#undef MTVAF_LEAN
}

}  // namespace mtvaf

using namespace mtvaf;

extern "C" {

size_t mtvaf_ln_bwd_workspace_bytes(int M, int H) { return (size_t)row_grid_bwd(M) * 4 * H * sizeof(float); }
// scratch of mtvaf_embed_ln_bwd: the LayerNorm column partials + the deterministic scatter's flags / segments ([M] each) and
// per-id count / cursor / offset tables ([max(vocab, max_pos)] each)
size_t mtvaf_embed_ln_bwd_workspace_bytes(int M, int H, int vocab, int max_pos) {
  return (size_t)row_grid_bwd(M) * 4 * H * sizeof(float) + ((size_t)2 * M + (size_t)3 * ((std::max(vocab, max_pos) + 4095) / 4096 * 4096)) * sizeof(int) + 64;
}

int mtvaf_roberta_position_ids(const int64_t* ids, int32_t* pos_ids, int B, int S, int pad_idx, hipStream_t st) {
  if (B <= 0 || S <= 0) return MTVAF_ERR_SHAPE;
  hipLaunchKernelGGL(roberta_pos_kernel, dim3(B), dim3(64), 0, st, ids, pos_ids, B, S, pad_idx);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

int mtvaf_embed_ln_fwd(const int64_t* ids, const int64_t* type_ids, const int32_t* pos_ids, const float* word,
                       const float* pos, const float* type, const float* gamma, const float* beta, float* out,
                       float* mean, float* rstd, int B, int S, int H, float eps, float p_drop, uint64_t seed,
                       uint64_t offset, void* out_bf16, hipStream_t st) {
  if (H % 4 || H > MAXC * 256 || B <= 0 || S <= 0) return MTVAF_ERR_SHAPE;
  const int M = B * S;
  hipLaunchKernelGGL((ln_fwd_kernel<1>), dim3(row_grid(M)), dim3(256), 0, st, nullptr, nullptr, ids, type_ids, pos_ids,
                     word, pos, type, gamma, beta, out, mean, rstd, M, S, H, eps, p_drop, seed, offset,
                     static_cast<__bf16*>(out_bf16), rng_epoch_ptr());
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

int mtvaf_dropout_res_ln_fwd_slabs(const float* slabs, int nslab, const float* bias, float* x_out, const float* res, const float* gamma,
                                   const float* beta, float* out, float* mean, float* rstd, int M, int H, float eps, float p_drop,
                                   uint64_t seed, uint64_t offset, void* out_bf16, hipStream_t st) {
  if (H % 4 || H > MAXC * 256 || M <= 0 || nslab < 1) return MTVAF_ERR_SHAPE;
  if (!slabs || !x_out) return MTVAF_ERR_ARG;
  hipLaunchKernelGGL((ln_fwd_kernel<2>), dim3(row_grid(M)), dim3(256), 0, st, slabs, res, nullptr, nullptr, nullptr, bias, x_out,
                     nullptr, gamma, beta, out, mean, rstd, M, nslab, H, eps, p_drop, seed, offset, static_cast<__bf16*>(out_bf16),
                     rng_epoch_ptr());
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

int mtvaf_dropout_res_ln_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* out,
                             float* mean, float* rstd, int M, int H, float eps, float p_drop, uint64_t seed,
                             uint64_t offset, void* out_bf16, hipStream_t st) {
  if (H % 4 || H > MAXC * 256 || M <= 0) return MTVAF_ERR_SHAPE;
  hipLaunchKernelGGL((ln_fwd_kernel<0>), dim3(row_grid(M)), dim3(256), 0, st, x, res, nullptr, nullptr, nullptr,
                     nullptr, nullptr, nullptr, gamma, beta, out, mean, rstd, M, 1, H, eps, p_drop, seed, offset,
                     static_cast<__bf16*>(out_bf16), rng_epoch_ptr());
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

// dgamma/dbeta: overwritten (accumulate = 0) or added to.  dres_accumulate: dres += instead of =.
// dbias_x (nullable): column sums of dx, i.e. the bias gradient of the dense layer whose output is x.
// dx_bf16 (nullable): dx rounded to bf16 (mixed-precision mode: dx is only a GEMM operand downstream); dx may then be NULL.
// The two halves of mtvaf_dropout_res_ln_bwd as separate calls (the executor runs the second on its weight-gradient stream: the
// column sums feed parameter gradients only, nothing on the main chain waits for them): _rows writes dx / dres and the
// per-block partial sums into `part` (mtvaf_ln_bwd_workspace_bytes(M, H) bytes, owned by the caller until _finish has run);
// _finish reduces them into dgamma / dbeta / dbias_x in fixed order.
int mtvaf_dropout_res_ln_bwd_rows(const float* dout, const float* x, const float* res, const float* gamma, const float* mean,
                                  const float* rstd, float* dx, float* dres, int dres_accumulate, int M, int H, float p_drop,
                                  uint64_t seed, uint64_t offset, float* part, void* dx_bf16, hipStream_t st) {
  if (H % 4 || H > MAXC * 256 || M <= 0) return MTVAF_ERR_SHAPE;
  if ((!dx && !dx_bf16) || !part) return MTVAF_ERR_ARG;
  const int g = row_grid_bwd(M);
  if (ln_lean(M, H, 1))
    ln_bwd_lean<false>(g, st, dout, nullptr, 0, x, res, gamma, mean, rstd, dx, dres, dres_accumulate, part, M, H, p_drop, seed, offset,
                       static_cast<__bf16*>(dx_bf16));
  else
    hipLaunchKernelGGL((ln_bwd_kernel<0>), dim3(g), dim3(256), 0, st, dout, x, res, nullptr, nullptr, nullptr, nullptr,
                       nullptr, nullptr, gamma, mean, rstd, dx, dres, dres_accumulate, part, M, 1, H, p_drop, seed,
                       offset, static_cast<__bf16*>(dx_bf16), rng_epoch_ptr());
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}
// _rows with dout = (slab 0 + ... + slab nslab-1) + dout_base: the split-K slabs of the accumulating dX product that feeds this
// LayerNorm (mtvaf_gemm_f32_slabs), added in the order of the reduction launch it replaces.
int mtvaf_dropout_res_ln_bwd_rows_slabs(const float* dout_base, const float* slabs, int nslab, const float* x, const float* res,
                                        const float* gamma, const float* mean, const float* rstd, float* dx, float* dres,
                                        int dres_accumulate, int M, int H, float p_drop, uint64_t seed, uint64_t offset, float* part,
                                        void* dx_bf16, hipStream_t st) {
  if (H % 4 || H > MAXC * 256 || M <= 0 || nslab < 1) return MTVAF_ERR_SHAPE;
  if ((!dx && !dx_bf16) || !part || !slabs || !dout_base) return MTVAF_ERR_ARG;
  const int g = row_grid_bwd(M);
  if (ln_lean(M, H, nslab))
    ln_bwd_lean<false>(g, st, dout_base, slabs, nslab, x, res, gamma, mean, rstd, dx, dres, dres_accumulate, part, M, H, p_drop, seed, offset,
                       static_cast<__bf16*>(dx_bf16));
  else
    hipLaunchKernelGGL((ln_bwd_kernel<0>), dim3(g), dim3(256), 0, st, dout_base, x, res, nullptr, nullptr, nullptr, slabs, nullptr,
                       nullptr, gamma, mean, rstd, dx, dres, dres_accumulate, part, M, nslab, H, p_drop, seed, offset,
                       static_cast<__bf16*>(dx_bf16), rng_epoch_ptr());
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}
// Round 5, pre-split operands (csrc/gemm_f32p.hip): the two LayerNorm kernels that ALSO write the tile-blocked plane image
// [H / 32][3][M][32] of their output -- the operand form of the products that read it next -- instead of a split pass behind them.
// _fwd_planes: nslab == 0 -> x is the dense output (as mtvaf_dropout_res_ln_fwd), nslab >= 1 -> x = the product's unreduced slabs
// (as mtvaf_dropout_res_ln_fwd_slabs: bias added, the sum stored to x_out).  _bwd_rows_planes: nslab == 0 -> dout = dout_base (as
// mtvaf_dropout_res_ln_bwd_rows), nslab >= 1 -> dout = slabs + dout_base; dx may be NULL (a gradient only GEMMs read).
int mtvaf_dropout_res_ln_fwd_planes(const float* x, int nslab, const float* bias, float* x_out, const float* res, const float* gamma,
                                    const float* beta, float* out, float* mean, float* rstd, int M, int H, float eps, float p_drop,
                                    uint64_t seed, uint64_t offset, void* out_planes, hipStream_t st) {
  if (H % 32 || H > MAXC * 256 || M <= 0 || nslab < 0) return MTVAF_ERR_SHAPE;
  if (!x || !out_planes || (nslab >= 1 && !x_out) || (((uintptr_t)out_planes) & 15)) return MTVAF_ERR_ARG;
  if (nslab >= 1)
    hipLaunchKernelGGL((ln_fwd_kernel<2, true>), dim3(row_grid(M)), dim3(256), 0, st, x, res, nullptr, nullptr, nullptr, bias, x_out,
                       nullptr, gamma, beta, out, mean, rstd, M, nslab, H, eps, p_drop, seed, offset, static_cast<__bf16*>(out_planes),
                       rng_epoch_ptr());
  else
    hipLaunchKernelGGL((ln_fwd_kernel<0, true>), dim3(row_grid(M)), dim3(256), 0, st, x, res, nullptr, nullptr, nullptr, nullptr, nullptr,
                       nullptr, gamma, beta, out, mean, rstd, M, 1, H, eps, p_drop, seed, offset, static_cast<__bf16*>(out_planes),
                       rng_epoch_ptr());
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}
int mtvaf_dropout_res_ln_bwd_rows_planes(const float* dout_base, const float* slabs, int nslab, const float* x, const float* res,
                                         const float* gamma, const float* mean, const float* rstd, float* dx, float* dres,
                                         int dres_accumulate, int M, int H, float p_drop, uint64_t seed, uint64_t offset, float* part,
                                         void* dx_planes, hipStream_t st) {
  if (H % 32 || H > MAXC * 256 || M <= 0 || nslab < 0) return MTVAF_ERR_SHAPE;
  if (!dx_planes || !part || !dout_base || (nslab >= 1 && !slabs) || (((uintptr_t)dx_planes) & 15)) return MTVAF_ERR_ARG;
  const int g = row_grid_bwd(M);
  if (ln_lean(M, H, nslab))
    ln_bwd_lean<true>(g, st, dout_base, nslab >= 1 ? slabs : nullptr, nslab, x, res, gamma, mean, rstd, dx, dres, dres_accumulate, part, M,
                      H, p_drop, seed, offset, static_cast<__bf16*>(dx_planes));
  else
    hipLaunchKernelGGL((ln_bwd_kernel<0, true>), dim3(g), dim3(256), 0, st, dout_base, x, res, nullptr, nullptr, nullptr,
                       nslab >= 1 ? slabs : nullptr, nullptr, nullptr, gamma, mean, rstd, dx, dres, dres_accumulate, part, M,
                       nslab >= 1 ? nslab : 1, H, p_drop, seed, offset, static_cast<__bf16*>(dx_planes), rng_epoch_ptr());
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}
int mtvaf_dropout_res_ln_bwd_finish(const float* part, int M, int H, float* dgamma, float* dbeta, float* dbias_x, int accumulate,
                                    hipStream_t st) {
  if (H % 4 || H > MAXC * 256 || M <= 0) return MTVAF_ERR_SHAPE;
  if (!part) return MTVAF_ERR_ARG;
  OutPtrs outs{{dgamma, dbeta, dbias_x, nullptr}};
  static const int rg32 = [] { const char* e = getenv("MTVAF_LN_FINISH_RG32"); return e ? atoi(e) : 1; }();
  if (rg32 && row_grid_bwd(M) >= 128)
    hipLaunchKernelGGL((colsum_final_multi_kernel<32>), dim3((3 * H + 31) / 32), dim3(1024), 0, st, part, row_grid_bwd(M), H, 3, outs,
                       accumulate);
  else
    hipLaunchKernelGGL((colsum_final_multi_kernel<8>), dim3((3 * H + 31) / 32), dim3(256), 0, st, part, row_grid_bwd(M), H, 3, outs,
                       accumulate);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

int mtvaf_dropout_res_ln_bwd(const float* dout, const float* x, const float* res, const float* gamma,
                             const float* mean, const float* rstd, float* dx, float* dres, int dres_accumulate,
                             float* dgamma, float* dbeta, float* dbias_x, int accumulate, int M, int H, float p_drop,
                             uint64_t seed, uint64_t offset, void* workspace, size_t workspace_bytes, void* dx_bf16,
                             hipStream_t st) {
  if (H % 4 || H > MAXC * 256 || M <= 0) return MTVAF_ERR_SHAPE;
  if (!dx && !dx_bf16) return MTVAF_ERR_ARG;
  if (workspace_bytes < (size_t)row_grid_bwd(M) * 3 * H * sizeof(float)) return MTVAF_ERR_WORKSPACE;
  const int rc = mtvaf_dropout_res_ln_bwd_rows(dout, x, res, gamma, mean, rstd, dx, dres, dres_accumulate, M, H, p_drop, seed, offset,
                                               static_cast<float*>(workspace), dx_bf16, st);
  if (rc != MTVAF_OK) return rc;
  return mtvaf_dropout_res_ln_bwd_finish(static_cast<const float*>(workspace), M, H, dgamma, dbeta, dbias_x, accumulate, st);
}

// 0 (default): deterministic scatter-add of the word / position table gradients; 1: float atomics (MTVAF_EMBED_ATOMIC=1)
static int g_embed_atomic = -1;
static bool embed_scatter_atomic() {
  if (g_embed_atomic < 0) {
    const char* e = getenv("MTVAF_EMBED_ATOMIC");
    g_embed_atomic = (e && e[0] == '1') ? 1 : 0;
  }
  return g_embed_atomic == 1;
}
int mtvaf_embed_scatter_mode(int mode) {  // mode 0 / 1 sets, anything else queries; -> the mode in force
  if (mode == 0 || mode == 1) g_embed_atomic = mode;
  return embed_scatter_atomic() ? 1 : 0;
}

// Backward of K1.  dz_ws: [M,H] scratch for the gradient of the summed embeddings.  Word-table rows
// equal to word_pad (and position rows equal to pos_pad when pos_ids != NULL) receive no gradient
// (nn.Embedding padding_idx, models/modeling_bert.py:170, models/modeling_roberta.py:97-100).
// accumulate = 0 zero-fills dword/dpos/dtype first.
int mtvaf_embed_ln_bwd(const float* dout, const int64_t* ids, const int64_t* type_ids, const int32_t* pos_ids,
                       const float* word, const float* pos, const float* type, const float* gamma, const float* mean,
                       const float* rstd, float* dword, float* dpos, float* dtype, float* dgamma, float* dbeta,
                       int accumulate, int B, int S, int H, int vocab, int max_pos, int type_vocab, int word_pad,
                       int pos_pad, float p_drop, uint64_t seed, uint64_t offset, float* dz_ws, void* workspace,
                       size_t workspace_bytes, hipStream_t st) {
  if (H % 4 || H > MAXC * 256 || B <= 0 || S <= 0 || type_vocab > 2) return MTVAF_ERR_SHAPE;
  const int M = B * S;
  const int g = row_grid_bwd(M);
  if (workspace_bytes < mtvaf_embed_ln_bwd_workspace_bytes(M, H, vocab, max_pos)) return MTVAF_ERR_WORKSPACE;
  float* part = (float*)workspace;
  hipLaunchKernelGGL((ln_bwd_kernel<1>), dim3(g), dim3(256), 0, st, dout, nullptr, nullptr, ids, type_ids, pos_ids,
                     word, pos, type, gamma, mean, rstd, dz_ws, nullptr, 0, part, M, S, H, p_drop, seed, offset,
                     (__bf16*)nullptr, rng_epoch_ptr());
  MTVAF_LAUNCH_CHECK();
  OutPtrs outs{{dgamma, dbeta, dtype, type_vocab > 1 ? dtype + H : nullptr}};
  hipLaunchKernelGGL((colsum_final_multi_kernel<8>), dim3((4 * H + 31) / 32), dim3(256), 0, st, part, g, H, 4, outs,
                     accumulate);
  if (!accumulate) {
    zero_f32(dword, (long)vocab * H, st);
    if (pos_ids) zero_f32(dpos, (long)max_pos * H, st);
  }
  // the owner scheme's one-block scan covers tables of up to 65536 rows: larger vocabularies (bert-base-multilingual 119547,
  // xlm-roberta 250002 -- the reference takes any `bert_name`) keep the atomic scatter (decided before anything is launched
  // for the scatter; not bit-reproducible there, as torch's own nn.Embedding backward)
  const int Vmax = (std::max(vocab, max_pos) + 4095) / 4096 * 4096;  // (the scan's int4 shares: 1024 threads x 4 * ceil(V / 4096))
  if (embed_scatter_atomic() || H > DET_NC * 64 || Vmax > 65536) {
    hipLaunchKernelGGL(embed_scatter_kernel, dim3(g), dim3(256), 0, st, dz_ws, ids, pos_ids, dword, dpos, M, H, word_pad,
                       pos_pad);
  } else {  // bit-reproducible: one owner per table row, gradient rows added in token order
    const int gd = (M + 3) / 4, gm = (M + 255) / 256;
    int* flags = reinterpret_cast<int*>(part + (size_t)g * 4 * H);
    int* seg = flags + M;
    int* count = reinterpret_cast<int*>((reinterpret_cast<uintptr_t>(seg + M) + 15) & ~(uintptr_t)15);  // [Vmax] count, [Vmax] cursor, [Vmax] offset (16-byte aligned: int4 scan)
    hipLaunchKernelGGL(row_nonzero_kernel, dim3(std::min(gd, 2048)), dim3(256), 0, st, dz_ws, flags, M, H);
    for (int key = 0; key < (pos_ids ? 2 : 1); ++key) {
      const int V = key == 0 ? vocab : max_pos;
      int* cursor = count + Vmax;
      int* offset = cursor + Vmax;
      zero_f32(reinterpret_cast<float*>(count), (long)2 * Vmax, st);  // count and cursor (a kernel, not a memset node: DESIGN.md section 5, HIP graphs)
      if (key == 0) {
        hipLaunchKernelGGL((det_count_kernel<0>), dim3(gm), dim3(256), 0, st, ids, pos_ids, flags, count, M, (long)word_pad);
        hipLaunchKernelGGL(det_scan_kernel, dim3(1), dim3(1024), 0, st, count, offset, V);
        hipLaunchKernelGGL((det_fill_kernel<0>), dim3(gm), dim3(256), 0, st, ids, pos_ids, flags, offset, cursor, seg, M, (long)word_pad);
        hipLaunchKernelGGL((embed_scatter_det_kernel<0>), dim3(gd), dim3(256), 0, st, dz_ws, ids, pos_ids, flags, count, offset, seg,
                           dword, M, H, (long)word_pad);
      } else {
        hipLaunchKernelGGL((det_count_kernel<1>), dim3(gm), dim3(256), 0, st, ids, pos_ids, flags, count, M, (long)pos_pad);
        hipLaunchKernelGGL(det_scan_kernel, dim3(1), dim3(1024), 0, st, count, offset, V);
        hipLaunchKernelGGL((det_fill_kernel<1>), dim3(gm), dim3(256), 0, st, ids, pos_ids, flags, offset, cursor, seg, M, (long)pos_pad);
        hipLaunchKernelGGL((embed_scatter_det_kernel<1>), dim3(gd), dim3(256), 0, st, dz_ws, ids, pos_ids, flags, count, offset, seg,
                           dpos, M, H, (long)pos_pad);
      }
    }
  }
  if (!pos_ids) {
    if (!accumulate && max_pos > S) zero_f32(dpos + (long)S * H, (long)(max_pos - S) * H, st);
    hipLaunchKernelGGL(pos_reduce_kernel, dim3(S), dim3(256), 0, st, dz_ws, dpos, B, S, H, accumulate);
  }
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

// dst[r][:] = map[r] >= 0 ? src[map[r]][:] : 0   -- packs the unmasked token rows of a padded [B*S, H] tensor (map = the
// kept rows, -1 for the rows that pad the packed image to a whole tile) and, with the inverse map, unpacks them again
// (zeros at the padded positions).  H % 4 == 0.
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ map,
                                                         float* __restrict__ dst, int rows, int h4) {
  const long n = (long)rows * h4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / h4), c = (int)(i % h4);
    const int sr = map[r];
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (sr >= 0) v = reinterpret_cast<const f32x4*>(src)[(long)sr * h4 + c];
    reinterpret_cast<f32x4*>(dst)[i] = v;
  }
}

// Token packing of a batch from its additive mask [B, T] (T = P + S; text keys at columns P..): cu [B+1] row offsets of
// the sentences, inv [B*S] flat token -> packed row (-1: masked), rowmap [B*S] packed row -> flat token (-1 beyond the
// mv_out[0] = cu[B] kept rows).  One block: a thread counts / numbers the tokens of a sentence, thread 0 scans the counts.
// ordered (round 6; cu holds 2 B + 1 ints): cu[B + 1 + z] = the sentence the attention launches run in slot z of their grid's
// slowest dimension -- longest sentence first (ties by index) -- and cu[0] = -1 says that the list is there.  Placement only: the
// blocks of a launch are dealt to the CUs in grid order, three or four to a CU, and a launch lasts as long as its busiest CU; in
// sorted order a CU's blocks come from the long, the middle and the short third of the batch instead of at random
// (tools/attn_balance_probe.py: forward 26.1 -> 22.8 us, backward 59.4 -> 52.9 us per layer at the bench shape).
__global__ __launch_bounds__(1024) void build_packing_kernel(const float* __restrict__ addmask, int B, int T, int P, int S,
                                                            int* __restrict__ cu, int* __restrict__ inv, int* __restrict__ rowmap,
                                                            int* __restrict__ mv_out, int ordered) {
  // one wave per sentence (16 waves take the sentences in turn): 64 mask values per coalesced read, counted / numbered
  // with a ballot; thread 0 scans the B counts in between
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int b = wave; b < B; b += nw) {
    const float* m = addmask + (long)b * T + P;
    int n = 0;
    for (int t0 = 0; t0 < S; t0 += 64) n += __popcll(__ballot(t0 + lane < S && m[t0 + lane] > -5000.f));
    if (lane == 0) cu[b + 1] = n;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int acc = 0;
    cu[0] = 0;
    for (int b = 0; b < B; ++b) {
      acc += cu[b + 1];
      cu[b + 1] = acc;
    }
    mv_out[0] = acc;
  }
  __syncthreads();
  const int mv = cu[B];
  for (int b = wave; b < B; b += nw) {
    const float* m = addmask + (long)b * T + P;
    int base = cu[b];  // (cu[0] is still 0 here)
    for (int t0 = 0; t0 < S; t0 += 64) {
      const bool in = t0 + lane < S;
      const bool keep = in && m[t0 + lane] > -5000.f;
      const unsigned long long bal = __ballot(keep);
      const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
      if (in) inv[b * S + t0 + lane] = keep ? pos : -1;
      if (keep) rowmap[pos] = b * S + t0 + lane;
      base += __popcll(bal);
    }
  }
  for (int r = mv + threadIdx.x; r < B * S; r += blockDim.x) rowmap[r] = -1;
  if (ordered) {  // rank sort of the sentence lengths (B is a batch size: a few hundred at most)
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
      const int nb = cu[b + 1] - cu[b];
      int rank = 0;
      for (int j = 0; j < B; ++j) {
        const int nj = cu[j + 1] - cu[j];
        rank += (nj > nb || (nj == nb && j < b)) ? 1 : 0;
      }
      cu[B + 1 + rank] = b;
    }
    __syncthreads();  // (every thread has read cu[0] = 0 by now)
    if (threadIdx.x == 0) cu[0] = -1;
  }
}

int mtvaf_build_packing(const float* addmask, int B, int T, int P, int S, int* cu, int* inv, int* rowmap, int* mv_out,
                        hipStream_t st) {
  if (B <= 0 || S <= 0 || P < 0 || T != P + S || (long)B * S >= (1L << 31)) return MTVAF_ERR_SHAPE;
  if (!addmask || !cu || !inv || !rowmap || !mv_out) return MTVAF_ERR_ARG;
  hipLaunchKernelGGL(build_packing_kernel, dim3(1), dim3(1024), 0, st, addmask, B, T, P, S, cu, inv, rowmap, mv_out, 0);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

// mtvaf_build_packing whose cu holds 2 B + 1 ints: behind the B + 1 row offsets the sentence order of the attention launches (longest
// first) and cu[0] = -1 as the mark that it is there (the attention kernels take row 0 for sentence 0 either way).
int mtvaf_build_packing_ordered(const float* addmask, int B, int T, int P, int S, int* cu, int* inv, int* rowmap, int* mv_out,
                                hipStream_t st) {
  if (B <= 0 || S <= 0 || P < 0 || T != P + S || (long)B * S >= (1L << 31)) return MTVAF_ERR_SHAPE;
  if (!addmask || !cu || !inv || !rowmap || !mv_out) return MTVAF_ERR_ARG;
  hipLaunchKernelGGL(build_packing_kernel, dim3(1), dim3(1024), 0, st, addmask, B, T, P, S, cu, inv, rowmap, mv_out, 1);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

// k-tile list of the weight-gradient products (mtvaf_gemm_f32_ktiles): the bk-row tiles of the [B*S] token axis that hold at
// least one unmasked token, in order; kcnt[0] = how many.  One block; 64 tiles per wave pass, compacted with ballots.
__global__ __launch_bounds__(1024) void build_ktiles_kernel(const float* __restrict__ addmask, int B, int T, int P, int S, int bk,
                                                           int* __restrict__ klist, int* __restrict__ kcnt) {
  __shared__ int chunk_base[1025];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int ntiles = (B * S) / bk, nchunks = (ntiles + 63) / 64;
  auto live = [&](int t) {
    bool any = false;
    for (int r = t * bk; r < (t + 1) * bk && !any; ++r) any = addmask[(long)(r / S) * T + P + (r % S)] > -5000.f;
    return any;
  };
  for (int c = wave; c < nchunks; c += nw) {
    const int t = c * 64 + lane;
    const unsigned long long bal = __ballot(t < ntiles && live(t));
    if (lane == 0) chunk_base[c + 1] = __popcll(bal);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int acc = 0;
    chunk_base[0] = 0;
    for (int c = 0; c < nchunks; ++c) {
      acc += chunk_base[c + 1];
      chunk_base[c + 1] = acc;
    }
    kcnt[0] = acc;
  }
  __syncthreads();
  for (int c = wave; c < nchunks; c += nw) {
    const int t = c * 64 + lane;
    const bool keep = t < ntiles && live(t);
    const unsigned long long bal = __ballot(keep);
    if (keep) klist[chunk_base[c] + __popcll(bal & ((1ull << lane) - 1ull))] = t;
  }
}

int mtvaf_build_ktiles(const float* addmask, int B, int T, int P, int S, int bk, int* klist, int* kcnt, hipStream_t st) {
  if (B <= 0 || S <= 0 || P < 0 || T != P + S || bk <= 0 || ((long)B * S) % bk || (long)B * S / bk > 64 * 1024) return MTVAF_ERR_SHAPE;
  if (!addmask || !klist || !kcnt) return MTVAF_ERR_ARG;
  hipLaunchKernelGGL(build_ktiles_kernel, dim3(1), dim3(1024), 0, st, addmask, B, T, P, S, bk, klist, kcnt);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

int mtvaf_zero_f32(float* p, long n, hipStream_t st) {
  if (n < 0 || (n && !p)) return MTVAF_ERR_ARG;
  if (n) zero_f32(p, n, st);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

int mtvaf_gather_rows(const float* src, const int* map, float* dst, int rows_dst, int H, hipStream_t st) {
  if (rows_dst <= 0 || H <= 0 || H % 4) return MTVAF_ERR_SHAPE;
  if (!src || !map || !dst) return MTVAF_ERR_ARG;
  const long n4 = (long)rows_dst * (H / 4);
  const int blocks = (int)std::min<long>((n4 + 255) / 256, 4096);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, st, src, map, dst, rows_dst, H / 4);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

size_t mtvaf_colsum_workspace_bytes(int rows, int cols) { return (size_t)64 * cols * sizeof(float); }

// few rows (the bs-4 configuration: M = 256 tokens): one launch -- 32 columns x 32 row groups per block (every thread
// has its <= 16 loads in flight at once: the kernel is one memory round trip, not a chain of them), combined in fixed
// order through LDS.  The two-stage form pays a second launch (~5 us on a stream of ~5-us kernels) for nothing here.
__global__ __launch_bounds__(1024) void colsum_direct_kernel(const float* __restrict__ x, int rows, int cols, long ld,
                                                            float* __restrict__ out, int accumulate) {
  __shared__ float red[32][33];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s = 0.f;
  if (c < cols) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = rg + 32 * i;
      v[i] = r < rows ? x[(long)r * ld + c] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
  }
  red[rg][cl] = s;
  __syncthreads();
  float t4 = 0.f;  // 32 partials per column: four threads add eight each, then one adds the four
  if (rg < 4) {
#pragma unroll
    for (int i = 0; i < 8; ++i) t4 += red[rg * 8 + i][cl];
  }
  __syncthreads();
  if (rg < 4) red[rg][cl] = t4;
  __syncthreads();
  if (rg == 0 && c < cols) {
    const float t = ((red[0][cl] + red[1][cl]) + red[2][cl]) + red[3][cl];
    out[c] = accumulate ? out[c] + t : t;
  }
}

// out[c] (+)= sum_r x[r*ld + c], deterministic two-stage reduction.
int mtvaf_colsum(const float* x, int rows, int cols, int ld, float* out, int accumulate, void* workspace,
                 size_t workspace_bytes, hipStream_t st) {
  if (rows <= 0 || cols <= 0) return MTVAF_ERR_SHAPE;
  if (rows <= 512) {
    hipLaunchKernelGGL(colsum_direct_kernel, dim3((cols + 31) / 32), dim3(1024), 0, st, x, rows, cols, (long)ld, out, accumulate);
    MTVAF_LAUNCH_CHECK();
    return MTVAF_OK;
  }
  int chunks = std::min(64, (rows + 63) / 64);
  if (workspace_bytes < (size_t)chunks * cols * sizeof(float)) return MTVAF_ERR_WORKSPACE;
  const int rpc = (rows + chunks - 1) / chunks;
  chunks = (rows + rpc - 1) / rpc;
  float* part = (float*)workspace;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((cols + 63) / 64, chunks), dim3(256), 0, st, x, rows, cols, (long)ld,
                     part, rpc);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((cols + 31) / 32), dim3(256), 0, st, part, chunks, cols, (long)cols,
                     out, accumulate);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

// y = dropout(x) (n % 4 == 0); the same call with the same (seed, offset) on a gradient is the backward.
int mtvaf_dropout(const float* x, float* y, long n, float p_drop, uint64_t seed, uint64_t offset, hipStream_t st) {
  if (n <= 0 || n % 4) return MTVAF_ERR_SHAPE;
  if (p_drop <= 0.f) {
    if (x != y) {
      hipError_t e = hipMemcpyAsync(y, x, n * sizeof(float), hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) return (int)e;
    }
    return MTVAF_OK;
  }
  const long n4 = n / 4;
  const int blocks = (int)std::min<long>((n4 + 255) / 256, 2048);
  hipLaunchKernelGGL(dropout_kernel, dim3(blocks), dim3(256), 0, st, x, y, n4, p_drop, seed, offset, rng_epoch_ptr());
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

}  // extern "C"
