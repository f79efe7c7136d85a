// Pieces shared by the fp32 (gemm.hip) and bf16-compute (gemm_bf16.hip) GEMM kernels: argument block,
// epilogue codes, XCD-aware tile order, checked 4-float load and the wide (LDS-transposed) epilogue.
#pragma once
#include "common.h"

namespace mtvaf {

enum { EPI_NONE = 0, EPI_GELU = 1, EPI_TANH = 2, EPI_DGELU = 3, EPI_DTANH = 4 };

// one product of a grouped launch (the weight gradients of one encoder layer: same layouts, same reduction axis)
struct GemmProb {
  const float* A;
  const float* B;
  float* C;  // result, or this product's split-K slabs
  int lda, ldb, ldc, tiles_n;
  long slab_stride;
  // grouped launch of the wave-specialised split kernel, unsplit: column sums of A over the reduction axis -> colsum[0 .. M)
  // (the bias gradient that belongs to this weight gradient: A is the dY of the dense layer), or NULL
  float* colsum;
};

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  float* aux;
  int M, N, K;
  int lda, ldb, ldc, ldaux;
  int k_chunk;
  long slab_stride;
  int epi, a_vec, b_vec, accumulate;
  int tiles_n;
  int wide;  // wide (LDS-transposed, dwordx4) epilogue allowed: ldc/ldaux % 4 == 0, 16-B aligned C/aux/bias
  // k-tile list (weight-gradient products, k-major operands): the reduction runs over the 32-row k-tiles klist[0 .. *kcnt)
  // only -- the caller vouches that every other k-tile of one operand is exactly zero (token rows whose gradient is zero:
  // padded positions).  NULL: the whole reduction range.  Both live on the device: no host sync.
  const int* klist;
  const int* kcnt;
  // grouped launch (template flag GROUP of gemm_f32_dma_kernel): blockIdx.x walks the tiles of ngrp products back to back;
  // product q owns tiles grp_tile_begin[q] .. grp_tile_begin[q+1]-1.  The table stays in the kernel-argument segment and is
  // indexed at run time (scalar loads on demand).
  int ngrp;
  int grp_tile_begin[4];
  GemmProb grp[4];
  // profiling hook of the wave-specialised split kernel (mtvaf_f32x3_trace): block 0 stores, per wave and k-tile, the shader
  // clock where it arrives at / leaves the tile barrier -- [8 waves][64 k-tiles][4] int64 -- or NULL (no cost but the test)
  long long* trace;
  int tile_walk;  // split kernels: 0 = rows of the tile grid per XCD (xcd_remap), G > 0 = groups of G tile rows column by column
};

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with a private 4 MiB L2), so
// consecutive blockIdx values never share an L2.  This bijective remap gives each XCD a CONTIGUOUS run of
// tiles (whole rows of the tile grid), so an A row-panel is fetched by one XCD only and the B panels it is
// multiplied with stay hot in that XCD's L2.  Placement is a speed hint only -- results never depend on it.
__device__ __forceinline__ int xcd_remap(int bid, int nb) {
  const int xcd = bid & 7, idx = bid >> 3;
  const int q = nb >> 3, r = nb & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// xcd_remap + a walk that keeps a B panel in the XCD's L2: inside an XCD's contiguous run of tiles, groups of G tile rows are
// walked COLUMN by column (the 32 CUs of an XCD hold G x 32/G tiles at a time: 8 B panels and G A panels live, each B panel
// fetched once per group instead of once per tile row).  Only when the XCD's run is whole groups of whole rows; otherwise the
// plain remap.  Placement is a speed hint only.
__device__ __forceinline__ int xcd_remap_cols(int bid, int nb, int tiles_n, int G) {
  const int t = xcd_remap(bid, nb);
  const int per = nb >> 3;
  if ((nb & 7) || per % (G * tiles_n)) return t;
  const int base = (bid & 7) * per, i = t - base;
  const int grp = i / (G * tiles_n), rem = i % (G * tiles_n);
  return base + (grp * G + rem % G) * tiles_n + rem / G;
}

__device__ __forceinline__ f32x4 ld4(const float* __restrict__ p, int nvalid, bool vec) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (nvalid >= 4 && vec) {
    v = *reinterpret_cast<const f32x4*>(p);
  } else {
    if (nvalid > 0) v.x = p[0];
    if (nvalid > 1) v.y = p[1];
    if (nvalid > 2) v.z = p[2];
    if (nvalid > 3) v.w = p[3];
  }
  return v;
}

// Wide epilogue for whole tiles: the accumulator tile is transposed through LDS (free after the main loop)
// so that every lane applies the epilogue to 4 consecutive columns and stores ONE dwordx4 (the MFMA C
// layout would give 16 dword stores per 32x32 block: the store tail is issue-bound, not bandwidth-bound).
// Requires ldc / ldaux / n0 multiples of 4 and 16-byte aligned C / aux / bias (checked by the launcher).
template <int BM, int BN, int WM, int WN, int TM, int TN, int NT>
__device__ __forceinline__ void epilogue_wide(const GemmArgs& p, f32x16 (&acc)[TM][TN], float* smem, int m0, int n0,
                                              int wm, int wn, int li, int h, int tid) {
  constexpr int LDE = BN + 4;
  __syncthreads();  // every wave is done reading the operand tiles
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        smem[row * LDE + (wn * TN + j) * 32 + li] = acc[i][j][r];
      }
  __syncthreads();
  float* C = p.C + (long)blockIdx.z * p.slab_stride;
  const bool split = gridDim.z > 1;
  constexpr int C4 = BN / 4;
#pragma unroll 2
  for (int idx = tid; idx < BM * C4; idx += NT) {
    const int r = idx / C4, c = (idx % C4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(smem + r * LDE + c);
    const long row = m0 + r;
    const int col = n0 + c;
    if (!split) {
      if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + col);
      if (p.epi == EPI_GELU) {
        *reinterpret_cast<f32x4*>(p.aux + row * p.ldaux + col) = v;
        v = f32x4{gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w)};
      } else if (p.epi == EPI_TANH) {
        v = f32x4{tanhf(v.x), tanhf(v.y), tanhf(v.z), tanhf(v.w)};
      } else if (p.epi == EPI_DGELU) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p.aux + row * p.ldaux + col);
        v = f32x4{v.x * gelu_erf_grad(a.x), v.y * gelu_erf_grad(a.y), v.z * gelu_erf_grad(a.z), v.w * gelu_erf_grad(a.w)};
      } else if (p.epi == EPI_DTANH) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p.aux + row * p.ldaux + col);
        v = v * (1.f - t * t);
      }
      if (p.accumulate) v += *reinterpret_cast<const f32x4*>(C + row * p.ldc + col);
    }
    *reinterpret_cast<f32x4*>(C + row * p.ldc + col) = v;
  }
}

}  // namespace mtvaf
