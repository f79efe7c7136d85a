// Argument block and split helpers shared by the pre-split fp32 GEMM kernels (gemm_f32p.hip: 128 x 128 tile; gemm_f32pw.hip: the
// 128 x 256 tile of round 6).
#pragma once
#include "gemm_bf16x.h"

namespace mtvaf {

typedef float f32x2p __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4p __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2p __attribute__((ext_vector_type(2)));
typedef unsigned fragp_t __attribute__((ext_vector_type(4)));

struct GemmArgsP {
  const unsigned char* Ap;  // plane 0 of A (bf16), byte addressed
  const unsigned char* Bp;
  long a_plane, a_row, a_kt;  // byte strides: plane -> plane, row -> row, k-tile (32 k) -> k-tile
  long a_col;                 // k-major A only: byte stride from one 128-column block of A to the next (natural image: 256)
  long b_plane, b_row, b_kt;
  long b_col;  // k-major B only: byte stride from one 128-column block of B to the next (natural image: 256)
  float* C;
  const float* bias;
  float* aux;
  int M, N, K;
  int ldc, ldaux;
  int k_chunk;
  long slab_stride;
  int epi, accumulate, tiles_n;
  // the result ALSO (C != NULL) or ONLY (C == NULL) as a tile-blocked plane image [N / 32][3][M][32] -- the operand form of the
  // product that reads it next (FFN-1 forward -> FFN-2 forward, FFN-2 dX -> FFN-1 dX and both weight gradients) -- written by the
  // epilogue of an UNSPLIT launch; colpart [M / 128][N]: per-tile column sums of the result (the bias gradient behind a dX product)
  unsigned char* Cpl;
  float* colpart;
  int ablate;        // research switches: 1 = no MFMAs, 2 = no DMA requests, 4 = no fragment reads
  // tile walk (placement only: results never depend on it): 0 = row-major; G > 0 = bands of G tile rows walked column by column, so
  // that the tiles an XCD runs at one time (its 32 CUs: a contiguous run of the walk) share G A panels and 32 / G B panels instead
  // of one or two A panels and a whole row of B panels
  int walk_g;
  long long* trace;  // [8 waves][64 k-tiles][2] shader-clock stamps of block 0 (arrive at / leave the tile barrier) + 17, or NULL
  // GROUP (weight gradients): blockIdx.x walks the 128 x 128 tiles of up to four products that share the reduction axis back to
  // back; product q owns tiles grp_tile_begin[q] .. grp_tile_begin[q + 1] - 1 (as GemmArgs::grp of the wave-specialised kernel)
  int ngrp;
  int grp_tile_begin[5];
  struct Prob {
    const unsigned char* Ap;
    const unsigned char* Bp;
    long a_plane, a_row, a_kt, a_col, b_plane, b_row, b_kt, b_col;
    float* C;
    int ldc, tiles_n;
  } grp[4];
  // GROUP: blocks behind the last tile are COLUMN-SUM items -- 64 columns of an fp32 matrix [cs_rows][cs_cols] (leading dimension
  // cs_ld) each, summed over all rows into cs_dst (the QKV bias gradient: the column sums of dQ|dK|dV).  They ride in the idle CUs of
  // the launch's last round of tiles (432 tiles on 256 CUs leave 80 of them free) instead of two launches of their own.
  // (up to eight jobs: job j owns blocks cs_blk0[j] .. cs_blk0[j + 1] - 1 behind the tiles -- besides the QKV bias gradient the two
  // LayerNorm-backward finishes of the layer (dgamma, dbeta, dense-bias gradient: three column blocks of 512 partial rows each) and
  // the FFN-1 bias gradient from the GELU' epilogue's per-tile sums)
  struct ColJob { const float* src; float* dst; int rows, cols, ld; } cs[8];
  int cs_n, cs_blk0[9], cs_tile0;
};

namespace f32p {

// the RNE three-way split of gemm_f32x3.hip (same planes bit for bit)
__device__ __forceinline__ unsigned cvt_pk(const f32x2p v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2p)); }
__device__ __forceinline__ f32x2p widen(const unsigned pk) {
  return f32x2p{__builtin_bit_cast(float, pk << 16), __builtin_bit_cast(float, pk & 0xffff0000u)};
}
__device__ __forceinline__ void split3_pair(const f32x2p x, unsigned& h, unsigned& m, unsigned& l) {
  h = cvt_pk(x);
  const f32x2p r = x - widen(h);
  m = cvt_pk(r);
  l = cvt_pk(r - widen(m));
}

__device__ __forceinline__ int swz(int row) {  // G[(row >> 2) & 3], G = {0, 2, 3, 1}
  const int q = (row >> 2) & 3;
  return (((q ^ (q >> 1)) & 1) << 1) | (q >> 1);
}

}  // namespace f32p

int launch_gemm_f32p16w(const GemmArgsP& a, int a_km, int b_km, dim3 grid, hipStream_t st, int bn);  // gemm_f32pw.hip: bn = 256 / 192
int launch_gemm_f32p16w_group(const GemmArgsP& a, dim3 grid, hipStream_t st);

}  // namespace mtvaf
