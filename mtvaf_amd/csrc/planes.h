// Plane images of fp32 tensors (round 5; csrc/gemm_f32p.hip reads them): x = x1 + x2 + x3 with x1 = RNE_bf16(x), x2 = RNE_bf16(x - x1),
// x3 = RNE_bf16(x - x1 - x2) -- the three-way split of csrc/gemm_f32x3.hip, bit for bit -- stored tile-blocked: [cols / 32][3 planes]
// [rows][32] bf16.  The kernels that PRODUCE a GEMM operand (LayerNorm, attention, AdamW, GEMM epilogues) write its image themselves
// with planes_store4 instead of a split pass (mtvaf_f32_split_planes) behind them.
#pragma once
#include "common.h"

namespace mtvaf {

typedef float pl_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pl_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pl_cvt(const pl_f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pl_bf16x2)); }
__device__ __forceinline__ pl_f32x2 pl_widen(const unsigned pk) {
  return pl_f32x2{__builtin_bit_cast(float, pk << 16), __builtin_bit_cast(float, pk & 0xffff0000u)};
}
__device__ __forceinline__ void pl_split(const pl_f32x2 x, unsigned& h, unsigned& m, unsigned& l) {
  h = pl_cvt(x);
  const pl_f32x2 r = x - pl_widen(h);
  m = pl_cvt(r);
  l = pl_cvt(r - pl_widen(m));
}
// 4 consecutive values of row `row` (columns col .. col + 3, col % 4 == 0) as 8 bytes of each plane of an image with M rows
__device__ __forceinline__ void planes_store4(unsigned char* img, long M, long row, int col, f32x4 v) {
  // (the values as they were ROUNDED for the fp32 store: under -ffp-contract=fast the residual x - bf16(x) would otherwise fuse
  // with the multiplication that produced x and split the unrounded product -- planes that differ from a split pass in the last bits)
  asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
  unsigned h0, m0, l0, h1, m1, l1;
  pl_split(pl_f32x2{v.x, v.y}, h0, m0, l0);
  pl_split(pl_f32x2{v.z, v.w}, h1, m1, l1);
  unsigned char* d = img + (long)(col >> 5) * 3 * M * 64 + row * 64 + (col & 31) * 2;
  *reinterpret_cast<uint2*>(d) = uint2{h0, h1};
  *reinterpret_cast<uint2*>(d + M * 64) = uint2{m0, m1};
  *reinterpret_cast<uint2*>(d + 2 * M * 64) = uint2{l0, l1};
}

}  // namespace mtvaf
