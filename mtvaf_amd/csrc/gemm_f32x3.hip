// fp32 GEMM on the bf16 matrix pipe: every fp32 operand value is split, while its tile is staged into the LDS, into THREE
// bf16 planes  x = x1 + x2 + x3  (x1 = RNE_bf16(x), x2 = RNE_bf16(x - x1), x3 = RNE_bf16(x - x1 - x2): 8 significant bits
// each, |x - x1 - x2 - x3| <= 2^-27 |x|), and a product a.b is the SIX MFMA products
//     a1 b1 + a1 b2 + a2 b1 + a1 b3 + a3 b1 + a2 b2          (v_mfma_f32_32x32x16_bf16, fp32 accumulation)
// -- every partial product of two bf16 values is exact in fp32, and the three dropped terms a2 b3 + a3 b2 + a3 b3 are
// bounded by 2^-26 |a| |b|: BELOW the rounding of a single fp32 product (2^-24).  Operands and results stay fp32 in HBM
// (same entry points, layouts, epilogues, split-K, k-tile lists as gemm.hip), nothing is stored in reduced precision.
// Why: MI355X's dense fp32 MFMA peak is 157 TFLOP/s, its bf16 peak 2.5 PFLOP/s -- six bf16 products per fp32 product
// are worth 417 TFLOP/s of fp32-equivalent work, 2.6x the fp32 pipe, at the accuracy of the fp32 pipe (measured against
// fp64: tests/test_ops_gpu.py::test_gemm_f32_split_accuracy).
//
// Two kernels.  gemm_f32x3_ws_kernel (128 x 128 x 32 or 128 x 96 x 32, the planner's choice): wave-specialised, see its header below.
// gemm_f32x3_kernel (BM x BN x 32, 64 x 64 or 32 x 32 wave tiles): every wave loads, splits, stores and multiplies; ONE LDS
// buffer of 3 planes x (BM + BN) rows x 80 B (conflict-free ds_read_b128 fragments, as gemm_bf16.hip), the next k-tile
// prefetched into registers during the MFMA phase, two blocks per CU -- the first form built (headline 18.55 -> 16.15 ms),
// kept for the 64 x 64 tile and for results without the wide epilogue's alignment.
#include "gemm_common.h"

namespace mtvaf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// MFMA operand fragments travel as four dwords: a loop-carried / conditional value of type <8 x bf16> is legalised piece by
// piece (48 v_perm_b32 + 48 v_lshrrev_b32 per k-tile in the consumer loop, found in round 4), a <4 x i32> is not
typedef unsigned frag_t __attribute__((ext_vector_type(4)));
#define MTVAF_FRAG(x) __builtin_bit_cast(bf16x8, x)
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace x3 {


// x -> three bf16 planes (round to nearest even at every level), two values at a time: v_cvt_pk_bf16_f32, the packed pair
// widened back by a shift / a mask, v_pk_add_f32 for the residual
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk(const f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
__device__ __forceinline__ f32x2 widen(const unsigned pk) {
  return f32x2{__builtin_bit_cast(float, pk << 16), __builtin_bit_cast(float, pk & 0xffff0000u)};
}
// Residual x - bf16 pair by two plain v_sub_f32 (beside MFMAs a v_pk_add_f32 costs several plain instructions' issue time):
// scalar subtractions, and this file is compiled with -fno-slp-vectorize (mtvaf_amd/build.py) so that they stay scalar.  Until
// round 4 they were inline assembly -- which kept them apart too, but made hipcc put an s_nop behind every one that feeds
// the next conversion (31 per staged k-tile: it cannot see inside the asm) and use the 8-byte encoding.
// Tried and dropped in round 4: v_dot2c_f32_bf16 with b = (-1, 0) / (0, -1) folds the widening into the subtraction (7
// instead of 11 vector instructions per pair) but issues slower than the three instructions it replaces -- the producers'
// staging went from 1840 to 2660 cycles per k-tile (tools/x3_trace.py) -- and hipcc 7.2 encodes the (-1, 0) operand as the
// inline constant -1.0, which the instruction reads as (0, -1): wrong planes.
__device__ __forceinline__ f32x2 resid2(const f32x2 x, const unsigned hpk) {
  const f32x2 w = widen(hpk);
  f32x2 r;
  r.x = x.x - w.x;
  r.y = x.y - w.y;
  return r;
}
__device__ __forceinline__ void split3_pair(const f32x2 x, unsigned& h, unsigned& m, unsigned& l) {
  h = cvt_pk(x);
  const f32x2 r = resid2(x, h);
  m = cvt_pk(r);
  l = cvt_pk(resid2(r, m));
}
__device__ __forceinline__ void split3(const f32x4 x, bf16x4& h, bf16x4& m, bf16x4& l) {
  // (the stages of several pairs interleaved by hand -- every operand several instructions old -- measured no better: 1115 vs
  // 1104 us over a layer's products; the vector instructions beside the matrix stream are bound by issue, not by latency)
  unsigned h0, m0, l0, h1, m1, l1;
  split3_pair(f32x2{x.x, x.y}, h0, m0, l0);
  split3_pair(f32x2{x.z, x.w}, h1, m1, l1);
  h = __builtin_bit_cast(bf16x4, uint2{h0, h1});
  m = __builtin_bit_cast(bf16x4, uint2{m0, m1});
  l = __builtin_bit_cast(bf16x4, uint2{l0, l1});
}

template <int ROWS, int NT, bool KM, int BK>
struct Stage {
  static constexpr int NKC = (ROWS * (BK / 4)) / NT;  // float4 per thread, KC
  static constexpr int UNITS = ROWS * (BK / 16);      // KM units of a tile: 4 k x 4 rows (four float4 each)
  static constexpr int NU = (UNITS + NT - 1) / NT;    // units per thread, KM (threads beyond UNITS idle: the 64-row tile)
  static constexpr int NREG = KM ? 4 * NU : NKC;
  static_assert((ROWS * (BK / 4)) % NT == 0, "whole float4s per thread");
};
// KC float4 idx -> row (BK = 32: 8 float4 per row): consecutive 8-lane groups take rows r, r+4, r+1, r+5, ... of each block of
// 8 rows, so that the two rows a 16-lane group of ds_write_b64 stores to are 320 B apart = 16 banks (mod 32): adjacent rows
// (80 B = 20 banks apart) overlap in 4 banks (SQ_LDS_BANK_CONFLICT: a third of the kernel's LDS cycles)
__device__ __forceinline__ int kc_row(int g) { return (g & ~7) | ((g & 1) << 2) | ((g >> 1) & 3); }
// KM unit u -> (k quad kq = u & (BK/4 - 1), first of 4 rows c4 = 4 (u / (BK/4))): a unit is 4 consecutive k of 4 consecutive rows
// (four float4 along the rows).  Lanes: kq fastest over 8 lanes, then the row groups -- a 16-lane group of the plane stores
// (ds_write_b64: 4 k of one row) covers all 8 k quads of two adjacent row groups = 32 distinct banks (conflict-free; the
// first form -- k PAIRS, 4-byte stores -- ran the weight-gradient kernel with 60 % of its LDS cycles in bank conflicts),
// and the 64 lanes of a load instruction cover 8 k-rows x 128 contiguous bytes.
// MTVAF_X3_KM_LANES (compile-time experiment switch): how many consecutive lanes load consecutive 16-byte pieces of one k-row.
// 1 = the mapping above (every lane of a load instruction starts its own 16-byte piece: the texture addresser sees 64 of them
// -- a KM request batch costs its wave 770 ticks where a KC batch costs 370, tools/x3_trace.py); 4 = four lanes cover 64
// contiguous bytes (kq above them): a quarter of the pieces, at the price of 2-way bank conflicts in the plane stores.
#ifndef MTVAF_X3_KM_LANES
#define MTVAF_X3_KM_LANES 4
#endif
template <int BK>
__device__ __forceinline__ void km_unit(int u, int& kq, int& c4) {
  constexpr int CL = MTVAF_X3_KM_LANES, NKQ = BK / 4;
  kq = (u / CL) & (NKQ - 1);
  c4 = ((u / (CL * NKQ)) * CL + (u & (CL - 1))) * 4;
}
// global -> registers (whole tiles only: the launcher checks alignment).  KC: float4 idx -> (row = idx / (BK/4), k = (idx %
// (BK/4)) * 4).  KM: unit u -> (k quad kq, 4 rows c4; km_unit): four float4 (the same 4 rows at k .. k + 3).
// rlim: the last row a load may start at (KC: rows - 1, KM: rows - 4; default: no limit) -- tiles that hang over the operand
// (ragged results of the wave-specialised kernel) re-read its last rows, finite values whose products are never stored; the
// clamp is loop-invariant (only k0 moves), so the k-loop does not see it.
// Addresses are a UNIFORM tile base (moves with k0: scalar arithmetic) + a per-thread 32-bit element offset inside the tile
// that does not depend on k0 (rows of the tile x leading dimension < 2^31: checked by the launcher) -- hipcc then forms
// global_load_dwordx4 v, v_offset, s[base] and the k-loop carries no 64-bit vector address arithmetic (16 v_lshl_add_u64
// per k-tile until round 4).
template <int ROWS, int NT, bool KM, int BK>
__device__ __forceinline__ void g_load(f32x4* reg, const float* base, int ld, int r0, int k0, int tid, int rlim = 0x7fffffff) {
  if constexpr (!KM) {
    const float* tb = base + (long)r0 * ld + k0;  // (uniform)
#pragma unroll
    for (int i = 0; i < Stage<ROWS, NT, KM, BK>::NKC; ++i) {
      const int idx = tid + i * NT;
      const int row = BK == 32 ? kc_row(idx >> 3) : idx / (BK / 4);
      // BYTE offset in 32 bits (so that it can be the load's vector offset beside a scalar base), opaque here: otherwise hipcc
      // hoists base + offset into a vector pair and adds k0 to each of them, one 64-bit vector add per load and k-tile
      unsigned off = (unsigned)((min(r0 + row, rlim) - r0) * ld + (idx % (BK / 4)) * 4) * 4u;
      asm("" : "+v"(off));
      reg[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(tb) + off);
    }
  } else {
    const float* tb = base + (long)k0 * ld + r0;  // (uniform)
#pragma unroll
    for (int i = 0; i < Stage<ROWS, NT, KM, BK>::NU; ++i) {
      const int u = tid + i * NT;
      if (Stage<ROWS, NT, KM, BK>::UNITS % NT != 0 && u >= Stage<ROWS, NT, KM, BK>::UNITS) continue;
      int kq, c4;
      km_unit<BK>(u, kq, c4);
      const unsigned off = (unsigned)(4 * kq * ld + (min(r0 + c4, rlim) - r0));
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        unsigned o = (off + (unsigned)(kk * ld)) * 4u;
        asm("" : "+v"(o));
        reg[4 * i + kk] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(tb) + o);
      }
    }
  }
}

// One unit of staging work: a KC float4 (4 consecutive k of one row), or a KM quad of float4 (4 rows at k .. k + 3).
// convert: registers -> six packed dwords per row (three planes x 4 k); put: the dwords -> LDS ([plane][row][LDH]; plane
// stride PS elements), 8 bytes per store either way.
template <bool KM>
__device__ __forceinline__ void convert_unit(const f32x4* reg, int unit, unsigned (&w)[KM ? 24 : 6]) {
  if constexpr (!KM) {
    bf16x4 h, m, l;
    split3(reg[unit], h, m, l);
    const uint2 hh = __builtin_bit_cast(uint2, h), mm = __builtin_bit_cast(uint2, m), ll = __builtin_bit_cast(uint2, l);
    w[0] = hh.x; w[1] = hh.y; w[2] = mm.x; w[3] = mm.y; w[4] = ll.x; w[5] = ll.y;
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // row c4 + q: its four k are element q of the four float4
      split3_pair(f32x2{reg[4 * unit][q], reg[4 * unit + 1][q]}, w[6 * q], w[6 * q + 2], w[6 * q + 4]);
      split3_pair(f32x2{reg[4 * unit + 2][q], reg[4 * unit + 3][q]}, w[6 * q + 1], w[6 * q + 3], w[6 * q + 5]);
    }
  }
}
template <int NT, bool KM, int PS, int BK, int ROWS>
__device__ __forceinline__ void put_unit(const unsigned (&w)[KM ? 24 : 6], int unit, __bf16* s, int tid) {
  constexpr int LDH = BK + 8;
  if constexpr (!KM) {
    const int idx = tid + unit * NT;
    const int row = BK == 32 ? kc_row(idx >> 3) : idx / (BK / 4);
    __bf16* d = s + row * LDH + (idx % (BK / 4)) * 4;
    *reinterpret_cast<uint2*>(d) = uint2{w[0], w[1]};
    *reinterpret_cast<uint2*>(d + PS) = uint2{w[2], w[3]};
    *reinterpret_cast<uint2*>(d + 2 * PS) = uint2{w[4], w[5]};
  } else {
    const int u = tid + unit * NT;
    if (u >= ROWS * (BK / 16)) return;
    int kq, c4;
    km_unit<BK>(u, kq, c4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __bf16* d = s + (c4 + q) * LDH + 4 * kq;
      *reinterpret_cast<uint2*>(d) = uint2{w[6 * q], w[6 * q + 1]};
      *reinterpret_cast<uint2*>(d + PS) = uint2{w[6 * q + 2], w[6 * q + 3]};
      *reinterpret_cast<uint2*>(d + 2 * PS) = uint2{w[6 * q + 4], w[6 * q + 5]};
    }
  }
}

}  // namespace x3

// Requirements (checked by the launcher): M % BM == 0, N % BN == 0, K and every k-chunk multiples of 32, 16-byte aligned
// operands with leading dimensions % 4 == 0.
template <int BM, int BN, int WM, int WN, bool A_KM, bool B_KM, bool KLIST, int BK>
__global__ __launch_bounds__(WM* WN * 64, 2) void gemm_f32x3_kernel(GemmArgs p) {
  using namespace x3;
  static_assert(!KLIST || (A_KM && B_KM), "the k-tile list addresses rows of k-major operands");
  static_assert(BK == 16 || BK == 32, "one or two MFMA k-slices per k-tile");
  constexpr int LDH = BK + 8;
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int A_SZ = BM * LDH, B_SZ = BN * LDH;  // bf16 elements per plane

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int BUF = 3 * (A_SZ + B_SZ);
  __bf16* sA = reinterpret_cast<__bf16*>(smem_raw);  // [3][A_SZ] | [3][B_SZ]
  __bf16* sB = sA + 3 * A_SZ;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, h = lane >> 5;
  const int bid = p.tile_walk > 0 ? xcd_remap_cols(blockIdx.x, gridDim.x, p.tiles_n, p.tile_walk) : xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / p.tiles_n) * BM;
  const int n0 = (bid % p.tiles_n) * BN;
  int kbeg = blockIdx.z * p.k_chunk;
  const int kend = min(p.K, kbeg + p.k_chunk);
  int nk = (kend - kbeg) / BK;
  int lbeg = 0;
  constexpr int SUB = 32 / BK;  // k-tiles of this kernel per listed 32-row tile
  if constexpr (KLIST) {  // this split's share of the listed k-tiles (the count lives on the device)
    const int cnt = *p.kcnt;
    const int per = (cnt + (int)gridDim.z - 1) / (int)gridDim.z;
    lbeg = blockIdx.z * per;
    nk = max(0, min(cnt - lbeg, per)) * SUB;
    kbeg = 0;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra0[Stage<BM, NT, A_KM, BK>::NREG], rb0[Stage<BN, NT, B_KM, BK>::NREG];
  constexpr int UA = A_KM ? Stage<BM, NT, true, BK>::NU : Stage<BM, NT, false, BK>::NKC;  // staging units per thread
  constexpr int UB = B_KM ? Stage<BN, NT, true, BK>::NU : Stage<BN, NT, false, BK>::NKC;

  auto gload = [&](int kt, f32x4* ra, f32x4* rb) __attribute__((always_inline)) {
    int k0;
    if constexpr (KLIST) k0 = p.klist[lbeg + kt / SUB] * 32 + (kt % SUB) * BK;
    else k0 = kbeg + kt * BK;
    g_load<BM, NT, A_KM, BK>(ra, p.A, p.lda, m0, k0, tid);
    g_load<BN, NT, B_KM, BK>(rb, p.B, p.ldb, n0, k0, tid);
  };
  // unit u of the UA + UB staging units of a k-tile: registers -> three planes -> LDS buffer `buf`
  auto stage = [&](int u, int buf, const f32x4* ra, const f32x4* rb) __attribute__((always_inline)) {
    if (u < UA) {
      unsigned w[A_KM ? 24 : 6];
      convert_unit<A_KM>(ra, u, w);
      put_unit<NT, A_KM, A_SZ, BK, BM>(w, u, sA + buf * BUF, tid);
    } else if (u < UA + UB) {
      unsigned w[B_KM ? 24 : 6];
      convert_unit<B_KM>(rb, u - UA, w);
      put_unit<NT, B_KM, B_SZ, BK, BN>(w, u - UA, sB + buf * BUF, tid);
    }
  };
  // the 2 x 6 MFMA groups of a k-tile (TM x TN independent accumulators each)
  auto compute = [&](int buf) __attribute__((always_inline)) {
    const __bf16* a = sA + buf * BUF;
    const __bf16* b = sB + buf * BUF;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 fa[3][TM], fb[3][TN];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[q][i] = *reinterpret_cast<const bf16x8*>(a + q * A_SZ + ((wm * TM + i) * 32 + li) * LDH + 16 * ks + 8 * h);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          fb[q][j] = *reinterpret_cast<const bf16x8*>(b + q * B_SZ + ((wn * TN + j) * 32 + li) * LDH + 16 * ks + 8 * h);
      }
      // smallest terms first; the TM x TN accumulators of a term are independent (no back-to-back dependent MFMAs)
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[t]][i], fb[PB[t]][j], acc[i][j], 0, 0, 0);
      }
    }
  };
  constexpr int NU_ALL = UA + UB;

  // ONE buffer, two blocks per CU: the next tile is requested before this tile's matrix work and split + stored behind it
  // (between two barriers) -- that phase of one block runs under the matrix phase of the other block of the CU
  if (nk > 0) {
    gload(0, ra0, rb0);
#pragma unroll
    for (int u = 0; u < NU_ALL; ++u) stage(u, 0, ra0, rb0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) gload(kt + 1, ra0, rb0);
    compute(0);
    __syncthreads();  // every wave has read this k-tile
    if (more) {
#pragma unroll
      for (int u = 0; u < NU_ALL; ++u) stage(u, 0, ra0, rb0);
    }
    __syncthreads();
  }

  if (p.wide) {
    // wide epilogue, one ROW OF WAVES per pass: the (BM / WM) x BN accumulator rows of the waves wm == pass are transposed
    // through the LDS so that every lane applies the epilogue to 4 consecutive columns and stores one dwordx4 (as
    // gemm_common.h: epilogue_wide, whose whole-tile image -- 66 KiB at 128 x 128 -- would cost the third block per CU)
    constexpr int LDE = BN + 4, RP = BM / WM, C4 = BN / 4;
    float* smem = reinterpret_cast<float*>(smem_raw);
    float* C = p.C + (long)blockIdx.z * p.slab_stride;
    const bool split = gridDim.z > 1;
#pragma unroll
    for (int pass = 0; pass < WM; ++pass) {
      __syncthreads();  // the operand tiles (pass 0) / the previous pass's image have been read
      if (wm == pass) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              smem[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * LDE + (wn * TN + j) * 32 + li] = acc[i][j][r];
      }
      __syncthreads();
#pragma unroll 2
      for (int idx = tid; idx < RP * C4; idx += NT) {
        const int r = idx / C4, c = (idx % C4) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(smem + r * LDE + c);
        const long row = m0 + pass * RP + r;
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
    return;
  }
  float* C = p.C + (long)blockIdx.z * p.slab_stride;
  const bool split = gridDim.z > 1;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + li;
      const float bv = (!split && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = acc[i][j][r] + bv;
        if (!split) {
          if (p.epi == EPI_GELU) {
            p.aux[(long)row * p.ldaux + col] = v;
            v = gelu_erf(v);
          } else if (p.epi == EPI_TANH) {
            v = tanhf(v);
          } else if (p.epi == EPI_DGELU) {
            v *= gelu_erf_grad(p.aux[(long)row * p.ldaux + col]);
          } else if (p.epi == EPI_DTANH) {
            const float t = p.aux[(long)row * p.ldaux + col];
            v *= (1.f - t * t);
          }
          if (p.accumulate) v += C[(long)row * p.ldc + col];
        }
        C[(long)row * p.ldc + col] = v;
      }
    }
  }
}

// WAVE-SPECIALISED form (128 x 128 x 32, 512 threads, one block per CU, two LDS buffers): waves 0-3 are CONSUMERS (2 x 2
// grid of 64 x 64 wave tiles: fragment reads + the 48 MFMAs of a k-tile, nothing else), waves 4-7 are PRODUCERS (global loads
// two k-tiles ahead into two register sets, split, plane stores into the other buffer).  A SIMD holds one wave of each kind
// (a workgroup's waves go to the SIMDs in cyclic order), so the producers' vector work runs on the SIMD's vector ALU while
// its matrix core is busy with the consumer's MFMAs -- measured on the forms above (rocprofv3 --pmc, FFN-1 forward):
// SQ_VALU_MFMA_COEXEC_CYCLES 3 % of the MFMA cycles, MFMA busy 0.41: with every wave running the same phases, the vector
// work (4.7 VALU instructions per MFMA) and the matrix work took turns instead of overlapping, whatever the occupancy.
// One barrier per k-tile; both kinds execute the same number of barriers.  Where it stands: the bare chain of
// 48 MFMAs per k-tile -- producers, fragment reads and barriers switched off -- takes 88-91 us on the K = 768 / 3072 products
// of 4096 token rows (18.4 ns per MFMA: 32 cycles at the ~1.75-1.9 GHz the chip holds under that stream), the whole kernel
// 108-114: ~80 % of its own matrix stream.
// BN = 128: consumers as 2 x 2 wave tiles of 64 x 64.  BN = 96 (round 4): consumers as 4 x 1 wave tiles of 32 x 96 -- the tile of
// the N = 768 / 2304 results of 4096 token rows (256 / 768 tiles = whole rounds of the 256 CUs, where 128 x 128 tiles are 192 /
// 576: three quarters of the chip idle in the last round).  BN = 64 (round 5): 4 x 1 wave tiles of 32 x 64 -- the N = 768 results
// of a PACKED (padding-free) batch: 2432 token rows are 19 x 12 = 228 tiles, where 128 x 96 gives 152 and 128 x 128 (the only
// layout of the dX products until then) 114 on 256 CUs.  A k-tile of it costs the producers 192 staged rows for 24 MFMAs per
// consumer (128 x 128: 256 rows for 48), so per flop it is the slowest layout: the planner takes it only where it saves a round.
// GROUP (round 4): blockIdx.x walks the 128 x 128 tiles of up to four products that share the reduction axis -- the four weight
// gradients of an encoder layer (36 + 144 + 144 + 108 tiles at BERT-base) -- back to back (GemmArgs::grp, as the fp32 pipe's
// gemm_f32_dma_group_kernel): at 4096 token rows they fill 1.7 rounds of the CUs UNSPLIT, where one launch per product needs
// 2 - 7 splits each to reach the idle CUs, pays the 6 us of prologue + epilogue per block that many times over and a slab
// reduction launch per product on top.
template <bool A_KM, bool B_KM, bool KLIST, int BN, bool GROUP = false>
__global__ __launch_bounds__(512, 1) void gemm_f32x3_ws_kernel(GemmArgs p) {
  using namespace x3;
  static_assert(!KLIST || (A_KM && B_KM), "the k-tile list addresses rows of k-major operands");
  static_assert(BN == 128 || BN == 96 || BN == 64, "consumer layouts exist for 128 x 128, 128 x 96 and 128 x 64");
  static_assert(!GROUP || (A_KM && B_KM && BN == 128), "grouped launches are weight gradients on 128 x 128 tiles");
  constexpr int BM = 128, BK = 32, LDH = BK + 8, NP = 256, NT = 512;
  constexpr int WN = BN == 128 ? 2 : 1, TM = BN == 128 ? 2 : 1, TN = BN == 128 ? 2 : BN / 32;
  constexpr int A_SZ = BM * LDH, B_SZ = BN * LDH, BUF = 3 * (A_SZ + B_SZ);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* sA = reinterpret_cast<__bf16*>(smem_raw);  // [2][ [3][A_SZ] | [3][B_SZ] ]
  __bf16* sB = sA + 3 * A_SZ;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool consumer = wave < 4;
  const int wm = (wave & 3) / WN, wn = (wave & 3) % WN;
  const int li = lane & 31, h = lane >> 5;
  int bid = (!GROUP && p.tile_walk > 0) ? xcd_remap_cols(blockIdx.x, gridDim.x, p.tiles_n, p.tile_walk) : xcd_remap(blockIdx.x, gridDim.x);
  const float* __restrict__ Ap = p.A;
  const float* __restrict__ Bp = p.B;
  float* Cp = p.C;
  int lda = p.lda, ldb = p.ldb, ldc = p.ldc, tiles_n = p.tiles_n;
  long slab_stride = p.slab_stride;
  float* colsum = nullptr;  // GROUP: where this product's column sums of A go (its blocks with n0 == 0 take them), or NULL
  if constexpr (GROUP) {
    // (a run-time index into the kernel-argument segment: scalar loads on demand)
    const int q = (bid >= p.grp_tile_begin[1]) + (bid >= p.grp_tile_begin[2]) + (bid >= p.grp_tile_begin[3]);
    const GemmProb& pb = p.grp[q];
    bid -= p.grp_tile_begin[q];
    Ap = pb.A; Bp = pb.B; Cp = pb.C;
    lda = pb.lda; ldb = pb.ldb; ldc = pb.ldc; tiles_n = pb.tiles_n;
    slab_stride = pb.slab_stride;
    colsum = pb.colsum;
  }
  const int m0 = (bid / tiles_n) * BM;
  const int n0 = (bid % tiles_n) * BN;
  int kbeg = blockIdx.z * p.k_chunk;
  const int kend = min(p.K, kbeg + p.k_chunk);
  int nk = (kend - kbeg) / BK;
  int lbeg = 0;
  if constexpr (KLIST) {
    const int cnt = *p.kcnt;
    const int per = (cnt + (int)gridDim.z - 1) / (int)gridDim.z;
    lbeg = blockIdx.z * per;
    nk = max(0, min(cnt - lbeg, per));
    kbeg = 0;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  long long* const tr = (p.trace && blockIdx.x == 0 && blockIdx.z == 0) ? p.trace : nullptr;  // (block-uniform)
  if (tr && tid == 0) tr[8 * 64 * 4] = __builtin_amdgcn_s_memtime();  // block start

  if (consumer) {
    // Fragment reads run ONE k-slice ahead of the MFMAs that consume them, across the barrier too: an in-order wave that
    // reads the 12 fragments of a slice and then issues its 24 MFMAs pays the LDS latency per slice (measured: reads +
    // barriers alone 48 us, MFMAs + reads 113 us on FFN-1 forward -- the matrix time ADDED to the read time).  Slice 1 of
    // tile t is read before the barrier that ends step t and multiplied after it, while the reads of tile t+1 / slice 0
    // are in flight (the producers refill the buffer of tile t only behind that barrier: its fragments are in registers).
    frag_t fa0[3][TM], fb0[3][TN], fa1[3][TM], fb1[3][TN];
    auto rd = [&](int buf, int ks, frag_t (&fa)[3][TM], frag_t (&fb)[3][TN]) __attribute__((always_inline)) {
      const __bf16* a = sA + buf * BUF;
      const __bf16* b = sB + buf * BUF;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[q][i] = *reinterpret_cast<const frag_t*>(a + q * A_SZ + ((wm * TM + i) * 32 + li) * LDH + 16 * ks + 8 * h);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          fb[q][j] = *reinterpret_cast<const frag_t*>(b + q * B_SZ + ((wn * TN + j) * 32 + li) * LDH + 16 * ks + 8 * h);
      }
    };
    auto mm = [&](const frag_t (&fa)[3][TM], const frag_t (&fb)[3][TN]) __attribute__((always_inline)) {
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // smallest terms first
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MTVAF_FRAG(fa[PA[t]][i]), MTVAF_FRAG(fb[PB[t]][j]), acc[i][j], 0, 0, 0);
    };
    __builtin_amdgcn_s_setprio(2);  // (the matrix stream first: measured -2..5 % against equal priorities, +3 % the other way round)
    __syncthreads();  // k-tile 0 is in buffer 0
    if (nk > 0) {
      rd(0, 0, fa0, fb0);
      for (int kt = 0; kt < nk; ++kt) {
        rd(kt & 1, 1, fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        mm(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        if (tr && kt < 64 && lane == 0) tr[(wave * 64 + kt) * 4 + 2] = __builtin_amdgcn_s_memtime();
        __syncthreads();  // the other buffer is complete, and this one may be refilled (its last fragments are in fa1 / fb1)
        if (tr && kt < 64 && lane == 0) tr[(wave * 64 + kt) * 4 + 3] = __builtin_amdgcn_s_memtime();
        if (kt + 1 < nk) rd((kt + 1) & 1, 0, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        mm(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
    const int ptid = tid - 256;
    f32x4 ra0[Stage<BM, NP, A_KM, BK>::NREG], rb0[Stage<BN, NP, B_KM, BK>::NREG], ra1[Stage<BM, NP, A_KM, BK>::NREG],
        rb1[Stage<BN, NP, B_KM, BK>::NREG];
    constexpr int UA = A_KM ? Stage<BM, NP, true, BK>::NU : Stage<BM, NP, false, BK>::NKC;
    constexpr int UB = B_KM ? Stage<BN, NP, true, BK>::NU : Stage<BN, NP, false, BK>::NKC;
    auto gload = [&](int kt, f32x4* ra, f32x4* rb) __attribute__((always_inline)) {
      int k0;
      if constexpr (KLIST) k0 = p.klist[lbeg + kt] * BK;
      else k0 = kbeg + kt * BK;
      g_load<BM, NP, A_KM, BK>(ra, Ap, lda, m0, k0, ptid, GROUP ? 0x7fffffff : p.M - (A_KM ? 4 : 1));
      g_load<BN, NP, B_KM, BK>(rb, Bp, ldb, n0, k0, ptid, GROUP ? 0x7fffffff : p.N - (B_KM ? 4 : 1));
    };
    auto stage_all = [&](int buf, const f32x4* ra, const f32x4* rb) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < UA; ++u) {
        unsigned w[A_KM ? 24 : 6];
        convert_unit<A_KM>(ra, u, w);
        put_unit<NP, A_KM, A_SZ, BK, BM>(w, u, sA + buf * BUF, ptid);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        unsigned w[B_KM ? 24 : 6];
        convert_unit<B_KM>(rb, u, w);
        put_unit<NP, B_KM, B_SZ, BK, BN>(w, u, sB + buf * BUF, ptid);
      }
    };
    // during step kt (the consumers work on tile kt): stage tile kt+1 from its register set, then request tile kt+3 into it
    // (the requests are UNCONDITIONAL -- past the end they fetch the last tile again: behind a conditional request hipcc
    // cannot count on the newer batch being in flight and waits for vmcnt(0), i.e. for the loads issued one step ago: every
    // step then lasted one global-load latency, ~1.6 us, whatever else overlapped)
    // GROUP: the column sums of A over the reduction axis (the bias gradient that goes with this weight gradient), taken by
    // the product's blocks with n0 == 0 from the A units they stage anyway -- every k-tile exactly once, in k order; a thread
    // owns four consecutive columns at one k quad, the eight k quads of a column group are eight consecutive lanes
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};
    const bool do_sum = GROUP && colsum != nullptr && n0 == 0 && gridDim.z == 1;  // (block-uniform)
    auto sum_a = [&](const f32x4* ra) __attribute__((always_inline)) {
      if constexpr (GROUP) {
        static_assert(!GROUP || UA == 1, "one A unit per producer thread");
        if (do_sum) csum += (ra[0] + ra[1]) + (ra[2] + ra[3]);
      }
    };
    auto pstep = [&](int kt, f32x4* ra, f32x4* rb) __attribute__((always_inline)) {
      if (tr && kt < 64 && lane == 0) tr[(wave * 64 + kt) * 4 + 0] = __builtin_amdgcn_s_memtime();
      if (kt + 1 < nk) sum_a(ra);
      stage_all((kt + 1) & 1, ra, rb);  // (past the end: a harmless copy of the last tile into the idle buffer)
      if (tr && kt < 64 && lane == 0) tr[(wave * 64 + kt) * 4 + 1] = __builtin_amdgcn_s_memtime();
      gload(min(kt + 3, nk - 1), ra, rb);
      if (tr && kt < 64 && lane == 0) tr[(wave * 64 + kt) * 4 + 2] = __builtin_amdgcn_s_memtime();
      __syncthreads();
      if (tr && kt < 64 && lane == 0) tr[(wave * 64 + kt) * 4 + 3] = __builtin_amdgcn_s_memtime();
    };
    if (nk > 0) {
      gload(0, ra0, rb0);
      gload(min(1, nk - 1), ra1, rb1);
      sum_a(ra0);
      stage_all(0, ra0, rb0);
      gload(min(2, nk - 1), ra0, rb0);
    }
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {  // tile kt+1 sits in set 1, tile kt+2 in set 0
      pstep(kt, ra1, rb1);
      pstep(kt + 1, ra0, rb0);
    }
    if (kt < nk) pstep(kt, ra1, rb1);
    if constexpr (GROUP) {
      if (do_sum) {
        int kq_, c4_;
        km_unit<BK>(ptid, kq_, c4_);  // (the eight k quads of a column group: lanes MTVAF_X3_KM_LANES * {0 .. 7} apart)
#pragma unroll
        for (int d = MTVAF_X3_KM_LANES; d < 8 * MTVAF_X3_KM_LANES; d <<= 1) {
          csum.x += __shfl_xor(csum.x, d, 64);
          csum.y += __shfl_xor(csum.y, d, 64);
          csum.z += __shfl_xor(csum.z, d, 64);
          csum.w += __shfl_xor(csum.w, d, 64);
        }
        if (kq_ == 0) *reinterpret_cast<f32x4*>(colsum + m0 + c4_) = csum;
      }
    }
  }

  if (tr && lane == 0) tr[8 * 64 * 4 + 1 + wave] = __builtin_amdgcn_s_memtime();  // this wave's k-loop is over
  // wide epilogue: the four consumer waves write their accumulators through the LDS at once (the whole 128-row image, 66 KB
  // at BN = 128, fits in the two operand buffers, both free behind the k-loop's last barrier), ONE barrier, stores by all 512
  // threads.  (Until round 4 in two passes of 64 rows with three barriers; measured level with this form -- 1186-1188 vs
  // 1188-1192 us over a layer's twelve products: the 4900 ticks of the epilogue are the stores, not the barriers.)
  {
    constexpr int LDE = BN + 4, RP = BM, C4 = BN / 4;
    float* smem = reinterpret_cast<float*>(smem_raw);
    float* C = Cp + (long)blockIdx.z * slab_stride;
    const bool split = gridDim.z > 1;
    const int wrow = wm * TM * 32;  // first result row of this consumer wave inside the tile
    {
      if (consumer) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              smem[(wrow + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * LDE + (wn * TN + j) * 32 + li] = acc[i][j][r];
      }
      __syncthreads();
#pragma unroll 2
      for (int idx = tid; idx < RP * C4; idx += NT) {
        const int r = idx / C4, c = (idx % C4) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(smem + r * LDE + c);
        const long row = m0 + r;
        const int col = n0 + c;
        if (!GROUP && (row >= p.M || col >= p.N)) continue;  // (a tile that hangs over the result: N, ldc multiples of 4)
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
          if (p.accumulate) v += *reinterpret_cast<const f32x4*>(C + row * ldc + col);
        }
        *reinterpret_cast<f32x4*>(C + row * ldc + col) = v;
      }
    }
  }
  if (tr && lane == 0) tr[8 * 64 * 4 + 9 + wave] = __builtin_amdgcn_s_memtime();  // this wave's stores are issued
}

template <int BN>
static int launch_x3_ws(const GemmArgs& a, int la, int lb, dim3 grid, hipStream_t st) {
  if (!a.wide) return MTVAF_ERR_ALIGN;
  const size_t smem = (size_t)2 * 3 * (128 + BN) * 40 * sizeof(__bf16);  // 122880 / 107520 (the epilogue image of 128 x (BN + 4) floats, 67584 / 51200 bytes, fits inside)
#define MTVAF_X3_WS(AK, BKM, KL)                                                                                      \
  do {                                                                                                               \
    auto kern = gemm_f32x3_ws_kernel<AK, BKM, KL, BN>;                                                               \
    static bool attr_set = false;                                                                                    \
    if (!attr_set) {                                                                                                 \
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);  \
      if (e != hipSuccess) return (int)e;                                                                            \
      attr_set = true;                                                                                               \
    }                                                                                                                \
    hipLaunchKernelGGL(kern, grid, dim3(512), smem, st, a);                                                          \
  } while (0)
  if (la == 0 && lb == 0) MTVAF_X3_WS(false, false, false);
  else if (la == 0 && lb == 1) MTVAF_X3_WS(false, true, false);
  else if (la == 1 && lb == 1) { if (a.klist) MTVAF_X3_WS(true, true, true); else MTVAF_X3_WS(true, true, false); }
  else return MTVAF_ERR_ARG;
#undef MTVAF_X3_WS
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}


// the grouped weight-gradient launch (GemmArgs::grp filled by mtvaf_gemm_f32_dw_group; no bias / activation)
int launch_gemm_f32x3_group(const GemmArgs& a, dim3 grid, hipStream_t st) {
  const size_t smem = (size_t)2 * 3 * (128 + 128) * 40 * sizeof(__bf16);
#define MTVAF_X3_GRP(KL)                                                                                             \
  do {                                                                                                               \
    auto kern = gemm_f32x3_ws_kernel<true, true, KL, 128, true>;                                                     \
    static bool attr_set = false;                                                                                    \
    if (!attr_set) {                                                                                                 \
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);  \
      if (e != hipSuccess) return (int)e;                                                                            \
      attr_set = true;                                                                                               \
    }                                                                                                                \
    hipLaunchKernelGGL(kern, grid, dim3(512), smem, st, a);                                                          \
  } while (0)
  if (a.klist) MTVAF_X3_GRP(true);
  else MTVAF_X3_GRP(false);
#undef MTVAF_X3_GRP
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

template <int BM, int BN, int WM, int WN, int BK>
static int launch_x3_tile(const GemmArgs& a, int la, int lb, dim3 grid, hipStream_t st) {
  size_t smem = (size_t)3 * (BM + BN) * (BK + 8) * sizeof(__bf16);
  const size_t epi = (size_t)(BM / WM) * (BN + 4) * sizeof(float);
  if (epi > smem) smem = epi;
  dim3 block(WM * WN * 64);
#define MTVAF_X3_LAUNCH(AK, BKM, KL)                                                                                  \
  do {                                                                                                               \
    auto kern = gemm_f32x3_kernel<BM, BN, WM, WN, AK, BKM, KL, BK>;                                                  \
    static bool attr_set = false;                                                                                    \
    if (smem > 64 * 1024 && !attr_set) {                                                                             \
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);  \
      if (e != hipSuccess) return (int)e;                                                                            \
      attr_set = true;                                                                                               \
    }                                                                                                                \
    hipLaunchKernelGGL(kern, grid, block, smem, st, a);                                                              \
  } while (0)
  if (la == 0 && lb == 0) MTVAF_X3_LAUNCH(false, false, false);
  else if (la == 0 && lb == 1) MTVAF_X3_LAUNCH(false, true, false);
  else if (la == 1 && lb == 1) { if (a.klist) MTVAF_X3_LAUNCH(true, true, true); else MTVAF_X3_LAUNCH(true, true, false); }
  else return MTVAF_ERR_ARG;
#undef MTVAF_X3_LAUNCH
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

// Called by the common launcher in gemm.hip (whole tiles only).  tile: 4 / 5 / 6 = 128x128x32 / 128x96x32 / 128x64x32 wave-specialised (the
// ones the planner uses; they need the wide epilogue's alignment); 1 / 3 = the same tiles, every wave doing everything, one
// buffer, two blocks per CU; 2 = 64x64x32 (2x2 waves of 32x32; small results).
int launch_gemm_f32x3(int tile, const GemmArgs& a, int la, int lb, dim3 grid, hipStream_t st) {
  switch (tile) {
    case 4: return launch_x3_ws<128>(a, la, lb, grid, st);
    case 5: return launch_x3_ws<96>(a, la, lb, grid, st);
    case 6: return launch_x3_ws<64>(a, la, lb, grid, st);
    case 1: return launch_x3_tile<128, 128, 2, 2, 32>(a, la, lb, grid, st);
    case 3: return launch_x3_tile<128, 96, 4, 1, 32>(a, la, lb, grid, st);
    default: return launch_x3_tile<64, 64, 2, 2, 32>(a, la, lb, grid, st);
  }
}

}  // namespace mtvaf
