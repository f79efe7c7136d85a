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
// Tile: BM x BN x 32 per block, WM x WN waves with 64 x 64 (or 32 x 32) wave tiles; TWO LDS buffers of 3 planes x (BM + BN)
// rows x 80 B (conflict-free ds_read_b128 fragments, as gemm_bf16.hip).  While the 48 MFMAs of k-tile t run, the same wave
// fetches tile t+1 into registers, splits it and writes its planes into the other buffer: the staging work is spread
// behind the twelve MFMA groups of the tile (hipcc keeps the interleaving), one barrier per k-tile.
// LDS traffic is the budget of this scheme (three planes per operand): 64 x 64 wave tiles read 24 fragments per 48 MFMAs.
#include "gemm_common.h"

namespace mtvaf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace x3 {

constexpr int BK = 32, LDH = BK + 8;

// x -> three bf16 planes (round to nearest even at every level), two values at a time: v_cvt_pk_bf16_f32, the packed pair
// widened back by a shift / a mask, v_pk_add_f32 for the residual
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk(const f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
__device__ __forceinline__ f32x2 widen(const unsigned pk) {
  return f32x2{__builtin_bit_cast(float, pk << 16), __builtin_bit_cast(float, pk & 0xffff0000u)};
}
__device__ __forceinline__ void split3_pair(const f32x2 x, unsigned& h, unsigned& m, unsigned& l) {
  h = cvt_pk(x);
  const f32x2 r = x - widen(h);
  m = cvt_pk(r);
  l = cvt_pk(r - widen(m));
}
__device__ __forceinline__ void split3(const f32x4 x, bf16x4& h, bf16x4& m, bf16x4& l) {
  unsigned h0, m0, l0, h1, m1, l1;
  split3_pair(f32x2{x.x, x.y}, h0, m0, l0);
  split3_pair(f32x2{x.z, x.w}, h1, m1, l1);
  h = __builtin_bit_cast(bf16x4, uint2{h0, h1});
  m = __builtin_bit_cast(bf16x4, uint2{m0, m1});
  l = __builtin_bit_cast(bf16x4, uint2{l0, l1});
}

template <int ROWS, int NT, bool KM>
struct Stage {
  static constexpr int NKC = (ROWS * 8) / NT;  // float4 per thread, KC
  static constexpr int NU = (ROWS * 4) / NT;   // units per thread, KM (two float4 each)
  static constexpr int NREG = KM ? 2 * NU : NKC;
  static_assert((ROWS * 8) % NT == 0 && (ROWS * 4) % NT == 0, "whole float4s / units per thread");
};

// global -> registers (whole tiles only: the launcher checks alignment).  KC: float4 idx -> (row = idx >> 3, k = (idx & 7) * 4).
// KM: unit u -> (k pair kp = (u >> 3) & 15, 4 rows c4 = ((u & 7) + 8 (u >> 7)) * 4): two float4 (same rows at k and k + 1);
// eight consecutive lanes fetch one whole 128-byte line of a k-row, and the transposed 4-byte LDS writes of a wave fall on
// 32 distinct banks (2-way conflicts; with the rows fastest over all lanes they were 8-way).
template <int ROWS, int NT, bool KM>
__device__ __forceinline__ void g_load(f32x4* reg, const float* base, int ld, int r0, int k0, int tid) {
  if constexpr (!KM) {
#pragma unroll
    for (int i = 0; i < Stage<ROWS, NT, KM>::NKC; ++i) {
      const int idx = tid + i * NT;
      reg[i] = *reinterpret_cast<const f32x4*>(base + (long)(r0 + (idx >> 3)) * ld + k0 + (idx & 7) * 4);
    }
  } else {
#pragma unroll
    for (int i = 0; i < Stage<ROWS, NT, KM>::NU; ++i) {
      const int u = tid + i * NT;
      const int kp = (u >> 3) & 15, c4 = ((u & 7) + (u >> 7) * 8) * 4;
      const float* q = base + (long)(k0 + 2 * kp) * ld + r0 + c4;
      reg[2 * i] = *reinterpret_cast<const f32x4*>(q);
      reg[2 * i + 1] = *reinterpret_cast<const f32x4*>(q + ld);
    }
  }
}

// One unit of staging work: a KC float4 (4 consecutive k of one row), or a KM pair of float4 (4 rows at k and k + 1).
// convert: registers -> six packed dwords (three planes); put: the dwords -> LDS ([plane][row][LDH]; plane stride PS elements).
template <bool KM>
__device__ __forceinline__ void convert_unit(const f32x4* reg, int unit, unsigned (&w)[KM ? 12 : 6]) {
  if constexpr (!KM) {
    bf16x4 h, m, l;
    split3(reg[unit], h, m, l);
    const uint2 hh = __builtin_bit_cast(uint2, h), mm = __builtin_bit_cast(uint2, m), ll = __builtin_bit_cast(uint2, l);
    w[0] = hh.x; w[1] = hh.y; w[2] = mm.x; w[3] = mm.y; w[4] = ll.x; w[5] = ll.y;
  } else {
    bf16x4 h0, m0, l0, h1, m1, l1;
    split3(reg[2 * unit], h0, m0, l0);
    split3(reg[2 * unit + 1], h1, m1, l1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      w[q] = __builtin_bit_cast(unsigned, bf16x2{h0[q], h1[q]});
      w[4 + q] = __builtin_bit_cast(unsigned, bf16x2{m0[q], m1[q]});
      w[8 + q] = __builtin_bit_cast(unsigned, bf16x2{l0[q], l1[q]});
    }
  }
}
template <int NT, bool KM, int PS>
__device__ __forceinline__ void put_unit(const unsigned (&w)[KM ? 12 : 6], int unit, __bf16* s, int tid) {
  if constexpr (!KM) {
    const int idx = tid + unit * NT;
    __bf16* d = s + (idx >> 3) * LDH + (idx & 7) * 4;
    *reinterpret_cast<uint2*>(d) = uint2{w[0], w[1]};
    *reinterpret_cast<uint2*>(d + PS) = uint2{w[2], w[3]};
    *reinterpret_cast<uint2*>(d + 2 * PS) = uint2{w[4], w[5]};
  } else {
    const int u = tid + unit * NT;
    const int kp = (u >> 3) & 15, c4 = ((u & 7) + (u >> 7) * 8) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __bf16* d = s + (c4 + q) * LDH + 2 * kp;
      *reinterpret_cast<unsigned*>(d) = w[q];
      *reinterpret_cast<unsigned*>(d + PS) = w[4 + q];
      *reinterpret_cast<unsigned*>(d + 2 * PS) = w[8 + q];
    }
  }
}

}  // namespace x3

// Requirements (checked by the launcher): M % BM == 0, N % BN == 0, K and every k-chunk multiples of 32, 16-byte aligned
// operands with leading dimensions % 4 == 0.
template <int BM, int BN, int WM, int WN, bool A_KM, bool B_KM, bool KLIST, bool TWO>
__global__ __launch_bounds__(WM* WN * 64, TWO ? 1 : 2) void gemm_f32x3_kernel(GemmArgs p) {
  using namespace x3;
  static_assert(!KLIST || (A_KM && B_KM), "the k-tile list addresses rows of k-major operands");
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int A_SZ = BM * LDH, B_SZ = BN * LDH;  // bf16 elements per plane

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int BUF = 3 * (A_SZ + B_SZ);             // elements per k-tile buffer (two of them)
  __bf16* sA = reinterpret_cast<__bf16*>(smem_raw);  // [2][ [3][A_SZ] | [3][B_SZ] ]
  __bf16* sB = sA + 3 * A_SZ;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, h = lane >> 5;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / p.tiles_n) * BM;
  const int n0 = (bid % p.tiles_n) * BN;
  int kbeg = blockIdx.z * p.k_chunk;
  const int kend = min(p.K, kbeg + p.k_chunk);
  int nk = (kend - kbeg) / BK;
  int lbeg = 0;
  if constexpr (KLIST) {  // this split's share of the listed k-tiles (the count lives on the device)
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

  // two register sets: the tile staged during step t was requested during step t-1 (a whole k-tile of matrix work between a
  // global load and its first use: requested in the step that consumes them, the loads stalled every step for their latency)
  f32x4 ra0[Stage<BM, NT, A_KM>::NREG], rb0[Stage<BN, NT, B_KM>::NREG], ra1[Stage<BM, NT, A_KM>::NREG], rb1[Stage<BN, NT, B_KM>::NREG];
  constexpr int UA = A_KM ? Stage<BM, NT, true>::NU : Stage<BM, NT, false>::NKC;  // staging units per thread
  constexpr int UB = B_KM ? Stage<BN, NT, true>::NU : Stage<BN, NT, false>::NKC;

  auto gload = [&](int kt, f32x4* ra, f32x4* rb) __attribute__((always_inline)) {
    int k0;
    if constexpr (KLIST) k0 = p.klist[lbeg + kt] * BK;
    else k0 = kbeg + kt * BK;
    g_load<BM, NT, A_KM>(ra, p.A, p.lda, m0, k0, tid);
    g_load<BN, NT, B_KM>(rb, p.B, p.ldb, n0, k0, tid);
  };
  // unit u of the UA + UB staging units of a k-tile: registers -> three planes -> LDS buffer `buf`
  auto stage = [&](int u, int buf, const f32x4* ra, const f32x4* rb) __attribute__((always_inline)) {
    if (u < UA) {
      unsigned w[A_KM ? 12 : 6];
      convert_unit<A_KM>(ra, u, w);
      put_unit<NT, A_KM, A_SZ>(w, u, sA + buf * BUF, tid);
    } else if (u < UA + UB) {
      unsigned w[B_KM ? 12 : 6];
      convert_unit<B_KM>(rb, u - UA, w);
      put_unit<NT, B_KM, B_SZ>(w, u - UA, sB + buf * BUF, tid);
    }
  };
  // the 2 x 6 MFMA groups of a k-tile (TM x TN independent accumulators each); `side(g)` is issued behind group g: the
  // split + store of the NEXT k-tile (into the other buffer) runs in the shadow of this tile's matrix work
  auto compute = [&](int buf, auto side) __attribute__((always_inline)) {
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
        side(ks * 6 + t);
      }
    }
  };
  constexpr int NU_ALL = UA + UB, PER = (NU_ALL + 11) / 12;  // units staged behind each of the 12 groups
  // step kt (not the last): compute tile kt from buffer kt & 1, stage tile kt+1 (registers `cur`) into the other buffer,
  // request tile kt+2 into the registers that held tile kt
  auto step = [&](int kt, f32x4* cura, f32x4* curb, f32x4* nxta, f32x4* nxtb) __attribute__((always_inline)) {
    if (kt + 2 < nk) gload(kt + 2, nxta, nxtb);
    compute(kt & 1, [&](int g) __attribute__((always_inline)) {
#pragma unroll
      for (int e = 0; e < PER; ++e) stage(g * PER + e, (kt & 1) ^ 1, cura, curb);
    });
    __syncthreads();  // the other buffer is complete, and every wave is done with this one
  };

  if (TWO) {
    if (nk > 0) {
      gload(0, ra0, rb0);
      if (nk > 1) gload(1, ra1, rb1);
#pragma unroll
      for (int u = 0; u < NU_ALL; ++u) stage(u, 0, ra0, rb0);
      __syncthreads();
      int kt = 0;
      for (; kt + 2 < nk; kt += 2) {  // tile kt+1 sits in set 1, tile kt+2 goes to set 0
        step(kt, ra1, rb1, ra0, rb0);
        step(kt + 1, ra0, rb0, ra1, rb1);
      }
      if (kt + 1 < nk) {  // two tiles left: kt (computed, staging kt+1 from set 1) and kt+1
        step(kt, ra1, rb1, ra0, rb0);
        ++kt;
      }
      compute(kt & 1, [&](int) __attribute__((always_inline)) {});
    }
  } else {
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
      compute(0, [&](int) __attribute__((always_inline)) {});
      __syncthreads();  // every wave has read this k-tile
      if (more) {
#pragma unroll
        for (int u = 0; u < NU_ALL; ++u) stage(u, 0, ra0, rb0);
      }
      __syncthreads();
    }
  }

  if (p.wide) {
    epilogue_wide<BM, BN, WM, WN, TM, TN, NT>(p, acc, reinterpret_cast<float*>(smem_raw), m0, n0, wm, wn, li, h, tid);
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

template <int BM, int BN, int WM, int WN, bool TWO>
static int launch_x3_tile(const GemmArgs& a, int la, int lb, dim3 grid, hipStream_t st) {
  size_t smem = (size_t)(TWO ? 2 : 1) * 3 * (BM + BN) * x3::LDH * sizeof(__bf16);
  const size_t epi = (size_t)BM * (BN + 4) * sizeof(float);
  if (epi > smem) smem = epi;
  dim3 block(WM * WN * 64);
#define MTVAF_X3_LAUNCH(AK, BKM, KL)                                                                                  \
  do {                                                                                                               \
    auto kern = gemm_f32x3_kernel<BM, BN, WM, WN, AK, BKM, KL, TWO>;                                                     \
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

// Called by the common launcher in gemm.hip (whole tiles only).  tile: 0 = 128x128 (2x2 waves), 1 = 64x128 (1x2), 2 = 64x64 (2x2 of 32x32).
int launch_gemm_f32x3(int tile, const GemmArgs& a, int la, int lb, dim3 grid, hipStream_t st) {
  switch (tile) {
    case 0: return launch_x3_tile<128, 128, 2, 2, true>(a, la, lb, grid, st);
    case 1: return launch_x3_tile<128, 128, 2, 2, false>(a, la, lb, grid, st);
    default: return launch_x3_tile<64, 64, 2, 2, false>(a, la, lb, grid, st);
  }
}

}  // namespace mtvaf
