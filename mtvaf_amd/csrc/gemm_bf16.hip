// bf16-compute GEMM for the mixed-precision configurations (BASELINE configs 3-4): operands and results stay
// fp32 in HBM (master weights, activations), tiles are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) while they are
// staged into LDS, and the products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Same operand
// layouts (KC / KM), tile plans, split-K and epilogues as the fp32 kernels (gemm.hip).
//
// LDS image of BOTH operand kinds is [row][k] bf16 with an 80-byte row stride (conflict-free ds_read_b128 of
// the 8-element MFMA fragment: lane (r = l & 31, h = l >> 5) reads k = 16*ks + 8*h .. +7 of row r).  A KM
// operand (k-major in memory) is transposed on the way in: a thread loads the same 4 rows at two consecutive
// k and writes four packed (k, k+1) bf16 pairs.
#include "gemm_common.h"

namespace mtvaf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// global -> registers.  KC: float4 idx -> (row = idx >> 3, k = (idx & 7) * 4).  KM: unit u -> (k pair kp, 4 rows):
// two float4 (same rows at k and k + 1).  Rows beyond R are clamped unless ALIGNED.
template <int ROWS, int NT, bool KM>
struct Stage {
  static constexpr int NKC = (ROWS * 8 + NT - 1) / NT;      // float4 per thread, KC
  static constexpr int NU = (ROWS * 4 + NT - 1) / NT;       // units per thread, KM
  static constexpr int NREG = KM ? 2 * NU : NKC;
};

template <int ROWS, int NT, bool KM, bool ALIGNED>
__device__ __forceinline__ void g_load(f32x4* reg, const float* base, int ld, int r0, int R, int k0, int tid) {
  if constexpr (!KM) {
#pragma unroll
    for (int i = 0; i < Stage<ROWS, NT, KM>::NKC; ++i) {
      const int idx = tid + i * NT;
      if ((ROWS * 8) % NT == 0 || idx < ROWS * 8) {
        const int row = r0 + (idx >> 3), k = k0 + (idx & 7) * 4;
        reg[i] = *reinterpret_cast<const f32x4*>(base + (long)(ALIGNED ? row : min(row, R - 1)) * ld + k);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < Stage<ROWS, NT, KM>::NU; ++i) {
      const int u = tid + i * NT;
      if ((ROWS * 4) % NT == 0 || u < ROWS * 4) {
        const int kp = u / (ROWS / 4), c4 = (u % (ROWS / 4)) * 4;
        const int row = r0 + c4;
        const float* q = base + (long)(k0 + 2 * kp) * ld + (ALIGNED ? row : min(row, R - 4));
        reg[2 * i] = *reinterpret_cast<const f32x4*>(q);
        reg[2 * i + 1] = *reinterpret_cast<const f32x4*>(q + ld);
      }
    }
  }
}

template <int ROWS, int NT, bool KM>
__device__ __forceinline__ void s_store(const f32x4* reg, __bf16* s, int tid) {
  constexpr int LDH = 40;
  if constexpr (!KM) {
#pragma unroll
    for (int i = 0; i < Stage<ROWS, NT, KM>::NKC; ++i) {
      const int idx = tid + i * NT;
      if ((ROWS * 8) % NT == 0 || idx < ROWS * 8) {
        const bf16x4 y = {(__bf16)reg[i].x, (__bf16)reg[i].y, (__bf16)reg[i].z, (__bf16)reg[i].w};
        *reinterpret_cast<bf16x4*>(s + (idx >> 3) * LDH + (idx & 7) * 4) = y;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < Stage<ROWS, NT, KM>::NU; ++i) {
      const int u = tid + i * NT;
      if ((ROWS * 4) % NT == 0 || u < ROWS * 4) {
        const int kp = u / (ROWS / 4), c4 = (u % (ROWS / 4)) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bf16x2 y = {(__bf16)reg[2 * i][q], (__bf16)reg[2 * i + 1][q]};
          *reinterpret_cast<bf16x2*>(s + (c4 + q) * LDH + 2 * kp) = y;
        }
      }
    }
  }
}

// Requirements (checked by the launcher): K and every k-chunk multiples of 32, 16-byte aligned operands,
// M (N) multiples of 4 for KM operands.  Rows beyond M / N are clamped (never stored).
template <int BM, int BN, int WM, int WN, bool A_KM, bool B_KM, bool ALIGNED>
__global__ __launch_bounds__(WM* WN * 64, 2) void gemm_bf16_kernel(GemmArgs p) {
  constexpr int BK = 32, LDH = BK + 8;
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int A_SZ = BM * LDH, B_SZ = BN * LDH;  // bf16 elements per buffer

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* sA = reinterpret_cast<__bf16*>(smem_raw);  // [2][A_SZ]
  __bf16* sB = sA + 2 * A_SZ;                        // [2][B_SZ]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, h = lane >> 5;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / p.tiles_n) * BM;
  const int n0 = (bid % p.tiles_n) * BN;
  const int kbeg = blockIdx.z * p.k_chunk;
  const int kend = min(p.K, kbeg + p.k_chunk);
  const int nk = (kend - kbeg) / BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[Stage<BM, NT, A_KM>::NREG], rb[Stage<BN, NT, B_KM>::NREG];

  auto gload = [&](int kt) {
    const int k0 = kbeg + kt * BK;
    g_load<BM, NT, A_KM, ALIGNED>(ra, p.A, p.lda, m0, p.M, k0, tid);
    g_load<BN, NT, B_KM, ALIGNED>(rb, p.B, p.ldb, n0, p.N, k0, tid);
  };
  auto sstore = [&](int buf) {
    s_store<BM, NT, A_KM>(ra, sA + buf * A_SZ, tid);
    s_store<BN, NT, B_KM>(rb, sB + buf * B_SZ, tid);
  };
  auto compute = [&](int buf) {
    const __bf16* a = sA + buf * A_SZ;
    const __bf16* b = sB + buf * B_SZ;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[i] = *reinterpret_cast<const bf16x8*>(a + ((wm * TM + i) * 32 + li) * LDH + 16 * ks + 8 * h);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb[j] = *reinterpret_cast<const bf16x8*>(b + ((wn * TN + j) * 32 + li) * LDH + 16 * ks + 8 * h);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  };

  if (nk > 0) {
    gload(0);
    sstore(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) gload(kt + 1);
    compute(kt & 1);
    if (more) sstore((kt + 1) & 1);
    __syncthreads();
  }

  if constexpr (ALIGNED) {
    if (p.wide) {
      epilogue_wide<BM, BN, WM, WN, TM, TN, NT>(p, acc, reinterpret_cast<float*>(smem_raw), m0, n0, wm, wn, li, h, tid);
      return;
    }
  }
  float* C = p.C + (long)blockIdx.z * p.slab_stride;
  const bool split = gridDim.z > 1;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + li;
      if (col >= p.N) continue;
      const float bv = (!split && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= p.M) continue;
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

template <int BM, int BN, int WM, int WN>
static int launch_bf16_tile(const GemmArgs& a, int la, int lb, dim3 grid, bool aligned, hipStream_t st) {
  constexpr int LDH = 40;
  size_t smem = (size_t)2 * (BM + BN) * LDH * sizeof(__bf16);
  const size_t epi = (size_t)BM * (BN + 4) * sizeof(float);
  if (aligned && epi > smem) smem = epi;
  dim3 block(WM * WN * 64);
#define MTVAF_BF16_LAUNCH(AK, BKM, AL) \
  hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, AK, BKM, AL>), grid, block, smem, st, a)
  if (la == 0 && lb == 0) { if (aligned) MTVAF_BF16_LAUNCH(false, false, true); else MTVAF_BF16_LAUNCH(false, false, false); }
  else if (la == 0 && lb == 1) { if (aligned) MTVAF_BF16_LAUNCH(false, true, true); else MTVAF_BF16_LAUNCH(false, true, false); }
  else if (la == 1 && lb == 1) { if (aligned) MTVAF_BF16_LAUNCH(true, true, true); else MTVAF_BF16_LAUNCH(true, true, false); }
  else return MTVAF_ERR_ARG;
#undef MTVAF_BF16_LAUNCH
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

// Called by the common launcher in gemm.hip.  tile: 0 = 128x96, 1 = 128x128, 2 = 64x64.
int launch_gemm_bf16(int tile, const GemmArgs& a, int la, int lb, dim3 grid, bool aligned, hipStream_t st) {
  switch (tile) {
    case 0: return launch_bf16_tile<128, 96, 4, 1>(a, la, lb, grid, aligned, st);
    case 1: return launch_bf16_tile<128, 128, 2, 2>(a, la, lb, grid, aligned, st);
    default: return launch_bf16_tile<64, 64, 2, 2>(a, la, lb, grid, aligned, st);
  }
}

}  // namespace mtvaf
