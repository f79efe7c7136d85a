// fp32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32), the dense-projection engine of the path:
//   K2  QKV projection          models/modeling_bert.py:266, 283-284
//   K5  attention output dense  models/modeling_bert.py:353
//   K6  intermediate dense+GELU models/modeling_bert.py:420-421
//   K7  output dense            models/modeling_bert.py:433
//   K9  encoder_conv            models/bert_model.py:446-454, 541-542
// and their backward products (dX = dY.W, dW = dY^T.X).
//
// One kernel template covers the three operand-layout combinations the path needs:
//   operand "KC": reduction index contiguous in memory   (X[m][k], W[n][k], dY[m][n] as A of dX)
//   operand "KM": reduction index is the row (k-major)    (W[n][k] as B of dX, dY/X as A/B of dW)
// Tiles are staged global -> registers -> LDS (double buffered, one barrier per k-tile); every lane
// feeds the MFMA with a k-permuted fragment (step s, lane half h <-> k = 8*kb + 4*h + s) so that a
// KC operand is fetched with one ds_read_b128 per four MFMAs.  A and B use the same permutation,
// so the sum over k is unchanged.
#include "gemm_common.h"

#include <climits>
#include <cstdlib>
#include <type_traits>

namespace mtvaf {

// FAST: every k-chunk a multiple of BK and 16-byte aligned operands (M, N multiples of 4 for KM operands):
// the main loop has no bounds checks or scalar tails -- rows beyond M / N are CLAMPED to the last valid row
// (their products land in accumulator rows/columns the epilogue never stores).
template <int BM, int BN, int WM, int WN, int BK, bool A_KM, bool B_KM, bool FAST, bool ALIGNED = false>
__global__ __launch_bounds__(WM* WN * 64, 2) void gemm_f32_kernel(GemmArgs p) {
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int KC_LD = BK + 4;
  constexpr int A_SZ = A_KM ? BK * BM : BM * KC_LD;
  constexpr int B_SZ = B_KM ? BK * BN : BN * KC_LD;
  constexpr int KQ = BK / 4;  // float4 per KC row
  constexpr int LA = (BM * KQ + NT - 1) / NT;
  constexpr int LB = (BN * KQ + NT - 1) / NT;
  static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "tile/wave mismatch");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                // [2][A_SZ]
  float* sB = smem + 2 * A_SZ;     // [2][B_SZ]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, h = lane >> 5;

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / p.tiles_n) * BM;
  const int n0 = (bid % p.tiles_n) * BN;
  const int kbeg = blockIdx.z * p.k_chunk;
  const int kend = min(p.K, kbeg + p.k_chunk);
  const int nk = (kend - kbeg + BK - 1) / BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[LA], rb[LB];

  auto gload = [&](int kt) {
    const int k0 = kbeg + kt * BK;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int idx = tid + i * NT;
      if (LA * NT == BM * KQ || idx < BM * KQ) {
        if (!A_KM) {
          const int r = idx / KQ, c = (idx % KQ) * 4;
          const int row = m0 + r, k = k0 + c;
          if (FAST) {
            ra[i] = *reinterpret_cast<const f32x4*>(p.A + (long)(ALIGNED ? row : min(row, p.M - 1)) * p.lda + k);
          } else {
            const int nv = row < p.M ? max(0, min(4, kend - k)) : 0;
            ra[i] = ld4(p.A + (long)row * p.lda + k, nv, p.a_vec);
          }
        } else {
          const int k = idx / (BM / 4), c = (idx % (BM / 4)) * 4;
          const int row = m0 + c;
          if (FAST) {
            ra[i] = *reinterpret_cast<const f32x4*>(p.A + (long)(k0 + k) * p.lda + (ALIGNED ? row : min(row, p.M - 4)));
          } else {
            const int nv = (k0 + k) < kend ? max(0, min(4, p.M - row)) : 0;
            ra[i] = ld4(p.A + (long)(k0 + k) * p.lda + row, nv, p.a_vec);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int idx = tid + i * NT;
      if (LB * NT == BN * KQ || idx < BN * KQ) {
        if (!B_KM) {
          const int r = idx / KQ, c = (idx % KQ) * 4;
          const int col = n0 + r, k = k0 + c;
          if (FAST) {
            rb[i] = *reinterpret_cast<const f32x4*>(p.B + (long)(ALIGNED ? col : min(col, p.N - 1)) * p.ldb + k);
          } else {
            const int nv = col < p.N ? max(0, min(4, kend - k)) : 0;
            rb[i] = ld4(p.B + (long)col * p.ldb + k, nv, p.b_vec);
          }
        } else {
          const int k = idx / (BN / 4), c = (idx % (BN / 4)) * 4;
          const int col = n0 + c;
          if (FAST) {
            rb[i] = *reinterpret_cast<const f32x4*>(p.B + (long)(k0 + k) * p.ldb + (ALIGNED ? col : min(col, p.N - 4)));
          } else {
            const int nv = (k0 + k) < kend ? max(0, min(4, p.N - col)) : 0;
            rb[i] = ld4(p.B + (long)(k0 + k) * p.ldb + col, nv, p.b_vec);
          }
        }
      }
    }
  };
  auto sstore = [&](int buf) {
    float* a = sA + buf * A_SZ;
    float* b = sB + buf * B_SZ;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int idx = tid + i * NT;
      if (LA * NT == BM * KQ || idx < BM * KQ) {
        if (!A_KM) {
          *reinterpret_cast<f32x4*>(a + (idx / KQ) * KC_LD + (idx % KQ) * 4) = ra[i];
        } else {
          *reinterpret_cast<f32x4*>(a + (idx / (BM / 4)) * BM + (idx % (BM / 4)) * 4) = ra[i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int idx = tid + i * NT;
      if (LB * NT == BN * KQ || idx < BN * KQ) {
        if (!B_KM) {
          *reinterpret_cast<f32x4*>(b + (idx / KQ) * KC_LD + (idx % KQ) * 4) = rb[i];
        } else {
          *reinterpret_cast<f32x4*>(b + (idx / (BN / 4)) * BN + (idx % (BN / 4)) * 4) = rb[i];
        }
      }
    }
  };
  auto compute = [&](int buf) {
    const float* a = sA + buf * A_SZ;
    const float* b = sB + buf * B_SZ;
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      f32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + li;
        if (!A_KM) {
          fa[i] = *reinterpret_cast<const f32x4*>(a + row * KC_LD + kb * 8 + 4 * h);
        } else {
          const float* q = a + (kb * 8 + 4 * h) * BM + row;
          fa[i].x = q[0]; fa[i].y = q[BM]; fa[i].z = q[2 * BM]; fa[i].w = q[3 * BM];
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = (wn * TN + j) * 32 + li;
        if (!B_KM) {
          fb[j] = *reinterpret_cast<const f32x4*>(b + col * KC_LD + kb * 8 + 4 * h);
        } else {
          const float* q = b + (kb * 8 + 4 * h) * BN + col;
          fb[j].x = q[0]; fb[j].y = q[BN]; fb[j].z = q[2 * BN]; fb[j].w = q[3 * BN];
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
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

  // ---- epilogue -------------------------------------------------------------------------
  if constexpr (ALIGNED && (2 * (A_SZ + B_SZ) >= BM * (BN + 4))) {
    if (p.wide) {
      epilogue_wide<BM, BN, WM, WN, TM, TN, NT>(p, acc, smem, m0, n0, wm, wn, li, h, tid);
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
      if (!ALIGNED && col >= p.N) continue;
      const float bv = (!split && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (!ALIGNED && row >= p.M) continue;
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

// ---------------------------------------------------------------------------------------------
// LDS-DMA pipelined variant (FAST shapes only): tiles go global -> LDS directly with
// global_load_lds_dwordx4 into a 3-stage ring; ONE raw s_barrier and one counted s_waitcnt vmcnt per
// k-tile, with the loads of tile kt+2 issued right after the barrier of iteration kt, so two tiles
// are always in flight behind the MFMAs (no register staging, no ds_write pass).
// A LDS-DMA write is lane-linear (wave-uniform base + lane*16 B), so KC tiles are stored UNPADDED
// ([rows][BK] floats, 128-B rows) and bank conflicts of the ds_read_b128 fragment reads are removed
// by an XOR swizzle of the 16-B chunk index applied on the per-lane SOURCE address and again on the
// read (chunk' = chunk ^ ((row >> 1) & 7)).  KM tiles ([BK][rows]) are read with conflict-free
// ds_read_b32 and need no swizzle.
// ---------------------------------------------------------------------------------------------
#define MTVAF_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) MTVAF_WAIT_VMCNT(0);
  else if constexpr (N == 4) MTVAF_WAIT_VMCNT(4);
  else if constexpr (N == 5) MTVAF_WAIT_VMCNT(5);
  else if constexpr (N == 6) MTVAF_WAIT_VMCNT(6);
  else if constexpr (N == 7) MTVAF_WAIT_VMCNT(7);
  else if constexpr (N == 8) MTVAF_WAIT_VMCNT(8);
  else if constexpr (N == 10) MTVAF_WAIT_VMCNT(10);
  else static_assert(N == 0, "add the count");
}

__device__ __forceinline__ void glds16(const float* src, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int BM, int BN, int WM, int WN, bool A_KM, bool B_KM, int NSTAGE, bool KLIST, bool GROUP>
__device__ __forceinline__ void gemm_f32_dma_body(GemmArgs p) {
  static_assert(!KLIST || (A_KM && B_KM), "the k-tile list addresses rows of k-major operands");
  constexpr int BK = 32;
  constexpr int NW = WM * WN;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int A_SZ = BM * BK, B_SZ = BN * BK, STAGE = A_SZ + B_SZ;  // floats
  constexpr int IA = A_SZ / 256 / NW, IB = B_SZ / 256 / NW;           // 1-KiB DMA instructions per wave
  static_assert(A_SZ % (256 * NW) == 0 && B_SZ % (256 * NW) == 0, "tile must split into whole DMA pieces per wave");

  extern __shared__ __attribute__((aligned(16))) float smem[];  // [NSTAGE][A_SZ + B_SZ]  (the ONLY LDS object)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, h = lane >> 5;

  int bid = xcd_remap(blockIdx.x, gridDim.x);
  if constexpr (GROUP) {  // which product of the group this tile belongs to (p is this block's private copy of the arguments)
    const int q = (bid >= p.grp_tile_begin[1]) + (bid >= p.grp_tile_begin[2]) + (bid >= p.grp_tile_begin[3]);
    const GemmProb& g = p.grp[q];
    bid -= p.grp_tile_begin[q];
    p.A = g.A; p.B = g.B; p.C = g.C; p.lda = g.lda; p.ldb = g.ldb; p.ldc = g.ldc; p.tiles_n = g.tiles_n;
    p.slab_stride = g.slab_stride;
  }
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

  // per-lane source pointers of this wave's DMA pieces (advance by one k-tile per issue)
  const float* pa[IA];
  const float* pb[IB];
  long stepA, stepB;
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int f = (wave * IA + i) * 64 + lane;  // float4 index inside the tile image
    if (!A_KM) {
      const int r = f >> 3, cp = f & 7;
      pa[i] = p.A + (long)(m0 + r) * p.lda + kbeg + ((cp ^ ((r >> 1) & 7)) << 2);
    } else {
      const int k = f / (BM / 4), c4 = f % (BM / 4);
      pa[i] = p.A + (long)(kbeg + k) * p.lda + m0 + c4 * 4;
    }
  }
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const int f = (wave * IB + i) * 64 + lane;
    if (!B_KM) {
      const int r = f >> 3, cp = f & 7;
      pb[i] = p.B + (long)(n0 + r) * p.ldb + kbeg + ((cp ^ ((r >> 1) & 7)) << 2);
    } else {
      const int k = f / (BN / 4), c4 = f % (BN / 4);
      pb[i] = p.B + (long)(kbeg + k) * p.ldb + n0 + c4 * 4;
    }
  }
  stepA = A_KM ? (long)BK * p.lda : BK;
  stepB = B_KM ? (long)BK * p.ldb : BK;

  int issued = 0;   // (k-tile list) ordinal of the next tile to issue ...
  int kt_next = 0;  // ... and its k-tile index, fetched one issue ahead (a uniform scalar load)
  if constexpr (KLIST) kt_next = nk > 0 ? p.klist[lbeg] : 0;
  auto issue = [&](int stage) {
    float* sa = smem + stage * STAGE + wave * IA * 256;
    float* sb = smem + stage * STAGE + A_SZ + wave * IB * 256;
    if constexpr (KLIST) {
      const long oa = (long)kt_next * stepA, ob = (long)kt_next * stepB;
      ++issued;
      kt_next = p.klist[lbeg + min(issued, nk - 1)];
#pragma unroll
      for (int i = 0; i < IA; ++i) glds16(pa[i] + oa, sa + i * 256);
#pragma unroll
      for (int i = 0; i < IB; ++i) glds16(pb[i] + ob, sb + i * 256);
      return;
    }
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      glds16(pa[i], sa + i * 256);
      pa[i] += stepA;
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      glds16(pb[i], sb + i * 256);
      pb[i] += stepB;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read offsets (floats) that do not depend on the stage
  int offA[TM], offB[TN], swA[TM], swB[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = (wm * TM + i) * 32 + li;
    offA[i] = A_KM ? row : row * BK;
    swA[i] = (row >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = (wn * TN + j) * 32 + li;
    offB[j] = B_KM ? col : col * BK;
    swB[j] = (col >> 1) & 7;
  }

  // Fragments are double-buffered in registers: the ds_reads of k-block kb+1 are issued BEFORE the MFMAs
  // of k-block kb, and the DMA pieces of tile kt+2 are issued behind the first MFMA group, so neither
  // LDS latency nor DMA issue sits between two MFMA bursts.
  auto load_frags = [&](const float* a, const float* b, int kb, f32x4 (&fa)[TM], f32x4 (&fb)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (!A_KM) {
        fa[i] = *reinterpret_cast<const f32x4*>(a + offA[i] + (((2 * kb + h) ^ swA[i]) << 2));
      } else {
        const float* q = a + (kb * 8 + 4 * h) * BM + offA[i];
        fa[i].x = q[0]; fa[i].y = q[BM]; fa[i].z = q[2 * BM]; fa[i].w = q[3 * BM];
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if (!B_KM) {
        fb[j] = *reinterpret_cast<const f32x4*>(b + offB[j] + (((2 * kb + h) ^ swB[j]) << 2));
      } else {
        const float* q = b + (kb * 8 + 4 * h) * BN + offB[j];
        fb[j].x = q[0]; fb[j].y = q[BN]; fb[j].z = q[2 * BN]; fb[j].w = q[3 * BN];
      }
    }
  };
  auto mfma_group = [&](const f32x4 (&fa)[TM], const f32x4 (&fb)[TN]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
  };
  // One straight-line body per k-tile.  An MFMA only occupies the issue port for a few cycles of its 64-cycle
  // pass, but the wave issues in order: a burst of LDS reads or DMA pieces between two MFMA groups leaves the
  // matrix pipe idle.  sched_group_barrier therefore puts ONE non-MFMA instruction group behind each MFMA:
  // the reads of k-blocks 1-2 ride behind the MFMAs of k-block 0, the DMA pieces of tile kt+2 behind k-block
  // 1, the reads of k-block 3 behind k-block 2.  The wait + barrier that publishes tile kt+1 sits BEFORE the
  // last MFMA group, and the first fragments of tile kt+1 are read behind that group -- no LDS latency is
  // exposed after the barrier.  (WAR: the stage of tile kt is next written by the DMA of tile kt+3, issued one
  // iteration later behind k-block 0; every wave has drained its reads of tile kt (lgkmcnt(0)) before the
  // barrier it must pass first.)
  constexpr int NMF = TM * TN * 4;                                    // MFMAs per k-block
  constexpr int NRD = (A_KM ? 4 : 1) * TM + (B_KM ? 4 : 1) * TN;      // ds_read instructions per k-block
  constexpr int R12 = (2 * NRD + NMF - 1) / NMF, R3 = (NRD + NMF - 1) / NMF;
  f32x4 fa0[TM], fb0[TN];  // k-block 0 fragments of the tile about to be computed (loop-carried)

  // PHASE 2: steady (issue tile kt+2, publish kt+1); 1: second-to-last (publish kt+1, all DMA landed); 0: last
  // PHASE 3 (2-stage ring, two blocks per CU): issue tile kt+1 into the stage tile kt-1 left, publish it
  auto tile_body = [&](auto phase_tag, int stage, int next_stage, int issue_stage) {
    constexpr int PHASE = decltype(phase_tag)::value;
    const float* a = smem + stage * STAGE;
    const float* b = a + A_SZ;
    f32x4 fa1[TM], fb1[TN], fa2[TM], fb2[TN], fa3[TM], fb3[TN], fan[TM], fbn[TN];
    load_frags(a, b, 1, fa1, fb1);
    load_frags(a, b, 2, fa2, fb2);
    mfma_group(fa0, fb0);
    if constexpr (PHASE >= 2) issue(issue_stage);  // (issuing the 2-stage ring's DMA one k-block earlier measured 3 % slower)
    load_frags(a, b, 3, fa3, fb3);
    mfma_group(fa1, fb1);
    mfma_group(fa2, fb2);
#pragma unroll
    for (int i = 0; i < NMF; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, R12, 0);
    }
#pragma unroll
    for (int i = 0; i < NMF; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (PHASE >= 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
#pragma unroll
    for (int i = 0; i < NMF; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, R3, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PHASE >= 1) {
      if constexpr (PHASE == 2) wait_vmcnt<IA + IB>(); else wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const float* an = smem + next_stage * STAGE;
      load_frags(an, an + A_SZ, 0, fan, fbn);
    }
    mfma_group(fa3, fb3);
    if constexpr (PHASE >= 1) {
#pragma unroll
      for (int i = 0; i < NMF; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, R3, 1);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) fa0[i] = fan[i];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb0[j] = fbn[j];
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  if (!KLIST || nk > 0) {  // (a split of an almost empty list has nothing to add: its slab is zeros)
  issue(0);
  if (NSTAGE == 3 && nk > 1) {
    issue(1);
    wait_vmcnt<IA + IB>();
  } else {
    wait_vmcnt<0>();
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  load_frags(smem, smem + A_SZ, 0, fa0, fb0);
  int st_c = 0;
  auto nxt = [](int st) { return st == NSTAGE - 1 ? 0 : st + 1; };
  int kt = 0;
  if constexpr (NSTAGE == 3) {
    for (; kt + 2 < nk; ++kt) {
      const int s1 = nxt(st_c);
      tile_body(std::integral_constant<int, 2>{}, st_c, s1, nxt(s1));
      st_c = s1;
    }
    if (kt + 1 < nk) {
      const int s1 = nxt(st_c);
      tile_body(std::integral_constant<int, 1>{}, st_c, s1, 0);
      st_c = s1;
      ++kt;
    }
  } else {
    for (; kt + 1 < nk; ++kt) {
      const int s1 = nxt(st_c);
      tile_body(std::integral_constant<int, 3>{}, st_c, s1, s1);
      st_c = s1;
    }
  }
  if (kt < nk) tile_body(std::integral_constant<int, 0>{}, st_c, 0, 0);
  }

  if (p.wide) {
    epilogue_wide<BM, BN, WM, WN, TM, TN, NW * 64>(p, acc, smem, m0, n0, wm, wn, li, h, tid);
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

// the kernels proper: the body above, instantiated as the ordinary launch (one product) and as the GROUPED launch (the tiles of
// up to four weight-gradient products back to back: mtvaf_gemm_f32_dw_group)
template <int BM, int BN, int WM, int WN, bool A_KM, bool B_KM, int NSTAGE = 3, bool KLIST = false>
__global__ __launch_bounds__(WM* WN * 64, NSTAGE == 3 ? 1 : 2) void gemm_f32_dma_kernel(GemmArgs p) {
  gemm_f32_dma_body<BM, BN, WM, WN, A_KM, B_KM, NSTAGE, KLIST, false>(p);
}
template <int BM, int BN, int WM, int WN, int NSTAGE, bool KLIST>
__global__ __launch_bounds__(WM* WN * 64, NSTAGE == 3 ? 1 : 2) void gemm_f32_dma_group_kernel(GemmArgs p) {
  gemm_f32_dma_body<BM, BN, WM, WN, true, true, NSTAGE, KLIST, true>(p);
}

// dst[r][c] (ld ldd) = (accumulate ? dst : 0) + epi(sum_z slabs[z][r][c] + bias[c]);  every element-wise epilogue of the
// main kernels (tanh, erf-GELU with the pre-activation saved to aux, multiply by GELU'(aux) or by 1 - aux^2): the
// few-token products (bs 4: M = 256) are worth splitting whatever follows them
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, int splits, long slab_stride, float* dst,
                                     int rows, int cols, int ldd, const float* bias, int accumulate, int epi,
                                     float* aux, int ldaux) {
  const long n = (long)rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += slabs[z * slab_stride + i];
    if (bias) s += bias[c];
    if (epi == EPI_TANH) {
      s = tanhf(s);
    } else if (epi == EPI_GELU) {
      aux[(long)r * ldaux + c] = s;
      s = gelu_erf(s);
    } else if (epi == EPI_DGELU) {
      s *= gelu_erf_grad(aux[(long)r * ldaux + c]);
    } else if (epi == EPI_DTANH) {
      const float t = aux[(long)r * ldaux + c];
      s *= (1.f - t * t);
    }
    float* d = dst + (long)r * ldd + c;
    if (accumulate) s += *d;
    *d = s;
  }
}

// the same reduction four columns per thread (cols, ldd, ldaux multiples of 4 and 16-byte aligned pointers: every product
// of the encoder): one dwordx4 per slab, all slab loads of a thread in flight together
__global__ __launch_bounds__(256) void splitk_reduce4_kernel(const float* __restrict__ slabs, int splits, long slab_stride,
                                                             float* dst, int rows, int cols, int ldd, const float* bias,
                                                             int accumulate, int epi, float* aux, int ldaux) {
  const int c4n = cols >> 2;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * c4n) return;
  const int r = (int)(i / c4n), c = (int)(i % c4n) << 2;
  const float* src = slabs + (long)r * cols + c;
  f32x4 s = *reinterpret_cast<const f32x4*>(src);
#pragma unroll 4
  for (int z = 1; z < splits; ++z) s += *reinterpret_cast<const f32x4*>(src + z * slab_stride);
  if (bias) s += *reinterpret_cast<const f32x4*>(bias + c);
  if (epi == EPI_TANH) {
    s = f32x4{tanhf(s.x), tanhf(s.y), tanhf(s.z), tanhf(s.w)};
  } else if (epi == EPI_GELU) {
    *reinterpret_cast<f32x4*>(aux + (long)r * ldaux + c) = s;
    s = f32x4{gelu_erf(s.x), gelu_erf(s.y), gelu_erf(s.z), gelu_erf(s.w)};
  } else if (epi == EPI_DGELU) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(aux + (long)r * ldaux + c);
    s = f32x4{s.x * gelu_erf_grad(a.x), s.y * gelu_erf_grad(a.y), s.z * gelu_erf_grad(a.z), s.w * gelu_erf_grad(a.w)};
  } else if (epi == EPI_DTANH) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(aux + (long)r * ldaux + c);
    s = s * (1.f - t * t);
  }
  float* d = dst + (long)r * ldd + c;
  if (accumulate) s += *reinterpret_cast<const f32x4*>(d);
  *reinterpret_cast<f32x4*>(d) = s;
}

// epilogues that commute with the ordered slab reduction (applied by splitk_reduce_kernel)
static inline bool splittable(int epi) { return epi >= EPI_NONE && epi <= EPI_DTANH; }

struct TileCfg { int bm, bn, bk; };
static const TileCfg kCfgs[] = {{128, 128, 16}, {128, 96, 16}, {128, 288, 16}, {64, 64, 16}, {128, 64, 16},
                                {128, 128, 32}, {128, 96, 32}, {128, 192, 16}, {128, 192, 32},
                                {128, 96, 32}, {128, 128, 32}, {128, 192, 32},   // 9..11: LDS-DMA pipeline (FAST only)
                                {128, 96, 32}, {128, 128, 32},   // 12..13: 2-stage LDS-DMA ring, two blocks per CU
                                {128, 64, 32}, {128, 64, 32},    // 14..15: 128x64 LDS-DMA tile (3-stage / 2-stage): row counts
                                                                 // that leave 128x96 tiles on half the CUs (padding-free runs)
                                {64, 64, 32}, {64, 64, 32}};     // 16..17: 64x64 LDS-DMA tile (3-stage / 2-stage): few-token
                                                                 // products (bs 4: M = 256), where the wave count is the limit
constexpr int kNumCfgs = 18;
constexpr int kFirstDma = 9;

template <int BM, int BN, int WM, int WN, int NSTAGE = 3>
static int launch_dma(const GemmArgs& a, int la, int lb, dim3 grid, hipStream_t st) {
  size_t smem = (size_t)NSTAGE * (BM + BN) * 32 * sizeof(float);
  if (smem < (size_t)BM * (BN + 4) * sizeof(float)) smem = (size_t)BM * (BN + 4) * sizeof(float);  // wide epilogue image
  dim3 block(WM * WN * 64);
#define MTVAF_DMA_LAUNCH(AK, BKM)                                                                                   \
  do {                                                                                                               \
    auto kern = gemm_f32_dma_kernel<BM, BN, WM, WN, AK, BKM, NSTAGE>;                                                      \
    static bool attr_set = false;                                                                                    \
    if (smem > 64 * 1024 && !attr_set) {                                                                             \
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);  \
      if (e != hipSuccess) return (int)e;                                                                            \
      attr_set = true;                                                                                               \
    }                                                                                                                \
    hipLaunchKernelGGL(kern, grid, block, smem, st, a);                                                              \
  } while (0)
  if (la == 0 && lb == 0) MTVAF_DMA_LAUNCH(false, false);
  else if (la == 0 && lb == 1) MTVAF_DMA_LAUNCH(false, true);
  else if (la == 1 && lb == 1) {
    if (a.klist) {
      auto kern = gemm_f32_dma_kernel<BM, BN, WM, WN, true, true, NSTAGE, true>;
      static bool attr_set_l = false;
      if (smem > 64 * 1024 && !attr_set_l) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_set_l = true;
      }
      hipLaunchKernelGGL(kern, grid, block, smem, st, a);
    } else {
      MTVAF_DMA_LAUNCH(true, true);
    }
  } else return MTVAF_ERR_ARG;
#undef MTVAF_DMA_LAUNCH
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

template <int BM, int BN, int WM, int WN, int BK, bool FAST, bool ALIGNED = false>
static int launch_l(const GemmArgs& a, int la, int lb, dim3 grid, hipStream_t st) {
  constexpr int KC_LD = BK + 4;
  const int asz = la ? BK * BM : BM * KC_LD, bsz = lb ? BK * BN : BN * KC_LD;
  const size_t smem = (size_t)2 * (asz + bsz) * sizeof(float);
  dim3 block(WM * WN * 64);
  if (la == 0 && lb == 0)
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, BK, false, false, FAST, ALIGNED>), grid, block, smem, st, a);
  else if (la == 0 && lb == 1)
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, BK, false, true, FAST, ALIGNED>), grid, block, smem, st, a);
  else if (la == 1 && lb == 1)
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, BK, true, true, FAST, ALIGNED>), grid, block, smem, st, a);
  else if (!FAST)  // KM x KC is not produced by the path; only the checked kernel carries it
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, BK, true, false, false>), grid, block, smem, st, a);
  else
    return MTVAF_ERR_ARG;
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

template <int BM, int BN, int WM, int WN, int BK>
static int launch_cfg(const GemmArgs& a, int la, int lb, dim3 grid, int mode, hipStream_t st) {
  // mode 2: k-aligned + whole tiles, 1: k-aligned with clamped ragged rows, 0: fully checked
  if (mode == 2 && !(la == 1 && lb == 0)) return launch_l<BM, BN, WM, WN, BK, true, true>(a, la, lb, grid, st);
  if (mode >= 1 && !(la == 1 && lb == 0)) return launch_l<BM, BN, WM, WN, BK, true, false>(a, la, lb, grid, st);
  return launch_l<BM, BN, WM, WN, BK, false, false>(a, la, lb, grid, st);
}

int launch_gemm_bf16(int tile, const GemmArgs& a, int la, int lb, dim3 grid, bool aligned, hipStream_t st);  // gemm_bf16.hip
int launch_gemm_f32x3(int tile, const GemmArgs& a, int la, int lb, dim3 grid, hipStream_t st);                // gemm_f32x3.hip
int launch_gemm_f32x3_group(const GemmArgs& a, dim3 grid, hipStream_t st);                                      // gemm_f32x3.hip

// ordered split-K slab reduction (+ bias / tanh / dtanh epilogue), shared with gemm_bf16kc.hip
int launch_splitk_reduce(const float* slabs, int splits, float* C, int M, int N, int ldc, const float* bias, int accumulate,
                         int epi, float* aux, int ldaux, hipStream_t stream) {
  const long n = (long)M * N;
  const bool vec = (N % 4 == 0) && (ldc % 4 == 0) && (((uintptr_t)slabs & 15) == 0) && (((uintptr_t)C & 15) == 0) &&
                   (!bias || (((uintptr_t)bias & 15) == 0)) && (!aux || ((ldaux % 4 == 0) && (((uintptr_t)aux & 15) == 0)));
  if (vec) {
    hipLaunchKernelGGL(splitk_reduce4_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, slabs, splits, n, C,
                       M, N, ldc, bias, accumulate, epi, aux, ldaux);
  } else {
    const int blocks = (int)std::min<long>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, slabs, splits, n, C, M, N, ldc, bias,
                       accumulate, epi, aux, ldaux);
  }
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

static inline long cdiv(long a, long b) { return (a + b - 1) / b; }

// ---- optional launch profiler (bench.py roofline): HIP events recorded on the launch stream directly
// around the main GEMM kernel (the split-K reduction is outside the bracket), so that the averages are
// comparable with rocprofv3's per-kernel durations.
struct ProfRec { hipEvent_t e0, e1; int key[8]; };
static ProfRec* g_prof = nullptr;
static int g_prof_cap = 0, g_prof_n = 0;

// begin / end of a profiled launch for the other GEMM translation units (gemm_bf16kc.hip): returns a record index
// or -1 when the profiler is off
int prof_begin(const int key[8], hipStream_t stream) {
  if (!g_prof || g_prof_n >= g_prof_cap) return -1;
  const int i = g_prof_n++;
  for (int j = 0; j < 8; ++j) g_prof[i].key[j] = key[j];
  (void)hipEventRecord(g_prof[i].e0, stream);
  return i;
}
void prof_end(int i, hipStream_t stream) {
  if (i >= 0 && g_prof) (void)hipEventRecord(g_prof[i].e1, stream);
}

// Cost model (units: fp32 MFMA cycles of one CU).  The MFMA pipe of a CU is shared by its resident
// blocks, so time ~ rounds over the 256 CUs x work per tile / efficiency of that tile shape (calibrated
// with tools/gemm_sweep.py on MI355X at M = 4096), plus, for split-K, the slab write + ordered reduce.
static bool x3_tile96_ok(int M, int N, int la, int lb) {
  static const int tile96 = [] { const char* e = getenv("MTVAF_X3_TILE96"); return e ? atoi(e) : 1; }();
  return M % 128 == 0 && N % 96 == 0 && (tile96 == 2 || (tile96 == 1 && la == 0 && lb == 0));
}

// 128x128 tiles of the wave-specialised split kernel may hang over the result (clamped loads, guarded stores): any M, N that
// are multiples of 4 and at least one tile -- the prompt generator's N = 800 (MTVAF_X3_RAGGED=0: whole tiles only)
static bool x3_tile64_ok(int M, int N) {
  static const int on = [] { const char* e = getenv("MTVAF_X3_TILE64"); return e ? atoi(e) : 1; }();
  return on && M % 128 == 0 && N % 64 == 0;
}

static bool x3_ragged_ok(int M, int N) {
  static const int on = [] { const char* e = getenv("MTVAF_X3_RAGGED"); return e ? atoi(e) : 1; }();
  return on && M % 4 == 0 && N % 4 == 0 && M >= 128 && N >= 128;
}

static double x3_e64() {
  static const double e = [] { const char* v = getenv("MTVAF_X3_E64"); return v ? atof(v) : 0.47; }();
  return e;
}

static void choose(int M, int N, int K, int allow_split, int la, int lb, int epi, int* cfg_out, int* splits_out,
                   int compute = 0, int ragged5 = 0) {
  //                                  128x128 128x96 128x288 64x64 128x64 128x128x32 128x96x32 128x192 128x192x32
  static const double eff_base[kNumCfgs] = {0.80, 0.86, 0.72, 0.45, 0.80, 0.70, 1.00, 0.92, 0.78,
                                            1.18, 1.00, 0.85,   // 9..11: LDS-DMA pipeline
                                            1.18, 1.00,         // 12..13: same tiles, 2-stage ring, two blocks per CU
                                            1.08, 1.08,         // 14..15: 128x64 (measured with tools/gemm_sweep.py --rows 2432)
                                            0.60, 0.60};        // 16..17: 64x64
  double eff[kNumCfgs];
  for (int c = 0; c < kNumCfgs; ++c) {
    eff[c] = eff_base[c];
    if (c >= kFirstDma) {
      const bool aligned = (M % kCfgs[c].bm == 0) && (N % kCfgs[c].bn == 0) && (K % 32 == 0);
      if (!aligned || (la == 1 && lb == 0)) eff[c] = 0.0;  // KM x KC never occurs on the path
      // one 84-KB block per CU cannot overlap a GELU-class epilogue with the next tile's main loop (measured:
      // tools/gemm_sweep.py): the 128x128 DMA tile amortises it best forward, the staged kernel backward
      if (epi == EPI_GELU) eff[c] *= ((c == 10 || c == 13) ? 1.10 : 0.90);
      if (epi == EPI_DGELU) eff[c] *= ((c == 10 || c == 13) ? 1.02 : 0.85);
    }
  }
  double best = 1e300;
  int bc = 0, bs = 1;
  for (int c = 0; c < kNumCfgs; ++c) {
    if (eff[c] <= 0.0) continue;
    if (compute == 1 && !(c == 6 || c == 5 || c == 3)) continue;  // bf16 kernels exist for 128x96, 128x128, 64x64
    // (the 128x96 split tile: forward products only by default -- with a k-major B operand or as a weight gradient it
    // measured level or behind 128x128 + split-K in the bench step; MTVAF_X3_TILE96 = 0 never, 2 every layout)
    const bool c6_ok = x3_tile96_ok(M, N, la, lb);
    // (128x64, round 5: the wave-specialised kernel's third layout, for row counts that leave the wider tiles on half the CUs --
    // packed batches; every layout; MTVAF_X3_TILE64=0 never)
    const bool c4_ok = x3_tile64_ok(M, N);
    if (compute == 2 && !((c == 5 && ((M % 128 == 0 && N % 128 == 0) || ragged5)) || (c == 6 && c6_ok) || (c == 4 && c4_ok) || (c == 3 && M % 64 == 0 && N % 64 == 0))) continue;
    const int bm = kCfgs[c].bm, bn = kCfgs[c].bn, bk = compute != 0 ? 32 : kCfgs[c].bk;
    const long tiles = cdiv(M, bm) * cdiv(N, bn);
    // split kernel, dX products (KC x KM) of >= 128 tiles: at most 2 splits -- in effect none: the [4096 x 768] products over
    // K = 2304 / 3072 now run as 192 unsplit tiles.  They sit on the main chain of a backward pass while the grouped weight
    // gradients share the chip on the second stream, so what a split costs is its CU time (6 us of prologue + epilogue per
    // block, a slab pass and a reduction launch) rather than what it saves in latency alone: 4 splits measured 15.20-15.26 ms
    // per step, this cap 14.85-14.97 (MTVAF_X3_DX_MAXSPLIT overrides)
    static const int dx_max_s = [] { const char* e = getenv("MTVAF_X3_DX_MAXSPLIT"); return e ? atoi(e) : 2; }();
    const int max_s = allow_split ? ((compute == 2 && la == 0 && lb == 1 && tiles >= 128) ? dx_max_s : 16) : 1;
    for (int s = 1; s <= max_s; ++s) {
      if (s > 1 && K / s < 256) break;
      const long rounds = cdiv(tiles * s, 256);
      const double kc = (double)cdiv(cdiv(K, s), bk) * bk;
      // per tile: (bm*bn*kc*2 flop) / (256 flop/clk/CU) cycles at 100 %
      // two co-resident blocks hide each other's prologue / epilogue / barrier stalls once every CU holds two
      // (measured +8..9 % at >= 2 tiles per CU, -3..5 % with a single tile per CU: the 2-stage ring is shallower;
      // preferring it there anyway for its smaller footprint measured 0.5 % slower on the whole step)
      const bool two_blocks = c == 12 || c == 13 || c == 15 || c == 17;
      const double occ2 = two_blocks ? (tiles * s >= 384 ? 1.08 : 0.95) : 1.0;  // (>= 1.5 blocks per CU: M = 2432 rows measured)
      // the 64x64 tile re-reads operands twice as often as 128x64 (its 0.60), which only costs once the CUs are full:
      // with at most one block per CU -- few-token products -- a wave's own MFMA chain is the limit and the small tile wins
      // split kernels (one block per CU, whole rounds decide): 128x128 a little ahead of 128x96 at equal rounds x area
      // (fewer operand bytes per flop), the 64x64 every-wave kernel well behind both
      // (128x64: 192 staged rows per k-tile for half the products of 128x128's 256 -- 0.70 x 0.5 / 0.75)
      const double e = compute == 2 ? (c == 5 ? 0.70 : (c == 6 ? 0.68 : (c == 4 ? x3_e64() : 0.45))) : ((c >= 16 && tiles * s <= 256) ? 1.0 : eff[c]);
      double cost = (double)rounds * bm * bn * kc / 128.0 / (e * occ2);
      cost += 3000.0;  // fill/drain + launch  (charging the split kernels' traced 6.2 us of prologue + epilogue per ROUND instead
                       // was tried in round 4: with the k-tile lists of the bench batch the weight gradients then take 4
                       // splits and 110 us where this model's 5 splits take 96)
      if (s > 1) {
        // slabs: s*M*N floats written then read once (plus the final write) at ~4 TB/s ~ 1.7 KB/clk chip-wide
        cost += ((double)s * 2.0 + 1.0) * M * N * 4.0 / 1700.0 + 8000.0;
      }
      if (cost < best) { best = cost; bc = c; bs = s; }
    }
  }
  *cfg_out = bc;
  *splits_out = bs;
}

}  // namespace mtvaf

using namespace mtvaf;

extern "C" {

// Bytes of workspace mtvaf_gemm_f32 may need for split-K slabs (upper bound over its heuristics).
size_t mtvaf_gemm_f32_workspace_bytes(int M, int N, int K, int allow_split) {
  if (!allow_split) return 0;
  return (size_t)16 * M * N * sizeof(float);
}

// Launch profiler: after mtvaf_prof_start(capacity) every mtvaf_gemm_f32 call records a HIP-event pair around
// its main kernel on the launch stream.  mtvaf_prof_stop synchronises those events and returns, per record,
// key = {tile cfg, layout_a, layout_b, fast path, M, N, K, splits} and the kernel duration in ms.
int mtvaf_prof_start(int capacity) {
  if (g_prof || capacity <= 0) return MTVAF_ERR_ARG;
  g_prof = new ProfRec[capacity];
  for (int i = 0; i < capacity; ++i) {
    if (hipEventCreate(&g_prof[i].e0) != hipSuccess || hipEventCreate(&g_prof[i].e1) != hipSuccess) return MTVAF_ERR_ARG;
  }
  g_prof_cap = capacity;
  g_prof_n = 0;
  return MTVAF_OK;
}

int mtvaf_prof_stop(int* n_out, int* keys, float* ms, int max_records) {
  if (!g_prof || !n_out) return MTVAF_ERR_ARG;
  const int n = g_prof_n < max_records ? g_prof_n : max_records;
  for (int i = 0; i < n; ++i) {
    (void)hipEventSynchronize(g_prof[i].e1);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, g_prof[i].e0, g_prof[i].e1);
    if (ms) ms[i] = t;
    if (keys)
      for (int j = 0; j < 8; ++j) keys[i * 8 + j] = g_prof[i].key[j];
  }
  *n_out = n;
  for (int i = 0; i < g_prof_cap; ++i) {
    (void)hipEventDestroy(g_prof[i].e0);
    (void)hipEventDestroy(g_prof[i].e1);
  }
  delete[] g_prof;
  g_prof = nullptr;
  g_prof_cap = g_prof_n = 0;
  return MTVAF_OK;
}

// Reports the tile configuration / split count the heuristic would pick (for profiling tools).
// tile (BMxBNxBK, 256 threads): 0 128x128x16, 1 128x96x16, 2 128x288x16, 3 64x64x16, 4 128x64x16,
// 5 128x128x32, 6 128x96x32, 7 128x192x16, 8 128x192x32; 9 128x96x32, 10 128x128x32, 11 128x192x32 on the
// LDS-DMA pipeline (aligned shapes, KC x KC / KC x KM).
int mtvaf_gemm_f32_plan(int layout_a, int layout_b, int M, int N, int K, int epi, int allow_split, int* cfg,
                        int* splits) {
  if (M <= 0 || N <= 0 || K <= 0 || !cfg || !splits) return MTVAF_ERR_ARG;
  choose(M, N, K, allow_split && splittable(epi), layout_a, layout_b, epi, cfg, splits);
  return MTVAF_OK;
}

// C[M,N] = opA[M,K] . opB[K,N] (+bias) with fused epilogue.  layout_a / layout_b: 0 = KC, 1 = KM.
// epi: 0 none, 1 bias+GELU (pre-activation stored to aux), 2 bias+tanh, 3 dGELU (multiply by
// gelu'(aux)), 4 dtanh (multiply by 1-aux^2).  accumulate: C += result.  allow_split: permit a
// deterministic split-K (slabs in workspace + ordered reduction); cfg/splits < 0 = heuristic.
static long long* g_x3_trace = nullptr;
static int g_x3_tile_walk = [] { const char* e = getenv("MTVAF_X3_TILE_WALK"); return e ? atoi(e) : 0; }();

static int gemm_dispatch(int compute, int layout_a, int layout_b, const float* A, int lda, const float* B, int ldb,
                         float* C, int ldc, int M, int N, int K, const float* bias, int epi, float* aux, int ldaux,
                         int accumulate, int allow_split, void* workspace, size_t workspace_bytes, int cfg, int splits,
                         hipStream_t stream, const int* klist = nullptr, const int* kcnt = nullptr, int* keep_slabs = nullptr) {
  if (M <= 0 || N <= 0 || K <= 0) return MTVAF_ERR_SHAPE;
  if (keep_slabs) *keep_slabs = 1;
  if (compute == 1) {
    // the bf16 kernels need k-aligned, vector-loadable operands; anything else runs the fp32 kernels
    const bool ok = (K % 32 == 0) && (lda % 4 == 0) && (ldb % 4 == 0) && (((uintptr_t)A & 15) == 0) &&
                    (((uintptr_t)B & 15) == 0) && (layout_a == 0 || M % 4 == 0) && (layout_b == 0 || N % 4 == 0) &&
                    M >= 4 && N >= 4 && !(layout_a == 1 && layout_b == 0);
    if (!ok) compute = 0;
    if (compute == 1 && !(cfg == 6 || cfg == 5 || cfg == 3)) cfg = -1;
  }
  bool rag5 = false;
  if (compute == 2) {
    // the split kernels take whole 64x64 tiles of k-aligned, vector-loadable operands; anything else runs the fp32 pipe
    const bool can5 = M % 128 == 0 && N % 128 == 0, can3 = M % 64 == 0 && N % 64 == 0;
    const bool can6 = M % 128 == 0 && N % 96 == 0;  // (forced; the planner takes it where x3_tile96_ok says so)
    // 128x128 tiles that hang over the result: the wave-specialised kernel only (it needs the wide epilogue's alignment)
    rag5 = !can5 && x3_ragged_ok(M, N) && (ldc % 4 == 0) && (((uintptr_t)C & 15) == 0) &&
           (!aux || ((ldaux % 4 == 0) && (((uintptr_t)aux & 15) == 0))) && (!bias || (((uintptr_t)bias & 15) == 0));
    const bool can4 = M % 128 == 0 && N % 64 == 0;
    const bool forced = (cfg == 5 && (can5 || rag5)) || (cfg == 6 && can6) || (cfg == 4 && can4) || (cfg == 3 && can3);
    const bool plannable = can5 || rag5 || can3 || x3_tile96_ok(M, N, layout_a, layout_b);
    // (leading dimensions < 2^23 elements: a tile's loads address it by 32-bit byte offsets from a scalar base)
    const bool ok = (K % 32 == 0) && (forced || plannable) && (lda % 4 == 0) && (ldb % 4 == 0) && (((uintptr_t)A & 15) == 0) &&
                    (((uintptr_t)B & 15) == 0) && !(layout_a == 1 && layout_b == 0) && lda < (1 << 23) && ldb < (1 << 23);
    // (few output tiles -- the 256-token products of BASELINE configs[0] -- leave most CUs without a block of the 128x128
    // kernel: the fp32 pipe's 64x64 tiles are faster there, measured 4.37 vs 4.53 ms per C1 step)
    const long tiles96 = can6 ? (long)(M / 128) * (N / 96) : 0;
    const long tiles64 = (can4 && x3_tile64_ok(M, N)) ? (long)(M / 128) * (N / 64) : 0;
    const long tiles = std::max(std::max(rag5 ? cdiv(M, 128) * cdiv(N, 128) : (long)(M / 128) * (N / 128), tiles96), tiles64);
    // ... unless the reduction is deep enough for split-K to fill the chip anyway (the [768 x 768] weight gradient over 4096
    // token rows: 36 tiles x 7 splits, 38-41 us against 46-51 on the fp32 pipe)
    const bool deep = allow_split && splittable(epi) && K >= 2048 && tiles * std::min(16, K / 256) >= 192;
    if (!ok || (!forced && tiles < 96 && !deep)) compute = 0;
    if (compute == 2 && !forced) cfg = -1;
    if (compute != 2) rag5 = false;
  }
  if (!A || !B || !C) return MTVAF_ERR_ARG;
  if ((epi == EPI_GELU || epi == EPI_DGELU || epi == EPI_DTANH) && !aux) return MTVAF_ERR_ARG;
  if (layout_a < 0 || layout_a > 1 || layout_b < 0 || layout_b > 1) return MTVAF_ERR_ARG;
  int c_auto, s_auto;
  choose(M, N, K, allow_split && splittable(epi), layout_a, layout_b, epi, &c_auto, &s_auto, compute, rag5);
  const bool cfg_forced = cfg >= 0 && cfg < kNumCfgs;
  if (!cfg_forced) cfg = c_auto;
  if (splits <= 0) splits = s_auto;
  if (!(allow_split && splittable(epi))) splits = 1;
  if (splits > 1 && (size_t)splits * M * N * sizeof(float) > workspace_bytes) {
    splits = (int)(workspace_bytes / ((size_t)M * N * sizeof(float)));
    if (splits < 1) splits = 1;
  }
  GemmArgs a;
  a.klist = nullptr; a.kcnt = nullptr; a.ngrp = 0;
  a.trace = g_x3_trace;
  a.tile_walk = g_x3_tile_walk;
  a.A = A; a.B = B; a.bias = bias; a.aux = aux;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldaux = ldaux;
  a.epi = epi; a.accumulate = accumulate;
  a.a_vec = (lda % 4 == 0) && (((uintptr_t)A & 15) == 0);
  a.b_vec = (ldb % 4 == 0) && (((uintptr_t)B & 15) == 0);
  const int bk = compute != 0 ? 32 : kCfgs[cfg].bk;
  int kc = (int)cdiv(cdiv(K, splits), bk) * bk;
  splits = (int)cdiv(K, kc);
  a.k_chunk = kc;
  if (splits > 1) {
    a.C = (float*)workspace; a.ldc = N; a.slab_stride = (long)M * N;
  } else {
    a.C = C; a.ldc = ldc; a.slab_stride = 0;
  }
  const int bm = kCfgs[cfg].bm, bn = kCfgs[cfg].bn;
  a.tiles_n = (int)cdiv(N, bn);
  dim3 grid((unsigned)(cdiv(M, bm) * a.tiles_n), 1, (unsigned)splits);
  // KM operands are loaded as float4 along the row index: a clamped tail needs M (N) % 4 == 0
  const bool fast = (K % bk == 0) && (kc % bk == 0) && a.a_vec && a.b_vec && (layout_a == 0 || M % 4 == 0) &&
                    (layout_b == 0 || N % 4 == 0) && M >= 4 && N >= 4;
  const bool aligned = fast && (M % bm == 0) && (N % bn == 0);
  const int mode = aligned ? 2 : (fast ? 1 : 0);
  // (rag5: the wave-specialised kernel guards its stores, so the wide epilogue does not need whole tiles there)
  a.wide = (aligned || (rag5 && cfg == 5 && fast)) && (a.ldc % 4 == 0) && (((uintptr_t)a.C & 15) == 0) &&
           (!aux || ((ldaux % 4 == 0) && (((uintptr_t)aux & 15) == 0))) && (!bias || (((uintptr_t)bias & 15) == 0)) &&
           (a.slab_stride % 4 == 0);
  if (compute == 2 && cfg == 5 && rag5 && !a.wide) return MTVAF_ERR_ALIGN;  // (only the guarded epilogue may hang over the result)
  if (compute == 0 && cfg >= kFirstDma && !(aligned && !(layout_a == 1 && layout_b == 0))) {
    if (cfg_forced) return MTVAF_ERR_SHAPE;
    static const int staged_twin[9] = {6, 5, 8, 6, 5, 4, 4, 3, 3};  // same tile, register-staged kernel (handles any alignment)
    cfg = staged_twin[cfg - kFirstDma];
  }
  // the k-tile list only reaches the LDS-DMA kernels (32-row k-tiles of k-major operands); any other plan reduces over
  // the whole range, which gives the same result (the skipped k-tiles are exact zeros by the caller's contract)
  if (klist && kcnt && ((compute == 0 && cfg >= kFirstDma) || compute == 2) && layout_a == 1 && layout_b == 1 && K % 32 == 0) {
    a.klist = klist;
    a.kcnt = kcnt;
  }
  ProfRec* pr = nullptr;
  if (g_prof && g_prof_n < g_prof_cap) {
    pr = &g_prof[g_prof_n++];
    // (key[3]: alignment mode 0..2, +8 when the launch walks a k-tile list: its flops are 2 M N 32 (*kcnt), not 2 M N K)
    const int key[8] = {compute == 1 ? 100 + cfg : (compute == 2 ? 200 + cfg + (((cfg == 5 || cfg == 6 || cfg == 4) && a.wide) ? 20 : 0) : cfg), layout_a, layout_b, mode + (a.klist ? 8 : 0), M, N, K, splits};
    for (int i = 0; i < 8; ++i) pr->key[i] = key[i];
    (void)hipEventRecord(pr->e0, stream);
  }
  int rc;
  if (compute == 1) {
    rc = launch_gemm_bf16(cfg == 6 ? 0 : (cfg == 5 ? 1 : 2), a, layout_a, layout_b, grid, aligned, stream);
  } else if (compute == 2) {
    // 128x128 / 128x96: the wave-specialised kernel; without the wide epilogue's alignment the every-wave-does-everything form
    if (cfg == 4 && !a.wide) return MTVAF_ERR_ALIGN;  // (the 128x64 layout exists in the wave-specialised kernel only)
    rc = launch_gemm_f32x3(cfg == 5 ? (a.wide ? 4 : 1) : (cfg == 6 ? (a.wide ? 5 : 3) : (cfg == 4 ? 6 : 2)), a, layout_a, layout_b, grid, stream);
  } else
  switch (cfg) {
    case 9: rc = launch_dma<128, 96, 4, 1>(a, layout_a, layout_b, grid, stream); break;
    case 10: rc = launch_dma<128, 128, 2, 2>(a, layout_a, layout_b, grid, stream); break;
    case 11: rc = launch_dma<128, 192, 2, 2>(a, layout_a, layout_b, grid, stream); break;
    case 12: rc = launch_dma<128, 96, 4, 1, 2>(a, layout_a, layout_b, grid, stream); break;
    case 13: rc = launch_dma<128, 128, 2, 2, 2>(a, layout_a, layout_b, grid, stream); break;
    case 14: rc = launch_dma<128, 64, 4, 1>(a, layout_a, layout_b, grid, stream); break;
    case 15: rc = launch_dma<128, 64, 4, 1, 2>(a, layout_a, layout_b, grid, stream); break;
    case 16: rc = launch_dma<64, 64, 2, 2>(a, layout_a, layout_b, grid, stream); break;
    case 17: rc = launch_dma<64, 64, 2, 2, 2>(a, layout_a, layout_b, grid, stream); break;
    case 0: rc = launch_cfg<128, 128, 2, 2, 16>(a, layout_a, layout_b, grid, mode, stream); break;
    case 1: rc = launch_cfg<128, 96, 4, 1, 16>(a, layout_a, layout_b, grid, mode, stream); break;
    case 2: rc = launch_cfg<128, 288, 4, 1, 16>(a, layout_a, layout_b, grid, mode, stream); break;
    case 3: rc = launch_cfg<64, 64, 2, 2, 16>(a, layout_a, layout_b, grid, mode, stream); break;
    case 4: rc = launch_cfg<128, 64, 4, 1, 16>(a, layout_a, layout_b, grid, mode, stream); break;
    case 5: rc = launch_cfg<128, 128, 2, 2, 32>(a, layout_a, layout_b, grid, mode, stream); break;
    case 6: rc = launch_cfg<128, 96, 4, 1, 32>(a, layout_a, layout_b, grid, mode, stream); break;
    case 7: rc = launch_cfg<128, 192, 2, 2, 16>(a, layout_a, layout_b, grid, mode, stream); break;
    default: rc = launch_cfg<128, 192, 2, 2, 32>(a, layout_a, layout_b, grid, mode, stream); break;
  }
  if (pr) (void)hipEventRecord(pr->e1, stream);
  if (rc != MTVAF_OK) return rc;
  if (splits > 1) {
    if (keep_slabs) {  // (mtvaf_gemm_f32_slabs: the caller's next kernel adds the slabs itself)
      *keep_slabs = splits;
      return MTVAF_OK;
    }
    return launch_splitk_reduce((const float*)workspace, splits, C, M, N, ldc, bias, accumulate, epi, aux, ldaux, stream);
  }
  return MTVAF_OK;
}

// fp32 products on the bf16 matrix pipe by three-way operand splitting (gemm_f32x3.hip: six bf16 MFMA products per fp32
// product, fp32 accumulation, fp32 operands and results in memory; accuracy of the fp32 pipe).  mtvaf_f32_split(1 / 0) makes
// mtvaf_gemm_f32 / mtvaf_gemm_f32_ktiles use it for every shape it covers (whole 64x64 tiles, K % 32 == 0, aligned
// operands) / never; (-1) queries.  Default since round 4: ON (the error against the fp64 product is at or below the fp32
// MFMA pipe's on the same operands: tests/test_ops_gpu.py::test_gemm_f32_split_accuracy, ..._adversarial); MTVAF_F32_SPLIT=0
// keeps every product on the fp32 pipe.
// profiling hook (tools/x3_trace.py): every following launch of the wave-specialised split kernel records, in block 0, the
// shader clock at which each wave arrives at / leaves each k-tile barrier into buf ([8][64][4] + 17 int64); NULL switches it off
int mtvaf_f32x3_trace(void* buf) {
  g_x3_trace = static_cast<long long*>(buf);
  return MTVAF_OK;
}

static int g_f32_split = -1;
int mtvaf_f32_split(int on) {
  if (on >= 0) g_f32_split = on ? 1 : 0;
  if (g_f32_split < 0) {
    const char* e = getenv("MTVAF_F32_SPLIT");
    g_f32_split = (e && atoi(e) == 0) ? 0 : 1;
  }
  return g_f32_split;
}

int mtvaf_gemm_f32x3(int layout_a, int layout_b, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                     int M, int N, int K, const float* bias, int epi, float* aux, int ldaux, int accumulate,
                     int allow_split, void* workspace, size_t workspace_bytes, int cfg, int splits,
                     hipStream_t stream) {
  return gemm_dispatch(2, layout_a, layout_b, A, lda, B, ldb, C, ldc, M, N, K, bias, epi, aux, ldaux, accumulate,
                       allow_split, workspace, workspace_bytes, cfg, splits, stream);
}

int mtvaf_gemm_f32(int layout_a, int layout_b, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                   int M, int N, int K, const float* bias, int epi, float* aux, int ldaux, int accumulate,
                   int allow_split, void* workspace, size_t workspace_bytes, int cfg, int splits,
                   hipStream_t stream) {
  return gemm_dispatch(mtvaf_f32_split(-1) ? 2 : 0, layout_a, layout_b, A, lda, B, ldb, C, ldc, M, N, K, bias, epi, aux, ldaux, accumulate,
                       allow_split, workspace, workspace_bytes, cfg, splits, stream);
}

// mtvaf_gemm_f32 (plain epilogue) that leaves a split-K plan's slabs UNREDUCED (round 5): *splits_out = 1 -> C holds the result
// (+ bias, accumulate) as usual; *splits_out = s > 1 -> `workspace` holds s slabs [M][N] (leading dimension N, slab stride M * N
// floats), neither bias nor accumulate applied, C untouched: the consumer adds them in the order the reduction launch would
// (slab 0 + slab 1 + ... + bias [+ C]) -- mtvaf_dropout_res_ln_fwd_slabs / mtvaf_dropout_res_ln_bwd_rows_slabs -- and must run
// before anything else uses the workspace.  Saves the reduction launch and one pass over the result per product.
int mtvaf_gemm_f32_slabs(int layout_a, int layout_b, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N,
                         int K, const float* bias, int accumulate, void* workspace, size_t workspace_bytes, int* splits_out,
                         hipStream_t stream) {
  if (!splits_out) return MTVAF_ERR_ARG;
  return gemm_dispatch(mtvaf_f32_split(-1) ? 2 : 0, layout_a, layout_b, A, lda, B, ldb, C, ldc, M, N, K, bias, EPI_NONE, nullptr, 0, accumulate,
                       1, workspace, workspace_bytes, -1, -1, stream, nullptr, nullptr, splits_out);
}

// mtvaf_gemm_f32 for a weight-gradient product (layouts KM x KM: C[M,N] = A[K,M]^T . B[K,N], the reduction index is the
// token row) whose operand A is known to be EXACTLY ZERO outside the listed 32-row k-tiles: the reduction runs over
// klist[0 .. *kcnt) only (device arrays: no host sync; k-tile t = rows 32 t .. 32 t + 31).  Gradients of token rows that
// nothing downstream reads -- padded positions -- are exact zeros, so skipping their k-tiles changes nothing but time.
// Plans that cannot use the list (unaligned shapes, other layouts) reduce over the whole range: same result.
int mtvaf_gemm_f32_ktiles(int layout_a, int layout_b, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                          int M, int N, int K, const float* bias, int epi, float* aux, int ldaux, int accumulate,
                          int allow_split, void* workspace, size_t workspace_bytes, int cfg, int splits, const int* klist,
                          const int* kcnt, hipStream_t stream) {
  return gemm_dispatch(mtvaf_f32_split(-1) ? 2 : 0, layout_a, layout_b, A, lda, B, ldb, C, ldc, M, N, K, bias, epi, aux, ldaux, accumulate,
                       allow_split, workspace, workspace_bytes, cfg, splits, stream, klist, kcnt);
}

// Up to four weight-gradient products dW_i[M_i, N_i] = A_i^T . B_i (layouts KM x KM: A_i [K, M_i], B_i [K, N_i] fp32 row-major)
// that share the reduction axis K -- the four products of one encoder layer -- in ONE launch of the 128x96 two-blocks-per-CU
// LDS-DMA kernel (its GROUP instantiation: blockIdx.x walks the tiles of the products back to back), with an optional k-tile
// list as mtvaf_gemm_f32_ktiles.  One launch instead of four (+ their split-K reductions when the tiles alone fill the chip):
// at 256 token rows (BASELINE configs[0]) the four products are 576 tiles of 8 k-tiles -- a single 16-us launch where four
// launches + four slab reductions took 80.  Deterministic split-K with per-product slabs in `workspace` when the reduction is
// long.  Requires M_i % 128 == 0, N_i % 96 == 0, K % 32 == 0, leading dimensions % 4 == 0, 16-byte aligned pointers;
// MTVAF_ERR_SHAPE / _ALIGN otherwise (no fallback inside the library: the caller launches the products one by one).
// splits of a grouped weight-gradient launch: fill the 512 block slots of the two-blocks-per-CU ring several times over when
// the reduction is long enough (requested > 0: that many, clamped to >= 4 k-tiles per split)
static int dw_group_plan(long tiles, int K, int requested) {
  const int ktiles = K / 32;
  int splits = requested;
  if (splits <= 0) {
    splits = 1;
    while (tiles * splits < 2048 && ktiles / (splits + 1) >= 16 && splits < 8) ++splits;
  }
  while (splits > 1 && ktiles / splits < 4) --splits;
  const int kc = (int)cdiv(cdiv(K, splits), 32) * 32;
  return (int)cdiv(K, kc);
}

// The grouped launch on the split kernel (gemm_f32x3_ws_kernel<.., GROUP>): split arithmetic in force, every product whole
// 128 x 128 tiles, and a reduction longer than the few-token case the fp32 pipe's grouped ring was tuned for (K <= 1024: kept)
static bool dw_group_on_split_kernel(int n, const int* M, const int* N, int K) {
  if (!mtvaf_f32_split(-1) || K <= 1024) return false;
  for (int i = 0; i < n; ++i)
    if (M[i] % 128 || N[i] % 128) return false;
  return true;
}
// its splits: none once the tiles alone cover three quarters of the CUs (the second stream shares the chip with the main one:
// what counts is CU time, and an unsplit block pays the ~6 us of prologue + epilogue once), otherwise enough to get there
static int dw_group_plan_x3(long tiles, int K, int requested) {
  const int ktiles = K / 32;
  int splits = requested;
  if (splits <= 0) {
    splits = 1;
    while (tiles * splits < 192 && ktiles / (splits + 1) >= 16 && splits < 8) ++splits;
  }
  while (splits > 1 && ktiles / splits < 4) --splits;
  const int kc = (int)cdiv(cdiv(K, splits), 32) * 32;
  return (int)cdiv(K, kc);
}

// Bytes of split-K slabs mtvaf_gemm_f32_dw_group needs for these products (0: no split planned).
size_t mtvaf_gemm_f32_dw_group_workspace_bytes(int n, const int* M, const int* N, int K, int splits) {
  if (n < 1 || n > 4 || K <= 0 || K % 32 || !M || !N) return 0;
  long tiles = 0, outs = 0;
  if (dw_group_on_split_kernel(n, M, N, K)) {
    for (int i = 0; i < n; ++i) {
      if (M[i] <= 0 || N[i] <= 0) return 0;
      tiles += (long)(M[i] / 128) * (N[i] / 128);
      outs += (long)M[i] * N[i];
    }
    const int s = dw_group_plan_x3(tiles, K, splits);
    return s > 1 ? (size_t)s * outs * sizeof(float) : 0;
  }
  for (int i = 0; i < n; ++i) {
    if (M[i] <= 0 || N[i] <= 0 || M[i] % 128 || N[i] % 96) return 0;
    tiles += (long)(M[i] / 128) * (N[i] / 96);
    outs += (long)M[i] * N[i];
  }
  const int s = dw_group_plan(tiles, K, splits);
  return s > 1 ? (size_t)s * outs * sizeof(float) : 0;
}

int mtvaf_colsum(const float* x, int rows, int cols, int ld, float* out, int accumulate, void* workspace, size_t workspace_bytes,
                 hipStream_t st);  // rowops.hip

// dbias (nullable; entries nullable): dbias[i][0 .. M_i) = column sums of A_i over the reduction axis -- the bias gradient of the
// dense layer whose weight gradient product i is (A_i = its dY).  The unsplit grouped launch of the split kernel takes them
// from the A tiles it stages anyway (no extra pass over dY, no extra launch); every other plan runs mtvaf_colsum behind the
// products (`workspace` is theirs again by then, in stream order).
static int dw_group_impl(int n, const float* const* A, const int* lda, const float* const* B, const int* ldb, float* const* C,
                         const int* ldc, const int* M, const int* N, int K, const int* klist, const int* kcnt, float* const* dbias,
                         void* workspace, size_t workspace_bytes, int splits, hipStream_t stream) {
  if (n < 1 || n > 4 || K <= 0 || K % 32) return MTVAF_ERR_SHAPE;
  GemmArgs a = {};
  long tiles = 0, outs = 0;
  for (int i = 0; i < n; ++i)
    if (M[i] <= 0 || N[i] <= 0) return MTVAF_ERR_SHAPE;
  if (dw_group_on_split_kernel(n, M, N, K)) {
    for (int i = 0; i < n; ++i) {
      if (lda[i] % 4 || ldb[i] % 4 || ldc[i] % 4 || (((uintptr_t)A[i] | (uintptr_t)B[i] | (uintptr_t)C[i]) & 15)) return MTVAF_ERR_ALIGN;
      if (lda[i] >= (1 << 23) || ldb[i] >= (1 << 23)) return MTVAF_ERR_SHAPE;  // (32-bit byte offsets inside a tile)
      a.grp_tile_begin[i] = (int)tiles;
      tiles += (long)(M[i] / 128) * (N[i] / 128);
      outs += (long)M[i] * N[i];
    }
    for (int i = n; i < 4; ++i) a.grp_tile_begin[i] = INT_MAX;
    splits = dw_group_plan_x3(tiles, K, splits);
    if (splits > 1 && (size_t)splits * outs * sizeof(float) > workspace_bytes) return MTVAF_ERR_WORKSPACE;
    float* slab = static_cast<float*>(workspace);
    for (int i = 0; i < n; ++i) {
      GemmProb& g = a.grp[i];
      g.A = A[i]; g.B = B[i]; g.lda = lda[i]; g.ldb = ldb[i]; g.tiles_n = N[i] / 128;
      if (splits > 1) {
        g.C = slab; g.ldc = N[i]; g.slab_stride = (long)M[i] * N[i];
        slab += (long)splits * M[i] * N[i];
      } else {
        g.C = C[i]; g.ldc = ldc[i]; g.slab_stride = 0;
      }
      static const int fuse_bias = [] { const char* e = getenv("MTVAF_X3_DW_BIAS"); return e ? atoi(e) : 1; }();  // (0: always mtvaf_colsum)
      g.colsum = (fuse_bias && splits == 1 && dbias && dbias[i] && (((uintptr_t)dbias[i] & 15) == 0)) ? dbias[i] : nullptr;
    }
    a.ngrp = n;
    a.A = a.grp[0].A; a.B = a.grp[0].B; a.C = a.grp[0].C;
    a.M = M[0]; a.N = N[0]; a.K = K; a.lda = lda[0]; a.ldb = ldb[0]; a.ldc = a.grp[0].ldc;
    a.k_chunk = (int)cdiv(cdiv(K, splits), 32) * 32;
    a.epi = EPI_NONE; a.a_vec = a.b_vec = 1; a.tiles_n = a.grp[0].tiles_n;
    a.klist = (klist && kcnt) ? klist : nullptr;
    a.kcnt = a.klist ? kcnt : nullptr;
    a.wide = 1;
    a.trace = nullptr; a.tile_walk = 0;
    const int key[8] = {1225, 1, 1, a.klist ? 10 : 2, (int)(outs / 768), 768, K, splits};  // (1000 + 225: the GROUP instantiation)
    const int rec = prof_begin(key, stream);
    const int rc = launch_gemm_f32x3_group(a, dim3((unsigned)tiles, 1, (unsigned)splits), stream);
    prof_end(rec, stream);
    if (rc != MTVAF_OK) return rc;
    if (splits > 1)
      for (int i = 0; i < n; ++i) {
        const int r2 = launch_splitk_reduce(a.grp[i].C, splits, C[i], M[i], N[i], ldc[i], nullptr, 0, EPI_NONE, nullptr, 0, stream);
        if (r2 != MTVAF_OK) return r2;
      }
    for (int i = 0; dbias && i < n; ++i)  // (the sums the launch did not take itself)
      if (dbias[i] && !a.grp[i].colsum) {
        const int r3 = mtvaf_colsum(A[i], K, M[i], lda[i], dbias[i], 0, workspace, workspace_bytes, stream);
        if (r3 != MTVAF_OK) return r3;
      }
    return MTVAF_OK;
  }
  for (int i = 0; i < n; ++i) {
    if (M[i] <= 0 || N[i] <= 0 || M[i] % 128 || N[i] % 96) return MTVAF_ERR_SHAPE;
    if (lda[i] % 4 || ldb[i] % 4 || ldc[i] % 4 || (((uintptr_t)A[i] | (uintptr_t)B[i] | (uintptr_t)C[i]) & 15)) return MTVAF_ERR_ALIGN;
    a.grp_tile_begin[i] = (int)tiles;
    tiles += (long)(M[i] / 128) * (N[i] / 96);
    outs += (long)M[i] * N[i];
  }
  for (int i = n; i < 4; ++i) a.grp_tile_begin[i] = INT_MAX;
  // the plan (dw_group_plan) never depends on the workspace the caller happens to hold: too little is an error, so that the
  // executor and the Python orchestration -- which size their scratch differently -- always cut the reduction alike
  splits = dw_group_plan(tiles, K, splits);
  if (splits > 1 && (size_t)splits * outs * sizeof(float) > workspace_bytes) return MTVAF_ERR_WORKSPACE;
  const int kc = (int)cdiv(cdiv(K, splits), 32) * 32;
  float* slab = static_cast<float*>(workspace);
  for (int i = 0; i < n; ++i) {
    GemmProb& g = a.grp[i];
    g.colsum = nullptr;
    g.A = A[i]; g.B = B[i]; g.lda = lda[i]; g.ldb = ldb[i]; g.tiles_n = N[i] / 96;
    if (splits > 1) {
      g.C = slab; g.ldc = N[i]; g.slab_stride = (long)M[i] * N[i];
      slab += (long)splits * M[i] * N[i];
    } else {
      g.C = C[i]; g.ldc = ldc[i]; g.slab_stride = 0;
    }
  }
  a.ngrp = n;
  a.A = a.grp[0].A; a.B = a.grp[0].B; a.C = a.grp[0].C;
  a.M = M[0]; a.N = N[0]; a.K = K; a.lda = lda[0]; a.ldb = ldb[0]; a.ldc = a.grp[0].ldc;
  a.k_chunk = kc; a.epi = EPI_NONE; a.a_vec = a.b_vec = 1; a.tiles_n = a.grp[0].tiles_n;
  a.klist = (klist && kcnt) ? klist : nullptr;
  a.kcnt = a.klist ? kcnt : nullptr;
  a.wide = 1;  // every product: whole tiles, ldc % 4 == 0, 16-byte aligned result / slabs
  const size_t smem = std::max((size_t)2 * (128 + 96) * 32 * sizeof(float), (size_t)128 * (96 + 4) * sizeof(float));
  const dim3 grid((unsigned)tiles, 1, (unsigned)splits), block(256);
  // (profiler key: the group's flops as one M x 768 x K product, fast bit 8 = the launch walks a k-tile list)
  const int key[8] = {1012, 1, 1, a.klist ? 10 : 2, (int)(outs / 768), 768, K, splits};  // (1000 + tile cfg: the GROUP instantiation)
  const int rec = prof_begin(key, stream);
  if (a.klist) {
    auto kern = gemm_f32_dma_group_kernel<128, 96, 4, 1, 2, true>;
    static bool attr = false;
    if (!attr) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return (int)e;
      attr = true;
    }
    hipLaunchKernelGGL(kern, grid, block, smem, stream, a);
  } else {
    auto kern = gemm_f32_dma_group_kernel<128, 96, 4, 1, 2, false>;
    static bool attr = false;
    if (!attr) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return (int)e;
      attr = true;
    }
    hipLaunchKernelGGL(kern, grid, block, smem, stream, a);
  }
  prof_end(rec, stream);
  MTVAF_LAUNCH_CHECK();
  if (splits > 1) {
    for (int i = 0; i < n; ++i) {
      const int rc = launch_splitk_reduce(a.grp[i].C, splits, C[i], M[i], N[i], ldc[i], nullptr, 0, EPI_NONE, nullptr, 0, stream);
      if (rc != MTVAF_OK) return rc;
    }
  }
  for (int i = 0; dbias && i < n; ++i)
    if (dbias[i]) {
      const int rc = mtvaf_colsum(A[i], K, M[i], lda[i], dbias[i], 0, workspace, workspace_bytes, stream);
      if (rc != MTVAF_OK) return rc;
    }
  return MTVAF_OK;
}

int mtvaf_gemm_f32_dw_group(int n, const float* const* A, const int* lda, const float* const* B, const int* ldb, float* const* C,
                            const int* ldc, const int* M, const int* N, int K, const int* klist, const int* kcnt,
                            void* workspace, size_t workspace_bytes, int splits, hipStream_t stream) {
  return dw_group_impl(n, A, lda, B, ldb, C, ldc, M, N, K, klist, kcnt, nullptr, workspace, workspace_bytes, splits, stream);
}

// mtvaf_gemm_f32_dw_group + the bias gradients that go with the weight gradients: dbias[i] (nullable) <- column sums of A_i
// over the K rows ([M_i] floats, 16-byte aligned, overwritten).  workspace_bytes must also cover mtvaf_colsum_workspace_bytes.
int mtvaf_gemm_f32_dw_group_bias(int n, const float* const* A, const int* lda, const float* const* B, const int* ldb, float* const* C,
                                 const int* ldc, const int* M, const int* N, int K, const int* klist, const int* kcnt,
                                 float* const* dbias, void* workspace, size_t workspace_bytes, int splits, hipStream_t stream) {
  return dw_group_impl(n, A, lda, B, ldb, C, ldc, M, N, K, klist, kcnt, dbias, workspace, workspace_bytes, splits, stream);
}

// Same contract as mtvaf_gemm_f32 (fp32 operands and results in memory), but the products run on the bf16
// MFMA with fp32 accumulation: operands are rounded to bf16 (RNE) while tiles are staged.  Shapes the bf16
// kernels cannot take (K % 32 != 0, unaligned operands) silently use the fp32 kernels.
int mtvaf_gemm_bf16(int layout_a, int layout_b, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                    int M, int N, int K, const float* bias, int epi, float* aux, int ldaux, int accumulate,
                    int allow_split, void* workspace, size_t workspace_bytes, int cfg, int splits,
                    hipStream_t stream) {
  return gemm_dispatch(1, layout_a, layout_b, A, lda, B, ldb, C, ldc, M, N, K, bias, epi, aux, ldaux, accumulate,
                       allow_split, workspace, workspace_bytes, cfg, splits, stream);
}

}  // extern "C"
