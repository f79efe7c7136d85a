// bf16-operand GEMM of the mixed-precision mode (BASELINE configs 3-4): every dense product of the path -- forward
// x.W^T, dX = dY.W, dW = dY^T.X -- on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, reading the SAME row-major bf16
// tensors in all three roles (no transposed copies, no cast passes):
//
//   operand "KC": reduction index contiguous in memory (x[m][k], W[n][k] forward, dY[m][n] as A of dX).  LDS image: one
//                 128-byte row of 64 k per tile row, fragments by ds_read_b128 (16-byte chunk 2*ks + h of row r holds
//                 the 8 consecutive k of lane (r, h)); bank conflicts removed by chunk ^ ((row >> 1) & 7).
//   operand "KM": reduction index is the ROW (W[n][k] as B of dX; dY[m][n] and x[m][k] as A / B of dW, reduction over
//                 the tokens m).  LDS image: 64 reduction rows of 128 output columns (256-byte rows); fragments by
//                 ds_read_b64_tr_b16, the gfx950 transposing LDS read: a 16-lane group fetches a 4 (reduction) x 16
//                 (column) block and every lane receives ITS column's 4 reduction values, two reads per MFMA operand.
//                 Conflict-free with the 16-byte chunk XOR  ((row & 3) << 2) | ((row >> 2) & 3).
//
// Tiles go HBM -> LDS with global_load_lds_dwordx4 (1 KiB per wave instruction; the LDS side is lane-linear, so both
// swizzles are applied to the per-lane SOURCE address and again on the read).  Ring of NSTAGE k-tiles:
//   NSTAGE 3: one block per CU, two k-tiles in flight across the barrier behind a COUNTED s_waitcnt vmcnt;
//   NSTAGE 2: two blocks per CU that hide each other's barrier / prologue / epilogue stalls.
// Epilogue (tile transposed through LDS, 8 columns per lane, 16-byte stores): bias, erf-GELU (pre-activation saved as
// bf16), * GELU'(pre), accumulate into fp32 C, fp32 and / or bf16 result, and per-tile column sums of the result (the
// bias gradient of the producing layer, finished by mtvaf_colsum_small) -- or fp32 split-K slabs + ordered reduction.
#include "gemm_bf16x.h"

#include <algorithm>
#include <mutex>

namespace mtvaf {

// DW (round 4; four-wave tiles): NW more waves that do nothing but issue the LDS-DMA requests.  A global_load_lds request
// costs the issuing wave 80-95 cycles of issue beside the matrix stream (traced on the split-fp32 kernels, tools/x3_trace.py),
// and a wave issues in order: with 7-8 requests per wave and k-tile in the MFMA waves, the requests took as long as the
// tile's 16 MFMAs -- which is what SQ_VALU_MFMA_BUSY 0.15-0.22 of these rings showed, not (only) the L2 -> LDS rate.
template <int BM, int BN, int WM, int WN, bool A_KM, bool B_KM, int NSTAGE, bool KLIST = false, bool DW = false>
__global__ __launch_bounds__((DW ? 2 : 1) * WM* WN * 64, (BM * BN > 256 * 128) ? 1 : ((WM * WN == 8) ? 2 : ((NSTAGE >= 3 ? 1 : 2) * (DW ? 2 : 1)))) void gemm_bf16x_kernel(GemmArgsX p) {
  static_assert(!DW || WM * WN == 4, "DMA waves: four-wave tiles");
  constexpr int BK = 64;  // bf16 elements per k-tile
  constexpr int NW = WM * WN, NT = (DW ? 2 : 1) * NW * 64;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int A_B = BM * 128, B_B = BN * 128, STAGE_B = A_B + B_B;  // bytes (KC: BM rows x 128 B; KM: 64 rows x 2*BM B)
  constexpr int IA = A_B / 1024 / NW, IB = B_B / 1024 / NW;          // 1-KiB DMA instructions per wave
  static_assert(A_B % (1024 * NW) == 0 && B_B % (1024 * NW) == 0, "tile must split into whole DMA pieces per wave");
  static_assert((!A_KM || BM == 128) && (!B_KM || BN == 128 || BN == 96 || BN == 192),
                "k-major images: 256-byte rows (XOR swizzle) or 192-byte rows (BN = 192: two 96-column images side by side)");
  constexpr int KM96_HALF_B = 64 * 192;  // bytes of one [64 k-rows][96 columns] image

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];  // the ONLY LDS object

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool dma_wave = DW && wave_all >= NW;  // (wave-uniform)
  const int wave = DW ? wave_all % NW : wave_all;  // MFMA wave index / owner index of a DMA wave's pieces
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, h = lane >> 5;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / p.tiles_n) * BM;
  const int n0 = (bid % p.tiles_n) * BN;
  int kbeg = blockIdx.z * p.k_chunk;
  const int kend = min(p.K, kbeg + p.k_chunk);
  int nk = (kend - kbeg) / BK;
  static_assert(!KLIST || (A_KM && B_KM), "the k-tile list addresses rows of k-major operands");
  constexpr bool klist_mode = KLIST;
  int lbeg = 0;
  if constexpr (klist_mode) {  // this split's share of the listed k-tiles (the count lives on the device)
    const int cnt = *p.kcnt;
    const int per = (cnt + (int)gridDim.z - 1) / (int)gridDim.z;
    lbeg = blockIdx.z * per;
    nk = max(0, min(cnt - lbeg, per));
    kbeg = 0;
  }

  // ---- per-lane DMA source addresses: LDS position (row, chunk position cp) receives source chunk cp ^ swizzle(row).
  // A request's address is a UNIFORM base (the operand's tile at the current k-tile: scalar arithmetic) + a per-lane 32-bit
  // byte offset that never changes (round 4: global_load_lds ... v_offset, s[base]; until then every lane carried IA + IB
  // 64-bit pointers and advanced each with a 64-bit vector add per request and k-tile).
  unsigned oa[IA], ob[IB];
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int f = (wave * IA + i) * 64 + lane;
    if (!A_KM) {
      const int r = f >> 3, cp = f & 7;
      oa[i] = (unsigned)r * (unsigned)p.lda * 2u + (unsigned)((cp ^ ((r >> 1) & 7)) << 4);
    } else {
      const int r = f >> 4, cp = f & 15;
      oa[i] = (unsigned)r * (unsigned)p.lda * 2u + (unsigned)((cp ^ km_swz(r)) << 4);
    }
  }
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const int f = (wave * IB + i) * 64 + lane;
    if (!B_KM) {
      const int r = f >> 3, cp = f & 7;
      ob[i] = (unsigned)r * (unsigned)p.ldb * 2u + (unsigned)((cp ^ ((r >> 1) & 7)) << 4);
    } else if (BN == 128) {
      const int r = f >> 4, cp = f & 15;
      ob[i] = (unsigned)r * (unsigned)p.ldb * 2u + (unsigned)((cp ^ km_swz(r)) << 4);
    } else {  // 192-byte rows: consecutive rows start 48 banks apart, the transposing reads are conflict-free unswizzled
      const int o = f << 4, half = o / KM96_HALF_B, oo = o % KM96_HALF_B, r = oo / 192, cb = oo % 192;
      ob[i] = (unsigned)r * (unsigned)p.ldb * 2u + (unsigned)(half * 192 + cb);
    }
  }
  const unsigned char* const baseA =
      reinterpret_cast<const unsigned char*>(A_KM ? p.A + (long)kbeg * p.lda + m0 : p.A + (long)m0 * p.lda + kbeg);
  const unsigned char* const baseB =
      reinterpret_cast<const unsigned char*>(B_KM ? p.B + (long)kbeg * p.ldb + n0 : p.B + (long)n0 * p.ldb + kbeg);
  const long stepA = A_KM ? (long)BK * p.lda * 2 : BK * 2;
  const long stepB = B_KM ? (long)BK * p.ldb * 2 : BK * 2;
  long posA = 0, posB = 0;  // (uniform) byte position of the next k-tile to request
  int issued = 0;
  int kt_next = (klist_mode && nk > 0) ? p.klist[lbeg] : 0;  // fetched one issue ahead (a uniform scalar load)
  auto issue = [&](int stage) {
    unsigned char* sa = smem_b + stage * STAGE_B + wave * IA * 1024;
    unsigned char* sb = smem_b + stage * STAGE_B + A_B + wave * IB * 1024;
    if constexpr (klist_mode) {
      posA = (long)kt_next * stepA;
      posB = (long)kt_next * stepB;
      ++issued;
      kt_next = p.klist[lbeg + min(issued, nk - 1)];
    }
    const unsigned char* ca = baseA + posA;
    const unsigned char* cb = baseB + posB;
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      unsigned o = oa[i];
      asm("" : "+v"(o));  // (opaque: keeps base + offset from being hoisted into a 64-bit vector pointer per request)
      glds16x(ca + o, sa + i * 1024);
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      unsigned o = ob[i];
      asm("" : "+v"(o));
      glds16x(cb + o, sb + i * 1024);
    }
    if constexpr (!klist_mode) {
      posA += stepA;
      posB += stepB;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- fragment read offsets (bytes inside an operand tile), k-step independent part
  int offA[TM][2], offB[TN][2];
  {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (!A_KM) {
        const int row = (wm * TM + i) * 32 + li;
        offA[i][0] = row * 128;
        offA[i][1] = (row >> 1) & 7;
      } else {
        const int ms = wm * TM + i;  // 32-row sub-tile of the 128 output rows
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          offA[i][jj] = 256 * (8 * (g >> 1) + 4 * jj + q) +
                        16 * ((((ms ^ q) & 3) << 2) | ((2 * (g & 1) + (pp >> 1)) ^ ((2 * (g >> 1) + jj) & 3))) + 8 * (pp & 1);
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if (!B_KM) {
        const int col = (wn * TN + j) * 32 + li;
        offB[j][0] = col * 128;
        offB[j][1] = (col >> 1) & 7;
      } else if (BN == 128) {
        const int ns = wn * TN + j;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          offB[j][jj] = 256 * (8 * (g >> 1) + 4 * jj + q) +
                        16 * ((((ns ^ q) & 3) << 2) | ((2 * (g & 1) + (pp >> 1)) ^ ((2 * (g >> 1) + jj) & 3))) + 8 * (pp & 1);
      } else {
        const int ns = wn * TN + j, half = ns / 3, nl = ns % 3;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          offB[j][jj] = half * KM96_HALF_B + 192 * (8 * (g >> 1) + 4 * jj + q) + nl * 64 + 32 * (g & 1) + 8 * pp;
      }
    }
  }

  bf16x8 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
  auto rdf = [&](const unsigned char* a, const unsigned char* b, int ks, bf16x8 (&fa)[TM], bf16x8 (&fb)[TN]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (!A_KM) fa[i] = *reinterpret_cast<const bf16x8*>(a + offA[i][0] + (((2 * ks + h) ^ offA[i][1]) << 4));
      else fa[i] = tr_read8(a + offA[i][0] + 4096 * ks, a + offA[i][1] + 4096 * ks);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if (!B_KM) fb[j] = *reinterpret_cast<const bf16x8*>(b + offB[j][0] + (((2 * ks + h) ^ offB[j][1]) << 4));
      else fb[j] = tr_read8(b + offB[j][0] + (BN == 128 ? 4096 : 3072) * ks, b + offB[j][1] + (BN == 128 ? 4096 : 3072) * ks);
    }
  };
  auto mm = [&](const bf16x8 (&fa)[TM], const bf16x8 (&fb)[TN]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
  };
  // one MFMA, then its share of the slice's fragment reads (a k-major fragment is two transposing reads).
  // MTVAF_BF16X_SPREAD=0 (compile-time A/B switch): the reads as a burst in front of the MFMA group, as until round 4.
#ifndef MTVAF_BF16X_SPREAD
#define MTVAF_BF16X_SPREAD 1
#endif
  auto burst = [&]() __attribute__((always_inline)) {
    if constexpr (!MTVAF_BF16X_SPREAD) __builtin_amdgcn_sched_barrier(0);
  };
  auto spread = [&]() __attribute__((always_inline)) {
    if constexpr (!MTVAF_BF16X_SPREAD) {
      __builtin_amdgcn_sched_barrier(0);
      return;
    }
    constexpr int NMF = TM * TN, NRD = TM * (A_KM ? 2 : 1) + TN * (B_KM ? 2 : 1), PER = (NRD + NMF - 1) / NMF;
#pragma unroll
    for (int i = 0; i < NMF; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, PER, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
#pragma unroll
  for (int i = 0; i < TM; ++i) fa1[i] = bf16x8{};
#pragma unroll
  for (int j = 0; j < TN; ++j) fb1[j] = bf16x8{};
  if (!DW || dma_wave) {
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
      if (s < nk) issue(s);
  }
  int st = 0;
  if (dma_wave) {  // the same barriers as the MFMA waves: one per k-tile
    for (int kt = 0; kt < nk; ++kt) {
      const int ahead = min(NSTAGE - 2, nk - 1 - kt);
      if (NSTAGE >= 5 && ahead == 3) wait_vm<3 * (IA + IB)>();
      else if (NSTAGE >= 4 && ahead == 2) wait_vm<2 * (IA + IB)>();
      else if (NSTAGE >= 3 && ahead == 1) wait_vm<IA + IB>();
      else wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + NSTAGE - 1 < nk) {
        int si = st + NSTAGE - 1;
        if (si >= NSTAGE) si -= NSTAGE;
        issue(si);
      }
      st = st + 1 == NSTAGE ? 0 : st + 1;
    }
    nk = 0;  // (no products in this wave)
  }
  for (int kt = 0; kt < nk; ++kt) {
    if constexpr (!DW) {
      // this wave's pieces of tile kt have landed; the (up to NSTAGE - 2) tiles issued after it stay in flight
      const int ahead = min(NSTAGE - 2, nk - 1 - kt);
      if (NSTAGE >= 5 && ahead == 3) wait_vm<3 * (IA + IB)>();
      else if (NSTAGE >= 4 && ahead == 2) wait_vm<2 * (IA + IB)>();
      else if (NSTAGE >= 3 && ahead == 1) wait_vm<IA + IB>();
      else wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();  // ... and everybody else's; every wave is done reading tile kt-1
    asm volatile("" ::: "memory");
    if constexpr (!DW) {
      if (kt + NSTAGE - 1 < nk) {
        int si = st + NSTAGE - 1;
        if (si >= NSTAGE) si -= NSTAGE;
        issue(si);  // into the stage tile kt-1 occupied
      }
    }
    const unsigned char* a = smem_b + st * STAGE_B;
    const unsigned char* b = a + A_B;
    // Fragment reads run ONE k-slice ahead of the MFMAs that consume them, across the tile barrier too (slice 3 of tile kt-1
    // is multiplied behind the barrier of tile kt, while slice 0 of tile kt is in flight): a slice is only TM x TN MFMAs
    // (128-512 cycles), and an in-order wave that reads and then multiplies pays the LDS latency per slice.
    // Round 5 (found on the pre-split fp32 kernel, csrc/gemm_f32p.hip): (1) the reads of a slice are SPREAD behind the MFMAs
    // of the slice before it (sched_group_barrier: one MFMA, then its share of the reads) -- issued as a burst between two MFMA
    // groups they kept the matrix pipe idle for the burst's issue time, which at 3-4 MFMAs per slice is half the loop;
    // (2) no conditional block: the first tile multiplies zero fragments instead of skipping the group (behind a branch hipcc
    // cannot count on the reads in flight and waits lgkmcnt(0) for the ones issued just before).
    rdf(a, b, 0, fa0, fb0);
    burst();
    mm(fa1, fb1);  // slice 3 of tile kt - 1 (kt == 0: zero fragments)
    spread();
    rdf(a, b, 1, fa1, fb1);
    burst();
    mm(fa0, fb0);
    spread();
    rdf(a, b, 2, fa0, fb0);
    burst();
    mm(fa1, fb1);
    spread();
    rdf(a, b, 3, fa1, fb1);
    burst();
    mm(fa0, fb0);
    spread();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragment reads of tile kt done before its stage can be refilled
    __builtin_amdgcn_sched_barrier(0);
    st = st + 1 == NSTAGE ? 0 : st + 1;
  }
  if (nk > 0) mm(fa1, fb1);  // slice 3 of the last tile (MFMA waves)

  // ---- epilogue: accumulator tile -> LDS (transposed to row-major) -> 8 columns per lane ----
  // in passes of 128 rows (one pass for the 128-row tiles): a [256][BN + 4] fp32 image does not fit the LDS at BN = 192,
  // and a pass is exactly one row of the per-128-row column-sum partials
  constexpr int LDE = BN + 4;
  constexpr int EPR = 128, NPASS = BM / EPR;  // rows per pass
  constexpr int WPP = WM / NPASS;             // wave rows (wm) per pass
  static_assert(BM % EPR == 0 && WM % NPASS == 0 && (WPP * TM * 32 == EPR), "whole wave rows per epilogue pass");
  float* smem = reinterpret_cast<float*>(smem_b);
  const bool split = gridDim.z > 1;
  float* C = p.C32 ? p.C32 + (long)blockIdx.z * p.slab_stride : nullptr;
  constexpr int C8 = BN / 8;
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    __syncthreads();
    if (!dma_wave && wm / WPP == pass) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = ((wm % WPP) * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            smem[row * LDE + (wn * TN + j) * 32 + li] = acc[i][j][r];
          }
    }
    __syncthreads();
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // a thread keeps ONE 8-column group and walks rows G apart (G = NT / C8 row groups; NT - G * C8 threads idle when C8
    // does not divide NT): its running sums are the column sums of its rows
    constexpr int G = NT / C8;
    const int c = (tid % C8) * 8;
#pragma unroll 2
    for (int r = tid / C8; r < EPR && tid < G * C8; r += G) {
      float v[8];
      *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(smem + r * LDE + c);
      *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(smem + r * LDE + c + 4);
      const long row = m0 + pass * EPR + r;
      const int col = n0 + c;
      if (!split) {
        if (p.bias) {
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + col), b1 = *reinterpret_cast<const f32x4*>(p.bias + col + 4);
#pragma unroll
          for (int t = 0; t < 4; ++t) { v[t] += b0[t]; v[4 + t] += b1[t]; }
        }
        if (p.epi == EPI_GELU) {
          bf16x8 pre;
#pragma unroll
          for (int t = 0; t < 8; ++t) pre[t] = (__bf16)v[t];
          *reinterpret_cast<bf16x8*>(p.aux16 + row * p.ldaux + col) = pre;
#pragma unroll
          for (int t = 0; t < 8; t += 2) {  // of the SAVED (rounded) pre-activation: fwd/bwd consistent
            const f32x2 y = gelu_fast2(f32x2{(float)pre[t], (float)pre[t + 1]});
            v[t] = y.x;
            v[t + 1] = y.y;
          }
        } else if (p.epi == EPI_DGELU) {
          const bf16x8 pre = *reinterpret_cast<const bf16x8*>(p.aux16 + row * p.ldaux + col);
#pragma unroll
          for (int t = 0; t < 8; t += 2) {
            const f32x2 y = gelu_fast_grad2(f32x2{(float)pre[t], (float)pre[t + 1]});
            v[t] *= y.x;
            v[t + 1] *= y.y;
          }
        }
        if (p.accumulate) {
          const f32x4 c0 = *reinterpret_cast<const f32x4*>(C + row * p.ldc32 + col), c1 = *reinterpret_cast<const f32x4*>(C + row * p.ldc32 + col + 4);
#pragma unroll
          for (int t = 0; t < 4; ++t) { v[t] += c0[t]; v[4 + t] += c1[t]; }
        }
      }
      if (C) {
        *reinterpret_cast<f32x4*>(C + row * p.ldc32 + col) = *reinterpret_cast<const f32x4*>(v);
        *reinterpret_cast<f32x4*>(C + row * p.ldc32 + col + 4) = *reinterpret_cast<const f32x4*>(v + 4);
      }
      if (p.C16 && !split) {
        bf16x8 o;
#pragma unroll
        for (int t = 0; t < 8; ++t) o[t] = (__bf16)v[t];
        *reinterpret_cast<bf16x8*>(p.C16 + row * p.ldc16 + col) = o;
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) cs[t] += v[t];
    }
    if (p.colpart && !split) {  // (uniform) per-pass column sums, combined over the G row groups in fixed order
      static_assert(G * BN <= EPR * LDE, "the partial sums reuse the epilogue image");
      __syncthreads();
      if (tid < G * C8) {
#pragma unroll
        for (int t = 0; t < 8; ++t) smem[(tid / C8) * BN + c + t] = cs[t];
      }
      __syncthreads();
      for (int cc = tid; cc < BN; cc += NT) {
        float sacc = 0.f;
#pragma unroll
        for (int gq = 0; gq < G; ++gq) sacc += smem[gq * BN + cc];
        p.colpart[(long)(m0 / EPR + pass) * p.N + n0 + cc] = sacc;
      }
    }
  }
}

// out[c] (+)= sum_r part[r][c], r < rows (<= a few hundred): the second stage of the epilogue column sums.
// Block = 32 columns x 8 row groups, combined in fixed order through LDS (deterministic).
__global__ __launch_bounds__(256) void colsum_small_kernel(const float* __restrict__ part, int rows, int cols, float* __restrict__ out,
                                                          int accumulate) {
  __shared__ float red[8][32];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  // eight loads in flight per thread (64 dependent round trips at 512 partial rows took 46 us; the sum order stays fixed)
  float s4[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < cols) {
    int r = rg;
    for (; r + 56 < rows; r += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(long)(r + 8 * u) * cols + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) s4[u & 3] += v[u];
    }
    for (; r < rows; r += 8) s4[0] += part[(long)r * cols + c];
  }
  const float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  red[rg][cl] = s;
  __syncthreads();
  if (rg == 0 && c < cols) {
    float t = red[0][cl];
#pragma unroll
    for (int i = 1; i < 8; ++i) t += red[i][cl];
    out[c] = accumulate ? out[c] + t : t;
  }
}

// out[r][c] = bf16(x[r][c]);  outT[c][r] = bf16(x[r][c]) (optional).  32x32 tiles through LDS for the transposed copy.
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, int ldx, __bf16* __restrict__ out, int ldo,
                                                       __bf16* __restrict__ outT, int ldt, int R, int C) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    float v = 0.f;
    if (r < R && c < C) {
      v = x[(long)r * ldx + c];
      if (out) out[(long)r * ldo + c] = (__bf16)v;
    }
    tile[ty + 8 * i][tx] = v;
  }
  if (!outT) return;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (r < R && c < C) outT[(long)c * ldt + r] = (__bf16)tile[tx][ty + 8 * i];
  }
}

static int g_bf16x_dma_waves = [] { const char* e = getenv("MTVAF_BF16X_DMA_WAVES"); return (e && atoi(e) == 0) ? 0 : 1; }();

// DMA waves pay where a product is at most two tiles per CU (measured, tools/bf16x_bench.py + bench.py: C3, 4096 token rows,
// 4252 -> 4668 sentences/s; with more tiles per CU the co-resident blocks already hide the requests, and the doubled block
// size costs: C4 at 8192 rows 5700 -> 5568 with the waves everywhere)
static bool bf16x_use_dma_waves(dim3 grid) { return g_bf16x_dma_waves && (long)grid.x * grid.z <= 512; }

template <int BM, int BN, int WM, int WN, bool A_KM, bool B_KM, int NSTAGE, bool KLIST = false, bool DW = false>
static int launch_x(const GemmArgsX& a, dim3 grid, hipStream_t st) {
  if constexpr (A_KM && B_KM && !KLIST && NSTAGE == 2) {  // (the k-tile list: 2-stage weight-gradient kernels only)
    if (a.klist) return launch_x<BM, BN, WM, WN, A_KM, B_KM, NSTAGE, true, DW>(a, grid, st);
  }
  if constexpr (WM * WN == 4 && !DW && NSTAGE <= 3) {  // four-wave tiles: the requests in waves of their own (MTVAF_BF16X_DMA_WAVES=0: off)
    if (bf16x_use_dma_waves(grid)) return launch_x<BM, BN, WM, WN, A_KM, B_KM, NSTAGE, KLIST, true>(a, grid, st);
  }
  size_t smem = (size_t)NSTAGE * (BM + BN) * 128;
  smem = std::max(smem, (size_t)128 * (BN + 4) * sizeof(float));  // epilogue image (one 128-row pass)
  auto kern = gemm_bf16x_kernel<BM, BN, WM, WN, A_KM, B_KM, NSTAGE, KLIST, DW>;
  static bool attr_set = false;
  if (smem > 64 * 1024 && !attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3((DW ? 2 : 1) * WM * WN * 64), smem, st, a);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

template <int BM, int BN, int WM, int WN, bool A_KM, bool B_KM>
static int launch_stages(const GemmArgsX& a, int stages, dim3 grid, hipStream_t st) {
  switch (stages) {
    case 3: return launch_x<BM, BN, WM, WN, A_KM, B_KM, 3>(a, grid, st);
    case 4: return launch_x<BM, BN, WM, WN, A_KM, B_KM, 4>(a, grid, st);
    case 5:
      if constexpr (5 * (BM + BN) * 128 <= 160 * 1024) return launch_x<BM, BN, WM, WN, A_KM, B_KM, 5>(a, grid, st);
      return MTVAF_ERR_SHAPE;
    default: return launch_x<BM, BN, WM, WN, A_KM, B_KM, 2>(a, grid, st);
  }
}

template <bool A_KM, bool B_KM>
static int launch_layout(const GemmArgsX& a, int bm, int bn, int stages, dim3 grid, hipStream_t st) {
  if (bm == 256 && bn == 128) {  // 8 waves (two per SIMD), one block per CU: half the L2 -> LDS bytes per flop of the 128-row tiles
    if constexpr (!A_KM) return launch_stages<256, 128, 4, 2, A_KM, B_KM>(a, stages > 3 ? 3 : stages, grid, st);
    return MTVAF_ERR_SHAPE;
  }
  if (bn == 192) {  // 8 waves, one block per CU: 0.57x the L2 -> LDS bytes per flop of the 128x128 tile (2-stage ring: 112 KB)
    if constexpr (!A_KM) return launch_x<256, 192, 4, 2, A_KM, B_KM, 2>(a, grid, st);
    return MTVAF_ERR_SHAPE;
  }
  if (bn == 128) return launch_stages<128, 128, 2, 2, A_KM, B_KM>(a, stages, grid, st);
  return launch_stages<128, 96, 4, 1, A_KM, B_KM>(a, stages, grid, st);
}


// ---- stream-K scratch: caller-owned, zero-initialised buffers attached per stream (mtvaf_streamk_attach) ----
// [0, 4096): flags (one 32-bit word per block, zero between launches) and the error word (last word); then one 256-KiB slab
// per block.  Never shared with the split-K workspaces, whose users scribble over their whole buffer.
struct StreamKScratch { hipStream_t stream; unsigned char* ptr; size_t bytes; };
static StreamKScratch g_sk[8];
static int g_sk_n = 0;
static std::mutex g_sk_mu;
constexpr size_t SK_HEADER = 4096, SK_SLAB = 256 * 256 * sizeof(float);

static bool streamk_lookup(hipStream_t stream, P256SK* sk, int* blocks) {
  std::lock_guard<std::mutex> lk(g_sk_mu);
  for (int i = 0; i < g_sk_n; ++i)
    if (g_sk[i].stream == stream) {
      const long b = std::min<long>({256L, (long)((g_sk[i].bytes - SK_HEADER) / SK_SLAB), (long)(SK_HEADER / 4 - 1)});
      if (b < 8) return false;
      sk->flags = reinterpret_cast<unsigned*>(g_sk[i].ptr);
      sk->err = sk->flags + SK_HEADER / 4 - 1;
      sk->slabs = reinterpret_cast<float*>(g_sk[i].ptr + SK_HEADER);
      *blocks = (int)b;
      return true;
    }
  return false;
}

// Pieces per tile for the in-launch-combine form of the 256x256 kernel: S divides the KT k-tiles of a tile's reduction, all
// tiles x S pieces are equal and run in ONE round of the `blocks` CUs (one contribution slab per piece).  Model calibrated with
// tools/p256_bench.py: 1.55 us per 256x256x64 step, ~3.5 us of block start + pipeline fill per piece, ~7 us for a piece's
// contribution (256-KiB slab out, flag) and the finisher's combine, ~3 us of epilogue.  -> microseconds, *S_out
static double p256_plan(long tiles, int KT, int blocks, int* S_out) {
  double best = 1e30;
  *S_out = 1;
  for (int S = 1; S <= KT; ++S) {
    if (KT % S || (S > 1 && (tiles * S > blocks || KT / S < 3))) continue;
    const double rounds = (double)((tiles * S + blocks - 1) / blocks);
    const double t = rounds * (KT / S * 1.55 + 3.5) + (S > 1 ? 7.0 : 0.0) + 3.0;
    if (t < best) { best = t; *S_out = S; }
  }
  return best;
}

}  // namespace mtvaf

using namespace mtvaf;

extern "C" {

// C[M,N] = opA[M,K] . opB[K,N] with bf16 operands (uint16 storage) and fp32 accumulation.
//   layout 0 (KC): reduction index contiguous -- A[m][k] (lda), B[n][k] (ldb);  layout 1 (KM): reduction index is the
//   row -- A[k][m], B[k][n].  Combinations of the path: (0,0) forward, (0,1) dX, (1,1) dW.
//   Results: C32 (fp32, ldc32) and / or C16 (bf16, ldc16); accumulate adds into C32.  epi: 0 none, 1 bias + erf-GELU
//   (pre-activation stored to aux16 as bf16; GELU evaluated on the stored value), 3 multiply by GELU'(aux16).
//   colpart [M/128][N] (optional, 128x128 tiles): per-tile column sums of the result for mtvaf_colsum_small.
//   allow_split: deterministic split-K (fp32 slabs in workspace + ordered reduction; fp32 result only, epi 0).
// Requirements (MTVAF_ERR_SHAPE / _ALIGN otherwise; no fallback): M % 128 == 0, K % 64 == 0, N % 128 == 0 or N % 96 == 0, leading dimensions % 8 == 0, 16-byte aligned pointers.  tile: 0 auto, 1 128x96,
// 2 128x128, 3 256x128 (8 waves; layout_a 0, M % 256 == 0, no colpart), 4 256x192 (8 waves; layout_a 0, M % 256 == 0, N % 192 == 0),
// 5 256x256 eight-phase (gemm_bf16p.hip; any layout pair, M % 256 == 0, N % 256 == 0).
// stages: 0 auto, 2 .. 5 (256x192: always 2).
static int gemm_bf16x_core(int layout_a, int layout_b, const void* A, int lda, const void* B, int ldb, float* C32, int ldc32,
                           void* C16, int ldc16, int M, int N, int K, const float* bias, int epi, void* aux16, int ldaux,
                           int accumulate, float* colpart, int allow_split, void* workspace, size_t workspace_bytes, int tile,
                           int splits, int stages, const int* klist, const int* kcnt, hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0) return MTVAF_ERR_SHAPE;
  if (!A || !B || (!C32 && !C16)) return MTVAF_ERR_ARG;
  if (layout_a < 0 || layout_a > 1 || layout_b < 0 || layout_b > 1 || (layout_a == 1 && layout_b == 0)) return MTVAF_ERR_ARG;
  if (epi != EPI_NONE && epi != EPI_GELU && epi != EPI_DGELU) return MTVAF_ERR_ARG;
  if ((epi == EPI_GELU || epi == EPI_DGELU) && !aux16) return MTVAF_ERR_ARG;
  if (accumulate && !C32) return MTVAF_ERR_ARG;
  if (M % 128 || K % 64 || (N % 96 && N % 128)) return MTVAF_ERR_SHAPE;
  // (tile addresses are a scalar base + a 32-bit per-lane byte offset of up to 256 rows x leading dimension x 2 bytes: ADVICE r4)
  if (lda >= (1 << 22) || ldb >= (1 << 22)) return MTVAF_ERR_SHAPE;
  if (lda % 8 || ldb % 8 || (C32 && ldc32 % 4) || (C16 && ldc16 % 8) || (aux16 && ldaux % 8)) return MTVAF_ERR_ALIGN;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C32 | (uintptr_t)C16 | (uintptr_t)bias | (uintptr_t)aux16 | (uintptr_t)colpart) & 15)
    return MTVAF_ERR_ALIGN;
  const bool can128 = N % 128 == 0, can96 = N % 96 == 0 && !colpart;
  const bool can256 = can128 && M % 256 == 0 && layout_a == 0 && !colpart;
  const bool can192 = N % 192 == 0 && M % 256 == 0 && layout_a == 0;
  const bool canp256 = M % 256 == 0 && N % 256 == 0;  // the eight-phase 256x256 kernel (gemm_bf16p.hip), all three layouts
  int bn = tile == 1 ? 96 : ((tile == 2 || tile == 3) ? 128 : (tile == 4 ? 192 : ((tile == 5 || tile == 6) ? 256 : 0)));
  int bm = (tile == 3 || tile == 4 || tile == 5 || tile == 6) ? 256 : 128;
  if ((bn == 96 && !can96) || (bn == 128 && !can128) || (bm == 256 && bn == 128 && !can256) || (bn == 192 && !can192) ||
      (bn == 256 && !canp256))
    return MTVAF_ERR_SHAPE;
  if (bn == 0) {
    const long t128 = can128 ? (long)(M / 128) * (N / 128) : 0;
    if (!can96) bn = 128;
    else if (!can128) bn = 96;
    else bn = (epi == EPI_GELU || epi == EPI_DGELU || t128 >= 512) ? 128 : 96;  // 96: more tiles for the 768-wide outputs
    // ... unless the 96-wide tiles spill into a second round of the CUs that the 128-wide ones avoid (round 5: the 768-wide
    // products of a packed bs-64 batch, 4864 rows: 304 tiles against 228 -- measured 36.3 -> 28.3 / 35.8 -> 27.4 / 28.3 -> 22.6 us
    // on FFN-2 forward / FFN-1 dX / QKV dX, tools/bf16x_bench.py 4864: one round of the 3-deep ring instead of two of the 2-deep)
    if (bn == 96 && can128 && (long)(M / 128) * (N / 96) > 256 && t128 <= 256) bn = 128;
    // big problems: the 256-row tile once it still gives every CU several tiles
    // (measured, tools/bf16x_bench.py at M = 65536: +6..10 % on the KC x KC products with the 3-deep ring; no gain with a
    // k-major B operand)
    if (can256 && layout_b == 0 && (long)(M / 256) * (N / 128) >= 1024) { bm = 256; bn = 128; }
    // wide outputs at one tile per CU: the 256x192 tile moves 0.57x the operand bytes of 128x128 through the L2 -> LDS
    // path (measured at M = 4096: QKV forward 32 -> 29 us, FFN-1 forward 39 -> 36, FFN-2 dX 45 -> 39; with two rounds of
    // tiles -- M = 8192 -- its single co-resident block per CU gives the gain back)
    const long t192 = can192 ? (long)(M / 256) * (N / 192) : 0;
    if (t192 >= 192 && t192 <= 256) { bm = 256; bn = 192; }
    // the eight-phase 256x256 kernel (gemm_bf16p.hip): 1.5 us per 256x256x64 step per CU in the loop (5.4 TFLOP/s per CU
    // against 2-3 of the rings above) but ~10 us of launch + pipeline fill + epilogue per tile round, and whole rounds of
    // 256 tiles (measured, tools/p256_bench.py): it wins once a forward / dX product has two rounds of tiles (M = 65536:
    // 0.78-1.19 PFLOP/s against 0.55-0.86), and on the weight-gradient products (long reductions over the tokens, few
    // output tiles) once tiles x splits fill the chip with at least 12 k-tiles per split
    const bool p256_epi = !(bias && (epi == EPI_DGELU || accumulate)) && !(colpart && epi != EPI_DGELU) && !(accumulate && (epi != EPI_NONE || C16)) &&
                          !(klist && kcnt);
    if (canp256 && p256_epi) {
      const long t256 = (long)(M / 256) * (N / 256);
      if (layout_a == 0) {
        if (t256 >= 512 && splits <= 1) { bm = 256; bn = 256; splits = 1; }
      } else if (splits <= 0 && allow_split && epi == EPI_NONE && C32 && !C16 && !colpart) {
        long sp = std::min<long>(32, std::max<long>(1, 256 / t256));
        while (sp > 1 && ((K / 64) / sp < 12 || (size_t)sp * M * N * sizeof(float) > workspace_bytes)) --sp;
        if (t256 * sp >= 160) { bm = 256; bn = 256; splits = (int)sp; }
      }
      // with a scratch attached to this stream (mtvaf_streamk_attach) the kernel can cut every tile's reduction into S equal
      // pieces that fill one round of the CUs and combine them inside the launch: priced against what the rings above reach on
      // these shapes (0.62 PFLOP/s, 0.78 at K >= 2048, 0.5 on weight gradients)
      if (bn != 256 && splits <= 1) {
        P256SK tmp = {};
        int blocks = 0, S = 1;
        if (streamk_lookup(stream, &tmp, &blocks)) {
          const double t_new = p256_plan(t256, K / 64, blocks, &S) + ((epi == EPI_GELU || epi == EPI_DGELU) ? 7.0 : 0.0);
          const double rate = layout_a == 1 ? 0.5e9 : (K >= 2048 ? 0.78e9 : 0.62e9);  // flop per microsecond
          const double t_old = 2.0 * M * N * K / rate + 3.0;
          if (S > 1 && t_new < 0.85 * t_old) { bm = 256; bn = 256; splits = 1; }
        }
      }
    }
  }
  const long tiles = (long)(M / bm) * (N / bn);
  // ---- the 256x256 kernel with its in-launch combine: S equal pieces per tile in one round of the CUs (tile 6 forces the
  // general stream-K cut: the step line in `blocks` equal runs wherever they fall) ----
  if ((bn == 256 && tile != 5) || tile == 6) {
    P256SK sk = {};
    int blocks = 0, S = 1;
    const int KT = K / 64;
    const bool have = streamk_lookup(stream, &sk, &blocks);
    if (tile == 6 && !have) return MTVAF_ERR_WORKSPACE;
    if (have) p256_plan(tiles, KT, blocks, &S);
    if (have && (tile == 6 || S > 1)) {
      GemmArgsX a = {};
      a.klist = a.kcnt = nullptr;
      a.A = static_cast<const __bf16*>(A); a.B = static_cast<const __bf16*>(B);
      a.C32 = C32; a.C16 = static_cast<__bf16*>(C16); a.bias = bias; a.aux16 = static_cast<__bf16*>(aux16); a.colpart = colpart;
      a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc32 = ldc32; a.ldc16 = ldc16; a.ldaux = ldaux;
      a.epi = epi; a.accumulate = accumulate; a.k_chunk = K; a.slab_stride = 0; a.tiles_n = N / 256;
      sk.nprob = 0; sk.KT = KT; sk.total = tiles * KT;
      const int key[8] = {300 + 1 + 4 * layout_a + 8 * layout_b + 16 + 64 + 128, layout_a, layout_b, 2, M, N, K, tile == 6 ? 0 : S};
      const int rec = prof_begin(key, stream);
      const int rc = launch_p256_streamk(a, sk, layout_a, layout_b, blocks, tile == 6 ? 0 : KT / S, stream);
      prof_end(rec, stream);
      return rc;
    }
  }
  const bool split_ok = allow_split && epi == EPI_NONE && C32 && !C16 && !colpart;
  if (splits <= 0) {
    splits = 1;
    if (split_ok && tiles < 384) splits = (int)std::min<long>(std::max<long>(512 / tiles, 1), 8);
  }
  if (!split_ok) splits = 1;
  while (splits > 1 && ((size_t)splits * M * N * sizeof(float) > workspace_bytes || (K / 64) / splits < (bn == 256 ? 1 : 4))) --splits;
  GemmArgsX a = {};
  a.klist = (klist && kcnt && layout_a == 1 && layout_b == 1) ? klist : nullptr;
  a.kcnt = a.klist ? kcnt : nullptr;
  a.A = static_cast<const __bf16*>(A);
  a.B = static_cast<const __bf16*>(B);
  a.C16 = static_cast<__bf16*>(C16);
  a.bias = bias;
  a.aux16 = static_cast<__bf16*>(aux16);
  a.colpart = colpart;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc16 = ldc16; a.ldaux = ldaux;
  a.epi = epi; a.accumulate = accumulate;
  int kc = (int)(((K / 64 + splits - 1) / splits) * 64);
  splits = (K + kc - 1) / kc;
  a.k_chunk = kc;
  if (splits > 1) {
    a.C32 = (float*)workspace; a.ldc32 = N; a.slab_stride = (long)M * N;
  } else {
    a.C32 = C32; a.ldc32 = ldc32; a.slab_stride = 0;
  }
  a.tiles_n = N / bn;
  // measured (tools/bf16x_bench.py): with 4-wave blocks two co-resident blocks beat the deeper ring on every shape of the
  // path; the 8-wave 256x128 block (one per CU) wants the 3-deep ring
  // ... WITHOUT the DMA waves.  With them (round 4) a product of at most one tile per CU takes the 3-deep ring: nothing
  // co-resides anyway, and two k-tiles in flight feed the MFMA waves that no longer stop to issue requests (M = 4096, 128x96:
  // FFN-2 forward 35.9 -> 25.6 us, FFN-1 dX 35.1 -> 25.4, QKV dX 27.9 -> 20.7; with several tiles per CU the two co-resident
  // blocks of the 2-deep ring stay ahead: FFN-1 forward 35.3 vs 46.6)
  if (stages < 2 || stages > 5) stages = bm == 256 ? 3 : ((g_bf16x_dma_waves && !a.klist && tiles * splits <= 256) ? 3 : 2);
  if (bn == 192) stages = 2;  // (the only ring that fits: 2 x 56 KB)
  if (bn == 256) { stages = 2; a.klist = a.kcnt = nullptr; }
  if (a.klist && stages != 2) a.klist = a.kcnt = nullptr;  // (list mode exists for the 2-stage kernels: otherwise reduce over everything)
  dim3 grid((unsigned)tiles, 1, (unsigned)splits);
  const int key[8] = {300 + (bn == 96 ? 0 : 1) + 2 * (stages >= 3) + 4 * layout_a + 8 * layout_b + 16 * (bm == 256) + 32 * (bn == 192) + 64 * (bn == 256), layout_a, layout_b, 2, M, N, K, splits};
  const int rec = prof_begin(key, stream);
  int rc;
  if (bn == 256) rc = launch_p256(a, layout_a, layout_b, grid, stream);
  else if (layout_a == 0 && layout_b == 0) rc = launch_layout<false, false>(a, bm, bn, stages, grid, stream);
  else if (layout_a == 0) rc = launch_layout<false, true>(a, bm, bn, stages, grid, stream);
  else rc = launch_layout<true, true>(a, bm, bn, stages, grid, stream);
  prof_end(rec, stream);
  if (rc != MTVAF_OK) return rc;
  if (splits > 1) return launch_splitk_reduce((const float*)workspace, splits, C32, M, N, ldc32, bias, accumulate, EPI_NONE, nullptr, 0, stream);
  return MTVAF_OK;
}

int mtvaf_gemm_bf16x(int layout_a, int layout_b, const void* A, int lda, const void* B, int ldb, float* C32, int ldc32,
                     void* C16, int ldc16, int M, int N, int K, const float* bias, int epi, void* aux16, int ldaux,
                     int accumulate, float* colpart, int allow_split, void* workspace, size_t workspace_bytes, int tile,
                     int splits, int stages, hipStream_t stream) {
  return gemm_bf16x_core(layout_a, layout_b, A, lda, B, ldb, C32, ldc32, C16, ldc16, M, N, K, bias, epi, aux16, ldaux, accumulate,
                         colpart, allow_split, workspace, workspace_bytes, tile, splits, stages, nullptr, nullptr, stream);
}

// mtvaf_gemm_bf16x for a weight-gradient product (layouts KM x KM) whose operand A is exactly zero outside the listed 64-row
// k-tiles of the reduction (token) axis: see mtvaf_gemm_f32_ktiles.  klist / kcnt: device int32 (mtvaf_build_ktiles, bk = 64).
int mtvaf_gemm_bf16x_ktiles(int layout_a, int layout_b, const void* A, int lda, const void* B, int ldb, float* C32, int ldc32,
                            void* C16, int ldc16, int M, int N, int K, const float* bias, int epi, void* aux16, int ldaux,
                            int accumulate, float* colpart, int allow_split, void* workspace, size_t workspace_bytes, int tile,
                            int splits, int stages, const int* klist, const int* kcnt, hipStream_t stream) {
  return gemm_bf16x_core(layout_a, layout_b, A, lda, B, ldb, C32, ldc32, C16, ldc16, M, N, K, bias, epi, aux16, ldaux, accumulate,
                         colpart, allow_split, workspace, workspace_bytes, tile, splits, stages, klist, kcnt, stream);
}

// Attach a caller-owned scratch to `stream` for the stream-K launches of the 256x256 kernel (gemm_bf16p.hip): at least
// mtvaf_streamk_scratch_bytes(blocks) bytes, 16-byte aligned, ZERO-INITIALISED by the caller (the kernel leaves the flag words
// zero behind every launch), alive until detached (ptr = NULL) or replaced.  Without a scratch the library uses the
// tile-per-block / split-K forms.
int mtvaf_streamk_attach(void* scratch, size_t bytes, hipStream_t stream) {
  std::lock_guard<std::mutex> lk(g_sk_mu);
  int slot = -1;
  for (int i = 0; i < g_sk_n; ++i)
    if (g_sk[i].stream == stream) slot = i;
  if (!scratch) {
    if (slot >= 0) g_sk[slot] = g_sk[--g_sk_n];
    return MTVAF_OK;
  }
  if (((uintptr_t)scratch & 15) || bytes < SK_HEADER + 8 * SK_SLAB) return MTVAF_ERR_WORKSPACE;
  if (slot < 0) {
    if (g_sk_n == 8) return MTVAF_ERR_WORKSPACE;
    slot = g_sk_n++;
  }
  g_sk[slot] = StreamKScratch{stream, static_cast<unsigned char*>(scratch), bytes};
  return MTVAF_OK;
}
int mtvaf_streamk_attached(hipStream_t stream) {
  P256SK sk = {};
  int blocks = 0;
  return streamk_lookup(stream, &sk, &blocks) ? blocks : 0;
}
size_t mtvaf_streamk_scratch_bytes(int blocks) { return SK_HEADER + (size_t)std::max(blocks, 8) * SK_SLAB; }

// Up to four weight-gradient products dW_i[M_i, N_i] = A_i^T . B_i (layouts KM x KM: A_i [K, M_i], B_i [K, N_i] bf16 row-major,
// fp32 results) that share the reduction length K -- the four products of one encoder layer -- in ONE stream-K launch of the
// 256x256 kernel: 36 + 36 + 27 + 9 output tiles cannot fill 256 CUs one product at a time.  Requires a scratch attached to
// `stream` (MTVAF_ERR_WORKSPACE otherwise), M_i % 256 == 0, N_i % 256 == 0, K % 64 == 0, leading dimensions % 8 == 0.
int mtvaf_gemm_bf16x_dw_group(int n, const void* const* A, const int* lda, const void* const* B, const int* ldb, float* const* C32,
                              const int* ldc32, const int* M, const int* N, int K, hipStream_t stream) {
  if (n < 1 || n > 4 || K <= 0 || K % 64) return MTVAF_ERR_SHAPE;
  P256SK sk = {};
  int blocks = 0;
  if (!streamk_lookup(stream, &sk, &blocks)) return MTVAF_ERR_WORKSPACE;
  long tiles = 0;
  for (int i = 0; i < n; ++i) {
    if (M[i] <= 0 || N[i] <= 0 || M[i] % 256 || N[i] % 256) return MTVAF_ERR_SHAPE;
    if (lda[i] >= (1 << 22) || ldb[i] >= (1 << 22)) return MTVAF_ERR_SHAPE;  // (32-bit lane offsets, as gemm_bf16x_core)
    if (lda[i] % 8 || ldb[i] % 8 || ldc32[i] % 4 || (((uintptr_t)A[i] | (uintptr_t)B[i] | (uintptr_t)C32[i]) & 15)) return MTVAF_ERR_ALIGN;
    sk.pr[i] = P256Prob{static_cast<const __bf16*>(A[i]), static_cast<const __bf16*>(B[i]), C32[i], lda[i], ldb[i], ldc32[i], N[i] / 256};
    sk.tile_begin[i] = (int)tiles;
    tiles += (long)(M[i] / 256) * (N[i] / 256);
  }
  sk.nprob = n; sk.KT = K / 64; sk.total = tiles * sk.KT;
  GemmArgsX a = {};
  a.A = sk.pr[0].A; a.B = sk.pr[0].B; a.C32 = sk.pr[0].C32;
  a.M = M[0]; a.N = N[0]; a.K = K; a.lda = lda[0]; a.ldb = ldb[0]; a.ldc32 = ldc32[0]; a.k_chunk = K; a.tiles_n = N[0] / 256;
  a.epi = EPI_NONE;
  long flop_m = 0;
  for (int i = 0; i < n; ++i) flop_m += (long)M[i] * N[i];
  const int key[8] = {300 + 1 + 4 + 8 + 16 + 64 + 128 + 256, 1, 1, 2, (int)(flop_m / 768), 768, K, n};  // (M x 768 x K: the group's flops)
  const int rec = prof_begin(key, stream);
  int S = 1;
  p256_plan(tiles, sk.KT, blocks, &S);
  const int rc = launch_p256_streamk(a, sk, 1, 1, blocks, sk.KT / S, stream);
  prof_end(rec, stream);
  return rc;
}

// out[c] (+)= sum over rows of part[rows][cols] (fixed order): finishes the epilogue column sums of mtvaf_gemm_bf16x
int mtvaf_colsum_small(const float* part, int rows, int cols, float* out, int accumulate, hipStream_t stream) {
  if (!part || !out || rows <= 0 || cols <= 0) return MTVAF_ERR_ARG;
  hipLaunchKernelGGL(colsum_small_kernel, dim3((cols + 31) / 32), dim3(256), 0, stream, part, rows, cols, out, accumulate);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

// out [R,C] bf16 (ld ldo) and / or outT [C,R] bf16 (ld ldt) from x [R,C] fp32 (ld ldx); either output may be NULL.
int mtvaf_cast_bf16(const float* x, int ldx, void* out, int ldo, void* outT, int ldt, int R, int C, hipStream_t stream) {
  if (R <= 0 || C <= 0 || !x || (!out && !outT)) return MTVAF_ERR_ARG;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, stream, x, ldx, static_cast<__bf16*>(out),
                     ldo, static_cast<__bf16*>(outT), ldt, R, C);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

}  // extern "C"
