// 256 x 256 bf16-operand GEMM tile of the mixed-precision mode: eight waves (2 x 4), BK = 64, 128 KiB of LDS, an
// EIGHT-PHASE k-loop (two k-tiles per iteration, four phases per k-tile) with the LDS-DMA prefetch kept in flight ACROSS
// the barriers behind a COUNTED s_waitcnt vmcnt -- the structure that keeps the matrix pipe fed at one workgroup per CU
// (the 128-row rings of gemm_bf16x.hip park their waves on vmcnt / the tile barrier: SQ_VALU_MFMA_BUSY 0.15-0.22).
// Same products, same row-major tensors and operand roles as gemm_bf16x.hip (KC: reduction index contiguous; KM: reduction
// index is the row, fragments by ds_read_b64_tr_b16), same epilogues.
//
// Work split.  Wave (wr, wc) of the 2 x 4 grid owns output rows wr*128 .. +127 and columns wc*64 .. +63 = 8 x 4
// accumulator blocks of v_mfma_f32_16x16x32_bf16, computed TRANSPOSED (the W fragment is the MFMA's A operand) so that a
// lane ends up with four CONSECUTIVE output columns of one row: the epilogue stores 16 / 8 bytes per lane straight from
// the registers and never touches the LDS.  A k-tile's work is cut into four quadrants (64 rows x 32 columns of the wave's
// block, 16 MFMAs each), one per phase.
//
// LDS.  Two buffers (even / odd k-tile) of four 16-KiB HALF-TILES: A-h0, A-h1, B-h0, B-h1.  Half h of A holds, for BOTH wave
// rows, the 64 output rows wr*128 + h*64 .. +63; half h of B, for all four wave columns, the 32 columns wc*64 + h*32 .. +31 --
// so that every half-tile is read in exactly ONE phase by every wave and is free to be refilled right after it:
//     phase 1 reads B-h0 + A-h0 (12 ds_read_b128), phase 2 B-h1 (4), phase 3 A-h1 (8), phase 4 nothing.
// KC half-tile image: 128 rows x 128 B, 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7)  (conflict-free for the
// 16x16x32 row reads: lane -> row l & 15, chunk 4*kh + (l >> 4)).  KM half-tile image: 64 k-rows x 256 B (128 outputs), chunk
// c of row r at c ^ (((r & 3) << 2) | ((r >> 2) & 3)), fragments by two ds_read_b64_tr_b16 (a 16-lane group fetches 4 k x 16
// outputs and each lane keeps ITS output's four k).  Both swizzles are applied to the per-lane SOURCE address of the DMA
// (the LDS side of global_load_lds is lane-linear) and again on the read.
//
// Schedule (tile t lives in buffer t & 1; every DMA is 2 instructions per wave = one half-tile per 8 waves):
//     phase 1: read B-h0, A-h0 of t | issue (t+1).A-h1                 | barrier, MFMA quadrant (0,0), barrier
//     phase 2: read B-h1            |                                  | barrier, MFMA (0,1), barrier
//     phase 3: read A-h1            | issue (t+2).B-h0, (t+2).A-h0     | barrier, MFMA (1,1), barrier
//     phase 4:                      | issue (t+2).B-h1, s_waitcnt vmcnt(6) | barrier, MFMA (1,0), barrier
// * RAW: tile t+1 is complete behind phase 4's vmcnt(6) (only the three half-tiles of t+2 stay in flight) AND the barrier
//   that follows it; it is first read in the NEXT phase.  LDS-DMA data is ordered for a ds_read by nothing else.
// * WAR: a half-tile is refilled no earlier than two phases after the phase that read it (B-h0 / A-h0: read in 1, refilled
//   in 3; B-h1: 2 -> 4; A-h1: 3 -> next 1), and every read is retired by the lgkmcnt(0) in front of its phase's MFMAs.
// * The two wave rows run staggered by one barrier (wr = 1 takes one extra barrier before the loop, wr = 0 one after it): each
//   SIMD holds one wave of either row, so while one of them issues its 16 MFMAs the other issues its LDS reads and DMA.
// The tail (last two k-tiles) issues less and drains vmcnt 6 -> 0; a block with a single k-tile skips the pipeline.
#include "gemm_bf16x.h"

#include <type_traits>

namespace mtvaf {

typedef float f32x4v __attribute__((ext_vector_type(4)));

#define P256_BAR()                      \
  do {                                  \
    __builtin_amdgcn_s_barrier();       \
    asm volatile("" ::: "memory");      \
  } while (0)
#define P256_SB() __builtin_amdgcn_sched_barrier(0)
#define P256_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <bool A_KM, bool B_KM>
__global__ __launch_bounds__(512) void gemm_bf16_p256_kernel(GemmArgsX p) {
  constexpr int HALF = 16384, BUF = 4 * HALF;  // bytes
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];  // the ONLY LDS object: 2 x 64 KiB

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int g = lane >> 4, l15 = lane & 15;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / p.tiles_n) * 256;
  const int n0 = (bid % p.tiles_n) * 256;
  const int kbeg = blockIdx.z * p.k_chunk;
  const int kend = min(p.K, kbeg + p.k_chunk);
  const int nk = (kend - kbeg) >> 6;

  // ---- DMA: per-lane source offsets (bytes from the block's operand base of k-tile 0, half 0) ----
  const unsigned char* Abase;
  const unsigned char* Bbase;
  long stepA, stepB;   // bytes per k-tile
  int hstepA, hstepB;  // bytes from half 0 to half 1
  unsigned voffA[2], voffB[2];
  if (!A_KM) {
    Abase = reinterpret_cast<const unsigned char*>(p.A + (long)m0 * p.lda + kbeg);
    stepA = 128;
    hstepA = 64 * p.lda * 2;
  } else {
    Abase = reinterpret_cast<const unsigned char*>(p.A + (long)kbeg * p.lda + m0);
    stepA = (long)64 * p.lda * 2;
    hstepA = 128;
  }
  if (!B_KM) {
    Bbase = reinterpret_cast<const unsigned char*>(p.B + (long)n0 * p.ldb + kbeg);
    stepB = 128;
    hstepB = 32 * p.ldb * 2;
  } else {
    Bbase = reinterpret_cast<const unsigned char*>(p.B + (long)kbeg * p.ldb + n0);
    stepB = (long)64 * p.ldb * 2;
    hstepB = 64;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = (wave * 2 + i) * 64 + lane;  // 16-byte position inside the half-tile image
    if (!A_KM) {
      const int r = f >> 3, cp = f & 7;
      voffA[i] = (unsigned)(((r >> 6) * 128 + (r & 63)) * p.lda * 2 + ((cp ^ ((r >> 1) & 7)) << 4));
    } else {
      const int r = f >> 4, cp = f & 15, o = (cp ^ km_swz(r)) * 8;  // o: first of the chunk's 8 outputs inside the half (0..127)
      voffA[i] = (unsigned)(r * p.lda * 2 + ((o >> 6) * 128 + (o & 63)) * 2);
    }
    if (!B_KM) {
      const int r = f >> 3, cp = f & 7;
      voffB[i] = (unsigned)(((r >> 5) * 64 + (r & 31)) * p.ldb * 2 + ((cp ^ ((r >> 1) & 7)) << 4));
    } else {
      const int r = f >> 4, cp = f & 15, o = (cp ^ km_swz(r)) * 8;
      voffB[i] = (unsigned)(r * p.ldb * 2 + ((o >> 5) * 64 + (o & 31)) * 2);
    }
  }
  unsigned char* const dma_dst = smem_b + wave * 2048;  // this wave's two 1-KiB pieces of a half-tile
  auto stage_A = [&](int buf, int h, int kt) {
    const unsigned char* s = Abase + (long)kt * stepA + h * hstepA;
    unsigned char* d = dma_dst + buf * BUF + h * HALF;
    glds16x(s + voffA[0], d);
    glds16x(s + voffA[1], d + 1024);
  };
  auto stage_B = [&](int buf, int h, int kt) {
    const unsigned char* s = Bbase + (long)kt * stepB + h * hstepB;
    unsigned char* d = dma_dst + buf * BUF + (2 + h) * HALF;
    glds16x(s + voffB[0], d);
    glds16x(s + voffB[1], d + 1024);
  };

  // ---- fragment read offsets (bytes inside a half-tile image) ----
  // KC: [kh] for block 0 of the wave's rows (+2048 per 16-row block); KM: [4 or 2 blocks][2 reads] (+8192 for kh = 1)
  int roA[A_KM ? 8 : 2], roB[B_KM ? 4 : 2];
  {
    const int s = l15 >> 1, q = l15 >> 2, pp = lane & 3;
    if (!A_KM) {
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) roA[kh] = (wr * 64 + l15) * 128 + ((((4 * kh) | g) ^ s) << 4);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int c = wr * 4 + i, row = 8 * g + 4 * jj + q;
          roA[i * 2 + jj] = 256 * row + 16 * ((2 * c + (pp >> 1)) ^ km_swz(row)) + 8 * (pp & 1);
        }
    }
    if (!B_KM) {
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) roB[kh] = (wc * 32 + l15) * 128 + ((((4 * kh) | g) ^ s) << 4);
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int c = wc * 2 + j, row = 8 * g + 4 * jj + q;
          roB[j * 2 + jj] = 256 * row + 16 * ((2 * c + (pp >> 1)) ^ km_swz(row)) + 8 * (pp & 1);
        }
    }
  }
  auto load_A = [&](bf16x8 (&F)[4][2], int buf, int h) {
    const unsigned char* a = smem_b + buf * BUF + h * HALF;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        if (!A_KM) F[i][kh] = *reinterpret_cast<const bf16x8*>(a + roA[kh] + i * 2048);
        else F[i][kh] = tr_read8(a + roA[i * 2] + kh * 8192, a + roA[i * 2 + 1] + kh * 8192);
      }
  };
  auto load_B = [&](bf16x8 (&F)[2][2], int buf, int h) {
    const unsigned char* b = smem_b + buf * BUF + (2 + h) * HALF;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        if (!B_KM) F[j][kh] = *reinterpret_cast<const bf16x8*>(b + roB[kh] + j * 2048);
        else F[j][kh] = tr_read8(b + roB[j * 2] + kh * 8192, b + roB[j * 2 + 1] + kh * 8192);
      }
  };

  f32x4v acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

  // 16 MFMAs of quadrant (mi, ni): D^T blocks -- the B fragment is the MFMA's A operand, so register r of a lane is output
  // column 16*J + 4*g + r of output row 16*I + (lane & 15)
#define P256_QUAD(mi, ni, FA_, FB_)                                                                             \
  do {                                                                                                          \
    __builtin_amdgcn_s_setprio(1);                                                                              \
    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                            \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                             \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                           \
          acc[(mi) * 4 + i][(ni) * 2 + j] =                                                                     \
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB_[j][kh], FA_[i][kh], acc[(mi) * 4 + i][(ni) * 2 + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                              \
  } while (0)

  // one k-tile (4 phases).  I1: issue (t+1).A-h1 in phase 1; I2: issue the three half-tiles of t+2 in phases 3 / 4;
  // WAIT: 6 (steady state), 0 (the last DMA of the block is in flight), -1 (nothing in flight)
#define P256_KTILE(I1, I2, WAIT)                                                  \
  do {                                                                            \
    const int cur = t & 1;                                                        \
    bf16x8 FA0[4][2], FA1[4][2], FB0[2][2], FB1[2][2];                            \
    /* phase 1 */                                                                 \
    load_B(FB0, cur, 0);                                                          \
    load_A(FA0, cur, 0);                                                          \
    if (I1) stage_A(cur ^ 1, 1, t + 1);                                           \
    P256_SB();                                                                    \
    P256_BAR();                                                                   \
    P256_LGKM0();                                                                 \
    P256_SB();                                                                    \
    P256_QUAD(0, 0, FA0, FB0);                                                    \
    P256_SB();                                                                    \
    P256_BAR();                                                                   \
    /* phase 2 */                                                                 \
    load_B(FB1, cur, 1);                                                          \
    P256_SB();                                                                    \
    P256_BAR();                                                                   \
    P256_LGKM0();                                                                 \
    P256_SB();                                                                    \
    P256_QUAD(0, 1, FA0, FB1);                                                    \
    P256_SB();                                                                    \
    P256_BAR();                                                                   \
    /* phase 3 */                                                                 \
    load_A(FA1, cur, 1);                                                          \
    if (I2) {                                                                     \
      stage_B(cur, 0, t + 2);                                                     \
      stage_A(cur, 0, t + 2);                                                     \
    }                                                                             \
    P256_SB();                                                                    \
    P256_BAR();                                                                   \
    P256_LGKM0();                                                                 \
    P256_SB();                                                                    \
    P256_QUAD(1, 1, FA1, FB1);                                                    \
    P256_SB();                                                                    \
    P256_BAR();                                                                   \
    /* phase 4 */                                                                 \
    if (I2) stage_B(cur, 1, t + 2);                                               \
    if ((WAIT) == 6) wait_vm<6>();                                                \
    else if ((WAIT) == 0) wait_vm<0>();                                           \
    P256_SB();                                                                    \
    P256_BAR();                                                                   \
    P256_SB();                                                                    \
    P256_QUAD(1, 0, FA1, FB0);                                                    \
    P256_SB();                                                                    \
    P256_BAR();                                                                   \
    P256_SB();                                                                    \
  } while (0)

  // ---- prologue: all of tile 0, three half-tiles of tile 1 ----
  stage_B(0, 0, 0);
  stage_A(0, 0, 0);
  stage_B(0, 1, 0);
  stage_A(0, 1, 0);
  if (nk > 1) {
    stage_B(1, 0, 1);
    stage_A(1, 0, 1);
    stage_B(1, 1, 1);
    wait_vm<6>();
  } else {
    wait_vm<0>();
  }
  P256_SB();
  P256_BAR();
  if (wr == 1) P256_BAR();  // stagger: this wave row runs one barrier behind the other (balanced after the loop)
  P256_SB();

  int t = 0;
  for (; t + 2 < nk; ++t) P256_KTILE(1, 1, 6);
  if (nk >= 2) {
    P256_KTILE(1, 0, 0);
    ++t;
  }
  P256_KTILE(0, 0, -1);
  if (wr == 0) P256_BAR();
#undef P256_KTILE
#undef P256_QUAD

  // ---- epilogue: straight from the registers (lane: row 16*I + l15, columns 16*J + 4*g .. +3 of the wave's block) ----
  // One straight-line body per epilogue KIND, chosen once: a runtime `if (epi == ..)` around each of the 32 loads of the
  // DGELU / accumulate forms makes hipcc branch around every load and drain vmcnt per element (32 dependent round trips:
  // measured 26 us per tile); inside a body the loads of half the tile are issued together, then consumed.
  typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
  const bool split = gridDim.z > 1;
  float* const C = p.C32 ? p.C32 + (long)blockIdx.z * p.slab_stride : nullptr;
  __bf16* const C16 = split ? nullptr : p.C16;
  const int colw = n0 + wc * 64 + 4 * g;
  const long row0 = m0 + wr * 128 + l15;
  // what the path combines (checked by the launcher): bias with the plain / GELU forms only, column sums with DGELU only
  enum { K_RAW = 0, K_PLAIN = 1, K_GELU = 2, K_DGELU = 3, K_ACC = 4 };
  auto body = [&](auto kind_c) {
    constexpr int KIND = decltype(kind_c)::value;
    constexpr bool BIAS = KIND == K_PLAIN || KIND == K_GELU, SUMS = KIND == K_DGELU;
    f32x4v cs[4], bias4[4];
#pragma unroll
    for (int J = 0; J < 4; ++J) {
      cs[J] = f32x4v{0.f, 0.f, 0.f, 0.f};
      bias4[J] = (BIAS && p.bias) ? *reinterpret_cast<const f32x4v*>(p.bias + colw + 16 * J) : f32x4v{0.f, 0.f, 0.f, 0.f};
    }
    constexpr int NB = KIND == K_ACC ? 2 : 4;  // 16-row blocks per batch of loads: 32 registers in flight either way
#pragma unroll
    for (int hI = 0; hI < 8 / NB; ++hI) {
      bf16x4v pre[NB][4];
      f32x4v old[NB][4];
      if (KIND == K_DGELU) {
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
          for (int J = 0; J < 4; ++J)
            pre[i][J] = *reinterpret_cast<const bf16x4v*>(p.aux16 + (row0 + 16 * (NB * hI + i)) * p.ldaux + colw + 16 * J);
      }
      if (KIND == K_ACC) {
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
          for (int J = 0; J < 4; ++J)
            old[i][J] = *reinterpret_cast<const f32x4v*>(C + (row0 + 16 * (NB * hI + i)) * p.ldc32 + colw + 16 * J);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int I = NB * hI + i;
        const long row = row0 + 16 * I;
#pragma unroll
        for (int J = 0; J < 4; ++J) {
          const int col = colw + 16 * J;
          f32x4v v = acc[I][J];
          if (BIAS) v += bias4[J];
          if (KIND == K_GELU) {
            bf16x4v pr;
#pragma unroll
            for (int e = 0; e < 4; ++e) pr[e] = (__bf16)v[e];
            *reinterpret_cast<bf16x4v*>(p.aux16 + row * p.ldaux + col) = pr;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf((float)pr[e]);  // of the SAVED (rounded) pre-activation: fwd/bwd consistent
          } else if (KIND == K_DGELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= gelu_erf_grad((float)pre[i][J][e]);
          } else if (KIND == K_ACC) {
            v += old[i][J];
          }
          if (C) *reinterpret_cast<f32x4v*>(C + row * p.ldc32 + col) = v;
          if (C16) {
            bf16x4v o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
            *reinterpret_cast<bf16x4v*>(C16 + row * p.ldc16 + col) = o;
          }
          if (SUMS) cs[J] += v;
        }
      }
    }
    if (SUMS && p.colpart) {  // column sums of this wave's 128 rows: over the 8 row blocks (above), then over the 16 row lanes
#pragma unroll
      for (int J = 0; J < 4; ++J) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float sum = cs[J][e];
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o, 64);
          cs[J][e] = sum;
        }
        if (l15 == 0) *reinterpret_cast<f32x4v*>(p.colpart + (long)(m0 / 128 + wr) * p.N + colw + 16 * J) = cs[J];
      }
    }
  };
  if (split) body(std::integral_constant<int, K_RAW>{});
  else if (p.epi == EPI_GELU) body(std::integral_constant<int, K_GELU>{});
  else if (p.epi == EPI_DGELU) body(std::integral_constant<int, K_DGELU>{});
  else if (p.accumulate) body(std::integral_constant<int, K_ACC>{});
  else body(std::integral_constant<int, K_PLAIN>{});
}

template <bool A_KM, bool B_KM>
static int launch_p256_t(const GemmArgsX& a, dim3 grid, hipStream_t st) {
  constexpr size_t smem = 128 * 1024;
  auto kern = gemm_bf16_p256_kernel<A_KM, B_KM>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(512), smem, st, a);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

// 256 x 256 tile, any of the three operand-layout pairs of the path.  Requires M % 256 == 0, N % 256 == 0, K % 64 == 0 and
// whole 64-element k-tiles per split (checked by the caller, gemm_bf16x_core).
int launch_p256(const GemmArgsX& a, int layout_a, int layout_b, dim3 grid, hipStream_t st) {
  // the epilogue bodies of the kernel cover the combinations the path uses
  if (a.bias && (a.epi == EPI_DGELU || a.accumulate)) return MTVAF_ERR_ARG;
  if (a.colpart && a.epi != EPI_DGELU) return MTVAF_ERR_ARG;
  if (a.accumulate && a.epi != EPI_NONE) return MTVAF_ERR_ARG;
  if (layout_a == 0 && layout_b == 0) return launch_p256_t<false, false>(a, grid, st);
  if (layout_a == 0) return launch_p256_t<false, true>(a, grid, st);
  return launch_p256_t<true, true>(a, grid, st);
}

}  // namespace mtvaf
