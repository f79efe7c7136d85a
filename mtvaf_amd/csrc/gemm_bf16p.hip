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
//
// STREAM-K mode (template flag SK; launch_p256_streamk).  The products of the path are small against the chip -- a
// [8192 x 3072] output is 384 tiles for 256 CUs (1.5 rounds), the four weight gradients of a layer are 36 + 36 + 27 + 9 tiles
// with reductions of 64 .. 1024 k-tiles -- so a tile-per-block launch leaves most CUs idle for part of it.  In this mode the
// unit of work is the k-tile STEP: all steps of all output tiles (of up to four products that share a launch: the weight
// gradients of one layer) are laid out tile after tile and cut into RUNS of W steps; a run may begin in the middle of a
// tile's reduction and end in the middle of another's, and every part of a run inside one tile is a PIECE -- one block.
//   * a piece that starts at k > 0 (always the FIRST piece of its run) is a CONTRIBUTION: its accumulators go to the run's
//     slab (256 KiB, in the register order of the kernel itself: lane-contiguous 16-byte stores, and the finisher's lanes
//     read back exactly what the same lanes of the contributor wrote), then the slab is published: every wave drains its
//     stores, the block's barrier, one lane's agent-scope release fence, a second drain, a relaxed agent-scope flag store;
//   * the piece that holds a tile's k = 0 is its FINISHER (always the LAST piece of its run): one lane polls the flags of
//     the following runs (relaxed, bounded), one agent-scope acquire fence + drain + the block's barrier, then every wave
//     adds the slabs to its accumulators in run order (= k order: the result depends on the cut, never on arrival order)
//     and runs the ordinary epilogue; it resets the flags it consumed, so a zero-initialised flag array stays zero
//     between launches (also under HIP-graph replay).
// No piece waits before it has published, and a finisher waits only for first pieces of LATER runs, which wait for nothing:
// no cycle, whatever the dispatch order or residency (runs are dispatched in reverse so that contributions usually run
// first).  The planner cuts every tile into S equal pieces that fill ONE round of the CUs (W = KT / S); W = ceil(total / 256)
// wherever it falls is the general cut (tile 6: tested, not planned -- a run's unequal pieces are separate blocks, and a
// persistent loop over them compiled to spills around the k-loop: DESIGN.md section 4.1d).  Slabs and flags live in a
// caller-owned scratch attached to the stream (mtvaf_streamk_attach), never shared with the split-K workspaces.
#include "gemm_bf16x.h"

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <type_traits>

namespace mtvaf {

typedef float f32x4v __attribute__((ext_vector_type(4)));

// MTVAF_P256_REBALANCE (compile-time A/B, a bit mask; where the 8 LDS-DMA requests of a k-tile sit among its four phases):
//   bit 0: (t+2).A-h0 is requested in phase 4 instead of phase 3 (phase 3: 8 fragment reads + 2 requests, phase 4: 4 requests;
//          0: 8 reads + 4 requests against 2 requests).  A-h0 of tile t was last read in phase 1;
//   bit 1: (t+1).A-h1 is requested in phase 2 instead of phase 1 (phase 1: 12 reads, phase 2: 4 reads + 2 requests).
// Request ORDER is unchanged (A-h1, then the three half-tiles of t+2), so the counted waits keep their meaning.
// Measured (round 6, tools/p256_bench.py 4864 38912, one box, profiles/r06_p256_rebalance.txt): mask 0 / 1 / 2 / 3 -> a layer's twelve
// products on this kernel at 38912 rows 2180 / 2152 / 2098 / 2113 us, the grouped weight gradients 666.7 / 663.5 / 656.8 / 659.5 us
// (4864 rows: 98.3 / 94.0 / 92.6 / 96.8): 2 is the default.
#ifndef MTVAF_P256_REBALANCE
#define MTVAF_P256_REBALANCE 2
#endif
#define P256_BAR()                      \
  do {                                  \
    __builtin_amdgcn_s_barrier();       \
    asm volatile("" ::: "memory");      \
  } while (0)
#define P256_SB() __builtin_amdgcn_sched_barrier(0)
#define P256_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

typedef __attribute__((address_space(1))) unsigned gu32;

template <bool A_KM, bool B_KM, bool SK>
__global__ __launch_bounds__(512) void gemm_bf16_p256_kernel(GemmArgsX p, P256SK sk) {
  constexpr int HALF = 16384, BUF = 4 * HALF;  // bytes
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];  // the ONLY LDS object: 2 x 64 KiB

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int g = lane >> 4, l15 = lane & 15;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);  // (stream-K: consecutive logical blocks -- which share tiles -- sit on one XCD)

  // ---- the piece of work in hand: operands, output tile, k-range (bound by `bind`) ----
  const unsigned char* Abase;
  const unsigned char* Bbase;
  long stepA, stepB;   // bytes per k-tile
  int hstepA, hstepB;  // bytes from half 0 to half 1
  unsigned voffA[2], voffB[2];
  int m0, n0, nk;
  float* Cout;         // fp32 result (or this split's slab)
  int ldc32;
  // DMA: per-lane source offsets (bytes from the piece's operand base of its first k-tile, half 0)
  auto bind = [&](const __bf16* A, const __bf16* B, int lda, int ldb, int m0_, int n0_, int kbeg, int nk_) {
    m0 = m0_; n0 = n0_; nk = nk_;
    if (!A_KM) {
      Abase = reinterpret_cast<const unsigned char*>(A + (long)m0 * lda + kbeg);
      stepA = 128;
      hstepA = 64 * lda * 2;
    } else {
      Abase = reinterpret_cast<const unsigned char*>(A + (long)kbeg * lda + m0);
      stepA = (long)64 * lda * 2;
      hstepA = 128;
    }
    if (!B_KM) {
      Bbase = reinterpret_cast<const unsigned char*>(B + (long)n0 * ldb + kbeg);
      stepB = 128;
      hstepB = 32 * ldb * 2;
    } else {
      Bbase = reinterpret_cast<const unsigned char*>(B + (long)kbeg * ldb + n0);
      stepB = (long)64 * ldb * 2;
      hstepB = 64;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = (wave * 2 + i) * 64 + lane;  // 16-byte position inside the half-tile image
      if (!A_KM) {
        const int r = f >> 3, cp = f & 7;
        voffA[i] = (unsigned)(((r >> 6) * 128 + (r & 63)) * lda * 2 + ((cp ^ ((r >> 1) & 7)) << 4));
      } else {
        const int r = f >> 4, cp = f & 15, o = (cp ^ km_swz(r)) * 8;  // o: first of the chunk's 8 outputs inside the half (0..127)
        voffA[i] = (unsigned)(r * lda * 2 + ((o >> 6) * 128 + (o & 63)) * 2);
      }
      if (!B_KM) {
        const int r = f >> 3, cp = f & 7;
        voffB[i] = (unsigned)(((r >> 5) * 64 + (r & 31)) * ldb * 2 + ((cp ^ ((r >> 1) & 7)) << 4));
      } else {
        const int r = f >> 4, cp = f & 15, o = (cp ^ km_swz(r)) * 8;
        voffB[i] = (unsigned)(r * ldb * 2 + ((o >> 5) * 64 + (o & 31)) * 2);
      }
    }
  };
  unsigned char* const dma_dst = smem_b + wave * 2048;  // this wave's two 1-KiB pieces of a half-tile
  auto stage_A = [&](int buf, int h, int kt) {
    const unsigned char* s = Abase + (long)kt * stepA + h * hstepA;
    unsigned char* d = dma_dst + buf * BUF + h * HALF;
    unsigned o0 = voffA[0], o1 = voffA[1];
    asm("" : "+v"(o0), "+v"(o1));  // (opaque: the requests stay scalar base + 32-bit lane offset instead of hoisted 64-bit pointers)
    glds16x(s + o0, d);
    glds16x(s + o1, d + 1024);
  };
  auto stage_B = [&](int buf, int h, int kt) {
    const unsigned char* s = Bbase + (long)kt * stepB + h * hstepB;
    unsigned char* d = dma_dst + buf * BUF + (2 + h) * HALF;
    unsigned o0 = voffB[0], o1 = voffB[1];
    asm("" : "+v"(o0), "+v"(o1));
    glds16x(s + o0, d);
    glds16x(s + o1, d + 1024);
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

  // one k-tile (4 phases).  I1: issue (t+1).A-h1 in phase 1; I2: issue the three half-tiles of t+2 in phases 3 / 4 and leave
  // them in flight (vmcnt 6), else drain (vmcnt 0: the last DMA of the piece, or nothing, is in flight).  Wave-uniform
  // run-time flags, ONE copy of the body: with the tail peeled into two further copies the three exits of the loop left the
  // accumulators in different registers and the allocator permuted ~120 of them through scratch after every k-loop
#define P256_KTILE(I1, I2)                                                        \
  do {                                                                            \
    const int cur = t & 1;                                                        \
    bf16x8 FA0[4][2], FA1[4][2], FB0[2][2], FB1[2][2];                            \
    /* phase 1 */                                                                 \
    load_B(FB0, cur, 0);                                                          \
    load_A(FA0, cur, 0);                                                          \
    if (I1 && !(MTVAF_P256_REBALANCE & 2)) stage_A(cur ^ 1, 1, t + 1);            \
    P256_SB();                                                                    \
    P256_BAR();                                                                   \
    P256_LGKM0();                                                                 \
    P256_SB();                                                                    \
    P256_QUAD(0, 0, FA0, FB0);                                                    \
    P256_SB();                                                                    \
    P256_BAR();                                                                   \
    /* phase 2 */                                                                 \
    load_B(FB1, cur, 1);                                                          \
    if (I1 && (MTVAF_P256_REBALANCE & 2)) stage_A(cur ^ 1, 1, t + 1);             \
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
      if (!(MTVAF_P256_REBALANCE & 1)) stage_A(cur, 0, t + 2);                    \
    }                                                                             \
    P256_SB();                                                                    \
    P256_BAR();                                                                   \
    P256_LGKM0();                                                                 \
    P256_SB();                                                                    \
    P256_QUAD(1, 1, FA1, FB1);                                                    \
    P256_SB();                                                                    \
    P256_BAR();                                                                   \
    /* phase 4 */                                                                 \
    if (I2) {                                                                     \
      if (MTVAF_P256_REBALANCE & 1) stage_A(cur, 0, t + 2);                       \
      stage_B(cur, 1, t + 2);                                                     \
      wait_vm<6>();                                                               \
    } else {                                                                      \
      wait_vm<0>();                                                               \
    }                                                                             \
    P256_SB();                                                                    \
    P256_BAR();                                                                   \
    P256_SB();                                                                    \
    P256_QUAD(1, 0, FA1, FB0);                                                    \
    P256_SB();                                                                    \
    P256_BAR();                                                                   \
    P256_SB();                                                                    \
  } while (0)

  // ---- the k-loop of the bound piece: accumulators <- sum over its nk k-tiles ----
  auto mainloop = [&]() __attribute__((always_inline)) {
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  // prologue: all of tile 0, three half-tiles of tile 1
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

  for (int t = 0; t < nk; ++t) {
    const bool i1 = t + 1 < nk, i2 = t + 2 < nk;
    P256_KTILE(i1, i2);
  }
  if (wr == 0) P256_BAR();
  };

  // ---- epilogue.  Two forms, chosen once per block (one straight-line body per KIND: a runtime `if (epi == ..)` around each
  // load made hipcc drain vmcnt per element).
  //  * fp32-only results (split-K / stream-K slabs, plain fp32, accumulate): straight from the registers -- the transposed
  //    MFMA leaves a lane with four consecutive columns of row 16*I + l15: 16-byte stores, 64 bytes per row per instruction.
  //  * anything with a bf16 array (bf16 result, GELU's saved pre-activation, GELU' reading it): through the LDS (free once the
  //    k-loop is over: eight wave-private 16-KiB regions) into a ROW layout -- lane (rl = lane >> 4, cl = lane & 15) holds the
  //    four consecutive columns 4*cl .. +3 of row 4*s + rl, s = 0..15 per pass of 64 rows -- so that one instruction of the wave
  //    covers FOUR WHOLE ROWS of its 64 columns (bf16: one full 128-byte line per row).  Straight from the MFMA layout a bf16
  //    store instruction touched 16 lines with 32 bytes each and the tile's stores took 8 us against a 17-us k-loop at
  //    K = 768 (measured: the same launches without their stores; skewing the blocks' start times changed nothing -- the cost
  //    sits in each CU's own store path, not in the HBM burst); in the row layout 4.6 us.  fp32 stores gained nothing from
  //    it (64 bytes per row already), hence the first form.  fp32 image of a pass: 64 rows x 256 B, 16-byte chunk c of row r at
  //    c ^ (r & 15) (conflict-free both ways).  Everything element-wise happens in the row layout; GELU / GELU' by the
  //    one-exponential form of common.h (the A&S erf cost 10 us of VALU time per tile: 128 evaluations per lane).
  typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
  // what the path combines (checked by the launcher): bias with the plain / GELU forms only, column sums with DGELU only
  enum { K_RAW = 0, K_PLAIN = 1, K_GELU = 2, K_DGELU = 3, K_ACC = 4 };
  auto epilogue = [&](bool raw, __bf16* C16) __attribute__((always_inline)) {
    unsigned char* const Cb = reinterpret_cast<unsigned char*>(Cout);
    unsigned char* const C16b = reinterpret_cast<unsigned char*>(C16);
    unsigned char* const auxb = reinterpret_cast<unsigned char*>(p.aux16);
    // addresses = wave-uniform base (advanced on the scalar unit) + ONE 32-bit per-lane offset per array: per-element 64-bit
    // address arithmetic in vector registers pushed the epilogue into spills
    auto direct = [&](auto kind_c) __attribute__((always_inline)) {
      constexpr int KIND = decltype(kind_c)::value;
      const int colw = n0 + wc * 64 + 4 * g;
      const int rowl = m0 + wr * 128 + l15;
      const unsigned vo32 = (unsigned)rowl * (unsigned)ldc32 * 4u + (unsigned)colw * 4u;
      const long rs32 = (long)ldc32 * 64;  // bytes per 16 rows
      f32x4v bias4[4];
#pragma unroll
      for (int J = 0; J < 4; ++J)
        bias4[J] = (KIND == K_PLAIN && p.bias) ? *reinterpret_cast<const f32x4v*>(p.bias + colw + 16 * J) : f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int hI = 0; hI < 4; ++hI) {
        f32x4v old[2][4];
        if (KIND == K_ACC) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int J = 0; J < 4; ++J) old[i][J] = *reinterpret_cast<const f32x4v*>(Cb + (2 * hI + i) * rs32 + 64 * J + vo32);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int I = 2 * hI + i;
#pragma unroll
          for (int J = 0; J < 4; ++J) {
            f32x4v v = acc[I][J];
            if (KIND == K_PLAIN) v += bias4[J];
            if (KIND == K_ACC) v += old[i][J];
            *reinterpret_cast<f32x4v*>(Cb + I * rs32 + 64 * J + vo32) = v;
          }
        }
      }
    };
    auto staged = [&](auto kind_c) __attribute__((always_inline)) {
      constexpr int KIND = decltype(kind_c)::value;
      constexpr bool BIAS = KIND == K_PLAIN || KIND == K_GELU, SUMS = KIND == K_DGELU;
      P256_BAR();  // every wave has read its last operand fragments: the tile buffers become staging space
      unsigned char* const stg = smem_b + wave * 16384;
      const int rl = lane >> 4, cl = lane & 15;
      const unsigned wr32 = (unsigned)(l15 * 256 + (((l15 & 12) | (g ^ (l15 & 3))) << 4));  // MFMA layout: block (i, J) at i*4096 + (wr32 ^ (J << 6))
      const unsigned rd32 = (unsigned)(rl * 256 + ((cl ^ rl) << 4));                        // row layout: step s at s*1024 + (rd32 ^ ((s & 3) << 6))
      const int colw = n0 + wc * 64 + 4 * cl;
      const int rowl = m0 + wr * 128 + rl;
      const unsigned vo32 = (unsigned)rowl * (unsigned)ldc32 * 4u + (unsigned)colw * 4u;
      const unsigned vo16 = (unsigned)rowl * (unsigned)p.ldc16 * 2u + (unsigned)colw * 2u;
      const unsigned voax = (unsigned)rowl * (unsigned)p.ldaux * 2u + (unsigned)colw * 2u;
      const long rs32 = (long)ldc32 * 16, rs16 = (long)p.ldc16 * 8, rsax = (long)p.ldaux * 8;  // bytes per 4 rows
      f32x4v cs = f32x4v{0.f, 0.f, 0.f, 0.f};
      const f32x4v bias4 = (BIAS && p.bias) ? *reinterpret_cast<const f32x4v*>(p.bias + colw) : f32x4v{0.f, 0.f, 0.f, 0.f};
      // GELU': the saved pre-activations of a batch of 8 steps are requested one batch AHEAD (behind the LDS round trip and
      // the arithmetic of the batch before), never waited for right after their issue
      bf16x4v pre[2][8];
      if (KIND == K_DGELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) pre[0][e] = *reinterpret_cast<const bf16x4v*>(auxb + e * rsax + voax);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int J = 0; J < 4; ++J) *reinterpret_cast<f32x4v*>(stg + i * 4096 + (wr32 ^ (unsigned)(J << 6))) = acc[4 * h + i][J];
        P256_LGKM0();
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
          const int bt = h * 2 + sb;  // batch 0..3
          f32x4v v[8];
          if (KIND == K_DGELU && bt < 3) {
#pragma unroll
            for (int e = 0; e < 8; ++e) pre[(bt + 1) & 1][e] = *reinterpret_cast<const bf16x4v*>(auxb + ((bt + 1) * 8 + e) * rsax + voax);
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<const f32x4v*>(stg + (sb * 8 + e) * 1024 + (rd32 ^ (unsigned)((e & 3) << 6)));
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int s = bt * 8 + e;
            f32x4v x = v[e];
            if (BIAS) x += bias4;
            if (KIND == K_GELU) {
              bf16x4v pr;
#pragma unroll
              for (int q = 0; q < 4; ++q) pr[q] = (__bf16)x[q];
              *reinterpret_cast<bf16x4v*>(auxb + s * rsax + voax) = pr;
              // of the SAVED (rounded) pre-activation: fwd/bwd consistent
              const f32x2 lo = gelu_fast2(f32x2{(float)pr[0], (float)pr[1]}), hi = gelu_fast2(f32x2{(float)pr[2], (float)pr[3]});
              x = f32x4v{lo.x, lo.y, hi.x, hi.y};
            } else if (KIND == K_DGELU) {
              const bf16x4v pq = pre[bt & 1][e];
              const f32x2 lo = gelu_fast_grad2(f32x2{(float)pq[0], (float)pq[1]});
              const f32x2 hi = gelu_fast_grad2(f32x2{(float)pq[2], (float)pq[3]});
              x *= f32x4v{lo.x, lo.y, hi.x, hi.y};
            }
            if (Cb) *reinterpret_cast<f32x4v*>(Cb + s * rs32 + vo32) = x;
            if (C16b) {
              bf16x4v o;
#pragma unroll
              for (int q = 0; q < 4; ++q) o[q] = (__bf16)x[q];
              *reinterpret_cast<bf16x4v*>(C16b + s * rs16 + vo16) = o;
            }
            if (SUMS) cs += x;
          }
        }
        P256_LGKM0();  // (the next pass overwrites the image)
      }
      if (SUMS && p.colpart) {  // column sums of this wave's 128 rows: over the 32 steps (above), then over the four row lanes
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float sum = cs[q];
          sum += __shfl_xor(sum, 16, 64);
          sum += __shfl_xor(sum, 32, 64);
          cs[q] = sum;
        }
        if (rl == 0) *reinterpret_cast<f32x4v*>(p.colpart + (long)(m0 / 128 + wr) * p.N + colw) = cs;
      }
    };
    if (raw) direct(std::integral_constant<int, K_RAW>{});
    else if (p.epi == EPI_GELU) staged(std::integral_constant<int, K_GELU>{});
    else if (p.epi == EPI_DGELU) staged(std::integral_constant<int, K_DGELU>{});
    else if (p.accumulate) direct(std::integral_constant<int, K_ACC>{});
    else if (C16b) staged(std::integral_constant<int, K_PLAIN>{});
    else direct(std::integral_constant<int, K_PLAIN>{});
  };

  if (!SK) {
    // one tile (and one split of its reduction) per block -- or, PERSISTENT (round 6; sk.total = the number of tiles, unsplit launches
    // of more than one round of the CUs): block b walks tiles b, b + gridDim.x, ...  A tile's epilogue stores are only ISSUED when the
    // wave moves on: they drain while the next tile's first operand tiles are on their way (one tile per block pays the store burst
    // of a whole round of tiles -- 64 MB at once -- and the next round's cold start one after the other).
    const int kbeg = blockIdx.z * p.k_chunk;
    const int kend = min(p.K, kbeg + p.k_chunk);
    const bool split = gridDim.z > 1;
    const int ntiles = sk.total > 0 ? (int)sk.total : (int)gridDim.x;
    Cout = p.C32 ? p.C32 + (long)blockIdx.z * p.slab_stride : nullptr;
    ldc32 = p.ldc32;
    for (int tile = bid; tile < ntiles; tile += (int)gridDim.x) {
      bind(p.A, p.B, p.lda, p.ldb, (tile / p.tiles_n) * 256, (tile % p.tiles_n) * 256, kbeg, (kend - kbeg) >> 6);
      mainloop();
      epilogue(split, split ? nullptr : p.C16);
      if (tile + (int)gridDim.x < ntiles) P256_BAR();  // (block-uniform) the staging images of the epilogue are free before the next tile's requests
    }
    return;
  }

  // ---- stream-K: ONE piece per block.  blockIdx.x -> run (reversed: later runs are dispatched first), blockIdx.y -> the
  // y-th piece of the run.  Piece 0 starts at the run's first step (possibly inside a tile: a CONTRIBUTION); the following
  // pieces start at the tile boundaries inside the run.  Every contribution is therefore a y = 0 piece of a LATER run than
  // its finisher: with x reversed and y slowest in the dispatch order, contributions are handed to CUs before the pieces
  // that wait for them (dispatch order is a speed matter only: waits are bounded and flagged) ----
  const int KT = sk.KT;
  // Which run a block takes is placement only (flags and slabs are indexed by RUN; every piece of a planned launch is resident at
  // once: tiles x S <= CUs).  bid is XCD-aware: consecutive values share an XCD, i.e. an L2.
  //  * tile-major (kmajor 0, the general cut): neighbouring runs = the pieces of ONE tile (different k-ranges: they share no
  //    operand byte) and of the tiles next to it;
  //  * k-major (round 6; equal pieces): an XCD's ~32 blocks take the SAME piece index -- the same rows of the reduction axis -- of
  //    ~32 consecutive tiles, which share their A / B panels: with S = 2 pieces and the [3 x 12] / [12 x 3] tile grids of a layer's
  //    weight gradients an XCD fetches ~15 panels per k-step for its 27 blocks where the tile-major order fetched ~28 for 32.
  int run = (int)gridDim.x - 1 - bid;
  if (sk.kmajor && gridDim.y == 1 && KT % sk.W == 0) {
    const int S = KT / sk.W, T = (int)gridDim.x / S;
    if (T * S == (int)gridDim.x) {
      const int s = bid / T, tp = bid - s * T;
      run = tp * S + (S - 1 - s);  // (contributions -- pieces with k0 > 0 -- take the low block ids: dispatched first)
    }
  }
  const long rs = (long)run * sk.W, re = min(rs + (long)sk.W, sk.total);
  long ps = rs;                                   // piece start
  if (blockIdx.y > 0) ps = (rs / KT + blockIdx.y) * KT;
  if (ps >= re) return;                           // (uniform: the run has fewer pieces)
  const int tg = (int)(ps / KT);
  const int k0 = (int)(ps - (long)tg * KT);
  const int k1 = (int)min((long)KT, k0 + (re - ps));
  {
    // the product this tile belongs to (up to four share a launch).  The table is indexed at run time: it stays in the
    // kernel-argument segment (scalar loads on demand) instead of occupying ~50 SGPRs for the whole kernel
    const int q = (tg >= sk.tile_begin[1]) + (tg >= sk.tile_begin[2]) + (tg >= sk.tile_begin[3]);
    const P256Prob& pb = sk.pr[q];
    const int tl = tg - sk.tile_begin[q], tiles_n = pb.tiles_n;
    Cout = pb.C32;
    ldc32 = pb.ldc32;
    bind(pb.A, pb.B, pb.lda, pb.ldb, (tl / tiles_n) * 256, (tl % tiles_n) * 256, k0 * 64, k1 - k0);
  }
  mainloop();
  const unsigned tid16 = (unsigned)tid * 16u;
  if (k0 > 0) {
    // CONTRIBUTION: accumulators -> this run's slab, in register order (8 KiB per store instruction of the block)
    unsigned char* const my_slab = reinterpret_cast<unsigned char*>(sk.slabs) + (long)run * (256 * 256 * 4);
#pragma unroll
    for (int I = 0; I < 8; ++I)
#pragma unroll
      for (int J = 0; J < 4; ++J) *reinterpret_cast<f32x4v*>(my_slab + (I * 4 + J) * 8192 + tid16) = acc[I][J];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains ...
    __syncthreads();                                    // ... before the one lane that publishes
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (keep: ROCm 7.2 can drop the fence's own wait)
      __hip_atomic_store((gu32*)(sk.flags + run), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  if (k1 < KT) {
    // FINISHER: the rest of this tile's reduction sits in the slabs of the following runs, in k order
    const int ncon = (KT - k1 + sk.W - 1) / sk.W;
    if (tid == 0) {
      for (int c = 1; c <= ncon; ++c) {
        unsigned spins = 0;
        while (__hip_atomic_load((gu32*)(sk.flags + run + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > (1u << 24)) {  // bounded: ~ seconds.  Never in a healthy launch; the error word says so
            __hip_atomic_store((gu32*)sk.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const unsigned char* const slab1 = reinterpret_cast<const unsigned char*>(sk.slabs) + (long)(run + 1) * (256 * 256 * 4);
    // eight 16-byte pieces per lane at a time: the run-time loop over the contributions carries 8 running sums, never the
    // 32 accumulator blocks (a loop that updates all of them made the allocator permute them through scratch)
#pragma unroll
    for (int b8 = 0; b8 < 4; ++b8) {
      f32x4v sum[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) sum[e] = acc[2 * b8 + (e >> 2)][e & 3];
      for (int c = 0; c < ncon; ++c) {
        const unsigned char* sl = slab1 + (long)c * (256 * 256 * 4) + b8 * 8 * 8192 + tid16;
        f32x4v part[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) part[e] = *reinterpret_cast<const f32x4v*>(sl + e * 8192);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum[e] += part[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[2 * b8 + (e >> 2)][e & 3] = sum[e];
    }
    if (tid == 0)
      for (int c = 1; c <= ncon; ++c) __hip_atomic_store((gu32*)(sk.flags + run + c), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // consumed
  }
  epilogue(false, p.C16);
}
#undef P256_KTILE
#undef P256_QUAD

template <bool A_KM, bool B_KM, bool SK>
static int launch_p256_t(const GemmArgsX& a, const P256SK& sk, dim3 grid, hipStream_t st) {
  constexpr size_t smem = 128 * 1024;
  auto kern = gemm_bf16_p256_kernel<A_KM, B_KM, SK>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(512), smem, st, a, sk);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

static int p256_check(const GemmArgsX& a) {
  // 32-bit byte offsets inside the result / pre-activation arrays (the epilogue's per-lane offsets)
  if ((double)a.M * std::max(std::max(a.ldc32 * 4, a.ldc16 * 2), a.ldaux * 2) >= 4294967296.0) return MTVAF_ERR_SHAPE;
  // the epilogue bodies of the kernel cover the combinations the path uses
  if (a.bias && (a.epi == EPI_DGELU || a.accumulate)) return MTVAF_ERR_ARG;
  if (a.colpart && a.epi != EPI_DGELU) return MTVAF_ERR_ARG;
  if (a.accumulate && (a.epi != EPI_NONE || a.C16 || !a.C32)) return MTVAF_ERR_ARG;
  return MTVAF_OK;
}

// 256 x 256 tile, any of the three operand-layout pairs of the path.  Requires M % 256 == 0, N % 256 == 0, K % 64 == 0 and
// whole 64-element k-tiles per split (checked by the caller, gemm_bf16x_core).
int launch_p256(const GemmArgsX& a, int layout_a, int layout_b, dim3 grid, hipStream_t st) {
  if (int rc = p256_check(a)) return rc;
  P256SK sk = {};
  // more than one round of tiles, unsplit: one block per CU walks the tiles (MTVAF_P256_PERSIST=0: one tile per block as before)
  static const int persist = [] { const char* e = getenv("MTVAF_P256_PERSIST"); return e ? atoi(e) : 1; }();
  if (persist && grid.z == 1 && grid.x > 256) {
    sk.total = grid.x;
    grid.x = 256;
  }
  if (layout_a == 0 && layout_b == 0) return launch_p256_t<false, false, false>(a, sk, grid, st);
  if (layout_a == 0) return launch_p256_t<false, true, false>(a, sk, grid, st);
  return launch_p256_t<true, true, false>(a, sk, grid, st);
}

// Stream-K launch: `sk` describes the step space (sk.nprob = 0: the single product of `a`, any epilogue; 1..4: products that
// share the launch, plain fp32 results); one block per CU of the scratch's grid.
int launch_p256_streamk(const GemmArgsX& a, P256SK sk, int layout_a, int layout_b, int grid_blocks, int steps_per_run, hipStream_t st) {
  if (int rc = p256_check(a)) return rc;
  if (grid_blocks <= 0 || !sk.slabs || !sk.flags || sk.KT <= 0 || sk.total <= 0) return MTVAF_ERR_ARG;
  if (sk.nprob == 0) {  // the single product of `a`
    sk.pr[0] = P256Prob{a.A, a.B, a.C32, a.lda, a.ldb, a.ldc32, a.tiles_n};
    sk.nprob = 1;
    sk.tile_begin[0] = 0;
  }
  for (int q = sk.nprob; q < 4; ++q) sk.tile_begin[q] = INT_MAX;
  // steps_per_run > 0: every tile's reduction is cut into KT / steps_per_run equal pieces (a divisor of KT: the planner's
  // choice -- equal pieces in whole rounds of the CUs); 0: the step line is cut into grid_blocks equal runs wherever they fall
  const long g = std::min<long>(grid_blocks, sk.total);
  sk.W = steps_per_run > 0 ? steps_per_run : (int)((sk.total + g - 1) / g);
  static const int kmajor_env = [] { const char* e = getenv("MTVAF_P256_SK_KMAJOR"); return e ? atoi(e) : 0; }();
  // (measured, round 6, profiles/r06_p256_sk_kmajor.txt: 2560 / 4864 / 38912 token rows 65.1 / 93.4 / 649 us tile-major against 70.7 /
  // 97.9 / 642 us k-major; C4 8387 vs 8402, C5 3385 vs 3399 sentences/s -- the launch is not bound by the fabric traffic the k-major
  // placement saves: off by default)
  sk.kmajor = kmajor_env;
  const long runs = (sk.total + sk.W - 1) / sk.W;
  if (steps_per_run > 0 && sk.KT % steps_per_run) return MTVAF_ERR_ARG;
  if (runs > grid_blocks && !(sk.W % sk.KT == 0)) return MTVAF_ERR_WORKSPACE;  // one slab / flag per run (runs of whole tiles need none)
  // runs x (pieces per run: the piece from the run's start, then one per tile boundary inside the run; blocks beyond a
  // run's last piece exit at once)
  const dim3 grid((unsigned)runs, (unsigned)(sk.KT % sk.W == 0 ? 1 : (sk.W + sk.KT - 2) / sk.KT + 1));
  if (layout_a == 0 && layout_b == 0) return launch_p256_t<false, false, true>(a, sk, grid, st);
  if (layout_a == 0) return launch_p256_t<false, true, true>(a, sk, grid, st);
  return launch_p256_t<true, true, true>(a, sk, grid, st);
}

}  // namespace mtvaf
