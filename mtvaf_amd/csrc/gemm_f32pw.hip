// fp32 GEMM on the bf16 matrix pipe from pre-split operands, 128 x 256 tile (round 6): the arithmetic, the plane images, the LDS
// images and the epilogues of gemm_f32p.hip's 128 x 128 kernel -- every output element sees the same MFMA products in the same
// order, so the two kernels agree BIT FOR BIT -- on a tile that moves a quarter less through the two paths that bound that kernel
// (profiles/r05_f32p_bench.txt: requests alone 48.9 us, MFMAs alone 61.4 us, both 82.7 us on QKV forward at 4096 rows):
//
//                                       128 x 128 x 32          128 x 256 x 32
//   L2 -> LDS requests per k-tile       48 KiB                  72 KiB  (36 per 128 x 128 of output)
//   fragment reads per MFMA wave        24 KiB for 96 MFMAs     36 KiB for 192 MFMAs
//   LDS traffic per 128 x 128 x 32      144 KiB                 108 KiB
//   tile prologues / epilogues          1 per 128 x 128         1 per 128 x 256
//
// Shape: 512 threads, waves 0-3 multiply (2 x 2 grid of 64 x 128 wave tiles = 4 x 8 blocks of v_mfma_f32_16x16x32_bf16: 128
// accumulator registers), waves 4-7 issue the LDS-DMA requests (18 each per k-tile).  TWO stages of 72 KiB (the A panel and the two
// 128-wide B panels, three planes each, every panel in the layout of the 128 x 128 kernel): the request for k-tile t + 2 leaves when
// the fragments of tile t are in registers and has one whole k-tile (192 MFMAs per wave, ~3100 cycles: as long as the two 96-MFMA
// tiles the three-stage ring of the small kernel allows) to land.
//
// Registers (256 per wave at two waves per SIMD): the 128 accumulators, ALL twelve A fragments of the k-tile (48 registers, single
// buffered) and the B fragments of two column blocks (24): a column block's 24 MFMAs run row block by row block (a chain of six
// products per accumulator: tools/micro/mfma_chain.hip -- dependent v_mfma_f32_16x16x32_bf16 issue at the full rate), so that under
// the LAST column block of a k-tile row block i's registers fall free after its six MFMAs and take row block i of the next k-tile.
#include "gemm_f32p.h"

namespace mtvaf {

// ABL / TRACE: the timing-only research switches and the shader-clock stamps of gemm_f32p16_kernel (1 = no MFMAs, 2 = no requests, 4 =
// no fragment reads; stamps of block 0: [wave][k-tile][arrive at / leave the tile barrier] + 17)
// NJ: 16-column blocks per wave = 8 (the 128 x 256 tile) or 6 (128 x 192, forward products only: QKV forward at 2432 rows is 228
// tiles instead of 171 on 256 CUs).
template <bool B_KM, bool A_KM, bool GROUP, int ABL = 0, bool TRACE = false, int NJ = 8>
__global__ __launch_bounds__(512, 1) void gemm_f32p16w_kernel(GemmArgsP p) {
  static_assert(!A_KM || B_KM, "k-major A comes with k-major B (weight gradients)");
  static_assert(!GROUP || A_KM, "grouped launches are weight gradients");
  static_assert(NJ == 8 || (NJ == 6 && !B_KM), "the 192-column tile serves k-contiguous operands");
  constexpr int BM = 128, BN = 32 * NJ;
  constexpr int PL_B = 128 * 64;              // one plane of the A panel (128 rows)
  constexpr int PAN_A = 3 * PL_B;             // the A panel: 24 KiB
  constexpr int PLB_B = NJ * 16 * 64;         // one plane of a B panel (16 NJ rows or columns)
  constexpr int PAN_B = 3 * PLB_B;            // a B panel: 24 KiB (18 at NJ = 6)
  constexpr int STAGE_B = PAN_A + 2 * PAN_B;  // A panel, B panel 0, B panel 1: 72 KiB (60)
  constexpr int IWA = 6, IWB = 3 * NJ / 2;    // requests per DMA wave and k-tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool dma_wave = wave >= 4;
  const int w4 = wave & 3;
  const int wm = w4 >> 1, wn = w4 & 1;
  if constexpr (GROUP) {
    if (p.cs_n > 0 && (int)blockIdx.x >= p.cs_tile0) {  // (block-uniform) a column-sum item: as gemm_f32p16_kernel
      float* red = reinterpret_cast<float*>(smem_w);
      const int bi = (int)blockIdx.x - p.cs_tile0;
      int j = 0;
#pragma unroll
      for (int q = 1; q < 8; ++q) j += (q < p.cs_n && bi >= p.cs_blk0[q]) ? 1 : 0;
      const GemmArgsP::ColJob& jb = p.cs[j];
      const int c = (bi - p.cs_blk0[j]) * 64 + (tid & 63), rg = tid >> 6;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      if (c < jb.cols) {
        const float* src = jb.src + c;
        int r = rg;
        for (; r + 24 < jb.rows; r += 32) {
          s0 += src[(long)r * jb.ld];
          s1 += src[(long)(r + 8) * jb.ld];
          s2 += src[(long)(r + 16) * jb.ld];
          s3 += src[(long)(r + 24) * jb.ld];
        }
        for (; r < jb.rows; r += 8) s0 += src[(long)r * jb.ld];
      }
      red[rg * 64 + (tid & 63)] = (s0 + s1) + (s2 + s3);
      __syncthreads();
      if (rg == 0 && c < jb.cols) {
        float t = red[tid & 63];
#pragma unroll
        for (int i = 1; i < 8; ++i) t += red[i * 64 + (tid & 63)];
        jb.dst[c] = t;
      }
      return;
    }
  }
  int bid = xcd_remap(blockIdx.x, GROUP ? p.cs_tile0 : (int)gridDim.x);
  const unsigned char* Apl = p.Ap;
  const unsigned char* Bpl = p.Bp;
  long a_plane = p.a_plane, a_row = p.a_row, a_kt = p.a_kt, a_col = p.a_col, b_plane = p.b_plane, b_row = p.b_row, b_kt = p.b_kt, b_col = p.b_col;
  float* Cp = p.C;
  int ldc = p.ldc, tiles_n = p.tiles_n;
  if constexpr (GROUP) {
    const int q = (bid >= p.grp_tile_begin[1]) + (bid >= p.grp_tile_begin[2]) + (bid >= p.grp_tile_begin[3]);
    const GemmArgsP::Prob& pb = p.grp[q];
    bid -= p.grp_tile_begin[q];
    Apl = pb.Ap; Bpl = pb.Bp; Cp = pb.C; ldc = pb.ldc; tiles_n = pb.tiles_n;
    a_plane = pb.a_plane; a_row = pb.a_row; a_kt = pb.a_kt; a_col = pb.a_col;
    b_plane = pb.b_plane; b_row = pb.b_row; b_kt = pb.b_kt; b_col = pb.b_col;
  }
  int tm = bid / tiles_n, tn = bid % tiles_n;
  if constexpr (!GROUP) {
    if (p.walk_g > 0) {
      const int tiles_m = (p.M + BM - 1) / BM, G = p.walk_g;
      const int band = bid / (G * tiles_n), first = band * G;
      const int gsz = tiles_m - first < G ? tiles_m - first : G;
      const int rem = bid - band * G * tiles_n;
      tm = first + rem % gsz;
      tn = rem / gsz;
    }
  }
  const int m0 = tm * BM;
  const int n0 = tn * BN;
  const int kbeg = blockIdx.z * p.k_chunk;
  const int kend = min(p.K, kbeg + p.k_chunk);
  const int nk = (kend - kbeg) / 32;
  long long* const tr = (TRACE && p.trace && blockIdx.x == 0 && blockIdx.z == 0) ? p.trace : nullptr;
  if (TRACE && tr && tid == 0) tr[8 * 64 * 2] = __builtin_amdgcn_s_memtime();

  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (dma_wave) {
    // ---- request issue.  A: piece I = w4 * 6 + i of the A panel (plane I / 8, 16 rows or 4 k-rows (I % 8) * ..) as in the 128 x 128
    // kernel; B: piece J = w4 * 12 + i of the 48 of the two B panels (panel J / 24 = w4 / 2)
    const unsigned char* pa[IWA];
    const unsigned char* pb[IWB];
#pragma unroll
    for (int i = 0; i < IWA; ++i) {
      const int I = w4 * IWA + i, plane = I >> 3;
      if constexpr (!A_KM) {
        const int row = (I & 7) * 16 + (lane >> 2), sc = (lane & 3) ^ f32p::swz(row);
        pa[i] = Apl + plane * a_plane + (long)(m0 + row) * a_row + (long)(kbeg / 32) * a_kt + sc * 16;
      } else {
        const int kr = (I & 7) * 4 + (lane >> 4), sca = (lane & 15) ^ km_swz(kr);
        pa[i] = Apl + plane * a_plane + (long)kr * a_row + (long)(kbeg / 32) * a_kt + (long)(m0 / 128) * a_col + (long)(sca >> 2) * (a_col >> 2) +
                ((sca & 3) << 4);
      }
    }
#pragma unroll
    for (int i = 0; i < IWB; ++i) {
      const int J = w4 * IWB + i, half = J / (3 * NJ), I = J % (3 * NJ), plane = I / NJ, sub = I % NJ;
      if constexpr (!B_KM) {
        const int row = sub * 16 + (lane >> 2), sc = (lane & 3) ^ f32p::swz(row);
        pb[i] = Bpl + plane * b_plane + (long)(n0 + half * (16 * NJ) + row) * b_row + (long)(kbeg / 32) * b_kt + sc * 16;
      } else {
        const int kr = sub * 4 + (lane >> 4), scb = (lane & 15) ^ km_swz(kr);
        pb[i] = Bpl + plane * b_plane + (long)kr * b_row + (long)(kbeg / 32) * b_kt + (long)(n0 / 128 + half) * b_col + (long)(scb >> 2) * (b_col >> 2) +
                ((scb & 3) << 4);
      }
    }
    auto issue = [&](int stage) __attribute__((always_inline)) {
      unsigned char* sa = smem_w + stage * STAGE_B + w4 * IWA * 1024;
      unsigned char* sb = smem_w + stage * STAGE_B + PAN_A + w4 * IWB * 1024;
#pragma unroll
      for (int i = 0; i < IWA; ++i) {
        glds16x(pa[i], sa + i * 1024);
        pa[i] += a_kt;
      }
#pragma unroll
      for (int i = 0; i < IWB; ++i) {
        glds16x(pb[i], sb + i * 1024);
        pb[i] += b_kt;
      }
    };
    constexpr bool go = !(ABL & 2);
    if (go && nk > 0) issue(0);
    if (go && nk > 1) issue(1);
    if (nk > 1) wait_vm<IWA + IWB>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();  // barrier -1: tile 0 is in stage 0
    asm volatile("" ::: "memory");
    int st = 0;
    for (int t = 0; t < nk; ++t) {
      // barrier t: tile t + 1 has landed (nothing else is in flight), every fragment of tile t is in registers
      if (TRACE && tr && t < 64 && lane == 0) tr[(wave * 64 + t) * 2 + 0] = __builtin_amdgcn_s_memtime();
      wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (TRACE && tr && t < 64 && lane == 0) tr[(wave * 64 + t) * 2 + 1] = __builtin_amdgcn_s_memtime();
      if (go && t + 2 < nk) issue(st);  // tile t + 2 into the stage tile t has left
      st ^= 1;
    }
  } else {
    const int r15 = lane & 15, ch = lane >> 4;
    const int offA = (wm * 64 + r15) * 64 + ((ch ^ f32p::swz(r15)) << 4);                       // + i * 1024 (+ plane)
    const int offB = PAN_A + PAN_B * wn + r15 * 64 + ((ch ^ f32p::swz(r15)) << 4);              // + j * 1024 (+ plane)
    // k-major images (32 k-rows x 256 B per plane and panel): lane (g, q, pp) addresses row 8 g + q (+ 4), the 4 columns 4 pp .. of
    // the 16-column block c >> 1 ...; for B the block is j itself (the panel is the wave's own): the chunk (2 j + (pp >> 1)) ^ swizzle
    // differs from the chunk of j = 0 in the bits of j alone -- offset(j) = offset(0) ^ (j << 5)
    int offA0[4], offA1[4];
    int offB0 = 0, offB1 = 0;
    if constexpr (B_KM) {
      const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
      const int r0 = 8 * g + q, r1 = r0 + 4;
      offB0 = PAN_A + PAN_B * wn + r0 * 256 + (((pp >> 1) ^ km_swz(r0)) << 4) + 8 * (pp & 1);
      offB1 = PAN_A + PAN_B * wn + r1 * 256 + (((pp >> 1) ^ km_swz(r1)) << 4) + 8 * (pp & 1);
      if constexpr (A_KM) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = 8 * wm + 2 * i + (pp >> 1);
          offA0[i] = r0 * 256 + ((c ^ km_swz(r0)) << 4) + 8 * (pp & 1);
          offA1[i] = r1 * 256 + ((c ^ km_swz(r1)) << 4) + 8 * (pp & 1);
        }
      }
    }
    fragp_t fa[3][4], fbs[2][3];
    constexpr bool do_rd = !(ABL & 4), do_mm = !(ABL & 1);
    auto rd_b = [&](const unsigned char* s, const int j, fragp_t (&f)[3]) __attribute__((always_inline)) {
      if constexpr (!do_rd) return;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if constexpr (!B_KM) f[q] = *reinterpret_cast<const fragp_t*>(s + q * PLB_B + offB + j * 1024);
        else f[q] = __builtin_bit_cast(fragp_t, tr_read8(s + q * PLB_B + (offB0 ^ (j << 5)), s + q * PLB_B + (offB1 ^ (j << 5))));
      }
    };
    auto rd_a = [&](const unsigned char* s, const int i) __attribute__((always_inline)) {
      if constexpr (!do_rd) return;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if constexpr (!A_KM) fa[q][i] = *reinterpret_cast<const fragp_t*>(s + q * PL_B + offA + i * 1024);
        else fa[q][i] = __builtin_bit_cast(fragp_t, tr_read8(s + q * PL_B + offA0[i], s + q * PL_B + offA1[i]));
      }
    };
    // the six products of row block i x column block j, smallest terms first (gemm_f32x3.hip's order)
    auto mm = [&](const int i, const int j, const fragp_t (&b)[3]) __attribute__((always_inline)) {
      if constexpr (!do_mm) return;
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[PA[t]][i]), __builtin_bit_cast(bf16x8, b[PB[t]]),
                                                            acc[i][j], 0, 0, 0);
    };
    // one fragment read behind each of the first MFMAs of a run of `nm` (see gemm_f32p16_kernel: a burst of reads idles the pipe)
    auto spread = [&](const int nm, const int nreads) __attribute__((always_inline)) {
      if constexpr (!do_rd || !do_mm) return;
#pragma unroll
      for (int i = 0; i < nm; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i < nreads - nm) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        else if (i < nreads) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    };
    constexpr int RA = A_KM ? 6 : 3, RB = B_KM ? 6 : 3;  // LDS reads per fragment triple
    if constexpr ((ABL & 4) != 0) {  // (defined operands for the timing-only ablation)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[q][i] = fragp_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        fbs[0][q] = fbs[1][q] = fragp_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
      }
    }
    __builtin_amdgcn_s_setprio(2);
    __builtin_amdgcn_s_barrier();  // barrier -1
    asm volatile("" ::: "memory");
    rd_a(smem_w, 0);
    rd_a(smem_w, 1);
    rd_a(smem_w, 2);
    rd_b(smem_w, 0, fbs[0]);
    int st = 0;
    // (-DMTVAF_PW_UNROLL2: two k-tiles per trip -- across the loop's back edge hipcc waits for EVERY outstanding fragment read, lgkmcnt(0)
    // in front of the first MFMA of a tile; unrolled, 3378 instead of 3429 ticks per k-tile in the block-0 trace -- and 5 % SLOWER as a
    // training step (3350 against 3555 sentences/s, same box): the unrolled kernels hold 214 - 248 registers instead of 204 - 228, and
    // at two waves per SIMD that is the difference between a CU that can take a wave of the other stream's kernel beside a GEMM block
    // and one that cannot; profiles/r06_p16_wide_tile_step_ab.txt)
    auto tile = [&](const int t) __attribute__((always_inline)) {
      const unsigned char* s = smem_w + st * STAGE_B;
      // column block 0 -- beside it the last row block of this tile's A and column block 1
      rd_a(s, 3);
      rd_b(s, 1, fbs[1]);
#pragma unroll
      for (int i = 0; i < 4; ++i) mm(i, 0, fbs[0]);
      spread(24, RA + RB);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 1; j < NJ - 1; ++j) {
        rd_b(s, j + 1, fbs[(j + 1) & 1]);
#pragma unroll
        for (int i = 0; i < 4; ++i) mm(i, j, fbs[j & 1]);
        spread(24, RB);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the last fragments of tile t are in registers: its stage may be refilled
      if (TRACE && tr && t < 64 && lane == 0) tr[(wave * 64 + t) * 2 + 0] = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_barrier();                         // barrier t
      asm volatile("" ::: "memory");
      if (TRACE && tr && t < 64 && lane == 0) tr[(wave * 64 + t) * 2 + 1] = __builtin_amdgcn_s_memtime();
      st ^= 1;
      {  // the last column block, row block by row block; behind each the row block's fragments of the NEXT tile (unconditional: past the end a
         // harmless read of the other stage -- see gemm_f32p16_kernel)
        const unsigned char* sn = smem_w + st * STAGE_B;
        rd_b(sn, 0, fbs[0]);
        mm(0, NJ - 1, fbs[1]);
        spread(6, RB);
        rd_a(sn, 0);
        mm(1, NJ - 1, fbs[1]);
        spread(6, RA);
        rd_a(sn, 1);
        mm(2, NJ - 1, fbs[1]);
        spread(6, RA);
        rd_a(sn, 2);
        mm(3, NJ - 1, fbs[1]);
        spread(6, RA);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    int t = 0;
#ifdef MTVAF_PW_UNROLL2
    for (; t + 1 < nk; t += 2) {
      tile(t);
      tile(t + 1);
    }
#endif
    for (; t < nk; ++t) tile(t);
  }

  if (TRACE && tr && lane == 0) tr[8 * 64 * 2 + 1 + wave] = __builtin_amdgcn_s_memtime();  // this wave's k-loop is over
  // ---- epilogue: the 128 x 256 image through the LDS (every request has landed: the last barrier waited for vmcnt(0)); per 128-column
  // half the thread mapping, the arithmetic and the order of gemm_f32p16_kernel's epilogue
  {
    constexpr int LDE = BN + 4, C4 = BN / 4;
    float* smem = reinterpret_cast<float*>(smem_w);
    float* C = Cp + (long)blockIdx.z * p.slab_stride;
    const bool split = gridDim.z > 1;
    __syncthreads();
    if (!dma_wave) {
      const int c15 = lane & 15, rq = lane >> 4;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) smem[(wm * 64 + i * 16 + rq * 4 + r) * LDE + wn * (16 * NJ) + j * 16 + c15] = acc[i][j][r];
    }
    __syncthreads();
    if (NJ == 8 && p.Cpl && !split) {  // (the 192-column tile is launched without a plane-image result)
      const int c8 = tid & 15, rr = tid >> 4;
      const long pl_b = (long)p.M * 64;
      f32x4 cs[2][2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int col = n0 + hh * 128 + 8 * c8;
        unsigned char* dstc = p.Cpl + (long)(col >> 5) * 3 * pl_b + (col & 31) * 2;
        f32x4 cs0 = {0.f, 0.f, 0.f, 0.f}, cs1 = {0.f, 0.f, 0.f, 0.f};
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) { b0 = *reinterpret_cast<const f32x4*>(p.bias + col); b1 = *reinterpret_cast<const f32x4*>(p.bias + col + 4); }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = rr + 32 * i;
          const long row = m0 + r;
          f32x4 v0 = *reinterpret_cast<const f32x4*>(smem + r * LDE + hh * 128 + 8 * c8) + b0;
          f32x4 v1 = *reinterpret_cast<const f32x4*>(smem + r * LDE + hh * 128 + 8 * c8 + 4) + b1;
          if (p.epi == EPI_GELU) {
            *reinterpret_cast<f32x4*>(p.aux + row * p.ldaux + col) = v0;
            *reinterpret_cast<f32x4*>(p.aux + row * p.ldaux + col + 4) = v1;
            v0 = f32x4{gelu_erf(v0.x), gelu_erf(v0.y), gelu_erf(v0.z), gelu_erf(v0.w)};
            v1 = f32x4{gelu_erf(v1.x), gelu_erf(v1.y), gelu_erf(v1.z), gelu_erf(v1.w)};
          } else if (p.epi == EPI_DGELU) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(p.aux + row * p.ldaux + col);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(p.aux + row * p.ldaux + col + 4);
            v0 = f32x4{v0.x * gelu_erf_grad(a0.x), v0.y * gelu_erf_grad(a0.y), v0.z * gelu_erf_grad(a0.z), v0.w * gelu_erf_grad(a0.w)};
            v1 = f32x4{v1.x * gelu_erf_grad(a1.x), v1.y * gelu_erf_grad(a1.y), v1.z * gelu_erf_grad(a1.z), v1.w * gelu_erf_grad(a1.w)};
          }
          if (Cp) {
            if (p.accumulate) {
              v0 += *reinterpret_cast<const f32x4*>(Cp + row * ldc + col);
              v1 += *reinterpret_cast<const f32x4*>(Cp + row * ldc + col + 4);
            }
            *reinterpret_cast<f32x4*>(Cp + row * ldc + col) = v0;
            *reinterpret_cast<f32x4*>(Cp + row * ldc + col + 4) = v1;
          }
          cs0 += v0;
          cs1 += v1;
          // (split the values as ROUNDED for the fp32 result: see gemm_f32p16_kernel)
          asm volatile("" : "+v"(v0.x), "+v"(v0.y), "+v"(v0.z), "+v"(v0.w), "+v"(v1.x), "+v"(v1.y), "+v"(v1.z), "+v"(v1.w));
          unsigned h[4], m[4], l[4];
          f32p::split3_pair(f32x2p{v0.x, v0.y}, h[0], m[0], l[0]);
          f32p::split3_pair(f32x2p{v0.z, v0.w}, h[1], m[1], l[1]);
          f32p::split3_pair(f32x2p{v1.x, v1.y}, h[2], m[2], l[2]);
          f32p::split3_pair(f32x2p{v1.z, v1.w}, h[3], m[3], l[3]);
          unsigned char* d = dstc + row * 64;
          *reinterpret_cast<uint4*>(d) = uint4{h[0], h[1], h[2], h[3]};
          *reinterpret_cast<uint4*>(d + pl_b) = uint4{m[0], m[1], m[2], m[3]};
          *reinterpret_cast<uint4*>(d + 2 * pl_b) = uint4{l[0], l[1], l[2], l[3]};
        }
        cs[hh][0] = cs0;
        cs[hh][1] = cs1;
      }
      if (p.colpart) {  // column sums of the tile: 32 row groups per half through the LDS, summed in a fixed order
        __syncthreads();  // (every thread has read its part of the image)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          *reinterpret_cast<f32x4*>(smem + (hh * 32 + rr) * 128 + 8 * c8) = cs[hh][0];
          *reinterpret_cast<f32x4*>(smem + (hh * 32 + rr) * 128 + 8 * c8 + 4) = cs[hh][1];
        }
        __syncthreads();
        if (tid < 256) {
          const int hh = tid >> 7, c = tid & 127;
          float t = 0.f;
#pragma unroll 8
          for (int g = 0; g < 32; ++g) t += smem[(hh * 32 + g) * 128 + c];
          p.colpart[(long)(m0 / 128) * p.N + n0 + tid] = t;
        }
      }
      if (TRACE && tr && lane == 0) tr[8 * 64 * 2 + 9 + wave] = __builtin_amdgcn_s_memtime();
      return;
    }
#pragma unroll 2
    for (int idx = tid; idx < BM * C4; idx += 512) {
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
        if (p.accumulate) v += *reinterpret_cast<const f32x4*>(C + row * ldc + col);
      }
      *reinterpret_cast<f32x4*>(C + row * ldc + col) = v;
    }
  }
  if (TRACE && tr && lane == 0) tr[8 * 64 * 2 + 9 + wave] = __builtin_amdgcn_s_memtime();
}

template <bool B_KM, bool A_KM, bool GROUP, int ABL = 0, bool TRACE = false, int NJ = 8>
static int launch_p16w(const GemmArgsP& a, dim3 grid, hipStream_t st) {
  constexpr size_t smem = (size_t)2 * 3 * (128 + 32 * NJ) * 64;  // 147456 (122880 at NJ = 6)
  auto kern = gemm_f32p16w_kernel<B_KM, A_KM, GROUP, ABL, TRACE, NJ>;
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

int launch_gemm_f32p16w_group(const GemmArgsP& a, dim3 grid, hipStream_t st) { return launch_p16w<true, true, true>(a, grid, st); }

int launch_gemm_f32p16w(const GemmArgsP& a, int a_km, int b_km, dim3 grid, hipStream_t st, int bn) {
  if (a_km && !b_km) return MTVAF_ERR_ARG;
  if (bn == 192) return (a_km || b_km || a.trace || a.ablate || a.Cpl) ? MTVAF_ERR_ARG : launch_p16w<false, false, false, 0, false, 6>(a, grid, st);
  if (bn != 256) return MTVAF_ERR_ARG;
  if (a_km) return a.trace ? launch_p16w<true, true, false, 0, true>(a, grid, st) : (a.ablate ? MTVAF_ERR_ARG : launch_p16w<true, true, false>(a, grid, st));
  if (b_km) return a.trace ? launch_p16w<true, false, false, 0, true>(a, grid, st) : (a.ablate ? MTVAF_ERR_ARG : launch_p16w<true, false, false>(a, grid, st));
  if (a.trace) return launch_p16w<false, false, false, 0, true>(a, grid, st);
  switch (a.ablate) {
    case 0: return launch_p16w<false, false, false>(a, grid, st);
    case 1: return launch_p16w<false, false, false, 1>(a, grid, st);
    case 2: return launch_p16w<false, false, false, 2>(a, grid, st);
    case 4: return launch_p16w<false, false, false, 4>(a, grid, st);
    case 5: return launch_p16w<false, false, false, 5>(a, grid, st);
    case 6: return launch_p16w<false, false, false, 6>(a, grid, st);
    default: return MTVAF_ERR_ARG;
  }
}

}  // namespace mtvaf
