// fp32 GEMM on the bf16 matrix pipe from PRE-SPLIT operands (round 5; research kernel behind mtvaf_gemm_f32p): both operands
// arrive as plane images -- the three bf16 planes x = x1 + x2 + x3 of gemm_f32x3.hip's split, written once per tensor by
// mtvaf_f32_split_planes (or, one day, by the kernel that produces the tensor) -- and a product a.b is the same six MFMA
// products a3 b1 + a1 b3 + a2 b2 + a2 b1 + a1 b2 + a1 b1 (fp32 accumulation, smallest first), here on
// v_mfma_f32_16x16x32_bf16.
//
// Why this form.  The wave-specialised kernel of gemm_f32x3.hip is bound by INSTRUCTION ISSUE on its SIMDs: per 32-deep k-tile a
// SIMD issues the consumer's 48 MFMAs and 24 fragment reads AND the producer's ~200 vector instructions (the split: 11 per pair
// of values), 24 plane stores and 16 loads -- 2200 - 2460 cycles per k-tile where the MFMAs need 1536.  With the
// split hoisted out of the k-loop nothing but MFMAs and fragment reads is left beside the requests, so the matrix pipe is the
// limit -- and then the MFMA SHAPE decides the rate, because a chip full of MFMAs runs at its power limit: a pure 32x32x16
// stream holds 1.51 GHz = 1545 TFLOP/s, a 16x16x32 stream 1.82 GHz = 1812 (profiles/r04_mfma_bf16_rate.txt).  In the
// wave-specialised kernel the 16x16 shape lost (twice the MFMA issues beside the producers); without producers it does not.
//
// Shape: 128 x 128 x 32 tile, 512 threads: waves 0-3 multiply (2 x 2 grid of 64 x 64 wave tiles = 4 x 4 blocks of 16 x 16; 96
// MFMAs and 24 ds_read_b128 per k-tile), waves 4-7 only issue LDS-DMA requests (global_load_lds_dwordx4: a request costs its wave
// 60 - 95 issue cycles, so they get waves of their own).  Three stages of 48 KiB (3 planes x (128 + 128) rows x 64 B), two
// k-tiles in flight behind a counted vmcnt, ONE raw s_barrier per k-tile placed so that the fragments of the next tile are read
// under the last 24 MFMAs of this one.
//
// LDS image of a k-contiguous (KC) operand plane: 128 rows x 64 B, unpadded; the 16-byte chunk c of row r sits at position
// c ^ G[(r >> 2) & 3] with G = {0, 2, 3, 1}: the ds_read_b128 of a 16x16x32 fragment (lane l: row l & 15, chunk l >> 4) is then
// conflict-free in every one of its four 16-lane groups.  LDS-DMA writes lane-linear, so the swizzle sits on the per-lane SOURCE
// address.  Plane images in memory are addressed by three byte strides (plane, row, k-tile): the natural form ([3][rows][ld], the
// fp32 tensor's own layout) and the tile-blocked form ([k-tile][plane][rows][32]: every 1-KiB request reads 1 KiB of contiguous
// memory) are both served.
#include "gemm_f32p.h"
#include <climits>

namespace mtvaf {

// fp32 [rows][cols] (leading dimension ld) -> plane image at dst with the byte strides (plane, row, k-tile); a thread takes 8
// consecutive columns (one 16-byte chunk of each plane)
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, int rows,
                                                           int cols, int ld, long s_plane, long s_row, long s_kt) {
  const long n8 = (long)rows * (cols / 8);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / (cols / 8)), c8 = (int)(i % (cols / 8));
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + (long)r * ld + 8 * c8);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + (long)r * ld + 8 * c8 + 4);
    unsigned h[4], m[4], l[4];
    f32p::split3_pair(f32x2p{v0.x, v0.y}, h[0], m[0], l[0]);
    f32p::split3_pair(f32x2p{v0.z, v0.w}, h[1], m[1], l[1]);
    f32p::split3_pair(f32x2p{v1.x, v1.y}, h[2], m[2], l[2]);
    f32p::split3_pair(f32x2p{v1.z, v1.w}, h[3], m[3], l[3]);
    unsigned char* d = dst + (long)(c8 >> 2) * s_kt + (long)r * s_row + (c8 & 3) * 16;
    *reinterpret_cast<uint4*>(d) = uint4{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<uint4*>(d + s_plane) = uint4{m[0], m[1], m[2], m[3]};
    *reinterpret_cast<uint4*>(d + 2 * s_plane) = uint4{l[0], l[1], l[2], l[3]};
  }
}

// ABL: timing-only research switches, compile-time so that the product instantiation (0) carries no branch in its k-loop (as
// run-time tests every MFMA group sat in a block of its own behind an s_waitcnt lgkmcnt(0): the fragment reads issued just
// before it were waited for at once): 1 = no MFMAs, 2 = no DMA requests, 4 = no fragment reads.  TRACE: shader-clock stamps.
// B_KM: the reduction index is the ROW of B (dX = dY . W with W [N][K] row-major: modeling_bert.py's nn.Linear backward): per plane
// the LDS image is 32 k-rows x 256 B (128 output columns), 16-byte chunk c of row r at c ^ km_swz(r) (gemm_bf16x.h), and a
// 16x16x32 B fragment (lane l: column l & 15, k = 8 (l >> 4) .. + 7) is two ds_read_b64_tr_b16 on the 4 x 16 blocks of rows
// 8 (l >> 4) .. + 3 and .. + 4 .. + 7 -- the two blocks a 32-lane half reads are 8 rows apart in the same columns: conflict-free.
// A_KM (with B_KM: the weight-gradient products dW = dY^T . X, both operands [tokens][features] with the token as the reduction
// index): the same image and the same transposing reads for A.
// k-major images are addressed through FOUR byte strides: plane, k-row, k-tile (32 k-rows) and a_col / b_col = the stride of a
// 128-column block, a quarter of which is the stride of a 32-column block: the natural image ([3][rows][cols]: 256) and the
// tile-blocked image ([cols / 32][3][rows][32]: a_row = 64, a_kt = 2048, a_col = 4 x 3 x rows x 64) are both served -- the blocked
// image of an activation then feeds its forward / dX product (k-contiguous A) AND its weight-gradient product (k-major).
// GROUP: see GemmArgsP::grp.
template <int ABL, bool TRACE, bool B_KM, bool A_KM = false, bool GROUP = false>
__global__ __launch_bounds__(512, 1) void gemm_f32p16_kernel(GemmArgsP p) {
  static_assert(!A_KM || B_KM, "k-major A comes with k-major B (weight gradients)");
  static_assert(!GROUP || A_KM, "grouped launches are weight gradients");
  constexpr int BM = 128, BN = 128, NS = 3;
  constexpr int PL_B = 128 * 64;       // bytes of one plane tile
  constexpr int OP_B = 3 * PL_B;       // one operand's stage: 24 KiB
  constexpr int STAGE_B = 2 * OP_B;    // 48 KiB
  constexpr int NPIECE = OP_B / 1024;  // 24 requests per operand and k-tile
  constexpr int IW = NPIECE / 4;       // 6 per DMA wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_p[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool dma_wave = wave >= 4;
  const int w4 = wave & 3;
  const int wm = w4 >> 1, wn = w4 & 1;
  if constexpr (GROUP) {
    if (p.cs_n > 0 && (int)blockIdx.x >= p.cs_tile0) {  // (block-uniform) a column-sum item: 64 columns, 8 row groups, fixed summation order
      float* red = reinterpret_cast<float*>(smem_p);
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
  int bid = xcd_remap(blockIdx.x, GROUP ? p.cs_tile0 : (int)gridDim.x);  // (GROUP: cs_tile0 = the number of tiles, items or not)
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

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (dma_wave) {
    // ---- request issue: piece I of an operand = plane I / 8, 16 rows (I % 8) * 16 ..; lane -> row + lane / 4, LDS chunk position
    // lane % 4, source chunk = position ^ swz(row)
    const unsigned char* pa[IW];
    const unsigned char* pb[IW];
#pragma unroll
    for (int i = 0; i < IW; ++i) {
      const int I = w4 * IW + i, plane = I >> 3, row = (I & 7) * 16 + (lane >> 2), cp = lane & 3;
      const int sc = cp ^ f32p::swz(row);
      if constexpr (!A_KM) {
        pa[i] = Apl + plane * a_plane + (long)(m0 + row) * a_row + (long)(kbeg / 32) * a_kt + sc * 16;
      } else {
        const int kr = (I & 7) * 4 + (lane >> 4), sca = (lane & 15) ^ km_swz(kr);  // source chunk: 8 columns, 4 chunks per 32-column block
        pa[i] = Apl + plane * a_plane + (long)kr * a_row + (long)(kbeg / 32) * a_kt + (long)(m0 / 128) * a_col + (long)(sca >> 2) * (a_col >> 2) +
                ((sca & 3) << 4);
      }
      if constexpr (!B_KM) {
        pb[i] = Bpl + plane * b_plane + (long)(n0 + row) * b_row + (long)(kbeg / 32) * b_kt + sc * 16;
      } else {  // piece I = plane I / 8, 4 k-rows (I % 8) * 4 ..; lane -> k-row + lane / 16, chunk position lane % 16
        const int kr = (I & 7) * 4 + (lane >> 4), scb = (lane & 15) ^ km_swz(kr);
        pb[i] = Bpl + plane * b_plane + (long)kr * b_row + (long)(kbeg / 32) * b_kt + (long)(n0 / 128) * b_col + (long)(scb >> 2) * (b_col >> 2) +
                ((scb & 3) << 4);
      }
    }
    auto issue = [&](int stage) __attribute__((always_inline)) {
      unsigned char* sa = smem_p + stage * STAGE_B + w4 * IW * 1024;
      unsigned char* sb = sa + OP_B;
#pragma unroll
      for (int i = 0; i < IW; ++i) {
        glds16x(pa[i], sa + i * 1024);
        pa[i] += a_kt;
      }
#pragma unroll
      for (int i = 0; i < IW; ++i) {
        glds16x(pb[i], sb + i * 1024);
        pb[i] += b_kt;
      }
    };
    constexpr bool go = !(ABL & 2);
    // prologue: tiles 0 and 1 are requested, tile 0 is waited for and published (barrier -1), THEN tile 2 is requested -- with all
    // three stages requested first the consumers started 12 requests (~1000 cycles of issue) later
    if (go) {
      if (nk > 0) issue(0);
      if (nk > 1) issue(1);
    }
    if (nk > 1) wait_vm<2 * IW>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (go && nk > 2) issue(2);
    int st = 0;
    for (int t = 0; t < nk; ++t) {
      // barrier t: tile t + 1 has landed (tile t + 2 may still be in flight), every fragment of tile t is in registers
      if (TRACE && tr && t < 64 && lane == 0) tr[(wave * 64 + t) * 2 + 0] = __builtin_amdgcn_s_memtime();
      if (t + 2 < nk) wait_vm<2 * IW>();
      else wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (TRACE && tr && t < 64 && lane == 0) tr[(wave * 64 + t) * 2 + 1] = __builtin_amdgcn_s_memtime();
      if (go && t + NS < nk) issue(st);  // tile t + 3 into the stage tile t has left
      st = st + 1 == NS ? 0 : st + 1;
    }
  } else {
    // ---- fragment read offsets inside a plane tile: row-block i of this wave's 64 rows, lane -> row l & 15, chunk l >> 4
    const int r15 = lane & 15, ch = lane >> 4;
    const int offA = (wm * 64 + r15) * 64 + ((ch ^ f32p::swz(r15)) << 4);
    const int offB = (wn * 64 + r15) * 64 + ((ch ^ f32p::swz(r15)) << 4);
    // k-major B: lane (g, q, pp) of a 16-lane group g addresses row 8 g + q (+ 4), columns 16 j + 4 pp .. + 3 of the wave's 64
    int offA0[4], offA1[4];
    if constexpr (A_KM) {
      const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
      const int r0 = 8 * g + q, r1 = r0 + 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 8 * wm + 2 * i + (pp >> 1);
        offA0[i] = r0 * 256 + ((c ^ km_swz(r0)) << 4) + 8 * (pp & 1);
        offA1[i] = r1 * 256 + ((c ^ km_swz(r1)) << 4) + 8 * (pp & 1);
      }
    }
    int offB0[4], offB1[4];
    if constexpr (B_KM) {
      const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
      const int r0 = 8 * g + q, r1 = r0 + 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = 8 * wn + 2 * j + (pp >> 1);
        offB0[j] = r0 * 256 + ((c ^ km_swz(r0)) << 4) + 8 * (pp & 1);
        offB1[j] = r1 * 256 + ((c ^ km_swz(r1)) << 4) + 8 * (pp & 1);
      }
    }
    fragp_t fb[2][3][4], fa[2][3];
    constexpr bool do_rd = !(ABL & 4), do_mm = !(ABL & 1);
    auto rd_b = [&](const unsigned char* s, fragp_t (&f)[3][4]) __attribute__((always_inline)) {
      if constexpr (!do_rd) return;
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (!B_KM) f[q][j] = *reinterpret_cast<const fragp_t*>(s + OP_B + q * PL_B + offB + j * 1024);
          else f[q][j] = __builtin_bit_cast(fragp_t, tr_read8(s + OP_B + q * PL_B + offB0[j], s + OP_B + q * PL_B + offB1[j]));
        }
    };
    auto rd_a = [&](const unsigned char* s, int i, fragp_t (&f)[3]) __attribute__((always_inline)) {
      if constexpr (!do_rd) return;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if constexpr (!A_KM) f[q] = *reinterpret_cast<const fragp_t*>(s + q * PL_B + offA + i * 1024);
        else f[q] = __builtin_bit_cast(fragp_t, tr_read8(s + q * PL_B + offA0[i], s + q * PL_B + offA1[i]));
      }
    };
    auto mm = [&](int i, const fragp_t (&a)[3], const fragp_t (&b)[3][4]) __attribute__((always_inline)) {
      if constexpr (!do_mm) return;
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // smallest terms first (as gemm_f32x3.hip)
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[PA[t]]), __builtin_bit_cast(bf16x8, b[PB[t]][j]),
                                                              acc[i][j], 0, 0, 0);
    };
    if constexpr ((ABL & 4) != 0) {  // (defined operands for the timing-only ablation)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          fa[u][q] = fragp_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
          for (int j = 0; j < 4; ++j) fb[u][q][j] = fragp_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        }
    }
    __builtin_amdgcn_s_setprio(2);
    __builtin_amdgcn_s_barrier();  // barrier -1: tile 0 is in stage 0
    asm volatile("" ::: "memory");
    if (nk > 0) {
      rd_b(smem_p, fb[0]);
      rd_a(smem_p, 0, fa[0]);
    }
    int st = 0;
    // (two k-tiles per trip so that the fragment buffers are indexed by constants)
    // A wave issues in order: a burst of fragment reads between two MFMA groups leaves the matrix pipe idle for the burst's issue
    // time (measured: + 22 % on the MFMA stream alone).  sched_group_barrier puts ONE read behind each of a group's first MFMAs
    // instead: the three fragments of the next row-block ride behind MFMAs 1-3 of a group of 24 (21 MFMAs = 340 cycles to land),
    // the fifteen of the next tile behind MFMAs 1-15 of the group behind the barrier.
    auto spread = [&](int nreads) __attribute__((always_inline)) {
      if constexpr (do_rd && do_mm) {
#pragma unroll
        for (int i = 0; i < 24; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (i < nreads - 24) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          else if (i < nreads) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    auto tile = [&](int t, const int cur) __attribute__((always_inline)) {
      const unsigned char* s = smem_p + st * STAGE_B;
      rd_a(s, 1, fa[1]);
      mm(0, fa[0], fb[cur]);
      spread(A_KM ? 6 : 3);
      rd_a(s, 2, fa[0]);
      mm(1, fa[1], fb[cur]);
      spread(A_KM ? 6 : 3);
      rd_a(s, 3, fa[1]);
      mm(2, fa[0], fb[cur]);
      spread(A_KM ? 6 : 3);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the last fragments of tile t are in registers: its stage may be refilled
      if (TRACE && tr && t < 64 && lane == 0) tr[(wave * 64 + t) * 2 + 0] = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_barrier();  // barrier t
      asm volatile("" ::: "memory");
      if (TRACE && tr && t < 64 && lane == 0) tr[(wave * 64 + t) * 2 + 1] = __builtin_amdgcn_s_memtime();
      st = st + 1 == NS ? 0 : st + 1;
      {  // (UNCONDITIONAL -- past the end a harmless read of the next stage: behind a conditional block of reads hipcc cannot
         // count on them and puts s_waitcnt lgkmcnt(0) in front of the next MFMA group, i.e. waits for reads issued just before it)
        const unsigned char* sn = smem_p + st * STAGE_B;
        rd_b(sn, fb[cur ^ 1]);
        rd_a(sn, 0, fa[0]);
      }
      mm(3, fa[1], fb[cur]);
      spread((B_KM ? 24 : 12) + (A_KM ? 6 : 3));
    };
    int t = 0;
    for (; t + 1 < nk; t += 2) {
      tile(t, 0);
      tile(t + 1, 1);
    }
    if (t < nk) tile(t, 0);
  }

  if (TRACE && tr && lane == 0) tr[8 * 64 * 2 + 1 + wave] = __builtin_amdgcn_s_memtime();  // this wave's k-loop is over
  // ---- epilogue: the 128 x 128 image through the LDS (every request has landed: the last barrier waited for vmcnt(0)), wide stores
  {
    constexpr int LDE = BN + 4, C4 = BN / 4;
    float* smem = reinterpret_cast<float*>(smem_p);
    float* C = Cp + (long)blockIdx.z * p.slab_stride;
    const bool split = gridDim.z > 1;
    __syncthreads();
    if (!dma_wave) {
      const int c15 = lane & 15, rq = lane >> 4;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) smem[(wm * 64 + i * 16 + rq * 4 + r) * LDE + wn * 64 + j * 16 + c15] = acc[i][j][r];
    }
    __syncthreads();
    if (p.Cpl && !split) {
      // plane-image output: a thread takes 8 consecutive columns (one 16-byte chunk of each plane) of 4 rows; the 16 threads of a
      // row cover its 128 columns, so a wave writes 4 rows x 64 contiguous bytes in each of the four 32-column blocks of the tile
      const int c8 = tid & 15, rr = tid >> 4;  // (the same 8 columns for all four rows of a thread)
      const int col = n0 + 8 * c8;
      const long pl_b = (long)p.M * 64;
      unsigned char* dstc = p.Cpl + (long)(col >> 5) * 3 * pl_b + (col & 31) * 2;
      f32x4 cs0 = {0.f, 0.f, 0.f, 0.f}, cs1 = {0.f, 0.f, 0.f, 0.f};
      f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) { b0 = *reinterpret_cast<const f32x4*>(p.bias + col); b1 = *reinterpret_cast<const f32x4*>(p.bias + col + 4); }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = rr + 32 * i;
        const long row = m0 + r;
        f32x4 v0 = *reinterpret_cast<const f32x4*>(smem + r * LDE + 8 * c8) + b0;
        f32x4 v1 = *reinterpret_cast<const f32x4*>(smem + r * LDE + 8 * c8 + 4) + b1;
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
        // (split the values as ROUNDED for the fp32 result: under -ffp-contract=fast the residual x - bf16(x) would otherwise fuse
        // with the multiplication that produced x)
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
      if (p.colpart) {  // column sums of the tile: 32 row groups through the LDS, summed in a fixed order
        __syncthreads();  // (every thread has read its part of the image)
        *reinterpret_cast<f32x4*>(smem + rr * 128 + 8 * c8) = cs0;
        *reinterpret_cast<f32x4*>(smem + rr * 128 + 8 * c8 + 4) = cs1;
        __syncthreads();
        if (tid < 128) {
          float t = 0.f;
#pragma unroll 8
          for (int g = 0; g < 32; ++g) t += smem[g * 128 + tid];
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

int launch_split_planes(const float* src, void* dst, int rows, int cols, int ld, long s_plane, long s_row, long s_kt, hipStream_t st) {
  const long n8 = (long)rows * (cols / 8);
  const int blocks = (int)std::min<long>((n8 + 255) / 256, 4096);
  hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, st, src, static_cast<unsigned char*>(dst), rows, cols, ld, s_plane,
                     s_row, s_kt);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

template <int ABL, bool TRACE, bool B_KM = false, bool A_KM = false>
static int launch_p16_t(const GemmArgsP& a, dim3 grid, hipStream_t st) {
  constexpr size_t smem = (size_t)3 * 2 * 3 * 128 * 64;  // 147456
  auto kern = gemm_f32p16_kernel<ABL, TRACE, B_KM, A_KM>;
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

template <int ABL, bool TRACE, bool B_KM, bool A_KM, bool GROUP>
static int launch_p16_g(const GemmArgsP& a, dim3 grid, hipStream_t st) {
  constexpr size_t smem = (size_t)3 * 2 * 3 * 128 * 64;
  auto kern = gemm_f32p16_kernel<ABL, TRACE, B_KM, A_KM, GROUP>;
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
int launch_gemm_f32p16_group(const GemmArgsP& a, dim3 grid, hipStream_t st) { return launch_p16_g<0, false, true, true, true>(a, grid, st); }

int launch_gemm_f32p16(const GemmArgsP& a, int a_km, int b_km, dim3 grid, hipStream_t st) {
  if (a_km && !b_km) return MTVAF_ERR_ARG;
  if (a_km) return a.trace ? launch_p16_t<0, true, true, true>(a, grid, st) : (a.ablate ? MTVAF_ERR_ARG : launch_p16_t<0, false, true, true>(a, grid, st));
  if (b_km) return a.trace ? launch_p16_t<0, true, true>(a, grid, st) : (a.ablate ? MTVAF_ERR_ARG : launch_p16_t<0, false, true>(a, grid, st));
  if (a.trace) return launch_p16_t<0, true>(a, grid, st);
  switch (a.ablate) {
    case 0: return launch_p16_t<0, false>(a, grid, st);
    case 1: return launch_p16_t<1, false>(a, grid, st);
    case 2: return launch_p16_t<2, false>(a, grid, st);
    case 4: return launch_p16_t<4, false>(a, grid, st);
    case 5: return launch_p16_t<5, false>(a, grid, st);
    case 6: return launch_p16_t<6, false>(a, grid, st);
    default: return MTVAF_ERR_ARG;
  }
}

}  // namespace mtvaf

using namespace mtvaf;

extern "C" {

// fp32 [rows][cols] (ld) -> its three bf16 planes (the RNE split of gemm_f32x3.hip) as a plane image with the given byte strides
// (plane -> plane, row -> row, 32-column k-tile -> k-tile); cols % 32 == 0, 16-byte aligned.
int mtvaf_f32_split_planes(const float* src, void* dst, int rows, int cols, int ld, long s_plane, long s_row, long s_kt, hipStream_t stream) {
  if (!src || !dst || rows <= 0 || cols <= 0 || cols % 32 || ld % 4) return MTVAF_ERR_SHAPE;
  if ((((uintptr_t)src | (uintptr_t)dst) & 15) || (s_plane % 16) || (s_row % 16) || (s_kt % 16)) return MTVAF_ERR_ALIGN;
  return launch_split_planes(src, dst, rows, cols, ld, s_plane, s_row, s_kt, stream);
}

static long long* g_f32p_trace = nullptr;
// Which products take the 128 x 256 tile (gemm_f32pw.hip): a mask -- 1: forward products (both operands k-contiguous), 2: dX products
// (k-major B), 4: weight gradients (k-major A and B; the grouped launch) -- among the tiles the mask admits a launch takes the
// cheapest by gemm_f32p_run's rounds x tile-time estimate; 8: never the 128 x 128 tile where another can serve (tests; research);
// 16: never the 128 x 192 tile (unsplit or split forward products without a plane-image result, N % 192 == 0).
static int g_p16_wide = -1;  // -1: MTVAF_P16_WIDE
static int p16_wide_mask() {
  static const int v = [] { const char* e = getenv("MTVAF_P16_WIDE"); return e ? atoi(e) & 31 : 7; }();
  return g_p16_wide < 0 ? v : g_p16_wide;
}
// mask >= 0: set; -1: only query.  Returns the mask in force.  Process-global, like mtvaf_f32_split; placement only -- the two
// kernels agree bit for bit.
int mtvaf_f32p_wide(int mask) {
  if (mask >= 0) g_p16_wide = mask & 31;
  return p16_wide_mask();
}

int mtvaf_f32p_trace(void* buf) {
  g_f32p_trace = static_cast<long long*>(buf);
  return MTVAF_OK;
}

// C[M,N] = A[M,K] . op(B) (+ bias, epilogue) from plane images of both operands.  layout_b 0: B [N][K], k contiguous (every
// nn.Linear forward, modeling_bert.py:266, 283-284, 353, 420-421, 433); layout_b 1: B [K][N], the reduction index is the row (their
// dX products: dY . W), b_row / b_kt = byte strides of a k-row / of 32 k-rows, b_col = of a 128-column block.  layout_a 1 (with
// layout_b 1): A [K][M] as well -- their weight-gradient products dW = dY^T . X, reduction over the token rows.  M, N % 128 == 0,
// K % 32 == 0.  splits > 1: split-K slabs
// in `workspace` (deterministic ordered reduction, as mtvaf_gemm_f32).
static int gemm_f32p_run(int layout_a, const void* Aplanes, long a_plane, long a_row, long a_kt, long a_col, int layout_b, const void* Bplanes,
                         long b_plane, long b_row, long b_kt, long b_col, float* C, int ldc, int M, int N, int K, const float* bias, int epi,
                         float* aux, int ldaux, int accumulate, int splits, void* workspace, size_t workspace_bytes, int ablate,
                         hipStream_t stream, int* keep_slabs, void* c_planes = nullptr, float* colpart = nullptr) {
  if (!Aplanes || !Bplanes || (!C && !c_planes) || M <= 0 || N <= 0 || K <= 0) return MTVAF_ERR_ARG;
  if (c_planes && (((uintptr_t)c_planes & 15) || (long)M * N * 6 >= (1L << 40))) return MTVAF_ERR_ALIGN;
  if (colpart && !c_planes) return MTVAF_ERR_ARG;
  if (c_planes) splits = 1;  // (the plane image is written by the epilogue of an unsplit launch)
  if (M % 128 || N % 128 || K % 32) return MTVAF_ERR_SHAPE;
  if ((epi == EPI_GELU || epi == EPI_DGELU || epi == EPI_DTANH) && !aux) return MTVAF_ERR_ARG;
  if ((ldc % 4) || (aux && ldaux % 4) || (((uintptr_t)Aplanes | (uintptr_t)Bplanes | (uintptr_t)C | (uintptr_t)aux | (uintptr_t)bias) & 15))
    return MTVAF_ERR_ALIGN;
  if ((a_plane | a_row | a_kt | a_col | b_plane | b_row | b_kt | b_col) & 15) return MTVAF_ERR_ALIGN;
  if (layout_a < 0 || layout_a > 1 || layout_b < 0 || layout_b > 1 || (layout_a == 1 && layout_b == 0)) return MTVAF_ERR_ARG;
  if (splits < 1) splits = 1;
  if (splits > 1 && ((size_t)splits * M * N * sizeof(float) > workspace_bytes || !workspace)) return MTVAF_ERR_WORKSPACE;
  GemmArgsP a = {};
  a.Ap = static_cast<const unsigned char*>(Aplanes); a.Bp = static_cast<const unsigned char*>(Bplanes);
  a.a_plane = a_plane; a.a_row = a_row; a.a_kt = a_kt; a.a_col = a_col; a.b_plane = b_plane; a.b_row = b_row; a.b_kt = b_kt; a.b_col = b_col;
  a.bias = bias; a.aux = aux; a.M = M; a.N = N; a.K = K; a.ldaux = ldaux; a.epi = epi; a.accumulate = accumulate;
  int kc = ((K + splits - 1) / splits + 31) / 32 * 32;
  splits = (K + kc - 1) / kc;
  a.k_chunk = kc;
  if (splits > 1) { a.C = (float*)workspace; a.ldc = N; a.slab_stride = (long)M * N; }
  else { a.C = C; a.ldc = ldc; a.slab_stride = 0; }
  // Tile choice (128 x 128 here; 128 x 256 / 128 x 192 in gemm_f32pw.hip: the same bits): the launch runs ceil(tiles / 256 CUs) rounds of
  // one tile each, a tile = nk k-tiles + prologue and epilogue, in shader-clock ticks from the block-0 traces (tools/p16_wide_probe.py,
  // profiles/r06_p16_wide_tile_probe.txt): per k-tile 1850 / 3378 / 2560 (k-major B: 1975 / 3493), fixed 13 k / 21 k / 17 k (+ 5 k with
  // a plane-image epilogue).  Ties go to the larger tile.  (At 2432 rows: QKV forward 2 rounds of 342 / 1 of 171 / 1 of 228 -> 192;
  // FFN-1 forward 2 of 456 / 1 of 228 / 2 of 304 -> 256; at 4096 rows 3 of 768 / 2 of 384 -> 128 x 128 again.)
  const int wmask = p16_wide_mask();
  const long tm = M / 128, nkb = kc / 32;
  auto cost = [&](int bn_, long per_k, long fixed) {
    const long tiles = tm * (N / bn_) * splits, t = nkb * per_k + fixed + (c_planes ? 5000 : 0) * (bn_ / 128);
    // (a few rounds run in lockstep; many rounds drift apart and the CUs stay busy to the last partial round)
    return tiles <= 1024 ? ((tiles + 255) / 256) * t : (tiles * t + 128 * t) / 256;
  };
  const bool ok256 = (wmask & (layout_a ? 4 : layout_b ? 2 : 1)) && N % 256 == 0;
  const bool ok192 = (wmask & 1) && !(wmask & 16) && !layout_a && !layout_b && !c_planes && !ablate && !g_f32p_trace && N % 192 == 0;
  int bn = 128;
  long best = cost(128, layout_b ? 1975 : 1850, 13000);
  if ((wmask & 8) && (ok256 || ok192)) best = LONG_MAX;  // (tests: never the 128 x 128 tile where another one can serve)
  if (ok256) {
    const long c = cost(256, layout_b ? 3493 : 3378, 21000);
    if (c <= best) { best = c; bn = 256; }
  }
  if (ok192) {
    const long c = cost(192, 2560, 17000);
    if (c < best) { best = c; bn = 192; }
  }
  const bool wide = bn != 128;
  a.tiles_n = N / bn;
  a.Cpl = static_cast<unsigned char*>(c_planes);
  a.colpart = colpart;
  a.ablate = ablate;
  // (measured at the headline shape, same box, two runs each: G = 0 / 2 / 4 / 8 -> 3394 - 3397 / 3406 - 3421 / 3418 - 3418 / 3407 - 3410
  // sentences/s, all GEMMs 0.436 - 0.452 / 0.453 / 0.454 - 0.455 / 0.454 of the split-product peak)
  static const int walk_env = [] { const char* e = getenv("MTVAF_P16_WALK_G"); return e ? atoi(e) : 4; }();
  a.walk_g = walk_env;
  a.trace = g_f32p_trace;
  dim3 grid((unsigned)((M / 128) * a.tiles_n), 1, (unsigned)splits);
  // (launch profiler of gemm.hip: key 400 + 4 [k-major A] + 8 [k-major B] + 32 [128 x 256 tile] + 64 [128 x 192]; hip.kernel_symbol names the instantiation)
  const int key[8] = {400 + 4 * layout_a + 8 * layout_b + (wide ? 32 : 0) + (bn == 192 ? 64 : 0), layout_a, layout_b, 2, M, N, K, splits};
  const int rec = prof_begin(key, stream);
  const int rc = wide ? launch_gemm_f32p16w(a, layout_a, layout_b, grid, stream, bn) : launch_gemm_f32p16(a, layout_a, layout_b, grid, stream);
  prof_end(rec, stream);
  if (rc != MTVAF_OK) return rc;
  if (keep_slabs) {  // (the caller's next kernel adds the slabs itself, in the reduction's order)
    *keep_slabs = splits;
    return MTVAF_OK;
  }
  if (splits > 1) return launch_splitk_reduce((const float*)workspace, splits, C, M, N, ldc, bias, accumulate, epi, aux, ldaux, stream);
  return MTVAF_OK;
}

int mtvaf_gemm_f32p(int layout_a, const void* Aplanes, long a_plane, long a_row, long a_kt, long a_col, int layout_b, const void* Bplanes,
                    long b_plane, long b_row, long b_kt, long b_col, float* C, int ldc, int M, int N, int K, const float* bias, int epi,
                    float* aux, int ldaux, int accumulate, int splits, void* workspace, size_t workspace_bytes, int ablate,
                    hipStream_t stream) {
  return gemm_f32p_run(layout_a, Aplanes, a_plane, a_row, a_kt, a_col, layout_b, Bplanes, b_plane, b_row, b_kt, b_col, C, ldc, M, N, K, bias, epi,
                       aux, ldaux, accumulate, splits, workspace, workspace_bytes, ablate, stream, nullptr);
}

// mtvaf_gemm_f32p (unsplit) whose result is written as a tile-blocked PLANE IMAGE c_planes [N / 32][3][M][32] -- beside the fp32
// result (C != NULL) or instead of it (C == NULL: a tensor that only GEMMs read) -- and, optionally, its per-tile column sums
// colpart [M / 128][N] (finished by mtvaf_colsum_small: the bias gradient behind a dX product).
int mtvaf_gemm_f32p_ep(int layout_a, const void* Aplanes, long a_plane, long a_row, long a_kt, long a_col, int layout_b, const void* Bplanes,
                       long b_plane, long b_row, long b_kt, long b_col, float* C, int ldc, void* c_planes, float* colpart, int M, int N, int K,
                       const float* bias, int epi, float* aux, int ldaux, int accumulate, hipStream_t stream) {
  if (!c_planes) return MTVAF_ERR_ARG;
  return gemm_f32p_run(layout_a, Aplanes, a_plane, a_row, a_kt, a_col, layout_b, Bplanes, b_plane, b_row, b_kt, b_col, C, ldc, M, N, K, bias, epi,
                       aux, ldaux, C ? accumulate : 0, 1, nullptr, 0, 0, stream, nullptr, c_planes, colpart);
}

// mtvaf_gemm_f32p with a plain epilogue that leaves a split-K plan's slabs UNREDUCED (as mtvaf_gemm_f32_slabs): *splits_out = 1 -> C
// holds the result (+ bias, accumulate); s > 1 -> `workspace` holds s slabs [M][N], neither bias nor accumulate applied, C untouched.
int mtvaf_gemm_f32p_slabs(int layout_a, const void* Aplanes, long a_plane, long a_row, long a_kt, long a_col, int layout_b, const void* Bplanes,
                          long b_plane, long b_row, long b_kt, long b_col, float* C, int ldc, int M, int N, int K, const float* bias,
                          int accumulate, int splits, void* workspace, size_t workspace_bytes, int* splits_out, hipStream_t stream) {
  if (!splits_out) return MTVAF_ERR_ARG;
  return gemm_f32p_run(layout_a, Aplanes, a_plane, a_row, a_kt, a_col, layout_b, Bplanes, b_plane, b_row, b_kt, b_col, C, ldc, M, N, K, bias,
                       EPI_NONE, nullptr, 0, accumulate, splits, workspace, workspace_bytes, 0, stream, splits_out);
}

// Up to four weight-gradient products C_i [M_i][N_i] = A_i^T . B_i from plane images, A_i [K][M_i] and B_i [K][N_i] k-major (the
// reduction index K -- the token rows -- is shared), in ONE launch that walks the 128 x 128 tiles of all of them back to back,
// unsplit (research entry: what mtvaf_gemm_f32_dw_group's launch of the wave-specialised kernel would become with pre-split
// operands; no bias sums, no k-tile list).  strides: per product eight byte strides a_plane, a_row, a_kt, a_col, b_plane, b_row,
// b_kt, b_col (see mtvaf_gemm_f32p; tile-blocked images: row 64, k-tile 2048, col 12 x K x 64).  M_i, N_i % 128 == 0, K % 32 == 0.
static int f32p_dw_group_run(int n, const void* const* Aplanes, const void* const* Bplanes, const long* strides, float* const* C, const int* ldc,
                             const int* M, const int* N, int K, int njobs, const float* const* cs_src, const int* cs_rows, const int* cs_cols,
                             const int* cs_ld, float* const* cs_dst, hipStream_t stream) {
  if (n < 1 || n > 4 || !Aplanes || !Bplanes || !strides || !C || !ldc || !M || !N || K <= 0 || K % 32) return MTVAF_ERR_ARG;
  GemmArgsP a = {};
  a.K = K; a.k_chunk = K; a.slab_stride = 0; a.epi = EPI_NONE; a.ngrp = n;
  long tiles = 0;
  bool wide = (p16_wide_mask() & 4) != 0;
  for (int i = 0; i < n && wide; ++i) wide = N[i] > 0 && N[i] % 256 == 0;
  const int bn = wide ? 256 : 128;
  for (int i = 0; i < n; ++i) {
    if (!Aplanes[i] || !Bplanes[i] || !C[i] || M[i] <= 0 || N[i] <= 0 || M[i] % 128 || N[i] % 128 || ldc[i] % 4) return MTVAF_ERR_SHAPE;
    if ((((uintptr_t)Aplanes[i] | (uintptr_t)Bplanes[i] | (uintptr_t)C[i]) & 15)) return MTVAF_ERR_ALIGN;
    for (int j = 0; j < 8; ++j)
      if (strides[8 * i + j] & 15) return MTVAF_ERR_ALIGN;
    GemmArgsP::Prob& q = a.grp[i];
    q.Ap = static_cast<const unsigned char*>(Aplanes[i]); q.Bp = static_cast<const unsigned char*>(Bplanes[i]);
    q.a_plane = strides[8 * i]; q.a_row = strides[8 * i + 1]; q.a_kt = strides[8 * i + 2]; q.a_col = strides[8 * i + 3];
    q.b_plane = strides[8 * i + 4]; q.b_row = strides[8 * i + 5]; q.b_kt = strides[8 * i + 6]; q.b_col = strides[8 * i + 7];
    q.C = C[i]; q.ldc = ldc[i]; q.tiles_n = N[i] / bn;
    a.grp_tile_begin[i] = (int)tiles;
    tiles += (long)(M[i] / 128) * (N[i] / bn);
  }
  for (int i = n; i < 5; ++i) a.grp_tile_begin[i] = (int)tiles;  // (absent products own no tiles)
  a.M = M[0]; a.N = N[0]; a.tiles_n = N[0] / bn;
  a.cs_tile0 = (int)tiles;
  long blocks = tiles;
  if (njobs < 0 || njobs > 8 || (njobs && (!cs_src || !cs_rows || !cs_cols || !cs_ld || !cs_dst))) return MTVAF_ERR_ARG;
  a.cs_n = njobs;
  int cb = 0;
  for (int j = 0; j < njobs; ++j) {
    if (!cs_src[j] || !cs_dst[j] || cs_rows[j] <= 0 || cs_cols[j] <= 0 || cs_ld[j] < cs_cols[j]) return MTVAF_ERR_ARG;
    a.cs[j].src = cs_src[j]; a.cs[j].dst = cs_dst[j]; a.cs[j].rows = cs_rows[j]; a.cs[j].cols = cs_cols[j]; a.cs[j].ld = cs_ld[j];
    a.cs_blk0[j] = cb;
    cb += (cs_cols[j] + 63) / 64;
  }
  for (int j = njobs; j < 9; ++j) a.cs_blk0[j] = cb;
  blocks += cb;
  const int key[8] = {400 + 4 + 8 + 16 + (wide ? 32 : 0), 1, 1, 2, (int)(tiles * 128 * bn / 768), 768, K, 1};  // (+16: the GROUP instantiation; M x 768 = all outputs)
  const int rec = prof_begin(key, stream);
  const int rc = wide ? launch_gemm_f32p16w_group(a, dim3((unsigned)blocks, 1, 1), stream) : launch_gemm_f32p16_group(a, dim3((unsigned)blocks, 1, 1), stream);
  prof_end(rec, stream);
  return rc;
}

int mtvaf_gemm_f32p_dw_group(int n, const void* const* Aplanes, const void* const* Bplanes, const long* strides, float* const* C, const int* ldc,
                             const int* M, const int* N, int K, hipStream_t stream) {
  return f32p_dw_group_run(n, Aplanes, Bplanes, strides, C, ldc, M, N, K, 0, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}
// ... with up to eight COLUMN-SUM jobs as extra blocks of the same launch: job j sums the fp32 matrix cs_src[j] [cs_rows[j]][cs_cols[j]]
// (leading dimension cs_ld[j]) over its rows into cs_dst[j] [cs_cols[j]], fixed summation order.  The small reductions of a layer's
// backward pass -- the QKV bias gradient (column sums of dQ|dK|dV), the FFN-1 bias gradient (the GELU' epilogue's per-tile sums), the
// two LayerNorm-backward finishes (dgamma, dbeta, dense-bias gradient from 512 partial rows each) -- fill CUs the last round of tiles
// leaves idle instead of five launches of their own.
int mtvaf_gemm_f32p_dw_group_colsum(int n, const void* const* Aplanes, const void* const* Bplanes, const long* strides, float* const* C,
                                    const int* ldc, const int* M, const int* N, int K, int njobs, const float* const* cs_src,
                                    const int* cs_rows, const int* cs_cols, const int* cs_ld, float* const* cs_dst, hipStream_t stream) {
  if (njobs < 1) return MTVAF_ERR_ARG;
  return f32p_dw_group_run(n, Aplanes, Bplanes, strides, C, ldc, M, N, K, njobs, cs_src, cs_rows, cs_cols, cs_ld, cs_dst, stream);
}

}  // extern "C"
