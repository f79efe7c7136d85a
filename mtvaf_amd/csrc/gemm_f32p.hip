// fp32 GEMM on the bf16 matrix pipe with BOTH operands given as plane images (round 4): every fp32 tensor that enters a
// dense product of the path exists, beside its fp32 form, as the three bf16 planes of its split  x = x1 + x2 + x3
// (mtvaf_f32_split_planes, or written by the kernel that produces the tensor: [3][...] bf16, plane q of element i at
// i + q * stride, the fp32 tensor's own row-major layout and leading dimension).  A product a.b is the six MFMA products
//     a3 b1 + a1 b3 + a2 b2 + a2 b1 + a1 b2 + a1 b1      (v_mfma_f32_32x32x16_bf16, fp32 accumulation, smallest terms first)
// exactly as in gemm_f32x3.hip -- the same planes, the same sequence, the same k order: results are bit-identical to the
// kernels that split fp32 tiles in-kernel.
//
// Why.  The wave-specialised kernel of gemm_f32x3.hip splits every operand tile in every block that reads it: ~200 vector
// instructions per producer wave and 32-deep k-tile, beside the 36-48 MFMAs of the consumer wave on the same SIMD.  Traced per
// wave and k-tile (tools/x3_trace.py, mtvaf_f32x3_trace): a k-tile takes 2330-2580 cycles where its MFMAs need 1150-1540; the
// producers need 1980-2200 cycles of it (their vector instructions issue at half rate beside the matrix stream), the consumers
// 1630-2030 (42-45 cycles per MFMA with the producers on their SIMD) and then wait 560-690 cycles at the tile barrier.  Here
// nothing is split in the k-loop: plane tiles travel L2 -> LDS by global_load_lds_dwordx4 (no registers, no vector work, no
// ds_write), through a 3-stage ring (two k-tiles in flight behind a counted vmcnt, one raw s_barrier per k-tile), and the four
// waves of a block do nothing but fragment reads and MFMAs.  48 KiB (40 KiB at BN = 96) of planes per 32-deep k-tile against 32
// KiB of fp32: more bytes from L2, no instruction issue.  The split itself happens ONCE per tensor instead of once per block
// that reads a tile of it (a [4096 x 768] activation: 6-24 blocks per tile forward, again in the weight-gradient product).
//
// LDS images (lane-linear LDS-DMA: both swizzles sit on the per-lane SOURCE address and again on the read):
//   KC operand (reduction index contiguous: x[m][k], W[n][k] forward, dY[m][n] as A of dX): per plane R rows x 64 B (32 k),
//      unpadded; 16-byte chunk c of row r at c ^ ((r >> 2) & 3): conflict-free for the ds_read_b128 fragments of the 32x32x16
//      MFMA (lane -> row l & 31, chunk 2 ks + (l >> 5)).
//   KM operand (reduction index is the ROW: W[n][k] as B of dX; dY and x as A / B of dW): per plane 32 rows x 256 B (128
//      columns), chunk c of row r at c ^ km_swz(r), fragments by two ds_read_b64_tr_b16 (gemm_bf16x.hip's image, 32 rows deep);
//      96 columns: 192-byte rows, unswizzled (consecutive rows start 48 banks apart).
// Tiles: 128 x 128 (2 x 2 waves of 64 x 64) and 128 x 96 (4 x 1 waves of 32 x 96: whole rounds of the 256 CUs for the N = 768 /
// 2304 results of 4096 token rows).  Epilogues of the path (bias, bias + erf-GELU with the fp32 pre-activation saved, x GELU',
// tanh, x (1 - t^2), accumulate), deterministic split-K slabs, k-tile lists (32-row tiles of k-major operands), and --
// optionally -- the RESULT's own plane image written beside / instead of the fp32 result (the operand of the next product is
// born split: the GELU output, the GELU' product).
#include <type_traits>

#include "gemm_bf16x.h"

namespace mtvaf {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned frag_t __attribute__((ext_vector_type(4)));  // (fragments as four dwords: see gemm_f32x3.hip)

namespace f32p {

// the RNE three-way split of gemm_f32x3.hip (kept textually identical: the planes must be the same bits)
__device__ __forceinline__ unsigned cvt_pk(const f32x2 v) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ f32x2 widen(const unsigned pk) {
  return f32x2{__builtin_bit_cast(float, pk << 16), __builtin_bit_cast(float, pk & 0xffff0000u)};
}
__device__ __forceinline__ f32x2 resid2(const f32x2 x, const unsigned hpk) { return x - widen(hpk); }
__device__ __forceinline__ void split3_pair(const f32x2 x, unsigned& h, unsigned& m, unsigned& l) {
  h = cvt_pk(x);
  const f32x2 r = resid2(x, h);
  m = cvt_pk(r);
  l = cvt_pk(resid2(r, m));
}
__device__ __forceinline__ void split3(const f32x4 x, bf16x4& h, bf16x4& m, bf16x4& l) {
  unsigned h0, m0, l0, h1, m1, l1;
  split3_pair(f32x2{x.x, x.y}, h0, m0, l0);
  split3_pair(f32x2{x.z, x.w}, h1, m1, l1);
  h = __builtin_bit_cast(bf16x4, uint2{h0, h1});
  m = __builtin_bit_cast(bf16x4, uint2{m0, m1});
  l = __builtin_bit_cast(bf16x4, uint2{l0, l1});
}

}  // namespace f32p

// DW: four more waves (one per SIMD) that do nothing but issue the LDS-DMA requests.  Why: a global_load_lds request costs the
// issuing wave 80-95 cycles beside the matrix stream (tools/x3_trace.py), and a wave issues in order -- with the requests in the
// MFMA waves a k-tile took 48 x ~42 + 12 x ~85 = ~3100 cycles; in waves of their own they overlap the MFMAs.
template <bool A_KM, bool B_KM, bool KLIST, int BN, bool DW>
__global__ __launch_bounds__(DW ? 512 : 256, 1) void gemm_f32p_kernel(GemmArgs p) {
  static_assert(BN == 128 || BN == 96, "tiles: 128 x 128 and 128 x 96");
  static_assert(!KLIST || (A_KM && B_KM), "the k-tile list addresses rows of k-major operands");
  constexpr int BM = 128, BK = 32, NT = DW ? 512 : 256, NS = 3, NW = 4;
  constexpr int WM = BN == 128 ? 2 : 4, WN = BN == 128 ? 2 : 1;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int AP_B = BM * 64;                       // bytes of one A plane tile (KC: 128 rows x 64 B; KM: 32 rows x 256 B)
  constexpr int BP_B = BN * 64;                       // ... of one B plane tile
  constexpr int A_B = 3 * AP_B;                       // 24 pieces of 1 KiB
  constexpr int BPIECES = 3 * BP_B / 1024;            // 24 / 18
  constexpr int IA = A_B / 1024 / NW;                 // pieces per wave: 6
  constexpr int IB = (BPIECES + NW - 1) / NW;         // 6 / 5 (BN = 96: two dummy pieces keep the count equal in every wave)
  constexpr int B_B = IB * NW * 1024;                 // 24576 / 20480 (the last 2 KiB receive the dummy pieces)
  constexpr int STAGE_B = A_B + B_B;                  // 49152 / 45056
  constexpr int KM_SLICE_A = 4096, KM_SLICE_B = BN == 128 ? 4096 : 3072;  // bytes of 16 k-rows of a KM plane tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];  // the ONLY LDS object

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool dma_wave = DW && wave >= 4;      // (wave-uniform)
  const int pw = DW ? (wave & 3) : wave;      // owner index of this wave's DMA pieces (DW: only the DMA waves issue them)
  const int wm = (wave & 3) / WN, wn = (wave & 3) % WN;
  const int li = lane & 31, h = lane >> 5;
  const int bid = p.tile_walk > 0 ? xcd_remap_cols(blockIdx.x, gridDim.x, p.tiles_n, p.tile_walk) : xcd_remap(blockIdx.x, gridDim.x);
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

  // ---- per-lane DMA source addresses (bytes): piece I of an operand = plane I / PPP, sub-piece I % PPP; its LDS slot
  // (16-byte unit) s = (I % PPP) * 64 + lane receives the source chunk  position ^ swizzle(row)
  const unsigned char* pa[IA];
  const unsigned char* pb[IB];
  const __bf16* Ap = reinterpret_cast<const __bf16*>(p.Ap);
  const __bf16* Bp = reinterpret_cast<const __bf16*>(p.Bp);
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int I = pw * IA + i, plane = I / (AP_B / 1024), slot = (I % (AP_B / 1024)) * 64 + lane;
    if constexpr (!A_KM) {
      const int row = slot >> 2, cp = slot & 3;
      pa[i] = reinterpret_cast<const unsigned char*>(Ap + plane * p.ap_stride + (long)(m0 + row) * p.lda + kbeg) + ((cp ^ ((row >> 2) & 3)) << 4);
    } else {
      const int row = slot >> 4, cp = slot & 15;
      pa[i] = reinterpret_cast<const unsigned char*>(Ap + plane * p.ap_stride + (long)(kbeg + row) * p.lda + m0) + ((cp ^ km_swz(row)) << 4);
    }
  }
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    int I = pw * IB + i;
    if (I >= BPIECES) I -= BPIECES;  // (BN = 96) dummy piece: piece I - 18 again, into the pad behind the planes
    const int plane = I / (BP_B / 1024), slot = (I % (BP_B / 1024)) * 64 + lane;
    if constexpr (!B_KM) {
      const int row = slot >> 2, cp = slot & 3;
      pb[i] = reinterpret_cast<const unsigned char*>(Bp + plane * p.bp_stride + (long)(n0 + row) * p.ldb + kbeg) + ((cp ^ ((row >> 2) & 3)) << 4);
    } else if constexpr (BN == 128) {
      const int row = slot >> 4, cp = slot & 15;
      pb[i] = reinterpret_cast<const unsigned char*>(Bp + plane * p.bp_stride + (long)(kbeg + row) * p.ldb + n0) + ((cp ^ km_swz(row)) << 4);
    } else {  // 192-byte rows, unswizzled
      const int o = slot << 4, row = o / 192, cb = o % 192;
      pb[i] = reinterpret_cast<const unsigned char*>(Bp + plane * p.bp_stride + (long)(kbeg + row) * p.ldb + n0) + cb;
    }
  }
  const long stepA = A_KM ? (long)BK * p.lda * 2 : BK * 2;
  const long stepB = B_KM ? (long)BK * p.ldb * 2 : BK * 2;
  int issued = 0;
  int kt_next = (KLIST && nk > 0) ? p.klist[lbeg] : 0;  // fetched one issue ahead (a uniform scalar load)
  auto issue = [&](int stage) __attribute__((always_inline)) {
    unsigned char* sa = smem_b + stage * STAGE_B;
    unsigned char* sb = sa + A_B;
    long oa = 0, ob = 0;
    if constexpr (KLIST) {
      oa = (long)kt_next * stepA;
      ob = (long)kt_next * stepB;
      ++issued;
      kt_next = p.klist[lbeg + min(issued, nk - 1)];
    }
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      const int I = pw * IA + i;
      glds16x(pa[i] + oa, sa + I * 1024);
      if constexpr (!KLIST) pa[i] += stepA;
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const int I = pw * IB + i;  // (dummy pieces land at 3 * BP_B + ..., behind the planes)
      glds16x(pb[i] + ob, sb + I * 1024);
      if constexpr (!KLIST) pb[i] += stepB;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- fragment read offsets inside a plane tile (bytes)
  int offA[TM][2], offB[TN][2];
  {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if constexpr (!A_KM) {
        const int row = (wm * TM + i) * 32 + li;
        offA[i][0] = row * 64;
        offA[i][1] = (row >> 2) & 3;
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
      if constexpr (!B_KM) {
        const int col = (wn * TN + j) * 32 + li;
        offB[j][0] = col * 64;
        offB[j][1] = (col >> 2) & 3;
      } else if constexpr (BN == 128) {
        const int ns = wn * TN + j;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          offB[j][jj] = 256 * (8 * (g >> 1) + 4 * jj + q) +
                        16 * ((((ns ^ q) & 3) << 2) | ((2 * (g & 1) + (pp >> 1)) ^ ((2 * (g >> 1) + jj) & 3))) + 8 * (pp & 1);
      } else {
        const int nl = wn * TN + j;  // 0..2
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) offB[j][jj] = 192 * (8 * (g >> 1) + 4 * jj + q) + nl * 64 + 32 * (g & 1) + 8 * pp;
      }
    }
  }

  frag_t fa0[3][TM], fb0[3][TN], fa1[3][TM], fb1[3][TN];
  auto rdf = [&](const unsigned char* a, const unsigned char* b, int ks, frag_t (&fa)[3][TM], frag_t (&fb)[3][TN]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (!A_KM) fa[q][i] = *reinterpret_cast<const frag_t*>(a + q * AP_B + offA[i][0] + (((2 * ks + h) ^ offA[i][1]) << 4));
        else fa[q][i] = __builtin_bit_cast(frag_t, tr_read8(a + q * AP_B + offA[i][0] + KM_SLICE_A * ks, a + q * AP_B + offA[i][1] + KM_SLICE_A * ks));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (!B_KM) fb[q][j] = *reinterpret_cast<const frag_t*>(b + q * BP_B + offB[j][0] + (((2 * ks + h) ^ offB[j][1]) << 4));
        else fb[q][j] = __builtin_bit_cast(frag_t, tr_read8(b + q * BP_B + offB[j][0] + KM_SLICE_B * ks, b + q * BP_B + offB[j][1] + KM_SLICE_B * ks));
      }
    }
  };
  auto mm = [&](const frag_t (&fa)[3][TM], const frag_t (&fb)[3][TN]) __attribute__((always_inline)) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // smallest terms first (as gemm_f32x3.hip)
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[PA[t]][i]), __builtin_bit_cast(bf16x8, fb[PB[t]][j]), acc[i][j], 0, 0, 0);
  };

  if (!DW || dma_wave) {
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
      if (s < nk) issue(s);
  }
  // One step = the barrier that publishes a tile + the request for the tile two ahead (into the stage the previous tile left).
  // No MFMA sits inside a conditional: the accumulators never pass through a phi (a conditional product group made the
  // compiler copy all 64 accumulator registers twice per k-tile).
  auto publish = [&](int kt, int st) __attribute__((always_inline)) {
    if constexpr (!DW) {
      if (kt + 1 < nk) wait_vm<IA + IB>();  // this wave's pieces of tile kt have landed; tile kt + 1 stays in flight
      else wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();  // ... and everybody else's; every wave is done reading tile kt - 1
    asm volatile("" ::: "memory");
    if constexpr (!DW) {
      if (kt + NS - 1 < nk) {
        int si = st + NS - 1;
        if (si >= NS) si -= NS;
        issue(si);
      }
    }
  };
  if (dma_wave) {
    int st = 0;
    for (int kt = 0; kt < nk; ++kt) {  // the same barrier count as the MFMA waves: one per k-tile
      if (kt + 1 < nk) wait_vm<IA + IB>();
      else wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + NS - 1 < nk) {
        int si = st + NS - 1;
        if (si >= NS) si -= NS;
        issue(si);
      }
      st = st + 1 == NS ? 0 : st + 1;
    }
  } else if (nk > 0) {
    publish(0, 0);
    rdf(smem_b, smem_b + A_B, 0, fa0, fb0);
    int st = 0;
    for (int kt = 0; kt < nk; ++kt) {
      // fragment reads run ONE k-slice ahead of the MFMAs that consume them, across the tile barrier too: slice 1 of tile kt is
      // read here and multiplied behind the barrier that publishes tile kt + 1, while slice 0 of that tile is in flight
      const unsigned char* a = smem_b + st * STAGE_B;
      rdf(a, a + A_B, 1, fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      mm(fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the fragments of tile kt are in registers before its stage is refilled
      st = st + 1 == NS ? 0 : st + 1;
      if (kt + 1 < nk) {
        publish(kt + 1, st);
        const unsigned char* an = smem_b + st * STAGE_B;
        rdf(an, an + A_B, 0, fa0, fb0);
      }
      __builtin_amdgcn_sched_barrier(0);
      mm(fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: 64 result rows per pass through the LDS (row-major image, 4 columns per lane, 16-byte stores) ----
  {
    constexpr int LDE = BN + 4, RP = 64, C4 = BN / 4;
    static_assert((size_t)RP * LDE * sizeof(float) <= (size_t)STAGE_B, "the epilogue image fits one stage");
    float* smem = reinterpret_cast<float*>(smem_b);
    float* C = p.C ? p.C + (long)blockIdx.z * p.slab_stride : nullptr;
    __bf16* Cp = reinterpret_cast<__bf16*>(p.Cp);
    const bool split = gridDim.z > 1;
    const int wrow = wm * TM * 32;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      __syncthreads();  // (every DMA piece has landed: the last publish waited for vmcnt(0))
      if (!dma_wave && wrow / RP == pass) {
        const int rofs = wrow % RP;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              smem[(rofs + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * LDE + (wn * TN + j) * 32 + li] = acc[i][j][r];
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
          if (Cp) {  // the result's own planes: the next product's operand is born split
            bf16x4 hh, mm_, ll;
            f32p::split3(v, hh, mm_, ll);
            __bf16* d = Cp + row * p.ldcp + col;
            *reinterpret_cast<bf16x4*>(d) = hh;
            *reinterpret_cast<bf16x4*>(d + p.cp_stride) = mm_;
            *reinterpret_cast<bf16x4*>(d + 2 * p.cp_stride) = ll;
          }
        }
        if (C) *reinterpret_cast<f32x4*>(C + row * p.ldc + col) = v;
      }
    }
  }
}

template <int BN, bool DW>
static int launch_f32p_t(const GemmArgs& a, int la, int lb, dim3 grid, hipStream_t st) {
  constexpr size_t smem = (size_t)3 * (3 * 128 * 64 + (BN == 128 ? 24576 : 20480));
#define MTVAF_F32P(AK, BKM, KL)                                                                                      \
  do {                                                                                                               \
    auto kern = gemm_f32p_kernel<AK, BKM, KL, BN, DW>;                                                               \
    static bool attr_set = false;                                                                                    \
    if (!attr_set) {                                                                                                 \
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);  \
      if (e != hipSuccess) return (int)e;                                                                            \
      attr_set = true;                                                                                               \
    }                                                                                                                \
    hipLaunchKernelGGL(kern, grid, dim3(DW ? 512 : 256), smem, st, a);                                               \
  } while (0)
  if (la == 0 && lb == 0) MTVAF_F32P(false, false, false);
  else if (la == 0 && lb == 1) MTVAF_F32P(false, true, false);
  else if (la == 1 && lb == 1) { if (a.klist) MTVAF_F32P(true, true, true); else MTVAF_F32P(true, true, false); }
  else return MTVAF_ERR_ARG;
#undef MTVAF_F32P
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

// Both operands as plane images (a.Ap / a.Bp): bn = 128 or 96.  Called by gemm.hip's dispatcher (whole tiles, wide epilogue).
int launch_gemm_f32p(int bn, const GemmArgs& a, int la, int lb, dim3 grid, hipStream_t st) {
  if (!a.wide || !a.Ap || !a.Bp) return MTVAF_ERR_ALIGN;
  static const bool dw = [] { const char* e = getenv("MTVAF_F32P_DMA_WAVES"); return !(e && atoi(e) == 0); }();
  if (dw) return bn == 128 ? launch_f32p_t<128, true>(a, la, lb, grid, st) : launch_f32p_t<96, true>(a, la, lb, grid, st);
  return bn == 128 ? launch_f32p_t<128, false>(a, la, lb, grid, st) : launch_f32p_t<96, false>(a, la, lb, grid, st);
}

}  // namespace mtvaf
