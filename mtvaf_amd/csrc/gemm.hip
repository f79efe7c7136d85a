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
#include "common.h"

namespace mtvaf {

enum { EPI_NONE = 0, EPI_GELU = 1, EPI_TANH = 2, EPI_DGELU = 3, EPI_DTANH = 4 };

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  float* aux;
  int M, N, K;
  int lda, ldb, ldc, ldaux;
  int k_chunk;
  long slab_stride;
  int epi, a_vec, b_vec, accumulate;
  int tiles_n;
};

__device__ __forceinline__ f32x4 ld4(const float* __restrict__ p, int nvalid, bool vec) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (nvalid >= 4 && vec) {
    v = *reinterpret_cast<const f32x4*>(p);
  } else {
    if (nvalid > 0) v.x = p[0];
    if (nvalid > 1) v.y = p[1];
    if (nvalid > 2) v.z = p[2];
    if (nvalid > 3) v.w = p[3];
  }
  return v;
}

// FAST: M % BM == 0, N % BN == 0, every k-chunk a multiple of BK, 16-byte aligned operands -> no bounds
// checks or scalar tails anywhere in the main loop.
template <int BM, int BN, int WM, int WN, int BK, bool A_KM, bool B_KM, bool FAST>
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

  const int bid = blockIdx.x;
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
            ra[i] = *reinterpret_cast<const f32x4*>(p.A + (long)row * p.lda + k);
          } else {
            const int nv = row < p.M ? max(0, min(4, kend - k)) : 0;
            ra[i] = ld4(p.A + (long)row * p.lda + k, nv, p.a_vec);
          }
        } else {
          const int k = idx / (BM / 4), c = (idx % (BM / 4)) * 4;
          const int row = m0 + c;
          if (FAST) {
            ra[i] = *reinterpret_cast<const f32x4*>(p.A + (long)(k0 + k) * p.lda + row);
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
            rb[i] = *reinterpret_cast<const f32x4*>(p.B + (long)col * p.ldb + k);
          } else {
            const int nv = col < p.N ? max(0, min(4, kend - k)) : 0;
            rb[i] = ld4(p.B + (long)col * p.ldb + k, nv, p.b_vec);
          }
        } else {
          const int k = idx / (BN / 4), c = (idx % (BN / 4)) * 4;
          const int col = n0 + c;
          if (FAST) {
            rb[i] = *reinterpret_cast<const f32x4*>(p.B + (long)(k0 + k) * p.ldb + col);
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
  float* C = p.C + (long)blockIdx.z * p.slab_stride;
  const bool split = gridDim.z > 1;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + li;
      if (!FAST && col >= p.N) continue;
      const float bv = (!split && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (!FAST && row >= p.M) continue;
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

// dst[r][c] (ld ldd) = (accumulate ? dst : 0) + sum_z slabs[z][r][c] (+ bias[c])
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, int splits, long slab_stride, float* dst,
                                     int rows, int cols, int ldd, const float* bias, int accumulate) {
  const long n = (long)rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += slabs[z * slab_stride + i];
    if (bias) s += bias[c];
    float* d = dst + (long)r * ldd + c;
    if (accumulate) s += *d;
    *d = s;
  }
}

struct TileCfg { int bm, bn, bk; };
static const TileCfg kCfgs[] = {{128, 128, 16}, {128, 96, 16}, {128, 288, 16}, {64, 64, 16}, {128, 64, 16},
                                {128, 128, 32}, {128, 96, 32}, {128, 192, 16}, {128, 192, 32}};
constexpr int kNumCfgs = 9;

template <int BM, int BN, int WM, int WN, int BK, bool FAST>
static int launch_l(const GemmArgs& a, int la, int lb, dim3 grid, hipStream_t st) {
  constexpr int KC_LD = BK + 4;
  const int asz = la ? BK * BM : BM * KC_LD, bsz = lb ? BK * BN : BN * KC_LD;
  const size_t smem = (size_t)2 * (asz + bsz) * sizeof(float);
  dim3 block(WM * WN * 64);
  if (la == 0 && lb == 0)
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, BK, false, false, FAST>), grid, block, smem, st, a);
  else if (la == 0 && lb == 1)
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, BK, false, true, FAST>), grid, block, smem, st, a);
  else if (la == 1 && lb == 1)
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, BK, true, true, FAST>), grid, block, smem, st, a);
  else if (!FAST)  // KM x KC is not produced by the path; only the checked kernel carries it
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, BK, true, false, false>), grid, block, smem, st, a);
  else
    return MTVAF_ERR_ARG;
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

template <int BM, int BN, int WM, int WN, int BK>
static int launch_cfg(const GemmArgs& a, int la, int lb, dim3 grid, bool fast, hipStream_t st) {
  if (fast && !(la == 1 && lb == 0)) return launch_l<BM, BN, WM, WN, BK, true>(a, la, lb, grid, st);
  return launch_l<BM, BN, WM, WN, BK, false>(a, la, lb, grid, st);
}

static inline long cdiv(long a, long b) { return (a + b - 1) / b; }

// Cost model (units: fp32 MFMA cycles of one CU).  The MFMA pipe of a CU is shared by its resident
// blocks, so time ~ rounds over the 256 CUs x work per tile / efficiency of that tile shape (calibrated
// with tools/gemm_sweep.py on MI355X at M = 4096), plus, for split-K, the slab write + ordered reduce.
static void choose(int M, int N, int K, int allow_split, int* cfg_out, int* splits_out) {
  //                                  128x128 128x96 128x288 64x64 128x64 128x128x32 128x96x32 128x192 128x192x32
  static const double eff[kNumCfgs] = {0.80,   0.86,  0.72,   0.45, 0.80,  0.70,      1.00,     0.92,   0.78};
  double best = 1e300;
  int bc = 0, bs = 1;
  for (int c = 0; c < kNumCfgs; ++c) {
    const int bm = kCfgs[c].bm, bn = kCfgs[c].bn, bk = kCfgs[c].bk;
    const long tiles = cdiv(M, bm) * cdiv(N, bn);
    const int max_s = allow_split ? 16 : 1;
    for (int s = 1; s <= max_s; ++s) {
      if (s > 1 && K / s < 256) break;
      const long rounds = cdiv(tiles * s, 256);
      const double kc = (double)cdiv(cdiv(K, s), bk) * bk;
      // per tile: (bm*bn*kc*2 flop) / (256 flop/clk/CU) cycles at 100 %
      double cost = (double)rounds * bm * bn * kc / 128.0 / eff[c];
      cost += 3000.0;  // fill/drain + launch
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

// Reports the tile configuration / split count the heuristic would pick (for profiling tools).
// tile (BMxBNxBK, 256 threads): 0 128x128x16, 1 128x96x16, 2 128x288x16, 3 64x64x16, 4 128x64x16,
// 5 128x128x32, 6 128x96x32, 7 128x192x16, 8 128x192x32.
int mtvaf_gemm_f32_plan(int M, int N, int K, int allow_split, int* cfg, int* splits) {
  if (M <= 0 || N <= 0 || K <= 0 || !cfg || !splits) return MTVAF_ERR_ARG;
  choose(M, N, K, allow_split, cfg, splits);
  return MTVAF_OK;
}

// C[M,N] = opA[M,K] . opB[K,N] (+bias) with fused epilogue.  layout_a / layout_b: 0 = KC, 1 = KM.
// epi: 0 none, 1 bias+GELU (pre-activation stored to aux), 2 bias+tanh, 3 dGELU (multiply by
// gelu'(aux)), 4 dtanh (multiply by 1-aux^2).  accumulate: C += result.  allow_split: permit a
// deterministic split-K (slabs in workspace + ordered reduction); cfg/splits < 0 = heuristic.
int mtvaf_gemm_f32(int layout_a, int layout_b, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                   int M, int N, int K, const float* bias, int epi, float* aux, int ldaux, int accumulate,
                   int allow_split, void* workspace, size_t workspace_bytes, int cfg, int splits,
                   hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0) return MTVAF_ERR_SHAPE;
  if (!A || !B || !C) return MTVAF_ERR_ARG;
  if ((epi == EPI_GELU || epi == EPI_DGELU || epi == EPI_DTANH) && !aux) return MTVAF_ERR_ARG;
  if (layout_a < 0 || layout_a > 1 || layout_b < 0 || layout_b > 1) return MTVAF_ERR_ARG;
  int c_auto, s_auto;
  choose(M, N, K, allow_split && epi == EPI_NONE, &c_auto, &s_auto);
  if (cfg < 0 || cfg >= kNumCfgs) cfg = c_auto;
  if (splits <= 0) splits = s_auto;
  if (!(allow_split && epi == EPI_NONE)) splits = 1;
  if (splits > 1 && (size_t)splits * M * N * sizeof(float) > workspace_bytes) {
    splits = (int)(workspace_bytes / ((size_t)M * N * sizeof(float)));
    if (splits < 1) splits = 1;
  }
  GemmArgs a;
  a.A = A; a.B = B; a.bias = bias; a.aux = aux;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldaux = ldaux;
  a.epi = epi; a.accumulate = accumulate;
  a.a_vec = (lda % 4 == 0) && (((uintptr_t)A & 15) == 0);
  a.b_vec = (ldb % 4 == 0) && (((uintptr_t)B & 15) == 0);
  const int bk = kCfgs[cfg].bk;
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
  const bool fast = (M % bm == 0) && (N % bn == 0) && (K % bk == 0) && a.a_vec && a.b_vec && (kc % bk == 0);
  int rc;
  switch (cfg) {
    case 0: rc = launch_cfg<128, 128, 2, 2, 16>(a, layout_a, layout_b, grid, fast, stream); break;
    case 1: rc = launch_cfg<128, 96, 4, 1, 16>(a, layout_a, layout_b, grid, fast, stream); break;
    case 2: rc = launch_cfg<128, 288, 4, 1, 16>(a, layout_a, layout_b, grid, fast, stream); break;
    case 3: rc = launch_cfg<64, 64, 2, 2, 16>(a, layout_a, layout_b, grid, fast, stream); break;
    case 4: rc = launch_cfg<128, 64, 4, 1, 16>(a, layout_a, layout_b, grid, fast, stream); break;
    case 5: rc = launch_cfg<128, 128, 2, 2, 32>(a, layout_a, layout_b, grid, fast, stream); break;
    case 6: rc = launch_cfg<128, 96, 4, 1, 32>(a, layout_a, layout_b, grid, fast, stream); break;
    case 7: rc = launch_cfg<128, 192, 2, 2, 16>(a, layout_a, layout_b, grid, fast, stream); break;
    default: rc = launch_cfg<128, 192, 2, 2, 32>(a, layout_a, layout_b, grid, fast, stream); break;
  }
  if (rc != MTVAF_OK) return rc;
  if (splits > 1) {
    const long n = (long)M * N;
    int blocks = (int)std::min<long>(cdiv(n, 256), 2048);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, (const float*)workspace, splits,
                       (long)M * N, C, M, N, ldc, bias, accumulate);
    MTVAF_LAUNCH_CHECK();
  }
  return MTVAF_OK;
}

}  // extern "C"
