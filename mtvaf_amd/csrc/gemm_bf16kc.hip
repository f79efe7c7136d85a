// bf16-operand GEMM for the mixed-precision configurations (BASELINE configs 3-4): C[M,N] (fp32) = A[M,K] . B[N,K]^T
// with BOTH operands stored as bf16, reduction index contiguous (KC x KC).  The fp32-operand bf16 kernel
// (gemm_bf16.hip) is bound by staging fp32 tiles through registers (per-CU L2->LDS bytes); here tiles go
// HBM -> LDS with global_load_lds_dwordx4 exactly like the fp32 DMA kernel -- a 64-element bf16 k-tile row is the
// same 128 bytes as a 32-element fp32 row, so the LDS image, the XOR swizzle (chunk ^ ((row >> 1) & 7)) and the
// fragment addressing carry over: lane (r = l & 31, h = l >> 5) reads the 16-byte chunk 2*ks + h of row r, i.e. the
// 8 consecutive k of one v_mfma_f32_32x32x16_bf16 operand.  The engine prepares the operands once per use with
// mtvaf_cast_bf16 (row-major and, for the weight-gradient products, transposed copies), which turns every product
// of the path (forward, dX = dY . W with W^T prepared, dW = dY^T . X with both transposed) into this one layout.
//
// 2-stage ring (57-67 KB), two blocks per CU; accumulators, epilogues and deterministic split-K as in gemm.hip.
#include "gemm_common.h"

#include <algorithm>

namespace mtvaf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

int launch_splitk_reduce(const float* slabs, int splits, float* C, int M, int N, int ldc, const float* bias, int accumulate,
                         int epi, const float* aux, int ldaux, hipStream_t stream);  // gemm.hip
int prof_begin(const int key[8], hipStream_t stream);                                // gemm.hip (launch profiler)
void prof_end(int rec, hipStream_t stream);

__device__ __forceinline__ void glds16b(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct GemmArgsH {
  const __bf16* A;
  const __bf16* B;
  GemmArgs g;  // C, bias, aux, M, N, K, ldc, ldaux, k_chunk, slab_stride, epi, accumulate, tiles_n, wide; lda/ldb in bf16 elements
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64, 2) void gemm_bf16kc_kernel(GemmArgsH ph) {
  const GemmArgs& p = ph.g;
  constexpr int BK = 64, NSTAGE = 2;                 // bf16 elements per k-tile row (128 bytes)
  constexpr int NW = WM * WN;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int A_B = BM * 128, B_B = BN * 128, STAGE_B = A_B + B_B;  // bytes
  constexpr int IA = A_B / 1024 / NW, IB = B_B / 1024 / NW;          // 1-KiB DMA instructions per wave
  static_assert(A_B % (1024 * NW) == 0 && B_B % (1024 * NW) == 0, "tile must split into whole DMA pieces per wave");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, h = lane >> 5;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / p.tiles_n) * BM;
  const int n0 = (bid % p.tiles_n) * BN;
  const int kbeg = blockIdx.z * p.k_chunk;
  const int kend = min(p.K, kbeg + p.k_chunk);
  const int nk = (kend - kbeg) / BK;

  // per-lane source addresses of this wave's DMA pieces: 16-byte chunk f of the tile image holds
  // row r = f >> 3, source chunk (f & 7) ^ ((r >> 1) & 7)
  const unsigned char* pa[IA];
  const unsigned char* pb[IB];
#pragma unroll
  for (int i = 0; i < IA; ++i) {
    const int f = (wave * IA + i) * 64 + lane, r = f >> 3, cp = f & 7;
    pa[i] = reinterpret_cast<const unsigned char*>(ph.A + (long)(m0 + r) * p.lda + kbeg) + ((cp ^ ((r >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const int f = (wave * IB + i) * 64 + lane, r = f >> 3, cp = f & 7;
    pb[i] = reinterpret_cast<const unsigned char*>(ph.B + (long)(n0 + r) * p.ldb + kbeg) + ((cp ^ ((r >> 1) & 7)) << 4);
  }
  auto issue = [&](int stage) {
    unsigned char* sa = smem_b + stage * STAGE_B + wave * IA * 1024;
    unsigned char* sb = smem_b + stage * STAGE_B + A_B + wave * IB * 1024;
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      glds16b(pa[i], sa + i * 1024);
      pa[i] += BK * 2;
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      glds16b(pb[i], sb + i * 1024);
      pb[i] += BK * 2;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int offA[TM], offB[TN], swA[TM], swB[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = (wm * TM + i) * 32 + li;
    offA[i] = row * 128;
    swA[i] = (row >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = (wn * TN + j) * 32 + li;
    offB[j] = col * 128;
    swB[j] = (col >> 1) & 7;
  }

  issue(0);
  for (int kt = 0; kt < nk; ++kt) {
    const int st = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile kt have landed
    __builtin_amdgcn_s_barrier();                      // ... and everybody else's; every wave is done with tile kt-1
    asm volatile("" ::: "memory");
    if (kt + 1 < nk) issue(st ^ 1);
    const unsigned char* a = smem_b + st * STAGE_B;
    const unsigned char* b = a + A_B;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(a + offA[i] + (((2 * ks + h) ^ swA[i]) << 4));
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(b + offB[j] + (((2 * ks + h) ^ swB[j]) << 4));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragment reads of tile kt done before its stage is refilled
  }
  epilogue_wide<BM, BN, WM, WN, TM, TN, NW * 64>(p, acc, reinterpret_cast<float*>(smem_b), m0, n0, wm, wn, li, h, tid);
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

template <int BM, int BN, int WM, int WN>
static int launch_kc(const GemmArgsH& a, dim3 grid, hipStream_t st) {
  size_t smem = (size_t)2 * (BM + BN) * 128;
  smem = std::max(smem, (size_t)BM * (BN + 4) * sizeof(float));  // wide epilogue image
  auto kern = gemm_bf16kc_kernel<BM, BN, WM, WN>;
  static bool attr_set = false;
  if (smem > 64 * 1024 && !attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, st, a);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

}  // namespace mtvaf

using namespace mtvaf;

extern "C" {

// C[M,N] fp32 = A[M,K] . B[N,K]^T, A and B bf16 (uint16 storage), K contiguous.  Requirements (MTVAF_ERR_SHAPE /
// MTVAF_ERR_ALIGN otherwise -- the caller falls back to mtvaf_gemm_bf16): M % 128 == 0, N % 96 == 0 or N % 128 == 0,
// K % 64 == 0, lda/ldb % 8 == 0, 16-byte aligned A/B/C/bias/aux, ldc/ldaux % 4 == 0.  tile: 0 = auto, 1 = 128x96,
// 2 = 128x128.  splits <= 0: auto.  Epilogues / split-K / workspace as mtvaf_gemm_f32.
int mtvaf_gemm_bf16kc(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K,
                      const float* bias, int epi, float* aux, int ldaux, int accumulate, int allow_split, void* workspace,
                      size_t workspace_bytes, int tile, int splits, hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0) return MTVAF_ERR_SHAPE;
  if (!A || !B || !C) return MTVAF_ERR_ARG;
  if ((epi == EPI_GELU || epi == EPI_DGELU || epi == EPI_DTANH) && !aux) return MTVAF_ERR_ARG;
  if (M % 128 || K % 64 || (N % 96 && N % 128)) return MTVAF_ERR_SHAPE;
  if (lda % 8 || ldb % 8 || ldc % 4 || (aux && ldaux % 4)) return MTVAF_ERR_ALIGN;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)bias | (uintptr_t)aux) & 15) return MTVAF_ERR_ALIGN;
  const bool can128 = N % 128 == 0, can96 = N % 96 == 0;
  int bn = tile == 1 ? 96 : (tile == 2 ? 128 : 0);
  if (bn == 96 && !can96) return MTVAF_ERR_SHAPE;
  if (bn == 128 && !can128) return MTVAF_ERR_SHAPE;
  if (bn == 0) {
    // prefer the tile that gives whole rounds of 512 resident blocks; GELU-class epilogues amortise better on 128x128
    const long t96 = can96 ? (long)(M / 128) * (N / 96) : 0, t128 = can128 ? (long)(M / 128) * (N / 128) : 0;
    if (!can96) bn = 128;
    else if (!can128) bn = 96;
    else bn = (epi == EPI_GELU || epi == EPI_DGELU || t128 >= 512) ? 128 : 96;
    (void)t96;
  }
  const long tiles = (long)(M / 128) * (N / bn);
  const bool split_ok = allow_split && (epi == EPI_NONE || epi == EPI_TANH || epi == EPI_DTANH);
  if (splits <= 0) {
    splits = 1;
    if (split_ok && tiles < 384) splits = (int)std::min<long>(std::max<long>(512 / tiles, 1), 8);
  }
  if (!split_ok) splits = 1;
  while (splits > 1 && ((size_t)splits * M * N * sizeof(float) > workspace_bytes || (K / 64) / splits < 4)) --splits;
  GemmArgsH h;
  h.A = static_cast<const __bf16*>(A);
  h.B = static_cast<const __bf16*>(B);
  GemmArgs& a = h.g;
  a.A = nullptr; a.B = nullptr; a.bias = bias; a.aux = aux;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldaux = ldaux;
  a.epi = epi; a.accumulate = accumulate; a.a_vec = a.b_vec = 1;
  int kc = (int)(((K / 64 + splits - 1) / splits) * 64);
  splits = (K + kc - 1) / kc;
  a.k_chunk = kc;
  if (splits > 1) {
    a.C = (float*)workspace; a.ldc = N; a.slab_stride = (long)M * N;
  } else {
    a.C = C; a.ldc = ldc; a.slab_stride = 0;
  }
  a.tiles_n = N / bn;
  a.wide = 1;
  dim3 grid((unsigned)tiles, 1, (unsigned)splits);
  const int key[8] = {bn == 96 ? 200 : 201, 0, 0, 2, M, N, K, splits};
  const int rec = prof_begin(key, stream);
  int rc = bn == 96 ? launch_kc<128, 96, 4, 1>(h, grid, stream) : launch_kc<128, 128, 2, 2>(h, grid, stream);
  prof_end(rec, stream);
  if (rc != MTVAF_OK) return rc;
  if (splits > 1) return launch_splitk_reduce((const float*)workspace, splits, C, M, N, ldc, bias, accumulate, epi, aux, ldaux, stream);
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
