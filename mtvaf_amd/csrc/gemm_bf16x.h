// Pieces shared by the bf16-operand GEMM kernels (gemm_bf16x.hip: 128-row tiles and the 256x128 / 256x192 rings;
// gemm_bf16p.hip: the 256x256 eight-wave, eight-phase kernel): argument block, LDS-DMA and transposing-read helpers.
#pragma once
#include "gemm_common.h"

namespace mtvaf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

int launch_splitk_reduce(const float* slabs, int splits, float* C, int M, int N, int ldc, const float* bias, int accumulate,
                         int epi, float* aux, int ldaux, hipStream_t stream);  // gemm.hip
int prof_begin(const int key[8], hipStream_t stream);                                // gemm.hip (launch profiler)
void prof_end(int rec, hipStream_t stream);

struct GemmArgsX {
  const __bf16* A;
  const __bf16* B;
  float* C32;        // fp32 result (or split-K slabs), may be NULL when C16 is given
  __bf16* C16;       // bf16 result, may be NULL
  const float* bias;
  __bf16* aux16;     // EPI_GELU: pre-activation out; EPI_DGELU: pre-activation in
  float* colpart;    // [M / BM][N] column sums of the result, or NULL
  int M, N, K;
  int lda, ldb, ldc32, ldc16, ldaux;  // elements
  int k_chunk;
  long slab_stride;
  int epi, accumulate, tiles_n;
  const int* klist;  // k-tile list (weight-gradient products, k-major operands): reduce over the 64-row k-tiles
  const int* kcnt;   // klist[0 .. *kcnt) only -- the rest of operand A is exactly zero (gemm_common.h); or NULL
};

__device__ __forceinline__ void glds16x(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm() {  // counted wait: the N most recent DMA instructions of this wave may stay in flight
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ int km_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// two transposing reads -> the 8 consecutive reduction values of this lane's output row / column
__device__ __forceinline__ bf16x8 tr_read8(const unsigned char* p0, const unsigned char* p1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}


// stream-K description of a launch of the 256x256 kernel (gemm_bf16p.hip): the k-tile steps of all output tiles, tile after
// tile, cut into equal runs of W steps per block
struct P256Prob {
  const __bf16* A;
  const __bf16* B;
  float* C32;
  int lda, ldb, ldc32, tiles_n;
};
struct P256SK {
  P256Prob pr[4];
  int tile_begin[4];  // global index of each product's first tile (INT_MAX: unused slot)
  int nprob;        // 0: the single product of the GemmArgsX; 1..4: products sharing the launch (same K, same layouts)
  int KT;           // k-tiles per output tile
  int W;            // steps per block (set by the launcher)
  int kmajor;       // equal pieces only (KT % W == 0): blocks that share an XCD take the SAME k-range of DIFFERENT tiles (see the kernel)
  long total;       // tiles * KT
  float* slabs;     // [blocks][256 * 256] fp32 contributions, in the kernel's register order
  unsigned* flags;  // [blocks], zero between launches
  unsigned* err;    // set if a bounded wait ran out
};
int launch_p256(const GemmArgsX& a, int layout_a, int layout_b, dim3 grid, hipStream_t st);  // gemm_bf16p.hip
int launch_p256_streamk(const GemmArgsX& a, P256SK sk, int layout_a, int layout_b, int grid_blocks, int steps_per_run, hipStream_t st);

}  // namespace mtvaf
