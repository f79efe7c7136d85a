// Argument block and the small device helpers shared by the fp32 prefix-attention kernels: csrc/attention.hip (fp32 MFMA pipe)
// and csrc/attention_f32s.hip (round 6: the same attention with its products formed as split bf16 products, the arithmetic of
// csrc/gemm_f32x3.hip).  Same launch geometry, same key order, same dropout hash, same outputs in both.
#pragma once
#include "common.h"
#include "planes.h"

namespace mtvaf {

constexpr int D = 64;      // head dim (asserted by the launcher)
constexpr int KT = 64;     // keys (or queries) per LDS tile
constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

struct AttnArgs {
  const float* qkv;
  const float* pk;
  const float* pv;
  const float* addmask;
  float* ctx;
  float* lse;
  // backward
  const float* dctx;
  float* delta;
  float* dqkv;
  float* dpk;
  float* dpv;
  int B, S, P, NH, H;
  float scale, p_drop;
  uint32_t drop_key, drop_thr;
  const uint64_t* epoch;  // device-side dropout epoch (captured launches), or NULL
  // PACKED token rows (padding-free execution): sentence b owns rows cu[b] .. cu[b+1]-1 of qkv / ctx / dctx / dqkv (its
  // unmasked tokens, in order), every kept key is unmasked (addmask is not read), queries beyond the sentence do not
  // exist.  NULL: the padded [B, S] layout.  lse / delta / the dropout row ids keep the [B, NH, S] indexing either way.
  const int* cu;
  int pad_rows;  // packed rows: this many rows behind the last sentence pad the image; the z-slice b == B zero-fills them
  // backward, padded layout: the caller vouches that dctx is EXACTLY zero for the queries behind the last unmasked text position
  // of a sentence (trailing padding: nothing downstream reads those rows -- the contract of the k-tile lists of the weight
  // gradients).  Their dQ is exactly zero and they add exactly nothing to dK / dV: the key side stops its query loop there.
  int zero_tail;
  // pre-split operands (round 5; packed rows only): ALSO write the tile-blocked plane image of the context ([H / 32][3][Mrows][32])
  // / of dQ | dK | dV ([3H / 32][3][Mrows][32]) -- what the Wo / QKV-dX products and the weight gradients read -- or NULL
  unsigned char* ctx_p;
  unsigned char* dqkv_p;
  long Mrows;
};

// the stores of the two results that are GEMM operands downstream: fp32, and the plane image when the caller asked for it
__device__ __forceinline__ void store_ctx(const AttnArgs& a, long row, int col, const f32x4 v) {
  *reinterpret_cast<f32x4*>(a.ctx + row * a.H + col) = v;
  if (a.ctx_p) planes_store4(a.ctx_p, a.Mrows, row, col, v);
}
__device__ __forceinline__ void store_dqkv(const AttnArgs& a, long row, int col, const f32x4 v) {
  *reinterpret_cast<f32x4*>(a.dqkv + row * 3 * a.H + col) = v;
  if (a.dqkv_p) planes_store4(a.dqkv_p, a.Mrows, row, col, v);
}

struct Sent {
  long tok0;  // first token row of the sentence
  int n;      // its text tokens (queries; text keys)
};
__device__ __forceinline__ Sent sentence(const AttnArgs& a, int b) {
  if (a.cu) {
    const int c0 = b ? a.cu[b] : 0;  // (cu[0] = -1 marks a launch order behind the offsets: slot_sentence)
    return Sent{(long)c0, a.cu[b + 1] - c0};
  }
  return Sent{(long)b * a.S, a.S};
}
// the sentence of grid slot z (mtvaf_build_packing_ordered: longest first; identity without the list)
__device__ __forceinline__ int slot_sentence(const AttnArgs& a, int z) { return (a.cu && a.cu[0] < 0) ? a.cu[a.B + 1 + z] : z; }
__device__ __forceinline__ float mask_at(const AttnArgs& a, int b, int Tf, int t) {
  return a.cu ? 0.f : a.addmask[(long)b * Tf + t];
}


// Keys behind the LAST unmasked text position of a sentence (trailing padding: additive mask -10000) contribute exactly 0
// to every probability sum -- exp2 underflows to 0 -- and leave the running maximum untouched, so whole key tiles made of
// them can be skipped with bit-identical results.  -> T_eff = P + 1 + max{s : addmask[b][P+s] > -5000}; the full T when
// no text key is unmasked (nothing is skipped then).  Masked keys BEFORE that position ("holes") stay in the loop.
__device__ __forceinline__ int effective_keys(const float* __restrict__ addmask_row, int P, int S, int* lds_slot) {
  if (threadIdx.x == 0) *lds_slot = -1;
  __syncthreads();
  int last = -1;
  for (int t = threadIdx.x; t < S; t += blockDim.x)
    if (addmask_row[P + t] > -5000.f) last = t;
  if (last >= 0) atomicMax(lds_slot, last);
  __syncthreads();
  const int l = *lds_slot;
  return l >= 0 ? P + l + 1 : P + S;
}

// Branch-free staging of [64][64] tiles of the [prefix ; text] key axis: rows beyond T re-read row T-1 (finite
// values; their probabilities are exactly 0 through the -1e30 entry of the mask tile).  The source pointer is
// selected with bit arithmetic (a ternary on pointers compiles to divergent branches whose loads the compiler
// then serialises with vmcnt(0) waits), and all loads of a tile are issued before the first LDS store.
struct KvSrc {
  const float* pre;  // prefix slab of this (b, h), + the thread's column offset
  const float* txt;  // text rows of this (b, h), + the thread's column offset
};
__device__ __forceinline__ const float* kv_row_ptr(const KvSrc& s, int t, int P, int ld_txt) {
  const bool ispre = t < P;
  const uint64_t m = ispre ? ~0ull : 0ull;
  const uint64_t base = (uint64_t)s.txt ^ (((uint64_t)s.txt ^ (uint64_t)s.pre) & m);
  const int off = ispre ? t * D : (t - P) * ld_txt;
  return reinterpret_cast<const float*>(base) + off;
}
// XCD-aware block order (round 5).  Workgroups are dealt round-robin over the 8 XCDs by their linear id, x fastest: the nx blocks
// of one (sentence, head) -- 2 query tiles forward; 2 query + 3 key tiles backward, which all read the same q / k / v / dO / O
// slices (3.5x the unique bytes: 165 - 205 MB per backward launch by the PMC counters against ~50 MB) -- landed on nx different
// XCDs, each with a private L2.  This bijective remap gives the k-th group of nx consecutive slots of ONE XCD to one
// (sentence, head), so the re-reads hit that XCD's L2.  Placement is a speed hint only: results never depend on it.
// Identity when the number of (sentence, head) pairs is not a multiple of 8.  nz = the sentences (the zero-fill slice z == B of a
// packed launch is dispatched behind them and keeps its index).
#ifndef MTVAF_ATTN_XCD_GROUP
#define MTVAF_ATTN_XCD_GROUP 1
#endif
__device__ __forceinline__ void xcd_group(int nx, int ny, int nz, int& x, int& y, int& z) {
  if (!MTVAF_ATTN_XCD_GROUP || ((ny * nz) & 7) || z >= nz) return;
  const int L = x + nx * (y + ny * z);
  const int xcd = L & 7, slot = L >> 3;
  const int gi = (slot / nx) * 8 + xcd;
  x = slot % nx;
  y = gi % ny;
  z = gi / ny;
}

// launchers of the split-product kernels (csrc/attention_f32s.hip); grid / block as the fp32-pipe kernels of attention.hip
int launch_attn_f32s_fwd(const AttnArgs& a, dim3 grid, hipStream_t st);
int launch_attn_f32s_bwd(const AttnArgs& a, int nq, dim3 grid, hipStream_t st);

}  // namespace mtvaf
