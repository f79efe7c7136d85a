// Native per-layer executor: one C call enqueues ALL kernels of a BertLayer forward or
// backward -- reference models/modeling_bert.py:439-522 (BertLayer = BertAttention + BertIntermediate + BertOutput) and
// its autograd backward -- instead of ~20 (forward) / ~45 (backward) Python-level calls, buffer allocations and stream /
// event objects per layer.  At bs 32 in the mixed-precision mode and at bs 4 the step is bound by the HOST (8.5 ms of
// Python to enqueue 7 ms of GPU work); the executor cuts that to one ctypes call per layer and direction.  The kernels
// and their order are exactly those of the Python engine (mtvaf_amd/engine.py keeps that path as MTVAF_NATIVE_EXEC=0
// for A/B tests): this file only composes the library's own entry points.
//
// Backward: the weight-gradient products (and the bias column sums) depend only on tensors the dX chain has already
// produced, so they are enqueued on `side` behind an event of `main`; `side == main` serialises.
#include "common.h"

#include <cstdlib>

#include <cstddef>

extern "C" {
// (the library's own C ABI; declared here instead of including the public header, which is C99 with void* streams)
int mtvaf_gemm_f32(int layout_a, int layout_b, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M,
                   int N, int K, const float* bias, int epi, float* aux, int ldaux, int accumulate, int allow_split,
                   void* workspace, size_t workspace_bytes, int cfg, int splits, hipStream_t stream);
int mtvaf_gemm_bf16x(int layout_a, int layout_b, const void* A, int lda, const void* B, int ldb, float* C32, int ldc32,
                     void* C16, int ldc16, int M, int N, int K, const float* bias, int epi, void* aux16, int ldaux,
                     int accumulate, float* colpart, int allow_split, void* workspace, size_t workspace_bytes, int tile,
                     int splits, int stages, hipStream_t stream);
int mtvaf_gemm_f32_dw_group(int n, const float* const* A, const int* lda, const float* const* B, const int* ldb, float* const* C,
                            const int* ldc, const int* M, const int* N, int K, const int* klist, const int* kcnt,
                            void* workspace, size_t workspace_bytes, int splits, hipStream_t stream);
int mtvaf_gemm_f32_dw_group_bias(int n, const float* const* A, const int* lda, const float* const* B, const int* ldb, float* const* C,
                                 const int* ldc, const int* M, const int* N, int K, const int* klist, const int* kcnt,
                                 float* const* dbias, void* workspace, size_t workspace_bytes, int splits, hipStream_t stream);
int mtvaf_streamk_attached(hipStream_t stream);
int mtvaf_gemm_bf16x_dw_group(int n, const void* const* A, const int* lda, const void* const* B, const int* ldb, float* const* C32,
                              const int* ldc32, const int* M, const int* N, int K, hipStream_t stream);
int mtvaf_gemm_bf16x_ktiles(int layout_a, int layout_b, const void* A, int lda, const void* B, int ldb, float* C32, int ldc32,
                            void* C16, int ldc16, int M, int N, int K, const float* bias, int epi, void* aux16, int ldaux,
                            int accumulate, float* colpart, int allow_split, void* workspace, size_t workspace_bytes, int tile,
                            int splits, int stages, const int* klist, const int* kcnt, hipStream_t stream);
int mtvaf_colsum_small(const float* part, int rows, int cols, float* out, int accumulate, hipStream_t stream);
int mtvaf_colsum(const float* x, int rows, int cols, int ld, float* out, int accumulate, void* workspace,
                 size_t workspace_bytes, hipStream_t st);
int mtvaf_prefix_attn_fwd(const float* qkv, const float* pk, const float* pv, const float* addmask, float* ctx, float* lse,
                          int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed, uint64_t offset,
                          hipStream_t st);
int mtvaf_prefix_attn_bwd_tail(const float* dctx, const float* qkv, const float* pk, const float* pv, const float* addmask,
                               const float* ctx, const float* lse, float* delta, float* dqkv, float* dpk, float* dpv, int B, int S,
                               int P, int NH, int head_dim, float p_drop, uint64_t seed, uint64_t offset, int zero_tail,
                               hipStream_t st);
int mtvaf_prefix_attn_bwd(const float* dctx, const float* qkv, const float* pk, const float* pv, const float* addmask,
                          const float* ctx, const float* lse, float* delta, float* dqkv, float* dpk, float* dpv, int B,
                          int S, int P, int NH, int head_dim, float p_drop, uint64_t seed, uint64_t offset, hipStream_t st);
int mtvaf_prefix_attn_bf16_fwd(const void* qkv16, const void* pk16, const void* pv16, const float* addmask, void* ctx16,
                               float* lse, int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed,
                               uint64_t offset, hipStream_t st);
int mtvaf_prefix_attn_bf16_bwd_tail(const void* dctx16, const void* qkv16, const void* pk16, const void* pv16, const float* addmask,
                                    const void* ctx16, const float* lse, void* dqkv16, float* dpk, float* dpv, float* partq,
                                    float* partkv, int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed,
                                    uint64_t offset, int zero_tail, hipStream_t st);
int mtvaf_prefix_attn_bf16_bwd(const void* dctx16, const void* qkv16, const void* pk16, const void* pv16,
                               const float* addmask, const void* ctx16, const float* lse, void* dqkv16, float* dpk,
                               float* dpv, float* partq, float* partkv, int B, int S, int P, int NH, int head_dim,
                               float p_drop, uint64_t seed, uint64_t offset, hipStream_t st);
int mtvaf_prefix_attn_varlen_fwd(const float* qkv, const float* pk, const float* pv, const int* cu, int pad_rows, float* ctx,
                                 float* lse, int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed, uint64_t offset,
                                 hipStream_t st);
int mtvaf_prefix_attn_varlen_bwd(const float* dctx, const float* qkv, const float* pk, const float* pv, const int* cu, int pad_rows,
                                 const float* ctx, const float* lse, float* delta, float* dqkv, float* dpk, float* dpv, int B, int S,
                                 int P, int NH, int head_dim, float p_drop, uint64_t seed, uint64_t offset, hipStream_t st);
int mtvaf_prefix_attn_varlen_fwd_planes(const float* qkv, const float* pk, const float* pv, const int* cu, int pad_rows, float* ctx,
                                        float* lse, int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed,
                                        uint64_t offset, void* ctx_planes, int rows, hipStream_t st);
int mtvaf_prefix_attn_varlen_bwd_planes(const float* dctx, const float* qkv, const float* pk, const float* pv, const int* cu, int pad_rows,
                                        const float* ctx, const float* lse, float* delta, float* dqkv, float* dpk, float* dpv, int B,
                                        int S, int P, int NH, int head_dim, float p_drop, uint64_t seed, uint64_t offset,
                                        void* dqkv_planes, int rows, hipStream_t st);
int mtvaf_zero_f32(float* p, long n, hipStream_t st);
int mtvaf_f32_split_planes(const float* src, void* dst, int rows, int cols, int ld, long s_plane, long s_row, long s_kt, hipStream_t stream);
int mtvaf_gemm_f32p(int layout_a, const void* Aplanes, long a_plane, long a_row, long a_kt, long a_col, int layout_b, const void* Bplanes,
                    long b_plane, long b_row, long b_kt, long b_col, float* C, int ldc, int M, int N, int K, const float* bias, int epi,
                    float* aux, int ldaux, int accumulate, int splits, void* workspace, size_t workspace_bytes, int ablate,
                    hipStream_t stream);
int mtvaf_gemm_f32p_slabs(int layout_a, const void* Aplanes, long a_plane, long a_row, long a_kt, long a_col, int layout_b, const void* Bplanes,
                          long b_plane, long b_row, long b_kt, long b_col, float* C, int ldc, int M, int N, int K, const float* bias,
                          int accumulate, int splits, void* workspace, size_t workspace_bytes, int* splits_out, hipStream_t stream);
int mtvaf_gemm_f32p_ep(int layout_a, const void* Aplanes, long a_plane, long a_row, long a_kt, long a_col, int layout_b, const void* Bplanes,
                       long b_plane, long b_row, long b_kt, long b_col, float* C, int ldc, void* c_planes, float* colpart, int M, int N, int K,
                       const float* bias, int epi, float* aux, int ldaux, int accumulate, hipStream_t stream);
int mtvaf_gemm_f32p_dw_group(int n, const void* const* Aplanes, const void* const* Bplanes, const long* strides, float* const* C, const int* ldc,
                             const int* M, const int* N, int K, hipStream_t stream);
int mtvaf_gemm_f32p_dw_group_colsum(int n, const void* const* Aplanes, const void* const* Bplanes, const long* strides, float* const* C,
                                    const int* ldc, const int* M, const int* N, int K, int njobs, const float* const* cs_src,
                                    const int* cs_rows, const int* cs_cols, const int* cs_ld, float* const* cs_dst, hipStream_t stream);
size_t mtvaf_ln_bwd_workspace_bytes(int M, int H);
int mtvaf_gemm_f32_ktiles(int layout_a, int layout_b, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                          int M, int N, int K, const float* bias, int epi, float* aux, int ldaux, int accumulate,
                          int allow_split, void* workspace, size_t workspace_bytes, int cfg, int splits, const int* klist,
                          const int* kcnt, hipStream_t stream);
int mtvaf_prefix_attn_bf16_varlen_fwd(const void* qkv16, const void* pk16, const void* pv16, const int* cu, int pad_rows, void* ctx16,
                                      float* lse, int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed,
                                      uint64_t offset, hipStream_t st);
int mtvaf_prefix_attn_bf16_varlen_bwd(const void* dctx16, const void* qkv16, const void* pk16, const void* pv16, const int* cu,
                                      int pad_rows, const void* ctx16, const float* lse, void* dqkv16, float* dpk, float* dpv, float* partq,
                                      float* partkv, int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed,
                                      uint64_t offset, hipStream_t st);
int mtvaf_dropout_res_ln_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* out,
                             float* mean, float* rstd, int M, int H, float eps, float p_drop, uint64_t seed,
                             uint64_t offset, void* out_bf16, hipStream_t st);
int mtvaf_dropout_res_ln_bwd_rows(const float* dout, const float* x, const float* res, const float* gamma, const float* mean,
                                  const float* rstd, float* dx, float* dres, int dres_accumulate, int M, int H, float p_drop,
                                  uint64_t seed, uint64_t offset, float* part, void* dx_bf16, hipStream_t st);
int mtvaf_dropout_res_ln_bwd_finish(const float* part, int M, int H, float* dgamma, float* dbeta, float* dbias_x, int accumulate,
                                    hipStream_t st);
int mtvaf_gemm_f32_slabs(int layout_a, int layout_b, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N,
                         int K, const float* bias, int accumulate, void* workspace, size_t workspace_bytes, int* splits_out,
                         hipStream_t stream);
int mtvaf_dropout_res_ln_fwd_slabs(const float* slabs, int nslab, const float* bias, float* x_out, const float* res, const float* gamma,
                                   const float* beta, float* out, float* mean, float* rstd, int M, int H, float eps, float p_drop,
                                   uint64_t seed, uint64_t offset, void* out_bf16, hipStream_t st);
int mtvaf_dropout_res_ln_bwd_rows_slabs(const float* dout_base, const float* slabs, int nslab, const float* x, const float* res,
                                        const float* gamma, const float* mean, const float* rstd, float* dx, float* dres,
                                        int dres_accumulate, int M, int H, float p_drop, uint64_t seed, uint64_t offset, float* part,
                                        void* dx_bf16, hipStream_t st);
int mtvaf_dropout_res_ln_fwd_planes(const float* x, int nslab, const float* bias, float* x_out, const float* res, const float* gamma,
                                    const float* beta, float* out, float* mean, float* rstd, int M, int H, float eps, float p_drop,
                                    uint64_t seed, uint64_t offset, void* out_planes, hipStream_t st);
int mtvaf_dropout_res_ln_bwd_rows_planes(const float* dout_base, const float* slabs, int nslab, const float* x, const float* res,
                                         const float* gamma, const float* mean, const float* rstd, float* dx, float* dres,
                                         int dres_accumulate, int M, int H, float p_drop, uint64_t seed, uint64_t offset, float* part,
                                         void* dx_planes, hipStream_t st);
int mtvaf_dropout_res_ln_bwd(const float* dout, const float* x, const float* res, const float* gamma, const float* mean,
                             const float* rstd, float* dx, float* dres, int dres_accumulate, float* dgamma, float* dbeta,
                             float* dbias_x, int accumulate, int M, int H, float p_drop, uint64_t seed, uint64_t offset,
                             void* workspace, size_t workspace_bytes, void* dx_bf16, hipStream_t st);
}

namespace mtvaf {
enum { X_KC = 0, X_KM = 1 };
enum { X_EPI_NONE = 0, X_EPI_GELU = 1, X_EPI_DGELU = 3 };

// events that order the second stream behind the main one: recycled ring (a wait captures the record that precedes it)
static hipEvent_t g_ev[64];
static int g_ev_n = 0, g_ev_i = 0;
static int fork_to(hipStream_t from, hipStream_t to) {
  if (from == to) return MTVAF_OK;
  if (g_ev_n == 0) {
    for (int i = 0; i < 64; ++i)
      if (hipEventCreateWithFlags(&g_ev[i], hipEventDisableTiming) != hipSuccess) return MTVAF_ERR_ARG;
    g_ev_n = 64;
  }
  hipEvent_t e = g_ev[g_ev_i];
  g_ev_i = (g_ev_i + 1) & 63;
  hipError_t rc = hipEventRecord(e, from);
  if (rc != hipSuccess) return (int)rc;
  rc = hipStreamWaitEvent(to, e, 0);
  return rc == hipSuccess ? MTVAF_OK : (int)rc;
}
}  // namespace mtvaf

using namespace mtvaf;

// Round 5: where a dense product in front of a LayerNorm runs as split-K slabs (the N = 768 products of a packed batch: 114
// tiles x 2 splits), the LayerNorm adds the slabs itself -- forward Wo / FFN-2 (+ bias), backward the accumulating FFN-1 dX -- instead
// of a reduction launch in between: 36 launches and as many passes over [M, H] less per step; same bits (the sum runs in the
// reduction launch's order).  MTVAF_LN_SLABS=0: the reduction launches, as before.
static bool ln_slabs_on() {
  static const int on = [] { const char* e = getenv("MTVAF_LN_SLABS"); return e ? atoi(e) : 1; }();
  return on != 0;
}

// dense product (+ bias) -> dropout + residual + LayerNorm, with the split-K slabs of the product handed to the LayerNorm
static int dense_ln_fwd(const float* A, int K, const float* W, const float* bias, float* x_out, const float* res, const float* gamma,
                        const float* beta, float* out, float* mean, float* rstd, int M, int H, float eps, float p_drop, uint64_t seed,
                        uint64_t offset, void* ws, size_t ws_bytes, hipStream_t st) {
  if (ln_slabs_on() && ws) {
    int ns = 1;
    const int rc = mtvaf_gemm_f32_slabs(0, 0, A, K, W, K, x_out, H, M, H, K, bias, 0, ws, ws_bytes, &ns, st);
    if (rc != MTVAF_OK) return rc;
    if (ns > 1)
      return mtvaf_dropout_res_ln_fwd_slabs(static_cast<const float*>(ws), ns, bias, x_out, res, gamma, beta, out, mean, rstd, M, H, eps,
                                            p_drop, seed, offset, nullptr, st);
  } else {
    const int rc = mtvaf_gemm_f32(0, 0, A, K, W, K, x_out, H, M, H, K, bias, 0, nullptr, 0, 0, 1, ws, ws_bytes, -1, -1, st);
    if (rc != MTVAF_OK) return rc;
  }
  return mtvaf_dropout_res_ln_fwd(x_out, res, gamma, beta, out, mean, rstd, M, H, eps, p_drop, seed, offset, nullptr, st);
}

// ---- pre-split operands (round 5; csrc/gemm_f32p.hip): tile-blocked plane images [cols / 32][3][rows][32] ----
static int planes_of(const float* src, void* dst, int rows, int cols, hipStream_t st) {
  return mtvaf_f32_split_planes(src, dst, rows, cols, cols, (long)rows * 64, 64, (long)3 * rows * 64, st);
}
// the split-K plan of a forward / dX product on the pre-split kernel: two slabs where the 128 x 128 tiles alone leave half the
// CUs idle (the 768-wide results of a packed batch: 114 tiles), measured best in tools/f32p_bench.py at 2432 and 4096 rows
static int p16_splits(int M, int N, int K) {
  const long tiles = (long)(M / 128) * (N / 128);
  return (tiles <= 128 && K / 32 >= 8) ? 2 : 1;
}
// C[M,N] = A . W^T (lb = 0: W [N][K], every forward product) or A . W (lb = 1: W [K][N], the dX products) from the blocked images of
// A [M][K] and of the whole weight; keep != NULL: a split plan's slabs stay in ws (mtvaf_gemm_f32p_slabs)
static int p16(int lb, const void* Ap, const void* Wp, float* C, int ldc, int M, int N, int K, const float* bias, int epi, float* aux,
               int ldaux, int accumulate, void* ws, size_t wsb, int* keep, hipStream_t st) {
  const long ap = (long)M * 64, akt = (long)3 * M * 64;
  const long bp = (long)(lb == 0 ? N : K) * 64, bkt = lb == 0 ? (long)3 * N * 64 : 2048, bc = lb == 0 ? 0 : (long)12 * K * 64;
  int splits = p16_splits(M, N, K);
  if (splits > 1 && (!ws || (size_t)splits * M * N * sizeof(float) > wsb)) splits = 1;
  if (keep)
    return mtvaf_gemm_f32p_slabs(0, Ap, ap, 64, akt, 0, lb, Wp, bp, 64, bkt, bc, C, ldc, M, N, K, bias, accumulate, splits, ws, wsb, keep, st);
  return mtvaf_gemm_f32p(0, Ap, ap, 64, akt, 0, lb, Wp, bp, 64, bkt, bc, C, ldc, M, N, K, bias, epi, aux, ldaux, accumulate, splits, ws, wsb, 0, st);
}

#define MTVAF_TRY(call)            \
  do {                             \
    const int rc_ = (call);        \
    if (rc_ != MTVAF_OK) return rc_; \
  } while (0)

extern "C" {

// One encoder layer.  `bf16` selects the kernel family: 0 = fp32 (qkv / cx / pre / act are float*, weights w*, prefix pk /
// pv float*), 1 = mixed precision (those four are bf16, weights w*_h bf16 images, x_h / h1_h / h2_h bf16 copies, prefix
// bf16).  Dropout sites of the layer: attention `offset`, attention-output `offset + 1`, FFN-output `offset + 2`.
struct mtvaf_layer_t {
  int B, S, P, NH, H, I;
  int bf16;
  float eps, p_hidden, p_attn;
  uint64_t seed, offset;
  const float *wqkv, *wo, *w1, *w2;            // fp32 masters [3H,H] [H,H] [I,H] [H,I] (fp32 mode)
  const void *wqkv_h, *wo_h, *w1_h, *w2_h;     // bf16 images (bf16 mode)
  const float *bqkv, *bo, *g1, *b1, *bi1, *bi2, *g2, *b2;
  const float* x;                              // [M,H] layer input
  const void* x_h;                             // bf16 copy (bf16 mode)
  const void *pk, *pv;                         // prefix slabs [B, P*H] or NULL
  const float* addmask;                        // [B, P+S]
  void *qkv, *cx;                              // [M,3H], [M,H]
  float *lse, *a, *h1;                         // [B,NH,S], [M,H], [M,H]
  void* h1_h;
  float *mean1, *rstd1;
  void *pre, *act;                             // [M,I]
  float *f, *h2;                               // [M,H]
  void* h2_h;
  float *mean2, *rstd2;
  void* ws; size_t ws_bytes;                   // main-stream scratch of the forward pass (split-K slabs of small-M products)
  const int* cu;                               // padding-free execution: [B+1] row offsets of the PACKED token tensors, or NULL
  int Mv, Mp;                                  // valid rows, rows of the packed image (Mv rounded up to whole 128-row tiles)
  const void* x_p;                             // fp32 mode, pre-split operands (round 5): tile-blocked plane image of x, or NULL
  void *cx_p, *h1_p, *act_p, *h2_p;            // ... and of cx / h1 / act / h2 (written by the forward; h2_p may be NULL)
};

struct mtvaf_layer_grads_t {
  float* dh;                                   // in: d/d h2, out: d/d x  [M,H]
  float* dh1;                                  // [M,H] scratch
  void *df, *dpre, *da, *dctx, *dqkv;          // [M,H] [M,I] [M,H] [M,H] [M,3H]  (bf16 in bf16 mode)
  float *part, *partq, *partkv;                // bf16 mode: [M/128, I], [B*ceil(S/64), H], [B*ceil((P+S)/64), 2H]
  float* delta;                                // fp32 mode: [B,NH,S]
  float *dwqkv, *dbqkv, *dwo, *dbo, *dg1, *db1, *dw1, *dbi1, *dw2, *dbi2, *dg2, *db2;
  float *dpk, *dpv;                            // [B, P*H] or NULL
  void* ws_main; size_t ws_main_bytes;         // scratch of the main stream (LayerNorm partials, split-K slabs)
  void* ws_side; size_t ws_side_bytes;         // scratch of the second stream (split-K slabs, column-sum partials)
  const int* klist;                            // optional: k-tile list of the token axis for the dW products (32-row tiles in fp32 mode, 64 in bf16)
  const int* kcnt;
  int zero_tail;                               // the caller's word: token rows behind a sentence's last unmasked position carry exactly-zero
                                               // gradients (what a k-tile list implies): attention backward stops its query loops there
  float *lnpart2, *lnpart1;                    // optional (both or neither): per-layer LayerNorm-backward partials of the FFN / attention
                                               // block (mtvaf_ln_bwd_workspace_bytes each) -- their column sums then run on `side`
  void *df_p, *dpre_p, *da_p, *dqkv_p;         // pre-split operands: scratch plane images of df / dpre / da / dqkv, or NULL
};

// fp32 mode with pre-split operands: every plane image present, packed rows (whole 128-row tiles), whole 128-column tiles
// p16 whose result leaves as a plane image (and per-tile column sums), no fp32 copy (mtvaf_gemm_f32p_ep)
static int p16_ep(int lb, const void* Ap, const void* Wp, void* Cp, float* colpart, int M, int N, int K, const float* bias, int epi, float* aux,
                  int ldaux, hipStream_t st) {
  const long ap = (long)M * 64, akt = (long)3 * M * 64;
  const long bp = (long)(lb == 0 ? N : K) * 64, bkt = lb == 0 ? (long)3 * N * 64 : 2048, bc = lb == 0 ? 0 : (long)12 * K * 64;
  return mtvaf_gemm_f32p_ep(0, Ap, ap, 64, akt, 0, lb, Wp, bp, 64, bkt, bc, nullptr, 0, Cp, colpart, M, N, K, bias, epi, aux, ldaux, 0, st);
}
// MTVAF_P16_EP=0: the GELU / GELU' results as fp32 tensors + a split pass each (the first form of the pre-split path)
static bool p16_ep_on() {
  static const int on = [] { const char* e = getenv("MTVAF_P16_EP"); return e ? atoi(e) : 1; }();
  return on != 0;
}

// MTVAF_ATTN_PLANES=0: the attention kernels write fp32 only and a split pass follows
static bool attn_planes_on() {
  static const int on = [] { const char* e = getenv("MTVAF_ATTN_PLANES"); return e ? atoi(e) : 1; }();
  return on != 0;
}

static bool planes_mode(const mtvaf_layer_t* L) {
  return !L->bf16 && L->cu && L->x_p && L->cx_p && L->h1_p && L->act_p && L->wqkv_h && L->wo_h && L->w1_h && L->w2_h && L->Mp % 128 == 0 &&
         L->H % 128 == 0 && L->I % 128 == 0 && L->ws;
}

// dense product on the pre-split kernel (+ bias) -> dropout + residual + LayerNorm, the product's split-K slabs handed to the LayerNorm
// MTVAF_LN_PLANES=0: the LayerNorm kernels write fp32 only and a split pass follows (the first form of the pre-split path)
static bool ln_planes_on() {
  static const int on = [] { const char* e = getenv("MTVAF_LN_PLANES"); return e ? atoi(e) : 1; }();
  return on != 0;
}
static int dense_ln_fwd_p(const void* Ap, int K, const void* Wp, const float* bias, float* x_out, const float* res, const float* gamma,
                          const float* beta, float* out, float* mean, float* rstd, int M, int H, float eps, float p_drop, uint64_t seed,
                          uint64_t offset, void* ws, size_t ws_bytes, void* out_planes, hipStream_t st) {
  int ns = 1;
  int rc = p16(0, Ap, Wp, x_out, H, M, H, K, bias, X_EPI_NONE, nullptr, 0, 0, ws, ws_bytes, ln_slabs_on() ? &ns : nullptr, st);
  if (rc != MTVAF_OK) return rc;
  if (out_planes && ln_planes_on())  // (the LayerNorm writes the plane image of its output itself)
    return mtvaf_dropout_res_ln_fwd_planes(ns > 1 ? static_cast<const float*>(ws) : x_out, ns > 1 ? ns : 0, bias, x_out, res, gamma, beta, out,
                                           mean, rstd, M, H, eps, p_drop, seed, offset, out_planes, st);
  if (out_planes) {
    rc = ns > 1 ? mtvaf_dropout_res_ln_fwd_slabs(static_cast<const float*>(ws), ns, bias, x_out, res, gamma, beta, out, mean, rstd, M, H, eps,
                                                 p_drop, seed, offset, nullptr, st)
                : mtvaf_dropout_res_ln_fwd(x_out, res, gamma, beta, out, mean, rstd, M, H, eps, p_drop, seed, offset, nullptr, st);
    if (rc != MTVAF_OK) return rc;
    return planes_of(out, out_planes, M, H, st);
  }
  if (ns > 1)
    return mtvaf_dropout_res_ln_fwd_slabs(static_cast<const float*>(ws), ns, bias, x_out, res, gamma, beta, out, mean, rstd, M, H, eps, p_drop,
                                          seed, offset, nullptr, st);
  return mtvaf_dropout_res_ln_fwd(x_out, res, gamma, beta, out, mean, rstd, M, H, eps, p_drop, seed, offset, nullptr, st);
}

int mtvaf_encoder_layer_fwd(const mtvaf_layer_t* L, hipStream_t st) {
  if (!L) return MTVAF_ERR_ARG;
  // padding-free execution: the row-wise kernels simply see Mp packed rows; only attention knows about sentences
  if (L->cu && (L->Mp <= 0 || L->Mp % 128 || L->Mv <= 0 || L->Mv > L->Mp)) return MTVAF_ERR_ARG;
  const int M = L->cu ? L->Mp : L->B * L->S, H = L->H, I = L->I;
  if (L->bf16) {
    MTVAF_TRY(mtvaf_gemm_bf16x(X_KC, X_KC, L->x_h, H, L->wqkv_h, H, nullptr, 0, L->qkv, 3 * H, M, 3 * H, H, L->bqkv, X_EPI_NONE,
                               nullptr, 0, 0, nullptr, 0, nullptr, 0, 0, -1, 0, st));
    if (L->cu) {
      // (the rows that pad the packed image belong to no sentence: an extra slice of the launch zero-fills them)
      MTVAF_TRY(mtvaf_prefix_attn_bf16_varlen_fwd(L->qkv, L->pk, L->pv, L->cu, L->Mp - L->Mv, L->cx, L->lse, L->B, L->S, L->P, L->NH, 64,
                                                  L->p_attn, L->seed, L->offset, st));
    } else {
      MTVAF_TRY(mtvaf_prefix_attn_bf16_fwd(L->qkv, L->pk, L->pv, L->addmask, L->cx, L->lse, L->B, L->S, L->P, L->NH, 64, L->p_attn,
                                           L->seed, L->offset, st));
    }
    MTVAF_TRY(mtvaf_gemm_bf16x(X_KC, X_KC, L->cx, H, L->wo_h, H, L->a, H, nullptr, 0, M, H, H, L->bo, X_EPI_NONE, nullptr, 0, 0,
                               nullptr, 0, nullptr, 0, 0, -1, 0, st));
    MTVAF_TRY(mtvaf_dropout_res_ln_fwd(L->a, L->x, L->g1, L->b1, L->h1, L->mean1, L->rstd1, M, H, L->eps, L->p_hidden, L->seed,
                                       L->offset + 1, L->h1_h, st));
    MTVAF_TRY(mtvaf_gemm_bf16x(X_KC, X_KC, L->h1_h, H, L->w1_h, H, nullptr, 0, L->act, I, M, I, H, L->bi1, X_EPI_GELU, L->pre, I, 0,
                               nullptr, 0, nullptr, 0, 0, -1, 0, st));
    MTVAF_TRY(mtvaf_gemm_bf16x(X_KC, X_KC, L->act, I, L->w2_h, I, L->f, H, nullptr, 0, M, H, I, L->bi2, X_EPI_NONE, nullptr, 0, 0,
                               nullptr, 0, nullptr, 0, 0, -1, 0, st));
    MTVAF_TRY(mtvaf_dropout_res_ln_fwd(L->f, L->h1, L->g2, L->b2, L->h2, L->mean2, L->rstd2, M, H, L->eps, L->p_hidden, L->seed,
                                       L->offset + 2, L->h2_h, st));
    return MTVAF_OK;
  }
  float* qkv = static_cast<float*>(L->qkv);
  float* cx = static_cast<float*>(L->cx);
  float* pre = static_cast<float*>(L->pre);
  float* act = static_cast<float*>(L->act);
  if (planes_mode(L)) {
    // pre-split operands (round 5): the same layer on the kernels of csrc/gemm_f32p.hip; every GEMM operand is read as a plane
    // image -- weights written once per optimizer step, activations by one pass behind the kernel that produces them
    MTVAF_TRY(p16(0, L->x_p, L->wqkv_h, qkv, 3 * H, M, 3 * H, H, L->bqkv, X_EPI_NONE, nullptr, 0, 0, L->ws, L->ws_bytes, nullptr, st));
    if (attn_planes_on()) {  // (the attention kernel writes the context's plane image itself)
      MTVAF_TRY(mtvaf_prefix_attn_varlen_fwd_planes(qkv, static_cast<const float*>(L->pk), static_cast<const float*>(L->pv), L->cu,
                                                    L->Mp - L->Mv, cx, L->lse, L->B, L->S, L->P, L->NH, 64, L->p_attn, L->seed, L->offset,
                                                    L->cx_p, M, st));
    } else {
      MTVAF_TRY(mtvaf_prefix_attn_varlen_fwd(qkv, static_cast<const float*>(L->pk), static_cast<const float*>(L->pv), L->cu,
                                             L->Mp - L->Mv, cx, L->lse, L->B, L->S, L->P, L->NH, 64, L->p_attn, L->seed, L->offset, st));
      MTVAF_TRY(planes_of(cx, L->cx_p, M, H, st));
    }
    MTVAF_TRY(dense_ln_fwd_p(L->cx_p, H, L->wo_h, L->bo, L->a, L->x, L->g1, L->b1, L->h1, L->mean1, L->rstd1, M, H, L->eps, L->p_hidden,
                             L->seed, L->offset + 1, L->ws, L->ws_bytes, L->h1_p, st));
    if (p16_ep_on()) {  // (the GELU output is read by GEMMs only: it leaves the FFN-1 epilogue as a plane image, no fp32 copy)
      MTVAF_TRY(p16_ep(0, L->h1_p, L->w1_h, L->act_p, nullptr, M, I, H, L->bi1, X_EPI_GELU, pre, I, st));
    } else {
      MTVAF_TRY(p16(0, L->h1_p, L->w1_h, act, I, M, I, H, L->bi1, X_EPI_GELU, pre, I, 0, L->ws, L->ws_bytes, nullptr, st));
      MTVAF_TRY(planes_of(act, L->act_p, M, I, st));
    }
    MTVAF_TRY(dense_ln_fwd_p(L->act_p, I, L->w2_h, L->bi2, L->f, L->h1, L->g2, L->b2, L->h2, L->mean2, L->rstd2, M, H, L->eps, L->p_hidden,
                             L->seed, L->offset + 2, L->ws, L->ws_bytes, L->h2_p, st));
    return MTVAF_OK;
  }
  // (plain-bias products may use the deterministic split-K: the planner only splits when the tile grid underfills the chip)
  MTVAF_TRY(mtvaf_gemm_f32(X_KC, X_KC, L->x, H, L->wqkv, H, qkv, 3 * H, M, 3 * H, H, L->bqkv, X_EPI_NONE, nullptr, 0, 0, 1, L->ws,
                           L->ws_bytes, -1, -1, st));
  if (L->cu) {
    // the rows that pad the packed image belong to no sentence: an extra slice of the launch zero-fills them (0 x NaN of an
    // unwritten row would poison dW)
    MTVAF_TRY(mtvaf_prefix_attn_varlen_fwd(qkv, static_cast<const float*>(L->pk), static_cast<const float*>(L->pv), L->cu,
                                           L->Mp - L->Mv, cx, L->lse, L->B, L->S, L->P, L->NH, 64, L->p_attn, L->seed, L->offset, st));
  } else {
    MTVAF_TRY(mtvaf_prefix_attn_fwd(qkv, static_cast<const float*>(L->pk), static_cast<const float*>(L->pv), L->addmask, cx, L->lse,
                                    L->B, L->S, L->P, L->NH, 64, L->p_attn, L->seed, L->offset, st));
  }
  MTVAF_TRY(dense_ln_fwd(cx, H, L->wo, L->bo, L->a, L->x, L->g1, L->b1, L->h1, L->mean1, L->rstd1, M, H, L->eps, L->p_hidden, L->seed,
                         L->offset + 1, L->ws, L->ws_bytes, st));
  MTVAF_TRY(mtvaf_gemm_f32(X_KC, X_KC, L->h1, H, L->w1, H, act, I, M, I, H, L->bi1, X_EPI_GELU, pre, I, 0, 1, L->ws, L->ws_bytes, -1, -1, st));
  MTVAF_TRY(dense_ln_fwd(act, I, L->w2, L->bi2, L->f, L->h1, L->g2, L->b2, L->h2, L->mean2, L->rstd2, M, H, L->eps, L->p_hidden, L->seed,
                         L->offset + 2, L->ws, L->ws_bytes, st));
  return MTVAF_OK;
}

// fp32 mode: the four weight-gradient products of a layer go out as ONE grouped launch (mtvaf_gemm_f32_dw_group) when the layer
// has at most this many token rows (0: never).  Few-token steps (BASELINE configs[0]: 256 rows) are bound by their launch
// count; at 4096 rows the four tuned launches + their split-K reductions are as fast (measured), so the default stops at 1024.
// MTVAF_DW_GROUP_ROWS overrides; mtvaf_dw_group_rows(rows >= 0) sets, (-1) queries.  The Python orchestration reads the same.
static int g_dw_group_rows = -1;
int mtvaf_dw_group_rows(int rows) {
  if (rows >= 0) g_dw_group_rows = rows;
  if (g_dw_group_rows < 0) {
    const char* e = getenv("MTVAF_DW_GROUP_ROWS");
    g_dw_group_rows = e ? atoi(e) : 1024;
  }
  return g_dw_group_rows;
}

// 1: a layer of `rows` token rows sends its four weight-gradient products as ONE grouped launch (mtvaf_gemm_f32_dw_group):
// few-token layers on the fp32 pipe's grouped ring (above), and -- under the split arithmetic, round 4 -- longer ones as one
// UNSPLIT launch of the wave-specialised split kernel's GROUP form when every product is whole 128 x 128 tiles (432 tiles at
// BERT-base: 1.7 rounds of the CUs; MTVAF_X3_DW_GROUP=0 keeps one launch + one slab reduction per product).  The Python
// orchestration asks the same function.
int mtvaf_f32_split(int on);
int mtvaf_dw_group_wanted(int rows, int H, int I) {
  static const int x3_group = [] { const char* e = getenv("MTVAF_X3_DW_GROUP"); return e ? atoi(e) : 1; }();
  if (rows <= 0 || rows % 32 || H % 128 || I % 128) return 0;
  if (rows <= mtvaf_dw_group_rows(-1) && H % 96 == 0 && I % 96 == 0) return 1;
  return (x3_group && rows > 1024 && mtvaf_f32_split(-1)) ? 1 : 0;
}

static bool attn_tail_on() {  // MTVAF_ATTN_TAIL=0: the attention backward keeps its full query loops under a k-tile list too
  static const int on = [] { const char* e = getenv("MTVAF_ATTN_TAIL"); return e ? atoi(e) : 1; }();
  return on != 0;
}

// LayerNorm backward of one block of the layer, then the fork the weight gradients behind it need anyway.  With per-layer
// partial buffers (g->lnpart*) and a second stream, the column sums (dgamma, dbeta, the dense bias gradient: parameter
// gradients only) run THERE: 24 launches of ~12 us leave the main chain of a step.
static int ln_bwd_forked(const float* dout, const float* x, const float* res, const float* gamma, const float* mean, const float* rstd,
                         float* dx, float* dres, float* dgamma, float* dbeta, float* dbias, int M, int H, float p_drop, uint64_t seed,
                         uint64_t offset, float* part, void* ws, size_t ws_bytes, void* dx16, hipStream_t mainS, hipStream_t side) {
  if (part && side != mainS) {
    MTVAF_TRY(mtvaf_dropout_res_ln_bwd_rows(dout, x, res, gamma, mean, rstd, dx, dres, 0, M, H, p_drop, seed, offset, part, dx16, mainS));
    MTVAF_TRY(fork_to(mainS, side));
    return mtvaf_dropout_res_ln_bwd_finish(part, M, H, dgamma, dbeta, dbias, 0, side);
  }
  MTVAF_TRY(mtvaf_dropout_res_ln_bwd(dout, x, res, gamma, mean, rstd, dx, dres, 0, dgamma, dbeta, dbias, 0, M, H, p_drop, seed, offset, ws,
                                     ws_bytes, dx16, mainS));
  return fork_to(mainS, side);
}

// Backward of one layer.  g->dh holds d loss / d h2 on entry and d loss / d x on return.  `settle` != 0: the second stream
// additionally waits for the layer's LAST main-stream kernel (an optimizer update hanging off the caller's hook must be
// behind every product that still reads the weights).
int mtvaf_encoder_layer_bwd(const mtvaf_layer_t* L, const mtvaf_layer_grads_t* g, hipStream_t mainS, hipStream_t side, int settle) {
  if (!L || !g) return MTVAF_ERR_ARG;
  if (L->cu && (L->Mp <= 0 || L->Mp % 128 || L->Mv <= 0 || L->Mv > L->Mp)) return MTVAF_ERR_ARG;
  const int M = L->cu ? L->Mp : L->B * L->S, H = L->H, I = L->I, B = L->B, S = L->S, P = L->P, NH = L->NH;
  if (L->bf16) {
    // the four weight-gradient products of the layer as ONE launch (mtvaf_gemm_bf16x_dw_group: every tile's reduction over the
    // tokens cut in two, combined inside the launch) when a stream-K scratch is attached to the second stream: 36 + 36 + 27 + 9
    // output tiles fill 256 CUs together, not one product at a time.  Enqueued behind the last of their operands (dqkv).
    const bool grp = mtvaf_streamk_attached(side) > 0 && !g->klist && M % 256 == 0 && H % 256 == 0 && I % 256 == 0;
    MTVAF_TRY(ln_bwd_forked(g->dh, L->f, L->h1, L->g2, L->mean2, L->rstd2, nullptr, g->dh1, g->dg2, g->db2, g->dbi2, M, H, L->p_hidden,
                            L->seed, L->offset + 2, g->lnpart2, g->ws_main, g->ws_main_bytes, g->df, mainS, side));
    if (!grp) MTVAF_TRY(mtvaf_gemm_bf16x_ktiles(X_KM, X_KM, g->df, H, L->act, I, g->dw2, I, nullptr, 0, H, I, M, nullptr, X_EPI_NONE, nullptr, 0, 0,
                               nullptr, 1, g->ws_side, g->ws_side_bytes, 0, -1, 0, g->klist, g->kcnt, side));
    MTVAF_TRY(mtvaf_gemm_bf16x(X_KC, X_KM, g->df, H, L->w2_h, I, nullptr, 0, g->dpre, I, M, I, H, nullptr, X_EPI_DGELU, L->pre, I, 0,
                               g->part, 0, nullptr, 0, 0, -1, 0, mainS));
    MTVAF_TRY(fork_to(mainS, side));
    MTVAF_TRY(mtvaf_colsum_small(g->part, M / 128, I, g->dbi1, 0, side));
    if (!grp) MTVAF_TRY(mtvaf_gemm_bf16x_ktiles(X_KM, X_KM, g->dpre, I, L->h1_h, H, g->dw1, H, nullptr, 0, I, H, M, nullptr, X_EPI_NONE, nullptr, 0, 0,
                               nullptr, 1, g->ws_side, g->ws_side_bytes, 0, -1, 0, g->klist, g->kcnt, side));
    MTVAF_TRY(mtvaf_gemm_bf16x(X_KC, X_KM, g->dpre, I, L->w1_h, H, g->dh1, H, nullptr, 0, M, H, I, nullptr, X_EPI_NONE, nullptr, 0, 1,
                               nullptr, 0, nullptr, 0, 0, -1, 0, mainS));
    MTVAF_TRY(ln_bwd_forked(g->dh1, L->a, L->x, L->g1, L->mean1, L->rstd1, nullptr, g->dh, g->dg1, g->db1, g->dbo, M, H, L->p_hidden,
                            L->seed, L->offset + 1, g->lnpart1, g->ws_main, g->ws_main_bytes, g->da, mainS, side));
    if (!grp) MTVAF_TRY(mtvaf_gemm_bf16x_ktiles(X_KM, X_KM, g->da, H, L->cx, H, g->dwo, H, nullptr, 0, H, H, M, nullptr, X_EPI_NONE, nullptr, 0, 0,
                               nullptr, 1, g->ws_side, g->ws_side_bytes, 0, -1, 0, g->klist, g->kcnt, side));
    MTVAF_TRY(mtvaf_gemm_bf16x(X_KC, X_KM, g->da, H, L->wo_h, H, nullptr, 0, g->dctx, H, M, H, H, nullptr, X_EPI_NONE, nullptr, 0, 0,
                               nullptr, 0, nullptr, 0, 0, -1, 0, mainS));
    if (L->cu) {
      MTVAF_TRY(mtvaf_prefix_attn_bf16_varlen_bwd(g->dctx, L->qkv, L->pk, L->pv, L->cu, L->Mp - L->Mv, L->cx, L->lse, g->dqkv, g->dpk,
                                                  g->dpv, g->partq, g->partkv, B, S, P, NH, 64, L->p_attn, L->seed, L->offset, mainS));
    } else {
      MTVAF_TRY(mtvaf_prefix_attn_bf16_bwd_tail(g->dctx, L->qkv, L->pk, L->pv, L->addmask, L->cx, L->lse, g->dqkv, g->dpk, g->dpv, g->partq,
                                                g->partkv, B, S, P, NH, 64, L->p_attn, L->seed, L->offset,
                                                ((g->klist != nullptr || g->zero_tail) && attn_tail_on()) ? 1 : 0, mainS));
    }
    MTVAF_TRY(fork_to(mainS, side));
    MTVAF_TRY(mtvaf_colsum_small(g->partq, B * ((S + 63) / 64), H, g->dbqkv, 0, side));
    MTVAF_TRY(mtvaf_colsum_small(g->partkv, B * ((P + S + 63) / 64), 2 * H, g->dbqkv + H, 0, side));
    if (!grp) MTVAF_TRY(mtvaf_gemm_bf16x_ktiles(X_KM, X_KM, g->dqkv, 3 * H, L->x_h, H, g->dwqkv, H, nullptr, 0, 3 * H, H, M, nullptr, X_EPI_NONE,
                               nullptr, 0, 0, nullptr, 1, g->ws_side, g->ws_side_bytes, 0, -1, 0, g->klist, g->kcnt, side));
    if (grp) {
      const void* const As[4] = {g->df, g->dpre, g->da, g->dqkv};
      const void* const Bs[4] = {L->act, L->h1_h, L->cx, L->x_h};
      float* const Cs[4] = {static_cast<float*>(g->dw2), static_cast<float*>(g->dw1), static_cast<float*>(g->dwo), static_cast<float*>(g->dwqkv)};
      const int lda[4] = {H, I, H, 3 * H}, ldb[4] = {I, H, H, H}, ldc[4] = {I, H, H, H};
      const int Ms[4] = {H, I, H, 3 * H}, Ns[4] = {I, H, H, H};
      MTVAF_TRY(mtvaf_gemm_bf16x_dw_group(4, As, lda, Bs, ldb, Cs, ldc, Ms, Ns, M, side));
    }
    MTVAF_TRY(mtvaf_gemm_bf16x(X_KC, X_KM, g->dqkv, 3 * H, L->wqkv_h, H, g->dh, H, nullptr, 0, M, H, 3 * H, nullptr, X_EPI_NONE,
                               nullptr, 0, 1, nullptr, 0, nullptr, 0, 0, -1, 0, mainS));
  } else {
    float* df = static_cast<float*>(g->df);
    float* dpre = static_cast<float*>(g->dpre);
    float* da = static_cast<float*>(g->da);
    float* dctx = static_cast<float*>(g->dctx);
    float* dqkv = static_cast<float*>(g->dqkv);
    const float* qkv = static_cast<const float*>(L->qkv);
    const float* cx = static_cast<const float*>(L->cx);
    float* pre = static_cast<float*>(L->pre);
    const float* act = static_cast<const float*>(L->act);
    if (planes_mode(L) && g->df_p && g->dpre_p && g->da_p && g->dqkv_p && g->ws_main) {
      // pre-split operands (round 5): the dX chain and the grouped weight gradients on the kernels of csrc/gemm_f32p.hip
      // (df and da are read by GEMMs only: with per-layer partial buffers and a second stream the LayerNorm backward kernels write
      // their plane images themselves and no fp32 copy)
      const bool lnp = ln_planes_on() && g->lnpart2 && g->lnpart1;
      if (lnp) {  // (its column sums -- dgamma, dbeta, the FFN-2 bias gradient -- are jobs of the grouped weight-gradient launch below)
        MTVAF_TRY(mtvaf_dropout_res_ln_bwd_rows_planes(g->dh, nullptr, 0, L->f, L->h1, L->g2, L->mean2, L->rstd2, nullptr, g->dh1, 0, M, H,
                                                       L->p_hidden, L->seed, L->offset + 2, g->lnpart2, g->df_p, mainS));
      } else {
        MTVAF_TRY(ln_bwd_forked(g->dh, L->f, L->h1, L->g2, L->mean2, L->rstd2, df, g->dh1, g->dg2, g->db2, g->dbi2, M, H, L->p_hidden, L->seed,
                                L->offset + 2, g->lnpart2, g->ws_main, g->ws_main_bytes, nullptr, mainS, side));
        MTVAF_TRY(planes_of(df, g->df_p, M, H, mainS));
      }
      const bool ep = p16_ep_on() && g->part != nullptr;
      if (ep) {  // (dpre is read by GEMMs only -- and summed over its rows for the FFN-1 bias gradient: per-tile sums from the epilogue)
        MTVAF_TRY(p16_ep(1, g->df_p, L->w2_h, g->dpre_p, g->part, M, I, H, nullptr, X_EPI_DGELU, pre, I, mainS));
      } else {
        MTVAF_TRY(p16(1, g->df_p, L->w2_h, dpre, I, M, I, H, nullptr, X_EPI_DGELU, pre, I, 0, g->ws_main, g->ws_main_bytes, nullptr, mainS));
        MTVAF_TRY(planes_of(dpre, g->dpre_p, M, I, mainS));
      }
      int ns1 = 1;
      const bool slabs1 = ln_slabs_on() && g->lnpart1 && (lnp || side != mainS);  // (the LayerNorm's partials must not share the slabs' scratch)
      MTVAF_TRY(p16(1, g->dpre_p, L->w1_h, g->dh1, H, M, H, I, nullptr, X_EPI_NONE, nullptr, 0, 1, g->ws_main, g->ws_main_bytes,
                    slabs1 ? &ns1 : nullptr, mainS));
      if (lnp) {
        MTVAF_TRY(mtvaf_dropout_res_ln_bwd_rows_planes(g->dh1, static_cast<const float*>(g->ws_main), ns1 > 1 ? ns1 : 0, L->a, L->x, L->g1,
                                                       L->mean1, L->rstd1, nullptr, g->dh, 0, M, H, L->p_hidden, L->seed, L->offset + 1,
                                                       g->lnpart1, g->da_p, mainS));
      } else {
        if (ns1 > 1) {
          MTVAF_TRY(mtvaf_dropout_res_ln_bwd_rows_slabs(g->dh1, static_cast<const float*>(g->ws_main), ns1, L->a, L->x, L->g1, L->mean1,
                                                        L->rstd1, da, g->dh, 0, M, H, L->p_hidden, L->seed, L->offset + 1, g->lnpart1, nullptr,
                                                        mainS));
          MTVAF_TRY(fork_to(mainS, side));
          MTVAF_TRY(mtvaf_dropout_res_ln_bwd_finish(g->lnpart1, M, H, g->dg1, g->db1, g->dbo, 0, side));
        } else {
          MTVAF_TRY(ln_bwd_forked(g->dh1, L->a, L->x, L->g1, L->mean1, L->rstd1, da, g->dh, g->dg1, g->db1, g->dbo, M, H, L->p_hidden, L->seed,
                                  L->offset + 1, g->lnpart1, g->ws_main, g->ws_main_bytes, nullptr, mainS, side));
        }
        MTVAF_TRY(planes_of(da, g->da_p, M, H, mainS));
      }
      MTVAF_TRY(p16(1, g->da_p, L->wo_h, dctx, H, M, H, H, nullptr, X_EPI_NONE, nullptr, 0, 0, g->ws_main, g->ws_main_bytes, nullptr, mainS));
      if (attn_planes_on()) {
        MTVAF_TRY(mtvaf_prefix_attn_varlen_bwd_planes(dctx, qkv, static_cast<const float*>(L->pk), static_cast<const float*>(L->pv), L->cu,
                                                      L->Mp - L->Mv, cx, L->lse, g->delta, dqkv, g->dpk, g->dpv, B, S, P, NH, 64, L->p_attn,
                                                      L->seed, L->offset, g->dqkv_p, M, mainS));
      } else {
        MTVAF_TRY(mtvaf_prefix_attn_varlen_bwd(dctx, qkv, static_cast<const float*>(L->pk), static_cast<const float*>(L->pv), L->cu,
                                               L->Mp - L->Mv, cx, L->lse, g->delta, dqkv, g->dpk, g->dpv, B, S, P, NH, 64, L->p_attn,
                                               L->seed, L->offset, mainS));
        MTVAF_TRY(planes_of(dqkv, g->dqkv_p, M, 3 * H, mainS));
      }
      MTVAF_TRY(fork_to(mainS, side));
      // second stream: the two bias gradients that are column sums of dY, then the four weight gradients as ONE launch
      if (!ep) MTVAF_TRY(mtvaf_colsum(dpre, M, I, I, g->dbi1, 0, g->ws_side, g->ws_side_bytes, side));
      // (the small reductions of the layer are COLUMN-SUM JOBS of the grouped launch below -- extra blocks that fill the CUs its
      // last round of tiles leaves idle: the QKV bias gradient = column sums of dQ|dK|dV, the FFN-1 bias gradient from the GELU'
      // epilogue's per-tile sums, the two LayerNorm finishes = three column blocks of their partial rows each)
      {
        const void* const As[4] = {g->df_p, g->dpre_p, g->da_p, g->dqkv_p};
        const void* const Bs[4] = {L->act_p, L->h1_p, L->cx_p, L->x_p};
        float* const Cs[4] = {g->dw2, g->dw1, g->dwo, g->dwqkv};
        const int ldc[4] = {I, H, H, H}, Ms[4] = {H, I, H, 3 * H}, Ns[4] = {I, H, H, H};
        long strides[32];
        for (int i = 0; i < 8; ++i) {  // (every image has M rows: plane M * 64, k-row 64, k-tile 2048, 128-column block 12 * M * 64)
          strides[4 * i] = (long)M * 64; strides[4 * i + 1] = 64; strides[4 * i + 2] = 2048; strides[4 * i + 3] = (long)12 * M * 64;
        }
        static const int jobs_on = [] { const char* e = getenv("MTVAF_DW_JOBS"); return e ? atoi(e) : 1; }();  // (0: one launch per reduction)
        if (!jobs_on) {
          MTVAF_TRY(mtvaf_colsum(dqkv, M, 3 * H, 3 * H, g->dbqkv, 0, g->ws_side, g->ws_side_bytes, side));
          if (ep) MTVAF_TRY(mtvaf_colsum_small(g->part, M / 128, I, g->dbi1, 0, side));
          if (lnp) {
            MTVAF_TRY(mtvaf_dropout_res_ln_bwd_finish(g->lnpart2, M, H, g->dg2, g->db2, g->dbi2, 0, side));
            MTVAF_TRY(mtvaf_dropout_res_ln_bwd_finish(g->lnpart1, M, H, g->dg1, g->db1, g->dbo, 0, side));
          }
        }
        const float* js[8]; float* jd[8]; int jr[8], jc[8], jl[8], nj = 0;
        auto job = [&](const float* src, int rows, int cols, int ld, float* dst) {
          if (dst) { js[nj] = src; jr[nj] = rows; jc[nj] = cols; jl[nj] = ld; jd[nj] = dst; ++nj; }
        };
        job(dqkv, M, 3 * H, 3 * H, g->dbqkv);
        if (ep) job(g->part, M / 128, I, I, g->dbi1);
        if (lnp) {
          const int G = (int)(mtvaf_ln_bwd_workspace_bytes(M, H) / ((size_t)16 * H));  // partial rows [G][3][H] of each LayerNorm backward
          job(g->lnpart2, G, H, 3 * H, g->dg2); job(g->lnpart2 + H, G, H, 3 * H, g->db2); job(g->lnpart2 + 2 * H, G, H, 3 * H, g->dbi2);
          job(g->lnpart1, G, H, 3 * H, g->dg1); job(g->lnpart1 + H, G, H, 3 * H, g->db1); job(g->lnpart1 + 2 * H, G, H, 3 * H, g->dbo);
        }
        if (!jobs_on) nj = 0;
        if (nj > 0) MTVAF_TRY(mtvaf_gemm_f32p_dw_group_colsum(4, As, Bs, strides, Cs, ldc, Ms, Ns, M, nj, js, jr, jc, jl, jd, side));
        else MTVAF_TRY(mtvaf_gemm_f32p_dw_group(4, As, Bs, strides, Cs, ldc, Ms, Ns, M, side));
      }
      MTVAF_TRY(p16(1, g->dqkv_p, L->wqkv_h, g->dh, H, M, H, 3 * H, nullptr, X_EPI_NONE, nullptr, 0, 1, g->ws_main, g->ws_main_bytes, nullptr,
                    mainS));
      if (settle) MTVAF_TRY(fork_to(mainS, side));
      return MTVAF_OK;
    }
    const bool grp = mtvaf_dw_group_wanted(M, H, I) != 0;
    MTVAF_TRY(ln_bwd_forked(g->dh, L->f, L->h1, L->g2, L->mean2, L->rstd2, df, g->dh1, g->dg2, g->db2, g->dbi2, M, H, L->p_hidden, L->seed,
                            L->offset + 2, g->lnpart2, g->ws_main, g->ws_main_bytes, nullptr, mainS, side));
    if (!grp) MTVAF_TRY(mtvaf_gemm_f32_ktiles(X_KM, X_KM, df, H, act, I, g->dw2, I, H, I, M, nullptr, X_EPI_NONE, nullptr, 0, 0, 1, g->ws_side,
                             g->ws_side_bytes, -1, -1, g->klist, g->kcnt, side));
    MTVAF_TRY(mtvaf_gemm_f32(X_KC, X_KM, df, H, L->w2, I, dpre, I, M, I, H, nullptr, X_EPI_DGELU, pre, I, 0, 1, g->ws_main, g->ws_main_bytes, -1, -1,
                             mainS));
    MTVAF_TRY(fork_to(mainS, side));
    // (grouped weight gradients: the bias gradients dbi1 / dbqkv come out of that launch -- column sums of the dY tiles it stages)
    if (!grp) MTVAF_TRY(mtvaf_colsum(dpre, M, I, I, g->dbi1, 0, g->ws_side, g->ws_side_bytes, side));
    if (!grp) MTVAF_TRY(mtvaf_gemm_f32_ktiles(X_KM, X_KM, dpre, I, L->h1, H, g->dw1, H, I, H, M, nullptr, X_EPI_NONE, nullptr, 0, 0, 1, g->ws_side,
                             g->ws_side_bytes, -1, -1, g->klist, g->kcnt, side));
    // (the LayerNorm backward behind this accumulating product adds its split-K slabs itself when its column sums have their
    // own partial buffer -- otherwise its scratch would be the workspace that holds the slabs)
    int ns1 = 1;
    if (ln_slabs_on() && g->lnpart1 && side != mainS && g->ws_main) {
      MTVAF_TRY(mtvaf_gemm_f32_slabs(X_KC, X_KM, dpre, I, L->w1, H, g->dh1, H, M, H, I, nullptr, 1, g->ws_main, g->ws_main_bytes, &ns1, mainS));
    } else {
      MTVAF_TRY(mtvaf_gemm_f32(X_KC, X_KM, dpre, I, L->w1, H, g->dh1, H, M, H, I, nullptr, X_EPI_NONE, nullptr, 0, 1, 1, g->ws_main,
                               g->ws_main_bytes, -1, -1, mainS));
    }
    if (ns1 > 1) {
      MTVAF_TRY(mtvaf_dropout_res_ln_bwd_rows_slabs(g->dh1, static_cast<const float*>(g->ws_main), ns1, L->a, L->x, L->g1, L->mean1, L->rstd1,
                                                    da, g->dh, 0, M, H, L->p_hidden, L->seed, L->offset + 1, g->lnpart1, nullptr, mainS));
      MTVAF_TRY(fork_to(mainS, side));
      MTVAF_TRY(mtvaf_dropout_res_ln_bwd_finish(g->lnpart1, M, H, g->dg1, g->db1, g->dbo, 0, side));
    } else {
      MTVAF_TRY(ln_bwd_forked(g->dh1, L->a, L->x, L->g1, L->mean1, L->rstd1, da, g->dh, g->dg1, g->db1, g->dbo, M, H, L->p_hidden, L->seed,
                              L->offset + 1, g->lnpart1, g->ws_main, g->ws_main_bytes, nullptr, mainS, side));
    }
    if (!grp) MTVAF_TRY(mtvaf_gemm_f32_ktiles(X_KM, X_KM, da, H, cx, H, g->dwo, H, H, H, M, nullptr, X_EPI_NONE, nullptr, 0, 0, 1, g->ws_side,
                             g->ws_side_bytes, -1, -1, g->klist, g->kcnt, side));
    MTVAF_TRY(mtvaf_gemm_f32(X_KC, X_KM, da, H, L->wo, H, dctx, H, M, H, H, nullptr, X_EPI_NONE, nullptr, 0, 0, 1, g->ws_main,
                             g->ws_main_bytes, -1, -1, mainS));
    if (L->cu) {
      MTVAF_TRY(mtvaf_prefix_attn_varlen_bwd(dctx, qkv, static_cast<const float*>(L->pk), static_cast<const float*>(L->pv), L->cu,
                                             L->Mp - L->Mv, cx, L->lse, g->delta, dqkv, g->dpk, g->dpv, B, S, P, NH, 64, L->p_attn,
                                             L->seed, L->offset, mainS));
    } else {
      // (a k-tile list = the caller's word that masked token rows carry exactly-zero gradients: the query loops stop at the
      // last unmasked position)
      MTVAF_TRY(mtvaf_prefix_attn_bwd_tail(dctx, qkv, static_cast<const float*>(L->pk), static_cast<const float*>(L->pv), L->addmask, cx,
                                           L->lse, g->delta, dqkv, g->dpk, g->dpv, B, S, P, NH, 64, L->p_attn, L->seed, L->offset,
                                           ((g->klist != nullptr || g->zero_tail) && attn_tail_on()) ? 1 : 0, mainS));
    }
    MTVAF_TRY(fork_to(mainS, side));
    if (!grp) MTVAF_TRY(mtvaf_colsum(dqkv, M, 3 * H, 3 * H, g->dbqkv, 0, g->ws_side, g->ws_side_bytes, side));
    if (!grp) MTVAF_TRY(mtvaf_gemm_f32_ktiles(X_KM, X_KM, dqkv, 3 * H, L->x, H, g->dwqkv, H, 3 * H, H, M, nullptr, X_EPI_NONE, nullptr, 0, 0, 1,
                             g->ws_side, g->ws_side_bytes, -1, -1, g->klist, g->kcnt, side));
    if (grp) {
      const float* const As[4] = {df, dpre, da, dqkv};
      const float* const Bs[4] = {act, static_cast<const float*>(L->h1), cx, static_cast<const float*>(L->x)};
      float* const Cs[4] = {g->dw2, g->dw1, g->dwo, g->dwqkv};
      const int lda[4] = {H, I, H, 3 * H}, ldb[4] = {I, H, H, H}, ldc[4] = {I, H, H, H};
      const int Ms[4] = {H, I, H, 3 * H}, Ns[4] = {I, H, H, H};
      float* const dbs[4] = {nullptr, g->dbi1, nullptr, g->dbqkv};
      MTVAF_TRY(mtvaf_gemm_f32_dw_group_bias(4, As, lda, Bs, ldb, Cs, ldc, Ms, Ns, M, g->klist, g->kcnt, dbs, g->ws_side, g->ws_side_bytes, -1,
                                             side));
    }
    MTVAF_TRY(mtvaf_gemm_f32(X_KC, X_KM, dqkv, 3 * H, L->wqkv, H, g->dh, H, M, H, 3 * H, nullptr, X_EPI_NONE, nullptr, 0, 1, 1,
                             g->ws_main, g->ws_main_bytes, -1, -1, mainS));
  }
  if (settle) MTVAF_TRY(fork_to(mainS, side));
  return MTVAF_OK;
}

size_t mtvaf_layer_struct_bytes(int which) { return which == 0 ? sizeof(mtvaf_layer_t) : sizeof(mtvaf_layer_grads_t); }

}  // extern "C"
