// Span-model heads of TVNetSAModel (reference models/bert_model.py:113-190, 288-305, 363-376):
//   * span_index      flatten_emb_by_sentence + the offset arithmetic of get_span_representation (:140-160)
//   * span_pool_fwd   span gather + unary_affine score + masked softmax pooling (:160-179, :367-369) fused:
//                     the [N*M, JR, D] span tensor of the reference is never materialised
//   * span_pool_bwd   two deterministic passes (per span, then per token) -- no float atomics
//   * distant_ce      distant_cross_entropy (:181-190), forward and backward
//   * ce              nn.CrossEntropyLoss (mean, ignore_index = -100) on the [N*M, C] polarity logits (:288, :302)
//
// The reference sizes JR (= widest span of the batch) and the flattened token list on the host (torch.max(...)
// .item(), nonzero()); here both live in a small device-side `meta` block so the step has no host sync.
// All kernels are HBM/latency-bound row operations: one wave64 per span / token / row, float4 accesses.
#include "common.h"
#include <algorithm>

namespace mtvaf {

constexpr int SPAN_NCH = 4;  // float4 chunks per lane  ->  H <= 1024

// meta[0] = text_length (valid tokens of the batch), meta[1] = JR (clamped to [0, S])
// rowmap[f] = b*S+s of the f-th valid token; fidx[b*S+s] = f or -1; soff[n] = span_starts + word_offset (:151-154);
// width[n] = end - start + 1 (:156)
__global__ __launch_bounds__(1024) void span_index_kernel(const uint8_t* __restrict__ mask, const int64_t* __restrict__ starts,
                                                          const int64_t* __restrict__ ends, int* __restrict__ rowmap,
                                                          int* __restrict__ fidx, int* __restrict__ woff,
                                                          int* __restrict__ soff, int* __restrict__ width,
                                                          int* __restrict__ meta, int B, int S, int M) {
  __shared__ int part[1024];
  __shared__ int jr_red[16];
  const int tid = threadIdx.x, n_tok = B * S;
  const int chunk = (n_tok + 1023) / 1024;
  const int beg = min(tid * chunk, n_tok), end = min(beg + chunk, n_tok);
  int cnt = 0;
  for (int i = beg; i < end; ++i) cnt += mask[i] != 0;
  part[tid] = cnt;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan
    const int v = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - cnt;
  for (int i = beg; i < end; ++i) {
    if (i % S == 0) woff[i / S] = run;
    if (mask[i]) {
      rowmap[run] = i;
      fidx[i] = run++;
    } else {
      fidx[i] = -1;
    }
  }
  if (tid == 1023) meta[0] = part[1023];
  __syncthreads();
  int jr = 0;
  for (int n = tid; n < B * M; n += 1024) {
    const int s = (int)starts[n], e = (int)ends[n];
    soff[n] = s + woff[n / M];
    width[n] = e - s + 1;
    jr = max(jr, e - s + 1);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) jr = max(jr, __shfl_xor(jr, o, 64));
  if ((tid & 63) == 0) jr_red[tid >> 6] = jr;
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w) jr = max(jr, jr_red[w]);
    meta[1] = min(jr, S);
  }
}

struct SpanRow {
  f32x4 v[SPAN_NCH];
};

__device__ __forceinline__ SpanRow load_row(const float* p, int H4, int lane) {
  SpanRow r;
#pragma unroll
  for (int c = 0; c < SPAN_NCH; ++c) {
    const int i = lane + 64 * c;
    r.v[c] = i < H4 ? reinterpret_cast<const f32x4*>(p)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  return r;
}
__device__ __forceinline__ float dot_row(const SpanRow& a, const SpanRow& b) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < SPAN_NCH; ++c) s += a.v[c].x * b.v[c].x + a.v[c].y * b.v[c].y + a.v[c].z * b.v[c].z + a.v[c].w * b.v[c].w;
  return wave_sum(s);
}

// token row of span position r: index into the flattened valid tokens, clipped to the last one (:161-162)
__device__ __forceinline__ int span_row(const int* rowmap, int so, int r, int T) {
  return rowmap[max(0, min(so + r, T - 1))];
}

// pooled[n] = sum_r softmax_r(x_r . w + b + (1 - [r < width]) * -10000) x_r     stats[n] = (max, sum exp)
__global__ __launch_bounds__(256) void span_pool_fwd_kernel(const float* __restrict__ seq, const float* __restrict__ wu,
                                                            const float* __restrict__ bu, const int* __restrict__ rowmap,
                                                            const int* __restrict__ soff, const int* __restrict__ width,
                                                            const int* __restrict__ meta, float* __restrict__ pooled,
                                                            float* __restrict__ stats, int NS, int H) {
  const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= NS) return;
  const int H4 = H >> 2, T = meta[0], JR = meta[1];
  const SpanRow w = load_row(wu, H4, lane);
  const float b = *bu;
  const int so = soff[n], wd = width[n];
  SpanRow acc;
#pragma unroll
  for (int c = 0; c < SPAN_NCH; ++c) acc.v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -3.0e38f, l = 0.f;
  if (T > 0) {
    for (int r = 0; r < JR; ++r) {
      const SpanRow x = load_row(seq + (long)span_row(rowmap, so, r, T) * H, H4, lane);
      const float sc = dot_row(x, w) + b + (r < wd ? 0.f : 1.f) * -10000.0f;
      const float mn = fmaxf(m, sc);
      const float scale = expf(m - mn), p = expf(sc - mn);
      l = l * scale + p;
#pragma unroll
      for (int c = 0; c < SPAN_NCH; ++c) acc.v[c] = acc.v[c] * scale + x.v[c] * p;
      m = mn;
    }
  }
  const float inv = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
  for (int c = 0; c < SPAN_NCH; ++c) {
    const int i = lane + 64 * c;
    if (i < H4) reinterpret_cast<f32x4*>(pooled + (long)n * H)[i] = acc.v[c] * inv;
  }
  if (lane == 0) {
    stats[2 * n] = m;
    stats[2 * n + 1] = l;
  }
}

// Backward pass 1 (one wave per span): p_r, ds_r = p_r (dpooled . x_r - dpooled . pooled) into pbuf/dsbuf [NS, S];
// dw partial [NS, H] = sum_r ds_r x_r; db partial [NS] = sum_r ds_r.
__global__ __launch_bounds__(256) void span_pool_bwd_span_kernel(
    const float* __restrict__ dpooled, const float* __restrict__ pooled, const float* __restrict__ stats,
    const float* __restrict__ seq, const float* __restrict__ wu, const float* __restrict__ bu, const int* __restrict__ rowmap,
    const int* __restrict__ soff, const int* __restrict__ width, const int* __restrict__ meta, float* __restrict__ pbuf,
    float* __restrict__ dsbuf, float* __restrict__ dwpart, float* __restrict__ dbpart, int NS, int H, int S) {
  const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= NS) return;
  const int H4 = H >> 2, T = meta[0], JR = meta[1];
  const SpanRow w = load_row(wu, H4, lane);
  const SpanRow g = load_row(dpooled + (long)n * H, H4, lane);
  const SpanRow o = load_row(pooled + (long)n * H, H4, lane);
  const float b = *bu, m = stats[2 * n], l = stats[2 * n + 1];
  const float inv = l > 0.f ? 1.f / l : 0.f;
  const float D = dot_row(g, o);
  const int so = soff[n], wd = width[n];
  SpanRow dw;
#pragma unroll
  for (int c = 0; c < SPAN_NCH; ++c) dw.v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float db = 0.f;
  if (T > 0) {
    for (int r = 0; r < JR; ++r) {
      const SpanRow x = load_row(seq + (long)span_row(rowmap, so, r, T) * H, H4, lane);
      const float sc = dot_row(x, w) + b + (r < wd ? 0.f : 1.f) * -10000.0f;
      const float p = expf(sc - m) * inv;
      const float ds = p * (dot_row(g, x) - D);
#pragma unroll
      for (int c = 0; c < SPAN_NCH; ++c) dw.v[c] += x.v[c] * ds;
      db += ds;
      if (lane == 0) {
        pbuf[(long)n * S + r] = p;
        dsbuf[(long)n * S + r] = ds;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < SPAN_NCH; ++c) {
    const int i = lane + 64 * c;
    if (i < H4) reinterpret_cast<f32x4*>(dwpart + (long)n * H)[i] = dw.v[c];
  }
  if (lane == 0) dbpart[n] = db;
}

// Backward pass 2 (one wave per token row): dseq[row] = sum over the (span, r) pairs that read this token of
// p dpooled[n] + ds w, in ascending (n, r) order -- deterministic.  Rows outside the mask get zeros.
__global__ __launch_bounds__(256) void span_pool_bwd_token_kernel(
    const float* __restrict__ dpooled, const float* __restrict__ wu, const int* __restrict__ fidx,
    const int* __restrict__ soff, const int* __restrict__ meta, const float* __restrict__ pbuf,
    const float* __restrict__ dsbuf, float* __restrict__ dseq, int n_rows, int NS, int H, int S) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int H4 = H >> 2, T = meta[0], JR = meta[1];
  const int f = fidx[row];
  SpanRow acc;
#pragma unroll
  for (int c = 0; c < SPAN_NCH; ++c) acc.v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (f >= 0 && JR > 0) {
    const SpanRow w = load_row(wu, H4, lane);
    const bool first = f == 0, last = f == T - 1;
    for (int base = 0; base < NS; base += 64) {
      const int n = base + lane;
      // positions r of span n that map to token f:  clip(so + r, 0, T-1) == f,  0 <= r < JR
      int rlo = 0, rhi = -1;
      if (n < NS) {
        const int so = soff[n];
        const int d = f - so;
        rlo = first ? 0 : max(d, 0);       // token 0 also receives the positions clipped from below
        rhi = last ? JR - 1 : min(d, JR - 1);  // the last token receives every position clipped from above
      }
      unsigned long long hits = __ballot(rlo <= rhi);
      while (hits) {
        const int src = __ffsll((long long)hits) - 1;
        hits &= hits - 1;
        const int nn = base + src;
        const int lo = __shfl(rlo, src, 64), hi = __shfl(rhi, src, 64);
        float cp = 0.f, cs = 0.f;
        for (int r = lo; r <= hi; ++r) {
          cp += pbuf[(long)nn * S + r];
          cs += dsbuf[(long)nn * S + r];
        }
        const SpanRow g = load_row(dpooled + (long)nn * H, H4, lane);
#pragma unroll
        for (int c = 0; c < SPAN_NCH; ++c) acc.v[c] += g.v[c] * cp + w.v[c] * cs;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < SPAN_NCH; ++c) {
    const int i = lane + 64 * c;
    if (i < H4) reinterpret_cast<f32x4*>(dseq + (long)row * H)[i] = acc.v[c];
  }
}

// distant_cross_entropy (:181-190): row value = sum_s pos log_softmax(z)_s / sum_s pos; loss = -mean_b.
// One wave per row; logits may be strided (start / end logits are the two columns of the binary_affine output).
// row_ws[b] = (value, lse, den)
__global__ __launch_bounds__(64) void distant_ce_fwd_kernel(const float* __restrict__ z, int ldz, const float* __restrict__ pos,
                                                           float* __restrict__ row_ws, int S) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* zr = z + (long)b * S * ldz;
  float m = -3.0e38f;
  for (int s = lane; s < S; s += 64) m = fmaxf(m, zr[(long)s * ldz]);
  m = wave_max(m);
  float e = 0.f, num = 0.f, den = 0.f;
  for (int s = lane; s < S; s += 64) {
    const float v = zr[(long)s * ldz], p = pos[(long)b * S + s];
    e += expf(v - m);
    num += p * v;
    den += p;
  }
  e = wave_sum(e);
  num = wave_sum(num);
  den = wave_sum(den);
  const float lse = m + logf(e);
  if (lane == 0) {
    row_ws[3 * b] = (num - den * lse) / den;
    row_ws[3 * b + 1] = lse;
    row_ws[3 * b + 2] = den;
  }
}

// loss = scale * -mean_b row value  (accumulated into *loss when accumulate)
__global__ __launch_bounds__(64) void distant_ce_mean_kernel(const float* __restrict__ row_ws, float* __restrict__ loss, int B,
                                                            float scale, int accumulate) {
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 64) s += row_ws[3 * b];
  s = wave_sum(s);
  if (threadIdx.x == 0) *loss = (accumulate ? *loss : 0.f) - scale * s / B;
}

// dz_s = -(g scale / B) (pos_s - softmax_s den) / den
__global__ __launch_bounds__(64) void distant_ce_bwd_kernel(const float* __restrict__ gout, float scale, const float* __restrict__ z,
                                                           int ldz, const float* __restrict__ pos,
                                                           const float* __restrict__ row_ws, float* __restrict__ dz, int lddz,
                                                           int B, int S) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float lse = row_ws[3 * b + 1], den = row_ws[3 * b + 2];
  const float g = -(*gout) * scale / B / den;
  for (int s = lane; s < S; s += 64) {
    const float v = z[((long)b * S + s) * ldz], p = pos[(long)b * S + s];
    dz[((long)b * S + s) * lddz] = g * (p - expf(v - lse) * den);
  }
}

// nn.CrossEntropyLoss(reduction='mean', ignore_index=-100) over [N, C] with C <= 64: one lane per class.
// ws[0] = sum of row losses, ws[1] = number of counted rows; rows processed by a single wave in order (N is a few
// hundred: the reduction stays deterministic).
__global__ __launch_bounds__(64) void ce_fwd_kernel(const float* __restrict__ z, const int64_t* __restrict__ labels,
                                                   float* __restrict__ loss, float* __restrict__ ws, int N, int C) {
  const int lane = threadIdx.x;
  float tot = 0.f, cnt = 0.f;
  for (int n = 0; n < N; ++n) {
    const long lab = labels[n];
    if (lab == -100) continue;
    const float v = lane < C ? z[(long)n * C + lane] : -3.0e38f;
    const float m = wave_max(v);
    const float e = wave_sum(lane < C ? expf(v - m) : 0.f);
    const float zl = __shfl(v, (int)lab, 64);
    tot += m + logf(e) - zl;
    cnt += 1.f;
  }
  if (lane == 0) {
    *loss = tot / cnt;
    ws[0] = tot;
    ws[1] = cnt;
  }
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ z,
                                                    const int64_t* __restrict__ labels, const float* __restrict__ ws,
                                                    float* __restrict__ dz, int N, int C) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const long lab = labels[n];
  const float g = *gout / ws[1];
  float m = -3.0e38f, e = 0.f;
  for (int c = 0; c < C; ++c) m = fmaxf(m, z[(long)n * C + c]);
  for (int c = 0; c < C; ++c) e += expf(z[(long)n * C + c] - m);
  for (int c = 0; c < C; ++c) {
    const float p = expf(z[(long)n * C + c] - m) / e;
    dz[(long)n * C + c] = lab == -100 ? 0.f : g * (p - (c == lab ? 1.f : 0.f));
  }
}

// Cutoff augmentation (modules/augument.py:99-159): out[b,s,:] = x[b,s,:] * row_keep[b,s] * col_keep[b,:]
// (span / token cutoff zero whole token rows, dim cutoff zeroes embedding dimensions per sample).  Its own
// backward: the same product applied to the incoming gradient.
__global__ __launch_bounds__(256) void mask_mul_kernel(const float* __restrict__ x, const float* __restrict__ row_keep,
                                                      const float* __restrict__ col_keep, float* __restrict__ out,
                                                      long n4, int S, int H4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long row = i / H4;
    const int c = (int)(i % H4);
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    if (row_keep) v *= row_keep[row];
    if (col_keep) v *= reinterpret_cast<const f32x4*>(col_keep)[(row / S) * H4 + c];
    reinterpret_cast<f32x4*>(out)[i] = v;
  }
}

}  // namespace mtvaf

using namespace mtvaf;

extern "C" {

// index block layout (int32): rowmap[B*S] | fidx[B*S] | woff[B] | soff[B*M] | width[B*M] | meta[2]
size_t mtvaf_span_index_ints(int B, int S, int M) { return (size_t)2 * B * S + B + (size_t)2 * B * M + 2; }

static void span_index_ptrs(int* index, int B, int S, int M, int*& rowmap, int*& fidx, int*& woff, int*& soff, int*& width,
                            int*& meta) {
  rowmap = index;
  fidx = rowmap + (size_t)B * S;
  woff = fidx + (size_t)B * S;
  soff = woff + B;
  width = soff + (size_t)B * M;
  meta = width + (size_t)B * M;
}

int mtvaf_span_index(const uint8_t* mask, const int64_t* span_starts, const int64_t* span_ends, int* index, int B, int S, int M,
                     hipStream_t st) {
  if (B <= 0 || S <= 0 || M <= 0) return MTVAF_ERR_SHAPE;
  int *rowmap, *fidx, *woff, *soff, *width, *meta;
  span_index_ptrs(index, B, S, M, rowmap, fidx, woff, soff, width, meta);
  hipLaunchKernelGGL(span_index_kernel, dim3(1), dim3(1024), 0, st, mask, span_starts, span_ends, rowmap, fidx, woff, soff,
                     width, meta, B, S, M);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

int mtvaf_span_pool_fwd(const float* seq, const float* w_unary, const float* b_unary, const int* index, float* pooled,
                        float* stats, int B, int S, int M, int H, hipStream_t st) {
  if (B <= 0 || S <= 0 || M <= 0 || H <= 0 || H % 4 || H > 256 * SPAN_NCH) return MTVAF_ERR_SHAPE;
  int *rowmap, *fidx, *woff, *soff, *width, *meta;
  span_index_ptrs(const_cast<int*>(index), B, S, M, rowmap, fidx, woff, soff, width, meta);
  const int NS = B * M;
  hipLaunchKernelGGL(span_pool_fwd_kernel, dim3((NS + 3) / 4), dim3(256), 0, st, seq, w_unary, b_unary, rowmap, soff, width,
                     meta, pooled, stats, NS, H);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

size_t mtvaf_span_pool_bwd_workspace_bytes(int B, int S, int M, int H) {
  return ((size_t)2 * B * M * S + (size_t)B * M * H + (size_t)B * M) * sizeof(float);
}

// dseq [B*S, H] is overwritten; dw_part [B*M, H] and db_part [B*M] live in the workspace and are returned through
// *dw_part_out / *db_part_out for the caller's column sums (mtvaf_colsum).
int mtvaf_span_pool_bwd(const float* dpooled, const float* pooled, const float* stats, const float* seq, const float* w_unary,
                        const float* b_unary, const int* index, float* dseq, float** dw_part_out, float** db_part_out, int B,
                        int S, int M, int H, void* ws, size_t ws_bytes, hipStream_t st) {
  if (B <= 0 || S <= 0 || M <= 0 || H <= 0 || H % 4 || H > 256 * SPAN_NCH) return MTVAF_ERR_SHAPE;
  if (ws_bytes < mtvaf_span_pool_bwd_workspace_bytes(B, S, M, H)) return MTVAF_ERR_WORKSPACE;
  int *rowmap, *fidx, *woff, *soff, *width, *meta;
  span_index_ptrs(const_cast<int*>(index), B, S, M, rowmap, fidx, woff, soff, width, meta);
  const int NS = B * M;
  float* dwpart = static_cast<float*>(ws);  // first: keeps the 16-byte alignment of the workspace
  float* pbuf = dwpart + (size_t)NS * H;
  float* dsbuf = pbuf + (size_t)NS * S;
  float* dbpart = dsbuf + (size_t)NS * S;
  hipLaunchKernelGGL(span_pool_bwd_span_kernel, dim3((NS + 3) / 4), dim3(256), 0, st, dpooled, pooled, stats, seq, w_unary,
                     b_unary, rowmap, soff, width, meta, pbuf, dsbuf, dwpart, dbpart, NS, H, S);
  hipLaunchKernelGGL(span_pool_bwd_token_kernel, dim3((B * S + 3) / 4), dim3(256), 0, st, dpooled, w_unary, fidx, soff, meta,
                     pbuf, dsbuf, dseq, B * S, NS, H, S);
  MTVAF_LAUNCH_CHECK();
  *dw_part_out = dwpart;
  *db_part_out = dbpart;
  return MTVAF_OK;
}

int mtvaf_distant_ce_fwd(const float* logits, int ld, const float* positions, float* loss, float* row_ws, int B, int S,
                         float scale, int accumulate, hipStream_t st) {
  if (B <= 0 || S <= 0 || ld <= 0) return MTVAF_ERR_SHAPE;
  hipLaunchKernelGGL(distant_ce_fwd_kernel, dim3(B), dim3(64), 0, st, logits, ld, positions, row_ws, S);
  hipLaunchKernelGGL(distant_ce_mean_kernel, dim3(1), dim3(64), 0, st, row_ws, loss, B, scale, accumulate);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

int mtvaf_distant_ce_bwd(const float* grad_out, float scale, const float* logits, int ld, const float* positions,
                         const float* row_ws, float* dlogits, int ldd, int B, int S, hipStream_t st) {
  if (B <= 0 || S <= 0 || ld <= 0 || ldd <= 0) return MTVAF_ERR_SHAPE;
  hipLaunchKernelGGL(distant_ce_bwd_kernel, dim3(B), dim3(64), 0, st, grad_out, scale, logits, ld, positions, row_ws, dlogits,
                     ldd, B, S);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

int mtvaf_ce_fwd(const float* logits, const int64_t* labels, float* loss, float* ws2, int N, int C, hipStream_t st) {
  if (N <= 0 || C <= 0 || C > 64) return MTVAF_ERR_SHAPE;
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(1), dim3(64), 0, st, logits, labels, loss, ws2, N, C);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

int mtvaf_ce_bwd(const float* grad_out, const float* logits, const int64_t* labels, const float* ws2, float* dlogits, int N,
                 int C, hipStream_t st) {
  if (N <= 0 || C <= 0 || C > 64) return MTVAF_ERR_SHAPE;
  hipLaunchKernelGGL(ce_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, st, grad_out, logits, labels, ws2, dlogits, N, C);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

int mtvaf_mask_mul(const float* x, const float* row_keep, const float* col_keep, float* out, int B, int S, int H,
                   hipStream_t st) {
  if (B <= 0 || S <= 0 || H <= 0 || H % 4) return MTVAF_ERR_SHAPE;
  const long n4 = (long)B * S * (H / 4);
  hipLaunchKernelGGL(mask_mul_kernel, dim3((unsigned)std::min<long>((n4 + 255) / 256, 4096)), dim3(256), 0, st, x, row_keep,
                     col_keep, out, n4, S, H / 4);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

}  // extern "C"
