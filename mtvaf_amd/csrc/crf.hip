// Linear-chain CRF head of TVNetSAModel2 (third-party `pytorch-crf` in the reference; call sites
// models/bert_model.py:464 ctor, :511 decode, :521 negative log-likelihood with reduction='mean').
//   K13 crf_nll fwd/bwd : gold-path score - log-partition (masked forward algorithm), analytic
//                         backward through the forward/backward marginals
//   K12 crf_viterbi     : masked Viterbi with back-pointers; one packed int32 [B,S] tensor out
// One wavefront per sequence: tag j lives on lane j (C <= 16), the C x C transition matrix sits in
// registers, and the sum / max over the previous tag is a loop of lane broadcasts (v_readlane).  The recursion
// over S is inherently serial, so these kernels are latency-bound by design (B waves of ~S*200 cycles).
#include "common.h"

namespace mtvaf {

constexpr int CMAX = 16;
constexpr float NEG = -1.0e30f;

__device__ __forceinline__ float lse2(float m, float s) { return m + __logf(s); }
// broadcast lane `i` (compile-time constant) of x to the whole wave: v_readlane_b32 into an SGPR instead of a
// ds_bpermute round trip through the LDS crossbar -- the tag recursions do 16-32 of these per time step
__device__ __forceinline__ float bcast(float x, int i) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), i));
}

// ---------------------------------------------------------------------------------------------
// forward / backward recursions in the SCALED LINEAR domain.  The log-domain form costs C exp + a log per lane and time
// step on a serial critical path (S steps); here
//     a_t[j] = (sum_i a_{t-1}[i] E[i][j]) x_t[j] / n_t,   E = exp(trans - tmax),  x_t = exp(emit_t - max_j emit_t)
// is C broadcasts + C FMAs per step, the per-step normaliser n_t (sum over the tag lanes: four DPP row rotations, no LDS
// crossbar) keeps every a_t at sum 1, and  logZ = c0 + sum_t (log n_t + max_j emit_t + tmax) + log sum_j a_last[j] e^{end[j]}
// takes its S logarithms in parallel after the loop.  The emission factors x_t are computed for all t up front, in
// parallel.  The backward pass runs the same way on b_t (scaled so that sum_i a_t[i] b_t[i] = 1): node and edge marginals
// are products of registers, normalised by one reciprocal per step -- no exp, no log on the serial path.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float row_sum16(float v) {  // sum over the 16 lanes of a DPP row, result in every lane of it
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true));  // row_ror:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, true));  // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, true));  // row_ror:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, true));  // row_ror:1
  return v;
}

// stage one sequence: xs[t][j] = exp(emit[t][j] - mx[t]), mx[t] = max_j emit[t][j]; tags; mask (lanes stride over t)
__device__ __forceinline__ void crf_stage(const float* __restrict__ em, const int64_t* __restrict__ tags,
                                          const uint8_t* __restrict__ mask, long b, int S, int C, float* xs, float* mxs, int* tg,
                                          uint8_t* mk, int lane) {
  for (int t = lane; t < S; t += 64) {
    const float* e = em + ((long)b * S + t) * C;
    float m = NEG;
    for (int j = 0; j < C; ++j) m = fmaxf(m, e[j]);
    for (int j = 0; j < C; ++j) xs[t * C + j] = __expf(e[j] - m);
    mxs[t] = m;
    tg[t] = (int)tags[(long)b * S + t];
    mk[t] = mask[(long)b * S + t];
  }
}

// forward: alpha_ws[b,t,:] = scaled alpha after step t (sum 1 for t >= 1), logZ[b], llh[b] = score(gold) - logZ
__global__ __launch_bounds__(64) void crf_fwd_kernel(const float* __restrict__ em, const int64_t* __restrict__ tags,
                                                    const uint8_t* __restrict__ mask, const float* __restrict__ start,
                                                    const float* __restrict__ end, const float* __restrict__ trans,
                                                    float* __restrict__ alpha_ws, float* __restrict__ logz,
                                                    float* __restrict__ llh, int S, int C) {
  extern __shared__ __attribute__((aligned(16))) float crf_lds[];
  float* xs = crf_lds;                                  // [S*C] emission factors
  float* mxs = xs + S * C;                              // [S]   per-step emission maxima
  float* nr = mxs + S;                                  // [S]   per-step normalisers (1 where masked)
  int* tg = reinterpret_cast<int*>(nr + S);             // [S]
  uint8_t* mk = reinterpret_cast<uint8_t*>(tg + S);     // [S]
  const int b = blockIdx.x, lane = threadIdx.x;
  crf_stage(em, tags, mask, b, S, C, xs, mxs, tg, mk, lane);
  for (int t = lane; t < S; t += 64) nr[t] = 1.f;
  __syncthreads();
  const bool act = lane < C;
  const int j = act ? lane : 0;
  float tm = NEG;
  for (int i = 0; i < C; ++i) tm = fmaxf(tm, act ? trans[i * C + j] : NEG);
  const float tmax = wave_max(tm);
  float tE[CMAX];
#pragma unroll
  for (int i = 0; i < CMAX; ++i) tE[i] = (i < C && act) ? __expf(trans[i * C + j] - tmax) : 0.f;
  const float e0 = em[((long)b * S) * C + j];
  const float a0l = act ? start[j] + e0 : NEG;
  const float c0 = wave_max(a0l);
  float a = act ? __expf(a0l - c0) : 0.f;
  if (act && alpha_ws) alpha_ws[((long)b * S) * C + j] = a;
  for (int t = 1; t < S; ++t) {
    if (mk[t]) {  // (wave-uniform)
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < CMAX; ++i)
        if (i < C) s = fmaf(bcast(a, i), tE[i], s);
      const float an = act ? s * xs[t * C + j] : 0.f;
      const float n = row_sum16(an);
      a = an * __frcp_rn(n);
      if (lane == 0) nr[t] = n;
    }
    if (act && alpha_ws) alpha_ws[((long)b * S + t) * C + j] = a;
  }
  const float em_ = wave_max(act ? end[j] : NEG);
  const float fin = wave_sum(act ? a * __expf(end[j] - em_) : 0.f);
  __syncthreads();
  // logZ: the S logarithms in parallel; gold path score, lanes stride over t
  float lz = 0.f, sc = 0.f;
  int cnt = 0;
  for (int t = lane; t < S; t += 64) {
    cnt += mk[t] ? 1 : 0;
    if (t >= 1 && mk[t]) {
      lz += __logf(nr[t]) + mxs[t] + tmax;
      sc += trans[tg[t - 1] * C + tg[t]] + em[((long)b * S + t) * C + tg[t]];
    }
  }
  lz = wave_sum(lz);
  sc = wave_sum(sc);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if (lane == 0) {
    const float z = c0 + lz + __logf(fin) + em_;
    sc += start[tg[0]] + em[((long)b * S) * C + tg[0]] + end[tg[cnt - 1]];
    logz[b] = z;
    llh[b] = sc - z;
  }
}

// loss = -(1/B) sum_b llh[b]   ('mean' reduction, negated at models/bert_model.py:521)
__global__ void crf_loss_kernel(const float* __restrict__ llh, float* __restrict__ loss, int B) {
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 64) s += llh[b];
  s = wave_sum(s);
  if (threadIdx.x == 0) *loss = -s / B;
}

// ---------------------------------------------------------------------------------------------
// backward: d(loss)/d(emissions) [B,S,C] and per-sequence partials of the parameter gradients
// partial layout per sequence: [start C | end C | trans C*C]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void crf_bwd_kernel(const float* __restrict__ em, const int64_t* __restrict__ tags,
                                                    const uint8_t* __restrict__ mask, const float* __restrict__ end,
                                                    const float* __restrict__ trans, const float* __restrict__ alpha_ws,
                                                    const float* __restrict__ gout, float* __restrict__ dem,
                                                    float* __restrict__ partial, int B, int S, int C) {
  extern __shared__ __attribute__((aligned(16))) float crf_lds[];
  float* xs = crf_lds;                                  // [S*C]
  float* al = xs + S * C;                               // [S*C] scaled alphas of the forward pass
  float* mxs = al + S * C;                              // [S] (unused here, filled by the shared staging code)
  float* gold = mxs + S;                                // [C*C] gold transition counts
  int* tg = reinterpret_cast<int*>(gold + CMAX * CMAX); // [S]
  uint8_t* mk = reinterpret_cast<uint8_t*>(tg + S);     // [S]
  const int b = blockIdx.x, lane = threadIdx.x;
  crf_stage(em, tags, mask, b, S, C, xs, mxs, tg, mk, lane);
  for (int i = lane; i < S * C; i += 64) al[i] = alpha_ws[(long)b * S * C + i];
  for (int i = lane; i < CMAX * CMAX; i += 64) gold[i] = 0.f;
  __syncthreads();
  for (int t = 1 + lane; t < S; t += 64)
    if (mk[t]) atomicAdd(&gold[tg[t - 1] * C + tg[t]], 1.f);  // (integer-valued sums: exact, order-independent)
  const bool act = lane < C;
  const int j = act ? lane : 0;
  const float g = (gout ? *gout : 1.f) / B;
  float tm = NEG;
  for (int i = 0; i < C; ++i) tm = fmaxf(tm, act ? trans[i * C + j] : NEG);
  const float tmax = wave_max(tm);
  float tcol[CMAX], trow[CMAX], eacc[CMAX];
#pragma unroll
  for (int i = 0; i < CMAX; ++i) {
    tcol[i] = (i < C && act) ? __expf(trans[i * C + j] - tmax) : 0.f;   // E[i][lane]
    trow[i] = (i < C && act) ? __expf(trans[j * C + i] - tmax) : 0.f;   // E[lane][i]
    eacc[i] = 0.f;
  }
  float* de = dem + (long)b * S * C;
  int cnt = 0;
  for (int t = lane; t < S; t += 64) cnt += mk[t] ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  const int last_tag = (int)tg[cnt - 1];

  const float em_ = wave_max(act ? end[j] : NEG);
  float bt = act ? __expf(end[j] - em_) : 0.f;  // beta of the last position, any positive scale
  float dend;
  {
    const float pe = act ? al[(long)(S - 1) * C + j] * bt : 0.f;
    const float pen = pe * __frcp_rn(row_sum16(pe));  // (DPP sum outside any lane-dependent control flow)
    dend = act ? pen - (j == last_tag ? 1.f : 0.f) : 0.f;
  }
  for (int t = S - 1; t >= 1; --t) {
    if (!mk[t]) {  // (wave-uniform)
      if (act) de[t * C + j] = 0.f;
      continue;
    }
    const float u = act ? xs[t * C + j] * bt : 0.f;            // x_t[j] b_t[j]
    const float ap = act ? al[(long)(t - 1) * C + j] : 0.f;    // a_{t-1}[lane]
    float pr[CMAX];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CMAX; ++i) {
      pr[i] = (i < C) ? bcast(ap, i) * tcol[i] : 0.f;          // a_{t-1}[i] E[i][j]
      s += pr[i];
    }
    const float w = s * u;
    const float inv = __frcp_rn(row_sum16(w));                 // 1 / sum_ij a_{t-1}[i] E[i][j] x_t[j] b_t[j]
    const float ui = u * inv;
    if (act) de[t * C + j] = g * (s * ui - (j == (int)tg[t] ? 1.f : 0.f));  // node marginal - gold
#pragma unroll
    for (int i = 0; i < CMAX; ++i)
      if (i < C) eacc[i] = fmaf(pr[i], ui, eacc[i]);           // edge marginals, lane j accumulates column j
    // b_{t-1}[lane] = sum_k E[lane][k] x_t[k] b_t[k], scaled so that sum_i a_{t-1}[i] b_{t-1}[i] = 1
    float nb = 0.f;
#pragma unroll
    for (int k = 0; k < CMAX; ++k)
      if (k < C) nb = fmaf(trow[k], bcast(ui, k), nb);
    bt = act ? nb : 0.f;
  }
  __syncthreads();
  const float p0 = act ? al[j] * bt : 0.f;
  const float p0n = p0 * __frcp_rn(row_sum16(p0));  // (all 16 lanes of the row take part in the DPP sum: not under `act`)
  if (act) {
    const float pm = p0n - (j == (int)tg[0] ? 1.f : 0.f);
    de[j] = g * pm;
    float* pp = partial + (long)b * (2 * C + C * C);
    pp[j] = pm;
    pp[C + j] = dend;
#pragma unroll
    for (int i = 0; i < CMAX; ++i)
      if (i < C) pp[2 * C + i * C + j] = eacc[i] - gold[i * C + j];
  }
}

__global__ void crf_param_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ gout, int B, int C,
                                        float* __restrict__ dstart, float* __restrict__ dend, float* __restrict__ dtrans,
                                        int accumulate) {
  const int n = 2 * C + C * C;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += partial[(long)b * n + i];
  s *= (gout ? *gout : 1.f) / B;
  float* d = i < C ? dstart + i : (i < 2 * C ? dend + (i - C) : dtrans + (i - 2 * C));
  if (accumulate) s += *d;
  *d = s;
}

// ---------------------------------------------------------------------------------------------
// Viterbi decode: tags_out[b, :len] best path, -1 beyond; lens_out[b] = sum(mask[b])
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void crf_viterbi_kernel(const float* __restrict__ em, const uint8_t* __restrict__ mask,
                                                        const float* __restrict__ start, const float* __restrict__ end,
                                                        const float* __restrict__ trans, int32_t* __restrict__ tags_out,
                                                        int32_t* __restrict__ lens_out, int S, int C) {
  extern __shared__ __attribute__((aligned(16))) float crf_lds[];
  float* e = crf_lds;                                   // [S*C]
  uint8_t* mk = reinterpret_cast<uint8_t*>(e + S * C);  // [S]
  uint8_t* bp = mk + ((S + 15) & ~15);                  // [S][CMAX] back-pointers
  const int b = blockIdx.x, lane = threadIdx.x;
  for (int i = lane; i < S * C; i += 64) e[i] = em[(long)b * S * C + i];
  for (int i = lane; i < S; i += 64) mk[i] = mask[(long)b * S + i];
  __syncthreads();
  const bool act = lane < C;
  const int j = act ? lane : 0;
  float tcol[CMAX];
#pragma unroll
  for (int i = 0; i < CMAX; ++i) tcol[i] = (i < C) ? trans[i * C + j] : 0.f;
  float score = act ? start[j] + e[j] : NEG;
  int cnt = 0;
  for (int t = lane; t < S; t += 64) cnt += mk[t] ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  for (int t = 1; t < S; ++t) {
    float best = NEG;
    int bi = 0;
#pragma unroll
    for (int i = 0; i < CMAX; ++i) {
      const float v = bcast(score, i) + tcol[i];
      if (i < C && v > best) { best = v; bi = i; }
    }
    if (act) bp[t * CMAX + j] = (uint8_t)bi;
    if (mk[t] && act) score = best + e[t * C + j];
  }
  float fin = act ? score + end[j] : NEG;
  int idx = act ? j : CMAX;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(fin, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ov > fin || (ov == fin && oi < idx)) { fin = ov; idx = oi; }
  }
  __syncthreads();
  if (lane == 0) {
    int32_t* out = tags_out + (long)b * S;
    int cur = idx;
    out[cnt - 1] = cur;
    // history[t-1] corresponds to step t; walk back over steps cnt-1 .. 1 (pytorch-crf: history[:seq_end])
    for (int t = cnt - 1; t >= 1; --t) {
      cur = bp[t * CMAX + cur];
      out[t - 1] = cur;
    }
    for (int t = cnt; t < S; ++t) out[t] = -1;
    lens_out[b] = cnt;
  }
}

}  // namespace mtvaf

using namespace mtvaf;

extern "C" {

size_t mtvaf_crf_workspace_bytes(int B, int S, int C) {
  return ((size_t)B * S * C + (size_t)B * 2 + (size_t)B * (2 * C + C * C)) * sizeof(float);
}

// loss[0] = -mean_b llh.  workspace keeps alpha/logZ/llh for mtvaf_crf_nll_bwd (same pointer).
int mtvaf_crf_nll_fwd(const float* emissions, const int64_t* tags, const uint8_t* mask, const float* start,
                      const float* end, const float* trans, float* loss, int B, int S, int C, void* workspace,
                      size_t workspace_bytes, hipStream_t st) {
  if (B <= 0 || S <= 0 || C <= 0 || C > CMAX) return MTVAF_ERR_SHAPE;
  if (workspace_bytes < mtvaf_crf_workspace_bytes(B, S, C)) return MTVAF_ERR_WORKSPACE;
  float* alpha = (float*)workspace;
  float* logz = alpha + (size_t)B * S * C;
  float* llh = logz + B;
  const size_t lds_f = ((size_t)S * C + 2 * (size_t)S) * sizeof(float) + (size_t)S * sizeof(int) + (size_t)S;
  if (lds_f > 64 * 1024) return MTVAF_ERR_SHAPE;  // S * C <= ~15000 (S = 512, C = 11 uses 29 KB)
  hipLaunchKernelGGL(crf_fwd_kernel, dim3(B), dim3(64), lds_f, st, emissions, tags, mask, start, end, trans, alpha, logz,
                     llh, S, C);
  hipLaunchKernelGGL(crf_loss_kernel, dim3(1), dim3(64), 0, st, llh, loss, B);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

// grad_out: device scalar d(total)/d(loss) (NULL = 1).  demissions overwritten; dstart/dend/dtrans
// overwritten or accumulated.
int mtvaf_crf_nll_bwd(const float* grad_out, const float* emissions, const int64_t* tags, const uint8_t* mask,
                      const float* start, const float* end, const float* trans, float* demissions, float* dstart,
                      float* dend, float* dtrans, int accumulate, int B, int S, int C, void* workspace,
                      size_t workspace_bytes, hipStream_t st) {
  if (B <= 0 || S <= 0 || C <= 0 || C > CMAX) return MTVAF_ERR_SHAPE;
  if (workspace_bytes < mtvaf_crf_workspace_bytes(B, S, C)) return MTVAF_ERR_WORKSPACE;
  float* alpha = (float*)workspace;
  float* logz = alpha + (size_t)B * S * C;
  float* partial = logz + 2 * B;
  const size_t lds_b = ((size_t)2 * S * C + S + CMAX * CMAX) * sizeof(float) + (size_t)S * sizeof(int) + (size_t)S;
  if (lds_b > 64 * 1024) return MTVAF_ERR_SHAPE;  // S * C <= ~7500 (S = 512, C = 11 uses 51 KB)
  hipLaunchKernelGGL(crf_bwd_kernel, dim3(B), dim3(64), lds_b, st, emissions, tags, mask, end, trans, alpha, grad_out,
                     demissions, partial, B, S, C);
  const int n = 2 * C + C * C;
  hipLaunchKernelGGL(crf_param_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st, partial, grad_out, B, C, dstart,
                     dend, dtrans, accumulate);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

// tags_out [B,S] int32 (best path, -1 padded), lens_out [B] int32.
int mtvaf_crf_viterbi(const float* emissions, const uint8_t* mask, const float* start, const float* end,
                      const float* trans, int32_t* tags_out, int32_t* lens_out, int B, int S, int C, hipStream_t st) {
  if (B <= 0 || S <= 0 || C <= 0 || C > CMAX) return MTVAF_ERR_SHAPE;
  if ((size_t)S * C * sizeof(float) + (size_t)S * (CMAX + 2) > 64 * 1024) return MTVAF_ERR_SHAPE;
  const size_t lds_v = (size_t)S * C * sizeof(float) + (size_t)((S + 15) & ~15) + (size_t)S * CMAX;
  hipLaunchKernelGGL(crf_viterbi_kernel, dim3(B), dim3(64), lds_v, st, emissions, mask, start, end, trans, tags_out,
                     lens_out, S, C);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

}  // extern "C"
