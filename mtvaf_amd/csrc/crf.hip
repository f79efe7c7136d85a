// Linear-chain CRF head of TVNetSAModel2 (third-party `pytorch-crf` in the reference; call sites
// models/bert_model.py:464 ctor, :511 decode, :521 negative log-likelihood with reduction='mean').
//   K13 crf_nll fwd/bwd : gold-path score - log-partition (masked forward algorithm), analytic
//                         backward through the forward/backward marginals
//   K12 crf_viterbi     : masked Viterbi with back-pointers; one packed int32 [B,S] tensor out
// One wavefront per sequence: tag j lives on lane j (C <= 16), the C x C transition matrix sits in
// registers, and the logsumexp / max over the previous tag is a shuffle loop.  The recursion over S
// is inherently serial, so these kernels are latency-bound by design (B waves of ~S*100 cycles).
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
// forward: alpha[b,t,:] (after step t), logZ[b], llh[b] = score(gold) - logZ
// ---------------------------------------------------------------------------------------------
// The recursion over S is serial, so every global load inside it would expose its full latency once per
// step: the per-sequence operands (emissions S x C, mask, tags, and alpha in the backward) are therefore
// staged in LDS first (dynamic LDS: S*C floats [+ S*C alpha] + S ints + S bytes).
__global__ __launch_bounds__(64) void crf_fwd_kernel(const float* __restrict__ em, const int64_t* __restrict__ tags,
                                                    const uint8_t* __restrict__ mask, const float* __restrict__ start,
                                                    const float* __restrict__ end, const float* __restrict__ trans,
                                                    float* __restrict__ alpha_ws, float* __restrict__ logz,
                                                    float* __restrict__ llh, int S, int C) {
  extern __shared__ __attribute__((aligned(16))) float crf_lds[];
  float* e = crf_lds;                                   // [S*C]
  int* tg = reinterpret_cast<int*>(e + S * C);          // [S]
  uint8_t* mk = reinterpret_cast<uint8_t*>(tg + S);     // [S]
  const int b = blockIdx.x, lane = threadIdx.x;
  for (int i = lane; i < S * C; i += 64) e[i] = em[(long)b * S * C + i];
  for (int i = lane; i < S; i += 64) {
    tg[i] = (int)tags[(long)b * S + i];
    mk[i] = mask[(long)b * S + i];
  }
  __syncthreads();
  const bool act = lane < C;
  const int j = act ? lane : 0;
  float tcol[CMAX];
#pragma unroll
  for (int i = 0; i < CMAX; ++i) tcol[i] = (i < C) ? trans[i * C + j] : 0.f;
  float alpha = act ? start[j] + e[j] : NEG;
  if (act && alpha_ws) alpha_ws[((long)b * S) * C + j] = alpha;
  for (int t = 1; t < S; ++t) {
    float v[CMAX];
    float m = NEG;
#pragma unroll
    for (int i = 0; i < CMAX; ++i) {
      const float ai = bcast(alpha, i);
      v[i] = (i < C) ? ai + tcol[i] : NEG;
      m = fmaxf(m, v[i]);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CMAX; ++i) s += (i < C) ? __expf(v[i] - m) : 0.f;
    const float nxt = lse2(m, s) + e[t * C + j];
    if (mk[t] && act) alpha = nxt;
    if (act && alpha_ws) alpha_ws[((long)b * S + t) * C + j] = alpha;
  }
  float fin = act ? alpha + end[j] : NEG;
  const float m = wave_max(fin);
  const float z = lse2(m, wave_sum(act ? __expf(fin - m) : 0.f));
  // gold path score, lanes stride over t
  float sc = 0.f;
  int cnt = 0;
  for (int t = lane; t < S; t += 64) {
    cnt += mk[t] ? 1 : 0;
    if (t >= 1 && mk[t]) sc += trans[tg[t - 1] * C + tg[t]] + e[t * C + tg[t]];
  }
  sc = wave_sum(sc);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if (lane == 0) {
    sc += start[tg[0]] + e[tg[0]] + end[tg[cnt - 1]];
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
                                                    const float* __restrict__ logz, const float* __restrict__ gout,
                                                    float* __restrict__ dem, float* __restrict__ partial, int B, int S,
                                                    int C) {
  extern __shared__ __attribute__((aligned(16))) float crf_lds[];
  float* e = crf_lds;                                   // [S*C]
  float* al = e + S * C;                                // [S*C]
  int* tg = reinterpret_cast<int*>(al + S * C);         // [S]
  uint8_t* mk = reinterpret_cast<uint8_t*>(tg + S);     // [S]
  const int b = blockIdx.x, lane = threadIdx.x;
  for (int i = lane; i < S * C; i += 64) {
    e[i] = em[(long)b * S * C + i];
    al[i] = alpha_ws[(long)b * S * C + i];
  }
  for (int i = lane; i < S; i += 64) {
    tg[i] = (int)tags[(long)b * S + i];
    mk[i] = mask[(long)b * S + i];
  }
  __syncthreads();
  const bool act = lane < C;
  const int j = act ? lane : 0;
  const float g = (gout ? *gout : 1.f) / B;
  float tcol[CMAX], trow[CMAX], eacc[CMAX];
#pragma unroll
  for (int i = 0; i < CMAX; ++i) {
    tcol[i] = (i < C) ? trans[i * C + j] : 0.f;   // trans[i][lane]
    trow[i] = (i < C) ? trans[j * C + i] : 0.f;   // trans[lane][i]
    eacc[i] = 0.f;
  }
  float* de = dem + (long)b * S * C;
  const float z = logz[b];
  int cnt = 0;
  for (int t = lane; t < S; t += 64) cnt += mk[t] ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  const int last_tag = (int)tg[cnt - 1];

  float beta = act ? end[j] : NEG;
  float dend = act ? __expf(al[(long)(S - 1) * C + j] + end[j] - z) - (j == last_tag ? 1.f : 0.f) : 0.f;
  for (int t = S - 1; t >= 1; --t) {
    if (!mk[t]) {
      if (act) de[t * C + j] = 0.f;
      continue;
    }
    const float et = e[t * C + j];
    // node marginal at t
    if (act) de[t * C + j] = g * (__expf(al[(long)t * C + j] + beta - z) - (j == (int)tg[t] ? 1.f : 0.f));
    const float eb = act ? et + beta : NEG;  // emit[t][j] + beta_t[j]
    const float aprev = act ? al[(long)(t - 1) * C + j] : NEG;  // alpha_{t-1}[lane]
    // edge marginals: lane j accumulates over previous tag i
    const int gi = (int)tg[t - 1], gj = (int)tg[t];
#pragma unroll
    for (int i = 0; i < CMAX; ++i) {
      const float ai = bcast(aprev, i);
      if (i < C && act) eacc[i] += __expf(ai + tcol[i] + eb - z) - ((i == gi && j == gj) ? 1.f : 0.f);
    }
    // beta_{t-1}[lane] = lse_k(trans[lane][k] + emit[t][k] + beta_t[k])
    float v[CMAX];
    float m = NEG;
#pragma unroll
    for (int k = 0; k < CMAX; ++k) {
      const float ebk = bcast(eb, k);
      v[k] = (k < C) ? trow[k] + ebk : NEG;
      m = fmaxf(m, v[k]);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < CMAX; ++k) s += (k < C) ? __expf(v[k] - m) : 0.f;
    beta = act ? lse2(m, s) : NEG;
  }
  if (act) {
    const float pm = __expf(al[j] + beta - z) - (j == (int)tg[0] ? 1.f : 0.f);
    de[j] = g * pm;
    float* pp = partial + (long)b * (2 * C + C * C);
    pp[j] = pm;
    pp[C + j] = dend;
#pragma unroll
    for (int i = 0; i < CMAX; ++i)
      if (i < C) pp[2 * C + i * C + j] = eacc[i];
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
  const size_t lds_f = (size_t)S * C * sizeof(float) + (size_t)S * sizeof(int) + (size_t)S;
  if (lds_f > 64 * 1024) return MTVAF_ERR_SHAPE;  // S * C <= ~16000 (S = 512, C = 11 uses 25 KB)
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
  const size_t lds_b = (size_t)2 * S * C * sizeof(float) + (size_t)S * sizeof(int) + (size_t)S;
  if (lds_b > 64 * 1024) return MTVAF_ERR_SHAPE;  // S * C <= ~8000 (S = 512, C = 11 uses 48 KB)
  hipLaunchKernelGGL(crf_bwd_kernel, dim3(B), dim3(64), lds_b, st, emissions, tags, mask, end, trans, alpha, logz, grad_out,
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
