// Linear-chain CRF head of TVNetSAModel2 (third-party `pytorch-crf` in the reference; call sites
// models/bert_model.py:464 ctor, :511 decode, :521 negative log-likelihood with reduction='mean').
//   K13 crf_nll fwd/bwd : gold-path score - log-partition (masked forward algorithm), analytic
//                         backward through the forward/backward marginals
//   K12 crf_viterbi     : masked Viterbi with back-pointers; one packed int32 [B,S] tensor out
// One wavefront per sequence: tag j lives on lane j of every 16-lane DPP row (C <= 16; the four rows of the wave carry
// identical copies, so a store by "all lanes" is a store of one value).  The recursion over S is inherently serial and a
// lone wave issues one instruction per ~4 cycles, so the kernels are built to MINIMISE THE INSTRUCTIONS PER TIME STEP:
//   * the 16-term matrix-vector product of a step is 16 DPP-fused multiply-adds (v_fmac_f32_dpp row_ror:k on the
//     vector, the matrix pre-rotated into per-lane registers) -- no lane broadcasts, no LDS crossbar;
//   * the mask is a 64-bit scalar word per 64 steps (wave-uniform branch on an SGPR bit), emission factors are fetched
//     one step ahead, everything a step produces leaves through stores with immediate offsets;
//   * whatever is not part of the serial chain is a small GEMM over all time steps on the matrix cores
//     (v_mfma_f32_16x16x4_f32): the predicted alphas of the backward pass and the summed edge marginals.
#include "common.h"

namespace mtvaf {

constexpr int CMAX = 16;
constexpr float NEG = -1.0e30f;
typedef float crf_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lse2(float m, float s) { return m + __logf(s); }
// broadcast lane `i` of x to the whole wave: v_readlane_b32 into an SGPR (Viterbi: max-plus has no fused DPP form)
__device__ __forceinline__ float bcast(float x, int i) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), i));
}
template <int K>
__device__ __forceinline__ float ror16(float v) {  // DPP row_ror:K -- a rotation of every 16-lane row by K lanes
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + K, 0xf, 0xf, true));
}
// direction of the rotation, probed instead of assumed: lane j of a row reads lane (j - K * rot_dir()) & 15
__device__ __forceinline__ int rot_dir(int lane) {
  const int j = lane & 15;
  return (j - __builtin_amdgcn_update_dpp(0, j, 0x121, 0xf, 0xf, true)) & 15;
}
// sum / max over the 16 lanes of a DPP row, result in every lane of it (bit-identical across the lanes: the rotation
// tree adds the same pairs everywhere).  Must run outside lane-divergent control flow.
__device__ __forceinline__ float row_sum16(float v) {
  v += ror16<8>(v);
  v += ror16<4>(v);
  v += ror16<2>(v);
  v += ror16<1>(v);
  return v;
}
__device__ __forceinline__ float row_max16(float v) {
  v = fmaxf(v, ror16<8>(v));
  v = fmaxf(v, ror16<4>(v));
  v = fmaxf(v, ror16<2>(v));
  v = fmaxf(v, ror16<1>(v));
  return v;
}
// One time step of the recursions, hand-scheduled: a lone wave issues one instruction per slot, so the step is written to
// be as few slots as possible with every hazard covered by useful work instead of s_nop:
//   * s[j] = sum_k v[(j - k d) & 15] * w[k] is 1 multiply + 15 v_fmac_f32_dpp (the compiler keeps v_mov_b32_dpp +
//     v_fmac + s_nop: three slots per term).  With w[k] = M[(j - k d) & 15][j] this is (v^T M)[j], with
//     w[k] = M[j][(j - k d) & 15] it is (M v)[j];
//   * the normaliser's row sum (4 DPP adds, each needing two slots of distance from the write it reads) and the
//     reciprocal (one slot of distance) are interleaved with that chain.
#define CRF_DPP(k) " row_ror:" #k " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define CRF_FMAC(k) "v_fmac_f32_dpp %[s], %[v], %[w" #k "]" CRF_DPP(k)
#define CRF_W(w) [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), [w4] "v"(w[4]), [w5] "v"(w[5]),          \
                 [w6] "v"(w[6]), [w7] "v"(w[7]), [w8] "v"(w[8]), [w9] "v"(w[9]), [w10] "v"(w[10]), [w11] "v"(w[11]),      \
                 [w12] "v"(w[12]), [w13] "v"(w[13]), [w14] "v"(w[14]), [w15] "v"(w[15])

// forward step:  n = sum_i a[i] y[i]  (= sum_j (a^T E)[j] x[j] with y = E x),  a <- (a^T E) x / n
__device__ __forceinline__ void crf_fwd_step(float& a, float x, float y, const float (&w)[16], float& n) {
  float s, r, xr;
  asm volatile(
      "v_mul_f32 %[p], %[v], %[y]\n"
      "v_mul_f32 %[s], %[v], %[w0]\n"
      CRF_FMAC(1) CRF_FMAC(2)
      "v_add_f32_dpp %[p], %[p], %[p]" CRF_DPP(8)
      CRF_FMAC(3) CRF_FMAC(4)
      "v_add_f32_dpp %[p], %[p], %[p]" CRF_DPP(4)
      CRF_FMAC(5) CRF_FMAC(6)
      "v_add_f32_dpp %[p], %[p], %[p]" CRF_DPP(2)
      CRF_FMAC(7) CRF_FMAC(8)
      "v_add_f32_dpp %[p], %[p], %[p]" CRF_DPP(1)
      CRF_FMAC(9)
      "v_rcp_f32 %[r], %[p]\n"
      CRF_FMAC(10) CRF_FMAC(11)
      "v_mul_f32 %[xr], %[x], %[r]\n"
      CRF_FMAC(12) CRF_FMAC(13) CRF_FMAC(14) CRF_FMAC(15)
      "v_mul_f32 %[v], %[s], %[xr]\n"
      : [v] "+v"(a), [p] "=&v"(n), [s] "=&v"(s), [r] "=&v"(r), [xr] "=&v"(xr)
      : [x] "v"(x), [y] "v"(y), CRF_W(w));
}

// backward step:  u = x b,  ui = u / (sp . u),  pm = sp ui,  b <- (E u) / (sp . u)
__device__ __forceinline__ void crf_bwd_step(float& bt, float x, float sp, const float (&w)[16], float& ui, float& pm) {
  float v, p, s, r;
  asm volatile(
      "v_mul_f32 %[v], %[x], %[b]\n"
      "v_mul_f32 %[p], %[sp], %[v]\n"
      "v_mul_f32 %[s], %[v], %[w0]\n"
      CRF_FMAC(1) CRF_FMAC(2)
      "v_add_f32_dpp %[p], %[p], %[p]" CRF_DPP(8)
      CRF_FMAC(3) CRF_FMAC(4)
      "v_add_f32_dpp %[p], %[p], %[p]" CRF_DPP(4)
      CRF_FMAC(5) CRF_FMAC(6)
      "v_add_f32_dpp %[p], %[p], %[p]" CRF_DPP(2)
      CRF_FMAC(7) CRF_FMAC(8)
      "v_add_f32_dpp %[p], %[p], %[p]" CRF_DPP(1)
      CRF_FMAC(9)
      "v_rcp_f32 %[r], %[p]\n"
      CRF_FMAC(10) CRF_FMAC(11)
      "v_mul_f32 %[ui], %[v], %[r]\n"
      CRF_FMAC(12)
      "v_mul_f32 %[pm], %[sp], %[ui]\n"
      CRF_FMAC(13) CRF_FMAC(14) CRF_FMAC(15)
      "v_mul_f32 %[b], %[s], %[r]\n"
      : [b] "+v"(bt), [ui] "=&v"(ui), [pm] "=&v"(pm), [v] "=&v"(v), [p] "=&v"(p), [s] "=&v"(s), [r] "=&v"(r)
      : [x] "v"(x), [sp] "v"(sp), CRF_W(w));
}

// ---------------------------------------------------------------------------------------------
// forward / backward recursions in the SCALED LINEAR domain.  The log-domain form costs C exp + a log per lane and time
// step on a serial critical path (S steps); here
//     a_t[j] = (sum_i a_{t-1}[i] E[i][j]) x_t[j] / n_t,   E = exp(trans - tmax),  x_t = exp(emit_t - max_j emit_t)
// is one rotated dot product per step, the per-step normaliser n_t (sum over the tag lanes: four DPP row rotations)
// keeps every a_t at sum 1, and  logZ = c0 + sum_t (log n_t + max_j emit_t + tmax) + log sum_j a_last[j] e^{end[j]}
// takes its S logarithms in parallel after the loop.  The emission factors x_t are computed for all t up front, in
// parallel.  The backward pass runs the same way on b_t (scaled so that sum_i a_t[i] b_t[i] = 1): node and edge marginals
// are products of registers, normalised by one reciprocal per step -- no exp, no log on the serial path.
// LDS rows are 16 floats per time step (tags beyond C hold zeros), so row t of every array sits at t * 64 bytes.
// ---------------------------------------------------------------------------------------------

// stage one sequence: xs16[t][j] = exp(emit[t][j] - mx[t]) (0 for j >= C and for rows S <= t < rows), mx[t], mask
__device__ __forceinline__ void crf_stage(const float* __restrict__ em, const uint8_t* __restrict__ mask, long b, int S,
                                          int rows, int C, float* xs16, float* mxs, uint8_t* mk, int lane) {
  for (int t = lane; t < rows; t += 64) {
    float v[CMAX];
    float m = NEG;
    if (t < S) {
      const float* e = em + ((long)b * S + t) * C;
#pragma unroll
      for (int j = 0; j < CMAX; ++j) {
        v[j] = j < C ? e[j] : NEG;
        m = fmaxf(m, v[j]);
      }
#pragma unroll
      for (int j = 0; j < CMAX; ++j) v[j] = j < C ? __expf(v[j] - m) : 0.f;
      if (mxs) mxs[t] = m;
      mk[t] = mask[(long)b * S + t];
    } else {
#pragma unroll
      for (int j = 0; j < CMAX; ++j) v[j] = 0.f;
    }
    crf_f4* x = reinterpret_cast<crf_f4*>(xs16 + t * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] = crf_f4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
  }
}

// the mask of steps [t0, t0 + 64) as a wave-uniform 64-bit word
__device__ __forceinline__ unsigned long long mask_word(const uint8_t* mk, int t0, int S, int lane) {
  return __ballot((t0 + lane < S) && mk[(t0 + lane < S) ? t0 + lane : 0] != 0);
}

// forward: alpha_ws[b,t,0:16] = scaled alpha after step t (sum 1 for t >= 1), logZ[b], llh[b] = score(gold) - logZ
__global__ __launch_bounds__(64) void crf_fwd_kernel(const float* __restrict__ em, const int64_t* __restrict__ tags,
                                                    const uint8_t* __restrict__ mask, const float* __restrict__ start,
                                                    const float* __restrict__ end, const float* __restrict__ trans,
                                                    float* __restrict__ alpha_ws, float* __restrict__ logz,
                                                    float* __restrict__ llh, int S, int C) {
  extern __shared__ __attribute__((aligned(16))) float crf_lds[];
  const int S16 = (S + 15) & ~15;
  float* xs16 = crf_lds;                                // [(S16+1)*16] emission factors x_t (zero rows beyond S)
  float* ys16 = xs16 + (S16 + 1) * 16;                  // [(S16+1)*16] y_t = E x_t; slot [t][*] takes n_t once step t ran
  float* mxs = ys16 + (S16 + 1) * 16;                   // [S]   per-step emission maxima
  int* tg = reinterpret_cast<int*>(mxs + S);            // [S]
  uint8_t* mk = reinterpret_cast<uint8_t*>(tg + S);     // [S]
  const int b = blockIdx.x, lane = threadIdx.x;
  crf_stage(em, mask, b, S, S16 + 1, C, xs16, mxs, mk, lane);
  for (int t = lane; t < S; t += 64) tg[t] = (int)tags[(long)b * S + t];
  __syncthreads();
  const int j = lane & 15, g4 = lane >> 4, d = rot_dir(lane);
  const bool act = j < C;
  float tm = NEG;
  for (int i = 0; i < C; ++i) tm = fmaxf(tm, act ? trans[i * C + j] : NEG);
  const float tmax = row_max16(tm);
  // y_t[i] = sum_j E[i][j] x_t[j] for all t on the matrix cores: A[m = t][k = j] = x, B[k = j][n = i] = E[i][j]
  {
    float eb[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int jj = 4 * kb + g4;  // (this lane's n is the FROM tag j = lane & 15)
      eb[kb] = (act && jj < C) ? __expf(trans[j * C + jj] - tmax) : 0.f;
    }
    for (int t0 = 0; t0 < S16; t0 += 16) {
      crf_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xs16[(t0 + j) * 16 + 4 * kb + g4], eb[kb], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) ys16[(t0 + 4 * g4 + r) * 16 + j] = acc[r];
    }
  }
  float w[CMAX];  // w[k] = E[i_k][j], i_k = (j - k d) & 15: the lane this row rotation brings to lane j
#pragma unroll
  for (int k = 0; k < CMAX; ++k) {
    const int i = (j - k * d) & 15;
    w[k] = (act && i < C) ? __expf(trans[i * C + j] - tmax) : 0.f;
  }
  const float a0l = act ? start[j] + em[((long)b * S) * C + j] : NEG;
  const float c0 = row_max16(a0l);
  float a = act ? __expf(a0l - c0) : 0.f;
  float* aw = alpha_ws + (long)b * S * 16 + j;
  aw[0] = a;
  __syncthreads();
  const float* xr = xs16 + j;
  float* yr = ys16 + j;
  float xn = xr[16], yn = yr[16];
#define CRF_FWD_STEP(t_, on_)                                                                                          \
  {                                                                                                                    \
    const int t_s = (t_);                                                                                              \
    const float x = xn, y = yn;                                                                                        \
    xn = xr[(t_s + 1) * 16];                                                                                           \
    yn = yr[(t_s + 1) * 16];                                                                                           \
    if (on_) { /* (wave-uniform, scalar) */                                                                            \
      float n;                                                                                                         \
      crf_fwd_step(a, x, y, w, n);                                                                                     \
      yr[t_s * 16] = n;                                                                                                \
    }                                                                                                                  \
    aw[(long)t_s * 16] = a;                                                                                            \
  }
  {
    int t = 1;
    unsigned long long mb = mask_word(mk, 0, S, lane);
    for (; t < S && (t & 7); ++t) CRF_FWD_STEP(t, (mb >> t) & 1)
    for (; t + 8 <= S; t += 8) {
      if ((t & 63) == 0) mb = mask_word(mk, t, S, lane);
      const unsigned m8 = (unsigned)(mb >> (t & 63)) & 0xffu;
#pragma unroll
      for (int q = 0; q < 8; ++q) CRF_FWD_STEP(t + q, (m8 >> q) & 1)
    }
    for (; t < S; ++t) {
      if ((t & 63) == 0) mb = mask_word(mk, t, S, lane);
      CRF_FWD_STEP(t, (mb >> (t & 63)) & 1)
    }
  }
#undef CRF_FWD_STEP
  const float em_ = row_max16(act ? end[j] : NEG);
  const float fin = row_sum16(act ? a * __expf(end[j] - em_) : 0.f);
  __syncthreads();
  // logZ: the S logarithms in parallel; gold path score, lanes stride over t
  float lz = 0.f, sc = 0.f;
  int cnt = 0;
  for (int t = lane; t < S; t += 64) {
    cnt += mk[t] ? 1 : 0;
    if (t >= 1 && mk[t]) {
      lz += __logf(ys16[t * 16]) + mxs[t] + tmax;
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
//   1. predicted alphas  sp_t[j] = sum_i a_{t-1}[i] E[i][j]  for ALL t: one [S,16] x [16,16] product on the matrix cores;
//   2. the serial part, t = S-1 .. 1:  u = x_t b_t,  b_{t-1} = (E u) / (sp_t . u)  -- one rotated dot product and one
//      row sum per step; the normalised  ui_t = u / (sp_t . u)  and the node marginal  sp_t ui_t  go to LDS;
//   3. summed edge marginals  sum_t a_{t-1}[i] E[i][j] ui_t[j] = E[i][j] (A^T UI)[i][j]: a [16,S] x [S,16] product on the
//      matrix cores; d(emissions) leaves LDS in one coalesced pass (gold one-hots subtracted there).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void crf_bwd_kernel(const float* __restrict__ em, const int64_t* __restrict__ tags,
                                                    const uint8_t* __restrict__ mask, const float* __restrict__ end,
                                                    const float* __restrict__ trans, const float* __restrict__ alpha_ws,
                                                    const float* __restrict__ gout, float* __restrict__ dem,
                                                    float* __restrict__ partial, int B, int S, int C) {
  extern __shared__ __attribute__((aligned(16))) float crf_lds[];
  const int S16 = (S + 15) & ~15;
  float* xs16 = crf_lds;                                // [S16*16] x_t, overwritten by the node marginals of step t
  float* al16 = xs16 + S16 * 16;                        // [S16*16] scaled alphas of the forward pass (zero rows beyond S)
  float* sp16 = al16 + S16 * 16;                        // [S16*16] predicted alphas, overwritten by ui_t
  float* gold = sp16 + S16 * 16;                        // [16*16] gold transition counts
  int* tg = reinterpret_cast<int*>(gold + CMAX * CMAX); // [S]
  uint8_t* mk = reinterpret_cast<uint8_t*>(tg + S);     // [S]
  const int b = blockIdx.x, lane = threadIdx.x;
  crf_stage(em, mask, b, S, S16, C, xs16, nullptr, mk, lane);
  {
    const crf_f4* src = reinterpret_cast<const crf_f4*>(alpha_ws + (long)b * S * 16);
    crf_f4* dst = reinterpret_cast<crf_f4*>(al16);
    for (int i = lane; i < S16 * 4; i += 64) dst[i] = i < S * 4 ? src[i] : crf_f4{0.f, 0.f, 0.f, 0.f};
  }
  for (int t = lane; t < S; t += 64) tg[t] = (int)tags[(long)b * S + t];
  for (int i = lane; i < CMAX * CMAX; i += 64) gold[i] = 0.f;
  __syncthreads();
  for (int t = 1 + lane; t < S; t += 64)
    if (mk[t]) atomicAdd(&gold[tg[t - 1] * C + tg[t]], 1.f);  // (integer-valued sums: exact, order-independent)
  const int j = lane & 15, g4 = lane >> 4, d = rot_dir(lane);
  const bool act = j < C;
  const float g = (gout ? *gout : 1.f) / B;
  float tm = NEG;
  for (int i = 0; i < C; ++i) tm = fmaxf(tm, act ? trans[i * C + j] : NEG);
  const float tmax = row_max16(tm);
  // 1. sp16[t][j] for t = 0 .. S16-1 (row 0 unused).  A[m = t][k = i] from al16 rows t-1, B[k = i][n = j] = E
  {
    float eb[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int i = 4 * kb + g4;
      eb[kb] = (i < C && act) ? __expf(trans[i * C + j] - tmax) : 0.f;
    }
    for (int t0 = 0; t0 < S16; t0 += 16) {
      const int ta = t0 + j - 1;  // (this lane's A row: time step t0 + j reads alpha of the step before)
      crf_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const float av = ta >= 0 ? al16[ta * 16 + 4 * kb + g4] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, eb[kb], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) sp16[(t0 + 4 * g4 + r) * 16 + j] = acc[r];
    }
  }
  float wr[CMAX];  // wr[k] = E[j][i_k], i_k = (j - k d) & 15
#pragma unroll
  for (int k = 0; k < CMAX; ++k) {
    const int i = (j - k * d) & 15;
    wr[k] = (act && i < C) ? __expf(trans[j * C + i] - tmax) : 0.f;
  }
  int cnt = 0;
  for (int t = lane; t < S; t += 64) cnt += mk[t] ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  const int last_tag = (int)tg[cnt - 1];
  const float em_ = row_max16(act ? end[j] : NEG);
  float bt = act ? __expf(end[j] - em_) : 0.f;  // beta of the last position, any positive scale
  __syncthreads();
  float dend;
  {
    const float pe = al16[(S - 1) * 16 + j] * bt;
    const float pen = pe * __frcp_rn(row_sum16(pe));
    dend = pen - (j == last_tag ? 1.f : 0.f);
  }
  // 2. the serial part
  {
    const float* ar = xs16 + j;  // (x rows are read, node marginals written through the same row pointer)
    float* xw = xs16 + j;
    float* sr = sp16 + j;
    float xp = ar[(S - 1) * 16], spp = sr[(S - 1) * 16];
#define CRF_BWD_STEP(t_, on_)                                                                                          \
  {                                                                                                                    \
    const int t_s = (t_);                                                                                              \
    const float x = xp, sp = spp;                                                                                      \
    xp = ar[(t_s - 1) * 16];                                                                                           \
    spp = sr[(t_s - 1) * 16];                                                                                          \
    float ui = 0.f, pm = 0.f;                                                                                          \
    if (on_) crf_bwd_step(bt, x, sp, wr, ui, pm); /* (wave-uniform, scalar) */                                        \
    sr[t_s * 16] = ui;                                                                                                 \
    xw[t_s * 16] = pm;                                                                                                 \
  }
    int t = S - 1;
    unsigned long long mb = mask_word(mk, t & ~63, S, lane);
    for (; t >= 1 && (t & 7) != 7; --t) CRF_BWD_STEP(t, (mb >> (t & 63)) & 1)
    for (; t >= 8; t -= 8) {  // (t & 7) == 7: steps t-7 .. t, all >= 1
      if ((t & 63) == 63) mb = mask_word(mk, t & ~63, S, lane);
      const unsigned m8 = (unsigned)(mb >> ((t - 7) & 63)) & 0xffu;
#pragma unroll
      for (int q = 7; q >= 0; --q) CRF_BWD_STEP(t - 7 + q, (m8 >> q) & 1)
    }
    for (; t >= 1; --t) {
      if ((t & 63) == 63) mb = mask_word(mk, t & ~63, S, lane);
      CRF_BWD_STEP(t, (mb >> (t & 63)) & 1)
    }
#undef CRF_BWD_STEP
  }
  {
    const float p0 = al16[j] * bt;
    const float p0n = p0 * __frcp_rn(row_sum16(p0));
    xs16[j] = p0n;
    sp16[j] = 0.f;  // no edge into step 0
    for (int i = S * 16 + lane; i < S16 * 16; i += 64) sp16[i] = 0.f;  // rows beyond S: no edges either
    float* pp = partial + (long)b * (2 * C + C * C);
    if (lane < C) {
      pp[j] = p0n - (j == (int)tg[0] ? 1.f : 0.f);
      pp[C + j] = dend;
    }
  }
  __syncthreads();
  // 3. G[i][j] = sum_t a_{t-1}[i] ui_t[j]: A[m = i][k = t] from al16 rows t-1, B[k = t][n = j] = ui rows t
  {
    crf_f4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = {0.f, 0.f, 0.f, 0.f};
    for (int t0 = 0; t0 < S16; t0 += 8) {
      const int ta = t0 + g4, tb = ta + 4;
      const float a0 = ta >= 1 ? al16[(ta - 1) * 16 + j] : 0.f;
      g0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, sp16[ta * 16 + j], g0, 0, 0, 0);
      g1 = __builtin_amdgcn_mfma_f32_16x16x4f32(al16[(tb - 1) * 16 + j], sp16[tb * 16 + j], g1, 0, 0, 0);
    }
    float* pp = partial + (long)b * (2 * C + C * C) + 2 * C;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * g4 + r;
      if (i < C && act) pp[i * C + j] = (g0[r] + g1[r]) * __expf(trans[i * C + j] - tmax) - gold[i * C + j];
    }
  }
  // d(emissions): node marginal - gold one-hot, coalesced
  float* de = dem + (long)b * S * C;
  for (int idx = lane; idx < S * C; idx += 64) {
    const int t = idx / C, jj = idx - t * C;
    de[idx] = (t == 0 || mk[t]) ? g * (xs16[t * 16 + jj] - (jj == tg[t] ? 1.f : 0.f)) : 0.f;
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
// Back-pointers are 4 bits per (step, tag): one packed word per lane and 8 steps; the backtrace walks them with one
// v_readlane per step (the current tag is wave-uniform) and the path leaves LDS in one coalesced store.
// ---------------------------------------------------------------------------------------------
// One Viterbi step for this lane's tag: best = max_i (score[i] + tc[i]) and the LOWEST i attaining it (the first maximum
// wins, as torch.max does).  Hand-scheduled in independent batches -- lane broadcasts into distinct SGPRs, adds, a
// max3 tree, compares into distinct SGPR pairs, selects in descending i -- because an instruction that reads an SGPR a
// VALU instruction has just written idles for two slots, and the compiler's schedule funnels every link of the chain
// through one SGPR (measured: ~100 slots per step instead of ~55).
template <int CT>
__device__ __forceinline__ void crf_vit_step(float score, const float (&tc)[CT], float& best, int& bi);
template <>
__device__ __forceinline__ void crf_vit_step<4>(float score, const float (&tc)[4], float& best, int& bi) {
  float v[4];
  asm volatile(
      "s_nop 0\n"  // a v_readlane must not directly follow the VALU write of the VGPR it reads (the score select)
      "v_readlane_b32 s40, %[sc], 0\n"
      "v_readlane_b32 s41, %[sc], 1\n"
      "v_readlane_b32 s42, %[sc], 2\n"
      "v_readlane_b32 s43, %[sc], 3\n"
      "v_add_f32 %[v0], s40, %[t0]\n"
      "v_add_f32 %[v1], s41, %[t1]\n"
      "v_add_f32 %[v2], s42, %[t2]\n"
      "v_add_f32 %[v3], s43, %[t3]\n"
      : [v0] "=&v"(v[0]), [v1] "=&v"(v[1]), [v2] "=&v"(v[2]), [v3] "=&v"(v[3])
      : [sc] "v"(score), [t0] "v"(tc[0]), [t1] "v"(tc[1]), [t2] "v"(tc[2]), [t3] "v"(tc[3])
      : "s40", "s41", "s42", "s43");
  asm volatile(
      "v_max3_f32 %[b], %[v0], %[v1], %[v2]\n"
      "v_max3_f32 %[b], %[b], %[v3], %[v3]\n"
      "v_mov_b32 %[i], 3\n"
      "v_cmp_eq_f32 s[40:41], %[v2], %[b]\n"
      "v_cmp_eq_f32 s[42:43], %[v1], %[b]\n"
      "v_cmp_eq_f32 s[44:45], %[v0], %[b]\n"
      "v_cndmask_b32 %[i], %[i], 2, s[40:41]\n"
      "v_cndmask_b32 %[i], %[i], 1, s[42:43]\n"
      "v_cndmask_b32 %[i], %[i], 0, s[44:45]\n"
      : [b] "=&v"(best), [i] "=&v"(bi)
      : [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3])
      : "s40", "s41", "s42", "s43", "s44", "s45");
}
template <>
__device__ __forceinline__ void crf_vit_step<8>(float score, const float (&tc)[8], float& best, int& bi) {
  float v[8];
  asm volatile(
      "s_nop 0\n"  // a v_readlane must not directly follow the VALU write of the VGPR it reads (the score select)
      "v_readlane_b32 s40, %[sc], 0\n"
      "v_readlane_b32 s41, %[sc], 1\n"
      "v_readlane_b32 s42, %[sc], 2\n"
      "v_readlane_b32 s43, %[sc], 3\n"
      "v_readlane_b32 s44, %[sc], 4\n"
      "v_readlane_b32 s45, %[sc], 5\n"
      "v_readlane_b32 s46, %[sc], 6\n"
      "v_readlane_b32 s47, %[sc], 7\n"
      "v_add_f32 %[v0], s40, %[t0]\n"
      "v_add_f32 %[v1], s41, %[t1]\n"
      "v_add_f32 %[v2], s42, %[t2]\n"
      "v_add_f32 %[v3], s43, %[t3]\n"
      "v_add_f32 %[v4], s44, %[t4]\n"
      "v_add_f32 %[v5], s45, %[t5]\n"
      "v_add_f32 %[v6], s46, %[t6]\n"
      "v_add_f32 %[v7], s47, %[t7]\n"
      : [v0] "=&v"(v[0]), [v1] "=&v"(v[1]), [v2] "=&v"(v[2]), [v3] "=&v"(v[3]), [v4] "=&v"(v[4]), [v5] "=&v"(v[5]), [v6] "=&v"(v[6]), [v7] "=&v"(v[7])
      : [sc] "v"(score), [t0] "v"(tc[0]), [t1] "v"(tc[1]), [t2] "v"(tc[2]), [t3] "v"(tc[3]), [t4] "v"(tc[4]), [t5] "v"(tc[5]), [t6] "v"(tc[6]), [t7] "v"(tc[7])
      : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
  asm volatile(
      "v_max3_f32 %[b], %[v0], %[v1], %[v2]\n"
      "v_max3_f32 %[b], %[b], %[v3], %[v4]\n"
      "v_max3_f32 %[b], %[b], %[v5], %[v6]\n"
      "v_max3_f32 %[b], %[b], %[v7], %[v7]\n"
      "v_mov_b32 %[i], 7\n"
      "v_cmp_eq_f32 s[40:41], %[v6], %[b]\n"
      "v_cmp_eq_f32 s[42:43], %[v5], %[b]\n"
      "v_cmp_eq_f32 s[44:45], %[v4], %[b]\n"
      "v_cmp_eq_f32 s[46:47], %[v3], %[b]\n"
      "v_cmp_eq_f32 s[48:49], %[v2], %[b]\n"
      "v_cmp_eq_f32 s[50:51], %[v1], %[b]\n"
      "v_cmp_eq_f32 s[52:53], %[v0], %[b]\n"
      "v_cndmask_b32 %[i], %[i], 6, s[40:41]\n"
      "v_cndmask_b32 %[i], %[i], 5, s[42:43]\n"
      "v_cndmask_b32 %[i], %[i], 4, s[44:45]\n"
      "v_cndmask_b32 %[i], %[i], 3, s[46:47]\n"
      "v_cndmask_b32 %[i], %[i], 2, s[48:49]\n"
      "v_cndmask_b32 %[i], %[i], 1, s[50:51]\n"
      "v_cndmask_b32 %[i], %[i], 0, s[52:53]\n"
      : [b] "=&v"(best), [i] "=&v"(bi)
      : [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [v4] "v"(v[4]), [v5] "v"(v[5]), [v6] "v"(v[6]), [v7] "v"(v[7])
      : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53");
}
template <>
__device__ __forceinline__ void crf_vit_step<12>(float score, const float (&tc)[12], float& best, int& bi) {
  float v[12];
  asm volatile(
      "s_nop 0\n"  // a v_readlane must not directly follow the VALU write of the VGPR it reads (the score select)
      "v_readlane_b32 s40, %[sc], 0\n"
      "v_readlane_b32 s41, %[sc], 1\n"
      "v_readlane_b32 s42, %[sc], 2\n"
      "v_readlane_b32 s43, %[sc], 3\n"
      "v_readlane_b32 s44, %[sc], 4\n"
      "v_readlane_b32 s45, %[sc], 5\n"
      "v_readlane_b32 s46, %[sc], 6\n"
      "v_readlane_b32 s47, %[sc], 7\n"
      "v_add_f32 %[v0], s40, %[t0]\n"
      "v_add_f32 %[v1], s41, %[t1]\n"
      "v_add_f32 %[v2], s42, %[t2]\n"
      "v_add_f32 %[v3], s43, %[t3]\n"
      "v_add_f32 %[v4], s44, %[t4]\n"
      "v_add_f32 %[v5], s45, %[t5]\n"
      "v_add_f32 %[v6], s46, %[t6]\n"
      "v_add_f32 %[v7], s47, %[t7]\n"
      : [v0] "=&v"(v[0]), [v1] "=&v"(v[1]), [v2] "=&v"(v[2]), [v3] "=&v"(v[3]), [v4] "=&v"(v[4]), [v5] "=&v"(v[5]), [v6] "=&v"(v[6]), [v7] "=&v"(v[7])
      : [sc] "v"(score), [t0] "v"(tc[0]), [t1] "v"(tc[1]), [t2] "v"(tc[2]), [t3] "v"(tc[3]), [t4] "v"(tc[4]), [t5] "v"(tc[5]), [t6] "v"(tc[6]), [t7] "v"(tc[7])
      : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
  asm volatile(
      "v_readlane_b32 s40, %[sc], 8\n"
      "v_readlane_b32 s41, %[sc], 9\n"
      "v_readlane_b32 s42, %[sc], 10\n"
      "v_readlane_b32 s43, %[sc], 11\n"
      "v_add_f32 %[v0], s40, %[t0]\n"
      "v_add_f32 %[v1], s41, %[t1]\n"
      "v_add_f32 %[v2], s42, %[t2]\n"
      "v_add_f32 %[v3], s43, %[t3]\n"
      : [v0] "=&v"(v[8]), [v1] "=&v"(v[9]), [v2] "=&v"(v[10]), [v3] "=&v"(v[11])
      : [sc] "v"(score), [t0] "v"(tc[8]), [t1] "v"(tc[9]), [t2] "v"(tc[10]), [t3] "v"(tc[11])
      : "s40", "s41", "s42", "s43");
  asm volatile(
      "v_max3_f32 %[b], %[v0], %[v1], %[v2]\n"
      "v_max3_f32 %[b], %[b], %[v3], %[v4]\n"
      "v_max3_f32 %[b], %[b], %[v5], %[v6]\n"
      "v_max3_f32 %[b], %[b], %[v7], %[v8]\n"
      "v_max3_f32 %[b], %[b], %[v9], %[v10]\n"
      "v_max3_f32 %[b], %[b], %[v11], %[v11]\n"
      "v_mov_b32 %[i], 11\n"
      "v_cmp_eq_f32 s[40:41], %[v10], %[b]\n"
      "v_cmp_eq_f32 s[42:43], %[v9], %[b]\n"
      "v_cmp_eq_f32 s[44:45], %[v8], %[b]\n"
      "v_cmp_eq_f32 s[46:47], %[v7], %[b]\n"
      "v_cmp_eq_f32 s[48:49], %[v6], %[b]\n"
      "v_cmp_eq_f32 s[50:51], %[v5], %[b]\n"
      "v_cmp_eq_f32 s[52:53], %[v4], %[b]\n"
      "v_cmp_eq_f32 s[54:55], %[v3], %[b]\n"
      "v_cmp_eq_f32 s[56:57], %[v2], %[b]\n"
      "v_cmp_eq_f32 s[58:59], %[v1], %[b]\n"
      "v_cmp_eq_f32 s[60:61], %[v0], %[b]\n"
      "v_cndmask_b32 %[i], %[i], 10, s[40:41]\n"
      "v_cndmask_b32 %[i], %[i], 9, s[42:43]\n"
      "v_cndmask_b32 %[i], %[i], 8, s[44:45]\n"
      "v_cndmask_b32 %[i], %[i], 7, s[46:47]\n"
      "v_cndmask_b32 %[i], %[i], 6, s[48:49]\n"
      "v_cndmask_b32 %[i], %[i], 5, s[50:51]\n"
      "v_cndmask_b32 %[i], %[i], 4, s[52:53]\n"
      "v_cndmask_b32 %[i], %[i], 3, s[54:55]\n"
      "v_cndmask_b32 %[i], %[i], 2, s[56:57]\n"
      "v_cndmask_b32 %[i], %[i], 1, s[58:59]\n"
      "v_cndmask_b32 %[i], %[i], 0, s[60:61]\n"
      : [b] "=&v"(best), [i] "=&v"(bi)
      : [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [v4] "v"(v[4]), [v5] "v"(v[5]), [v6] "v"(v[6]), [v7] "v"(v[7]), [v8] "v"(v[8]), [v9] "v"(v[9]), [v10] "v"(v[10]), [v11] "v"(v[11])
      : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61");
}
template <>
__device__ __forceinline__ void crf_vit_step<16>(float score, const float (&tc)[16], float& best, int& bi) {
  float v[16];
  asm volatile(
      "s_nop 0\n"  // a v_readlane must not directly follow the VALU write of the VGPR it reads (the score select)
      "v_readlane_b32 s40, %[sc], 0\n"
      "v_readlane_b32 s41, %[sc], 1\n"
      "v_readlane_b32 s42, %[sc], 2\n"
      "v_readlane_b32 s43, %[sc], 3\n"
      "v_readlane_b32 s44, %[sc], 4\n"
      "v_readlane_b32 s45, %[sc], 5\n"
      "v_readlane_b32 s46, %[sc], 6\n"
      "v_readlane_b32 s47, %[sc], 7\n"
      "v_add_f32 %[v0], s40, %[t0]\n"
      "v_add_f32 %[v1], s41, %[t1]\n"
      "v_add_f32 %[v2], s42, %[t2]\n"
      "v_add_f32 %[v3], s43, %[t3]\n"
      "v_add_f32 %[v4], s44, %[t4]\n"
      "v_add_f32 %[v5], s45, %[t5]\n"
      "v_add_f32 %[v6], s46, %[t6]\n"
      "v_add_f32 %[v7], s47, %[t7]\n"
      : [v0] "=&v"(v[0]), [v1] "=&v"(v[1]), [v2] "=&v"(v[2]), [v3] "=&v"(v[3]), [v4] "=&v"(v[4]), [v5] "=&v"(v[5]), [v6] "=&v"(v[6]), [v7] "=&v"(v[7])
      : [sc] "v"(score), [t0] "v"(tc[0]), [t1] "v"(tc[1]), [t2] "v"(tc[2]), [t3] "v"(tc[3]), [t4] "v"(tc[4]), [t5] "v"(tc[5]), [t6] "v"(tc[6]), [t7] "v"(tc[7])
      : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
  asm volatile(
      "v_readlane_b32 s40, %[sc], 8\n"
      "v_readlane_b32 s41, %[sc], 9\n"
      "v_readlane_b32 s42, %[sc], 10\n"
      "v_readlane_b32 s43, %[sc], 11\n"
      "v_readlane_b32 s44, %[sc], 12\n"
      "v_readlane_b32 s45, %[sc], 13\n"
      "v_readlane_b32 s46, %[sc], 14\n"
      "v_readlane_b32 s47, %[sc], 15\n"
      "v_add_f32 %[v0], s40, %[t0]\n"
      "v_add_f32 %[v1], s41, %[t1]\n"
      "v_add_f32 %[v2], s42, %[t2]\n"
      "v_add_f32 %[v3], s43, %[t3]\n"
      "v_add_f32 %[v4], s44, %[t4]\n"
      "v_add_f32 %[v5], s45, %[t5]\n"
      "v_add_f32 %[v6], s46, %[t6]\n"
      "v_add_f32 %[v7], s47, %[t7]\n"
      : [v0] "=&v"(v[8]), [v1] "=&v"(v[9]), [v2] "=&v"(v[10]), [v3] "=&v"(v[11]), [v4] "=&v"(v[12]), [v5] "=&v"(v[13]), [v6] "=&v"(v[14]), [v7] "=&v"(v[15])
      : [sc] "v"(score), [t0] "v"(tc[8]), [t1] "v"(tc[9]), [t2] "v"(tc[10]), [t3] "v"(tc[11]), [t4] "v"(tc[12]), [t5] "v"(tc[13]), [t6] "v"(tc[14]), [t7] "v"(tc[15])
      : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
  asm volatile(
      "v_max3_f32 %[b], %[v0], %[v1], %[v2]\n"
      "v_max3_f32 %[b], %[b], %[v3], %[v4]\n"
      "v_max3_f32 %[b], %[b], %[v5], %[v6]\n"
      "v_max3_f32 %[b], %[b], %[v7], %[v8]\n"
      "v_max3_f32 %[b], %[b], %[v9], %[v10]\n"
      "v_max3_f32 %[b], %[b], %[v11], %[v12]\n"
      "v_max3_f32 %[b], %[b], %[v13], %[v14]\n"
      "v_max3_f32 %[b], %[b], %[v15], %[v15]\n"
      "v_mov_b32 %[i], 15\n"
      "v_cmp_eq_f32 s[40:41], %[v14], %[b]\n"
      "v_cmp_eq_f32 s[42:43], %[v13], %[b]\n"
      "v_cmp_eq_f32 s[44:45], %[v12], %[b]\n"
      "v_cmp_eq_f32 s[46:47], %[v11], %[b]\n"
      "v_cmp_eq_f32 s[48:49], %[v10], %[b]\n"
      "v_cmp_eq_f32 s[50:51], %[v9], %[b]\n"
      "v_cmp_eq_f32 s[52:53], %[v8], %[b]\n"
      "v_cmp_eq_f32 s[54:55], %[v7], %[b]\n"
      "v_cmp_eq_f32 s[56:57], %[v6], %[b]\n"
      "v_cmp_eq_f32 s[58:59], %[v5], %[b]\n"
      "v_cmp_eq_f32 s[60:61], %[v4], %[b]\n"
      "v_cmp_eq_f32 s[62:63], %[v3], %[b]\n"
      "v_cmp_eq_f32 s[64:65], %[v2], %[b]\n"
      "v_cmp_eq_f32 s[66:67], %[v1], %[b]\n"
      "v_cmp_eq_f32 s[68:69], %[v0], %[b]\n"
      "v_cndmask_b32 %[i], %[i], 14, s[40:41]\n"
      "v_cndmask_b32 %[i], %[i], 13, s[42:43]\n"
      "v_cndmask_b32 %[i], %[i], 12, s[44:45]\n"
      "v_cndmask_b32 %[i], %[i], 11, s[46:47]\n"
      "v_cndmask_b32 %[i], %[i], 10, s[48:49]\n"
      "v_cndmask_b32 %[i], %[i], 9, s[50:51]\n"
      "v_cndmask_b32 %[i], %[i], 8, s[52:53]\n"
      "v_cndmask_b32 %[i], %[i], 7, s[54:55]\n"
      "v_cndmask_b32 %[i], %[i], 6, s[56:57]\n"
      "v_cndmask_b32 %[i], %[i], 5, s[58:59]\n"
      "v_cndmask_b32 %[i], %[i], 4, s[60:61]\n"
      "v_cndmask_b32 %[i], %[i], 3, s[62:63]\n"
      "v_cndmask_b32 %[i], %[i], 2, s[64:65]\n"
      "v_cndmask_b32 %[i], %[i], 1, s[66:67]\n"
      "v_cndmask_b32 %[i], %[i], 0, s[68:69]\n"
      : [b] "=&v"(best), [i] "=&v"(bi)
      : [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [v4] "v"(v[4]), [v5] "v"(v[5]), [v6] "v"(v[6]), [v7] "v"(v[7]), [v8] "v"(v[8]), [v9] "v"(v[9]), [v10] "v"(v[10]), [v11] "v"(v[11]), [v12] "v"(v[12]), [v13] "v"(v[13]), [v14] "v"(v[14]), [v15] "v"(v[15])
      : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69");
}

template <int CT>  // C rounded up to a multiple of 4: the candidate loop is unrolled over registers
__global__ __launch_bounds__(64) void crf_viterbi_kernel(const float* __restrict__ em, const uint8_t* __restrict__ mask,
                                                        const float* __restrict__ start, const float* __restrict__ end,
                                                        const float* __restrict__ trans, int32_t* __restrict__ tags_out,
                                                        int32_t* __restrict__ lens_out, int S, int C) {
  extern __shared__ __attribute__((aligned(16))) float crf_lds[];
  const int S8 = (S + 7) >> 3;
  float* e = crf_lds;                                               // [S*C + 16] (+ look-ahead padding)
  uint32_t* bpw = reinterpret_cast<uint32_t*>(e + S * C + 16);       // [S8][16] packed back-pointers
  int* path = reinterpret_cast<int*>(bpw + S8 * 16);                // [S]
  uint8_t* mk = reinterpret_cast<uint8_t*>(path + S);               // [S]
  const int b = blockIdx.x, lane = threadIdx.x;
  for (int i = lane; i < S * C; i += 64) e[i] = em[(long)b * S * C + i];
  if (lane < 16) e[S * C + lane] = 0.f;
  for (int i = lane; i < S; i += 64) mk[i] = mask[(long)b * S + i];
  __syncthreads();
  const int j = lane & 15;
  const bool act = j < C;
  const int jc = act ? j : 0;  // (lanes beyond C shadow tag 0: never read by a broadcast, excluded from the final argmax)
  float tcol[CT];
#pragma unroll
  for (int i = 0; i < CT; ++i) tcol[i] = (i < C) ? trans[i * C + jc] : NEG;  // NEG: a padding candidate never wins
  float score = start[jc] + e[jc];
  int cnt = 0;
  for (int t = lane; t < S; t += 64) cnt += mk[t] ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  const float* er = e + jc;
  float en = er[C];
#define CRF_VIT_STEP(t_, on_, q_)                                                                                      \
  {                                                                                                                    \
    const int t_s = (t_);                                                                                              \
    const float et = en;                                                                                               \
    en = er[(t_s + 1) * C];                                                                                            \
    float best;                                                                                                        \
    int bi;                                                                                                            \
    crf_vit_step<CT>(score, tcol, best, bi);                                                                           \
    wb |= (uint32_t)bi << (4 * (q_));                                                                                  \
    if (on_) score = best + et; /* (wave-uniform, scalar) */                                                           \
  }
  {
    int t = 1;
    unsigned long long mb = mask_word(mk, 0, S, lane);
    uint32_t wb = 0;
    for (; t < S && (t & 7); ++t) CRF_VIT_STEP(t, (mb >> t) & 1, t & 7)
    bpw[j] = wb;
    for (; t + 8 <= S; t += 8) {
      if ((t & 63) == 0) mb = mask_word(mk, t, S, lane);
      const unsigned m8 = (unsigned)(mb >> (t & 63)) & 0xffu;
      wb = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) CRF_VIT_STEP(t + q, (m8 >> q) & 1, q)
      bpw[(t >> 3) * 16 + j] = wb;
    }
    if (t < S) {
      wb = 0;
      const int t8 = t;
      for (; t < S; ++t) {
        if ((t & 63) == 0) mb = mask_word(mk, t, S, lane);
        CRF_VIT_STEP(t, (mb >> (t & 63)) & 1, t & 7)
      }
      bpw[(t8 >> 3) * 16 + j] = wb;
    }
  }
#undef CRF_VIT_STEP
  float fin = act ? score + end[jc] : NEG;
  int idx = act ? j : CMAX;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {  // (the four rows hold copies: reduce within a row)
    const float ov = __shfl_xor(fin, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ov > fin || (ov == fin && oi < idx)) { fin = ov; idx = oi; }
  }
  __syncthreads();
  // history[t-1] corresponds to step t; walk back over steps cnt-1 .. 1 (pytorch-crf: history[:seq_end])
  int cur = __builtin_amdgcn_readfirstlane(idx);
  path[cnt - 1] = cur;
  uint32_t word = bpw[((cnt - 1) >> 3) * 16 + j];
  for (int t = cnt - 1; t >= 1; --t) {
    if ((t & 7) == 7) word = bpw[(t >> 3) * 16 + j];
    cur = (__builtin_amdgcn_readlane((int)word, cur) >> (4 * (t & 7))) & 15;
    path[t - 1] = cur;
  }
  __syncthreads();
  int32_t* out = tags_out + (long)b * S;
  for (int t = lane; t < S; t += 64) out[t] = t < cnt ? path[t] : -1;
  if (lane == 0) lens_out[b] = cnt;
}

}  // namespace mtvaf

using namespace mtvaf;

namespace {
constexpr size_t CRF_LDS_MAX = 160 * 1024;
template <typename K>
int crf_allow_lds(K kernel, size_t bytes) {  // dynamic LDS beyond the 64 KB default needs the opt-in, once per kernel
  if (bytes > CRF_LDS_MAX) return MTVAF_ERR_SHAPE;
  if (bytes > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CRF_LDS_MAX) != hipSuccess)
    return MTVAF_ERR_SHAPE;
  return MTVAF_OK;
}
}  // namespace

extern "C" {

// alpha [B,S,16] | logZ [B] | llh [B] | parameter-gradient partials [B, 2C + C*C]
size_t mtvaf_crf_workspace_bytes(int B, int S, int C) {
  return ((size_t)B * S * CMAX + (size_t)B * 2 + (size_t)B * (2 * C + C * C)) * sizeof(float);
}

// loss[0] = -mean_b llh.  workspace keeps alpha/logZ/llh for mtvaf_crf_nll_bwd (same pointer).
int mtvaf_crf_nll_fwd(const float* emissions, const int64_t* tags, const uint8_t* mask, const float* start,
                      const float* end, const float* trans, float* loss, int B, int S, int C, void* workspace,
                      size_t workspace_bytes, hipStream_t st) {
  if (B <= 0 || S <= 0 || C <= 0 || C > CMAX) return MTVAF_ERR_SHAPE;
  if (workspace_bytes < mtvaf_crf_workspace_bytes(B, S, C)) return MTVAF_ERR_WORKSPACE;
  float* alpha = (float*)workspace;
  float* logz = alpha + (size_t)B * S * CMAX;
  float* llh = logz + B;
  const size_t S16 = ((size_t)S + 15) & ~(size_t)15;
  const size_t lds_f = (2 * (S16 + 1) * 16 + (size_t)S) * sizeof(float) + (size_t)S * sizeof(int) + (size_t)S;
  if (int rc = crf_allow_lds(crf_fwd_kernel, lds_f)) return rc;  // S <= ~1100
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
  float* logz = alpha + (size_t)B * S * CMAX;
  float* partial = logz + 2 * B;
  const size_t S16 = ((size_t)S + 15) & ~(size_t)15;
  const size_t lds_b = (3 * S16 * 16 + CMAX * CMAX) * sizeof(float) + (size_t)S * sizeof(int) + (size_t)S;
  if (int rc = crf_allow_lds(crf_bwd_kernel, lds_b)) return rc;  // S <= ~800
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
  const size_t S8 = ((size_t)S + 7) >> 3;
  const size_t lds_v = ((size_t)S * C + 16) * sizeof(float) + S8 * 16 * sizeof(uint32_t) + (size_t)S * sizeof(int) + (size_t)S;
#define CRF_VIT_LAUNCH(CT)                                                                                            \
  {                                                                                                                    \
    if (int rc = crf_allow_lds(crf_viterbi_kernel<CT>, lds_v)) return rc;                                              \
    hipLaunchKernelGGL(crf_viterbi_kernel<CT>, dim3(B), dim3(64), lds_v, st, emissions, mask, start, end, trans,        \
                       tags_out, lens_out, S, C);                                                                       \
  }
  if (C <= 4) CRF_VIT_LAUNCH(4)
  else if (C <= 8) CRF_VIT_LAUNCH(8)
  else if (C <= 12) CRF_VIT_LAUNCH(12)
  else CRF_VIT_LAUNCH(16)
#undef CRF_VIT_LAUNCH
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

}  // extern "C"
