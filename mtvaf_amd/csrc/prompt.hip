// Visual prompt generator kernels (TVNetSAModel2.get_visual_prompt, models/bert_model.py:534-588):
//   K9   split-mean -> 12 x gate softmax(leaky_relu(Linear(6144->4))) -> gated sum of the 4 splits,
//        written straight into the per-layer prefix slabs the attention kernel reads
//        (bert_model.py:566-585).  The reference recomputes the (layer-invariant) split mean 12 times
//        and launches ~150 tiny ops; here it is one mean kernel, one [N,48] GEMM and one mix kernel.
//   K10  VAO loss: KLDivLoss(batchmean)(log softmax(z), target) (bert_model.py:553-554), fused
//        log-softmax + KL forward and its gradient.
// Layouts: enc  [NI, B, L=4, 4*W]   encoder_conv output, NI = 1 + n_aux images, W = 2*hidden (1536)
//          sm   [NI*B, L*W]          mean over the 4 splits (the `.view(bsz,-1)` of :567)
//          gate [NI*B, NL*4]         softmax(leaky_relu(logits)) per layer
//          pkv  [NL, 2, B, P*hid]    P = NI*L slots; slot = img*L + l  (torch.cat(dim=1) of :583)
#include "common.h"

namespace mtvaf {

// sm[n][l*W + c] = 1/4 sum_k enc[n][l][k*W + c]
__global__ void split_mean_kernel(const float* __restrict__ enc, float* __restrict__ sm, long rows, int W) {
  const long n4 = rows * (W / 4);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long r = i / (W / 4);
    const int c = (int)(i % (W / 4)) * 4;
    const float* e = enc + r * 4 * W + c;
    const f32x4 v = (*reinterpret_cast<const f32x4*>(e) + *reinterpret_cast<const f32x4*>(e + W) +
                     *reinterpret_cast<const f32x4*>(e + 2 * W) + *reinterpret_cast<const f32x4*>(e + 3 * W)) * 0.25f;
    *reinterpret_cast<f32x4*>(sm + r * W + c) = v;
  }
}

// gate = softmax4(leaky_relu(logit, 0.01)), in groups of 4
__global__ void gate_fwd_kernel(const float* __restrict__ logits, float* __restrict__ gate, long ngroups) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ngroups) return;
  f32x4 z = *reinterpret_cast<const f32x4*>(logits + i * 4);
#pragma unroll
  for (int k = 0; k < 4; ++k) z[k] = z[k] > 0.f ? z[k] : 0.01f * z[k];
  const float m = fmaxf(fmaxf(z.x, z.y), fmaxf(z.z, z.w));
  f32x4 e = {__expf(z.x - m), __expf(z.y - m), __expf(z.z - m), __expf(z.w - m)};
  const float inv = 1.f / (e.x + e.y + e.z + e.w);
  *reinterpret_cast<f32x4*>(gate + i * 4) = e * inv;
}

// dlogit from dgate partials [n][L][NL*4] (summed over l here)
__global__ void gate_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ gate,
                                const float* __restrict__ dgate_part, float* __restrict__ dlogits, long n, int NL, int L) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * NL) return;
  const long r = i / NL;
  const int idx = (int)(i % NL);
  f32x4 dg = {0.f, 0.f, 0.f, 0.f};
  for (int l = 0; l < L; ++l) dg += *reinterpret_cast<const f32x4*>(dgate_part + ((r * L + l) * NL + idx) * 4);
  const f32x4 gt = *reinterpret_cast<const f32x4*>(gate + i * 4);
  const f32x4 z = *reinterpret_cast<const f32x4*>(logits + i * 4);
  const float dot = gt.x * dg.x + gt.y * dg.y + gt.z * dg.z + gt.w * dg.w;
  f32x4 dz = gt * (dg - dot);
#pragma unroll
  for (int k = 0; k < 4; ++k) dz[k] *= z[k] > 0.f ? 1.f : 0.01f;
  *reinterpret_cast<f32x4*>(dlogits + i * 4) = dz;
}

// one block per (img, b, l); thread t handles columns 4t..4t+3 of the W-wide split
// (launch bound stated: with the default 1024-thread assumption the unrolled layer loop spilled 144 VGPRs to scratch)
template <int NL>
__global__ __launch_bounds__(384) void prompt_mix_fwd_kernel(const float* __restrict__ enc, const float* __restrict__ gate,
                                      float* __restrict__ pkv, int NI, int B, int L, int W) {
  __shared__ float gs[NL * 4];
  const int blk = blockIdx.x;
  const int l = blk % L, b = (blk / L) % B, img = blk / (L * B);
  const long n = (long)img * B + b;
  for (int i = threadIdx.x; i < NL * 4; i += blockDim.x) gs[i] = gate[n * NL * 4 + i];
  __syncthreads();
  const int hid = W / 2, P = NI * L, slot = img * L + l;
  const float* e = enc + (n * L + l) * 4 * W;
  for (int c = threadIdx.x * 4; c < W; c += blockDim.x * 4) {
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(e + c), s1 = *reinterpret_cast<const f32x4*>(e + W + c),
                s2 = *reinterpret_cast<const f32x4*>(e + 2 * W + c), s3 = *reinterpret_cast<const f32x4*>(e + 3 * W + c);
    const int kv = c >= hid ? 1 : 0, cc = c - kv * hid;
    float* dst = pkv + ((long)kv * B + b) * P * hid + (long)slot * hid + cc;
    const long lstride = 2L * B * P * hid;  // one layer of [2, B, P*hid]
#pragma unroll
    for (int idx = 0; idx < NL; ++idx) {
      const f32x4 v = s0 * gs[idx * 4] + s1 * gs[idx * 4 + 1] + s2 * gs[idx * 4 + 2] + s3 * gs[idx * 4 + 3];
      *reinterpret_cast<f32x4*>(dst + idx * lstride) = v;
    }
  }
}

// Pass A of the backward: dgate_part[(n*L+l)][idx*4+k] = sum_c enc[n][l][k*W+c] * dpkv[idx][..][c]
template <int NL>
__global__ __launch_bounds__(256) void prompt_mix_bwd_gate_kernel(const float* __restrict__ enc,
                                                                 const float* __restrict__ dpkv,
                                                                 float* __restrict__ dgate_part, int NI, int B, int L,
                                                                 int W) {
  __shared__ float red[4][NL * 4];
  const int blk = blockIdx.x;
  const int l = blk % L, b = (blk / L) % B, img = blk / (L * B);
  const long n = (long)img * B + b;
  const int hid = W / 2, P = NI * L, slot = img * L + l;
  const float* e = enc + (n * L + l) * 4 * W;
  float acc[NL][4];
#pragma unroll
  for (int i = 0; i < NL; ++i) acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.f;
  for (int c = threadIdx.x * 4; c < W; c += blockDim.x * 4) {
    f32x4 s[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = *reinterpret_cast<const f32x4*>(e + k * W + c);
    const int kv = c >= hid ? 1 : 0, cc = c - kv * hid;
#pragma unroll
    for (int idx = 0; idx < NL; ++idx) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(dpkv + (((long)idx * 2 + kv) * B + b) * P * hid + (long)slot * hid + cc);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[idx][k] += s[k].x * d.x + s[k].y * d.y + s[k].z * d.z + s[k].w * d.w;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int idx = 0; idx < NL; ++idx)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float v = wave_sum(acc[idx][k]);
      if (lane == 0) red[wave][idx * 4 + k] = v;
    }
  __syncthreads();
  if (threadIdx.x < NL * 4)
    dgate_part[(long)blk * NL * 4 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// Pass B: denc[n][l][k*W+c] = sum_idx gate[n][idx][k] * dpkv[idx][..][c] + dsm[n][l*W+c]/4
template <int NL>
__global__ __launch_bounds__(384) void prompt_mix_bwd_enc_kernel(const float* __restrict__ gate, const float* __restrict__ dpkv,
                                          const float* __restrict__ dsm, float* __restrict__ denc, int NI, int B, int L,
                                          int W) {
  __shared__ float gs[NL * 4];
  const int blk = blockIdx.x;
  const int l = blk % L, b = (blk / L) % B, img = blk / (L * B);
  const long n = (long)img * B + b;
  for (int i = threadIdx.x; i < NL * 4; i += blockDim.x) gs[i] = gate[n * NL * 4 + i];
  __syncthreads();
  const int hid = W / 2, P = NI * L, slot = img * L + l;
  float* de = denc + (n * L + l) * 4 * W;
  for (int c = threadIdx.x * 4; c < W; c += blockDim.x * 4) {
    const int kv = c >= hid ? 1 : 0, cc = c - kv * hid;
    f32x4 o[4];
    const f32x4 m = *reinterpret_cast<const f32x4*>(dsm + n * L * W + (long)l * W + c) * 0.25f;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = m;
    const float* src = dpkv + ((long)kv * B + b) * P * hid + (long)slot * hid + cc;
    const long lstride = 2L * B * P * hid;
#pragma unroll
    for (int idx = 0; idx < NL; ++idx) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(src + idx * lstride);
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] += d * gs[idx * 4 + k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(de + k * W + c) = o[k];
  }
}

// ---- K10: fused log-softmax + KL(batchmean) -------------------------------------------------
// row_loss[r] = sum_k t_k (log t_k - logp_k) (0 where t_k == 0);  dlogits = g/B (softmax * sum(t) - t)
__global__ __launch_bounds__(256) void kl_logsoftmax_kernel(const float* __restrict__ logits,
                                                           const float* __restrict__ target, float* __restrict__ row_loss,
                                                           float* __restrict__ dlogits, const float* __restrict__ gout,
                                                           float gscale, int B, int N) {
  __shared__ float red[4];
  const int r = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* z = logits + (long)r * N;
  const float* t = target + (long)r * N;
  auto block_reduce = [&](float v, bool is_max) {
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : red[0] + red[1] + red[2] + red[3];
  };
  float m = -1.0e30f;
  for (int i = threadIdx.x; i < N; i += 256) m = fmaxf(m, z[i]);
  m = block_reduce(m, true);
  float s = 0.f, ts = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) { s += __expf(z[i] - m); ts += t[i]; }
  s = block_reduce(s, false);
  ts = block_reduce(ts, false);
  const float lz = m + __logf(s);
  if (row_loss) {
    float l = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) {
      const float tv = t[i];
      if (tv > 0.f) l += tv * (__logf(tv) - (z[i] - lz));
    }
    l = block_reduce(l, false);
    if (threadIdx.x == 0) row_loss[r] = l;
  }
  if (dlogits) {
    const float g = (gout ? *gout : 1.f) * gscale / B;
    for (int i = threadIdx.x; i < N; i += 256) dlogits[(long)r * N + i] = g * (__expf(z[i] - lz) * ts - t[i]);
  }
}

__global__ void mean_rows_kernel(const float* __restrict__ row_loss, float* __restrict__ loss, int B) {
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 64) s += row_loss[b];
  s = wave_sum(s);
  if (threadIdx.x == 0) *loss = s / B;
}

// out[n][c] = mean_l enc[n][l][c]   (prefix_guids.mean(dim=1), bert_model.py:550)  / and its backward
__global__ void mean_l_kernel(const float* __restrict__ enc, float* __restrict__ out, long n, int L, int W4) {
  const long tot = n * (W4 / 4);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
    const long r = i / (W4 / 4);
    const int c = (int)(i % (W4 / 4)) * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < L; ++l) a += *reinterpret_cast<const f32x4*>(enc + (r * L + l) * W4 + c);
    *reinterpret_cast<f32x4*>(out + r * W4 + c) = a * (1.f / L);
  }
}
// denc[n][l][c] += dmean[n][c] / L
__global__ void mean_l_bwd_kernel(const float* __restrict__ dmean, float* __restrict__ denc, long n, int L, int W4) {
  const long tot = n * L * (W4 / 4);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
    const long rl = i / (W4 / 4);
    const int c = (int)(i % (W4 / 4)) * 4;
    const long r = rl / L;
    float* d = denc + rl * W4 + c;
    *reinterpret_cast<f32x4*>(d) = *reinterpret_cast<const f32x4*>(d) +
                                   *reinterpret_cast<const f32x4*>(dmean + r * W4 + c) * (1.f / L);
  }
}

}  // namespace mtvaf

using namespace mtvaf;

// the mixing kernels unroll over the encoder depth: 12 (base), 24 (large) and the small depths the tests use
#define MTVAF_NL_DISPATCH(NLV, LAUNCH)            \
  switch (NLV) {                                  \
    case 2: LAUNCH(2); break;                     \
    case 3: LAUNCH(3); break;                     \
    case 4: LAUNCH(4); break;                     \
    case 6: LAUNCH(6); break;                     \
    case 8: LAUNCH(8); break;                     \
    case 12: LAUNCH(12); break;                   \
    case 24: LAUNCH(24); break;                   \
    default: return MTVAF_ERR_SHAPE;              \
  }

extern "C" {

int mtvaf_split_mean(const float* enc, float* sm, long rows, int W, hipStream_t st) {
  if (rows <= 0 || W % 4) return MTVAF_ERR_SHAPE;
  const long n4 = rows * (W / 4);
  hipLaunchKernelGGL(split_mean_kernel, dim3((unsigned)std::min<long>((n4 + 255) / 256, 2048)), dim3(256), 0, st, enc, sm,
                     rows, W);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

int mtvaf_gate_fwd(const float* logits, float* gate, long ngroups, hipStream_t st) {
  hipLaunchKernelGGL(gate_fwd_kernel, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, st, logits, gate, ngroups);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

// enc [NI,B,L,4W] + gate [NI*B, NL*4] -> pkv [NL,2,B,(NI*L)*(W/2)].  NL in {2,3,4,6,8,12,24}.
int mtvaf_prompt_mix_fwd(const float* enc, const float* gate, float* pkv, int NI, int B, int L, int W, int NL,
                         hipStream_t st) {
  if (W % 8 || NI <= 0 || B <= 0 || L <= 0) return MTVAF_ERR_SHAPE;
  dim3 grid(NI * B * L), block(W / 4 <= 384 && (W / 4) % 64 == 0 ? W / 4 : 256);
#define MTVAF_L(N_) hipLaunchKernelGGL((prompt_mix_fwd_kernel<N_>), grid, block, 0, st, enc, gate, pkv, NI, B, L, W)
  MTVAF_NL_DISPATCH(NL, MTVAF_L)
#undef MTVAF_L
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

// pass A: dgate_part [NI*B*L, NL*4]; then dlogits [NI*B, NL*4]
int mtvaf_prompt_mix_bwd_gate(const float* enc, const float* dpkv, const float* logits, const float* gate,
                              float* dgate_part, float* dlogits, int NI, int B, int L, int W, int NL, hipStream_t st) {
  if (W % 8 || NI <= 0 || B <= 0 || L <= 0) return MTVAF_ERR_SHAPE;
  dim3 grid(NI * B * L), block(256);
#define MTVAF_L(N_) hipLaunchKernelGGL((prompt_mix_bwd_gate_kernel<N_>), grid, block, 0, st, enc, dpkv, dgate_part, NI, B, L, W)
  MTVAF_NL_DISPATCH(NL, MTVAF_L)
#undef MTVAF_L
  const long n = (long)NI * B;
  hipLaunchKernelGGL(gate_bwd_kernel, dim3((unsigned)((n * NL + 255) / 256)), dim3(256), 0, st, logits, gate, dgate_part,
                     dlogits, n, NL, L);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

// pass B: denc [NI,B,L,4W] (overwritten) from gate, dpkv and dsm [NI*B, L*W]
int mtvaf_prompt_mix_bwd_enc(const float* gate, const float* dpkv, const float* dsm, float* denc, int NI, int B, int L,
                             int W, int NL, hipStream_t st) {
  if (W % 8 || NI <= 0 || B <= 0 || L <= 0) return MTVAF_ERR_SHAPE;
  dim3 grid(NI * B * L), block(W / 4 <= 384 && (W / 4) % 64 == 0 ? W / 4 : 256);
#define MTVAF_L(N_) hipLaunchKernelGGL((prompt_mix_bwd_enc_kernel<N_>), grid, block, 0, st, gate, dpkv, dsm, denc, NI, B, L, W)
  MTVAF_NL_DISPATCH(NL, MTVAF_L)
#undef MTVAF_L
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

// loss[0] = KLDiv(batchmean)(log_softmax(logits), target); row_ws: B floats scratch.
int mtvaf_kl_logsoftmax_fwd(const float* logits, const float* target, float* loss, float* row_ws, int B, int N,
                            hipStream_t st) {
  if (B <= 0 || N <= 0) return MTVAF_ERR_SHAPE;
  hipLaunchKernelGGL(kl_logsoftmax_kernel, dim3(B), dim3(256), 0, st, logits, target, row_ws, (float*)nullptr,
                     (const float*)nullptr, 1.f, B, N);
  hipLaunchKernelGGL(mean_rows_kernel, dim3(1), dim3(64), 0, st, row_ws, loss, B);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

// dlogits = (*grad_out) * gscale / B * (softmax(logits) * sum(target) - target)
int mtvaf_kl_logsoftmax_bwd(const float* grad_out, float gscale, const float* logits, const float* target,
                            float* dlogits, int B, int N, hipStream_t st) {
  if (B <= 0 || N <= 0) return MTVAF_ERR_SHAPE;
  hipLaunchKernelGGL(kl_logsoftmax_kernel, dim3(B), dim3(256), 0, st, logits, target, (float*)nullptr, dlogits, grad_out,
                     gscale, B, N);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

int mtvaf_mean_l_fwd(const float* enc, float* out, long n, int L, int W4, hipStream_t st) {
  if (n <= 0 || W4 % 4) return MTVAF_ERR_SHAPE;
  hipLaunchKernelGGL(mean_l_kernel, dim3((unsigned)std::min<long>((n * (W4 / 4) + 255) / 256, 2048)), dim3(256), 0, st, enc,
                     out, n, L, W4);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

int mtvaf_mean_l_bwd(const float* dmean, float* denc, long n, int L, int W4, hipStream_t st) {
  if (n <= 0 || W4 % 4) return MTVAF_ERR_SHAPE;
  hipLaunchKernelGGL(mean_l_bwd_kernel, dim3((unsigned)std::min<long>((n * L * (W4 / 4) + 255) / 256, 2048)), dim3(256), 0,
                     st, dmean, denc, n, L, W4);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

}  // extern "C"
