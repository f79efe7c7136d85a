// Fused prefix self-attention for gfx950 (fp32, v_mfma_f32_16x16x4_f32):
//   K3/K4  softmax(Q.[Kp;K]^T / sqrt(D) + mask) . [Vp;V]  with the visual prefix K/V slab in FRONT of the
//          text keys, attention-prob dropout and head merge -- models/modeling_bert.py:282-286, 303,
//          320-337 (identical in models/modeling_roberta.py:218-222).
// The [B,NH,S,T] score/probability tensors of the reference are never materialised: the forward keeps
// a flash-style running (max, sum) per query and saves only the log-sum-exp; the backward recomputes
// the probabilities from it.
//
// Data layout (all fp32):
//   qkv   [B*S, 3H]  token-major output of the fused QKV projection (Q | K | V column blocks)
//   pk,pv [B, P*H]   one layer's prefix slab; head h, slot p, dim d at h*(P*64) + p*64 + d  -- the raw
//                    reshape(bsz, 12, -1, 64) of models/bert_model.py:585
//   addmask [B, T]   additive mask, T = P + S: (1 - mask) * -10000  (models/modeling_bert.py:1134-1137)
//   ctx   [B*S, H]   merged heads (modeling_bert.py:335-337)
//
// MFMA mapping: scores are produced TRANSPOSED (S^T[key][q] = K.Q^T) so that a query lives on a lane:
// softmax statistics are lane-local plus two shuffles, and the probability registers are directly
// the B operand of the P.V product (O^T[d][q] = V^T.P^T) -- no LDS round trip for P.  The k index of
// every product is permuted (step t, lane group g <-> k = 4g + t) identically on both operands.
#include "common.h"

namespace mtvaf {

constexpr int D = 64;      // head dim (asserted by the launcher)
constexpr int LDT = 68;    // LDS row stride (floats) for 64-wide tiles: conflict-free b32 column reads
constexpr int KT = 64;     // keys (or queries) per LDS tile
constexpr float NEG_BIG = -1.0e30f;

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

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
};

// Stage a [64][64] tile of K (which = 1) or V (which = 2) rows t0..t0+63 of the concatenated
// [prefix ; text] key axis into LDS (row stride LDT); rows >= T are zero-filled.
__device__ __forceinline__ void stage_kv(float* dst, const AttnArgs& a, int b, int h, int t0, int which) {
  const int T = a.P + a.S;
  const float* pre = which == 1 ? a.pk : a.pv;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + 256 * i;
    const int r = idx >> 4, c = (idx & 15) * 4;
    const int t = t0 + r;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (t < a.P) {
      v = *reinterpret_cast<const f32x4*>(pre + ((long)b * a.P * a.NH + (long)h * a.P + t) * D + c);
    } else if (t < T) {
      v = *reinterpret_cast<const f32x4*>(a.qkv + ((long)b * a.S + (t - a.P)) * 3 * a.H + which * a.H + h * D + c);
    }
    *reinterpret_cast<f32x4*>(dst + r * LDT + c) = v;
  }
}

// Stage a [64][64] tile of rows q0..q0+63 of a token-major [B*S, ld] matrix (head column block h).
__device__ __forceinline__ void stage_rows(float* dst, const float* src, int ld, int col0, int b, int S, int q0) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + 256 * i;
    const int r = idx >> 4, c = (idx & 15) * 4;
    const int q = q0 + r;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (q < S) v = *reinterpret_cast<const f32x4*>(src + ((long)b * S + q) * ld + col0 + c);
    *reinterpret_cast<f32x4*>(dst + r * LDT + c) = v;
  }
}

// ---------------------------------------------------------------------------------------------
// forward: grid (ceil(S/64), NH, B), 256 threads; wave w owns queries q0+16w .. +15
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float Ks[KT * LDT];
  __shared__ __attribute__((aligned(16))) float Vs[KT * LDT];
  __shared__ __attribute__((aligned(16))) float Ms[KT];  // additive mask of the tile's keys (NEG_BIG beyond T)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lq = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q = blockIdx.x * 64 + wave * 16 + lq;
  const int T = a.P + a.S;
  const bool qok = q < a.S;
  const float inv_keep = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  const uint32_t drow = (uint32_t)((b * a.NH + h) * a.S + q);

  f32x4 qreg[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    qreg[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (qok) qreg[db] = *reinterpret_cast<const f32x4*>(a.qkv + ((long)b * a.S + q) * 3 * a.H + h * D + 16 * db + 4 * g);
  }
  f32x4 oacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = NEG_BIG, l_run = 0.f;

  for (int t0 = 0; t0 < T; t0 += KT) {
#if EXP != 3
    __syncthreads();
#endif
#if EXP == 1 || EXP == 2
    if (t0 == 0)
#endif
    {
    stage_kv(Ks, a, b, h, t0, 1);
    stage_kv(Vs, a, b, h, t0, 2);
    }
    if (threadIdx.x < KT) Ms[threadIdx.x] = (t0 + (int)threadIdx.x < T) ? a.addmask[(long)b * T + t0 + threadIdx.x] : NEG_BIG;
#if EXP != 3
    __syncthreads();
#endif
    f32x4 s[4];
    float tmax = NEG_BIG;
    const int nsub = min(4, (T - t0 + 15) >> 4);  // 16-key sub-tiles of this tile that hold real keys
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[j] = f32x4{NEG_BIG, NEG_BIG, NEG_BIG, NEG_BIG};
      if (j >= nsub) continue;
      s[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + (16 * j + lq) * LDT + 16 * db + 4 * g);
#pragma unroll
        for (int t = 0; t < 4; ++t) s[j] = MFMA16(kf[t], qreg[db][t], s[j]);
      }
      const f32x4 mv = *reinterpret_cast<const f32x4*>(Ms + 16 * j + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[j][r] = mv[r] > -1.0e29f ? s[j][r] * a.scale + mv[r] : NEG_BIG;
        tmax = fmaxf(tmax, s[j][r]);
      }
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#if EXP == 2 || EXP == 4
        const float p = s[j][r] - m_new;
#else
        const float p = __expf(s[j][r] - m_new);
#endif
        psum += p;
        float pd = p;
        if (a.p_drop > 0.f)
          pd = attn_dropout_keep(a.drop_key, drow, (uint32_t)(t0 + 16 * j + 4 * g + r), a.drop_thr) ? p * inv_keep : 0.f;
        s[j][r] = pd;
      }
    psum += __shfl_xor(psum, 16, 64);
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] *= alpha;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j >= nsub) continue;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float* vrow = Vs + (16 * j + 4 * g + t) * LDT + lq;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oacc[dt] = MFMA16(vrow[16 * dt], s[j][t], oacc[dt]);
      }
    }
  }
  if (qok) {
    const float inv_l = 1.f / l_run;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *reinterpret_cast<f32x4*>(a.ctx + ((long)b * a.S + q) * a.H + h * D + 16 * dt + 4 * g) = oacc[dt] * inv_l;
    if (g == 0) a.lse[((long)b * a.NH + h) * a.S + q] = m_run + __logf(l_run);
  }
}

// ---------------------------------------------------------------------------------------------
// backward, query side: dQ (and delta = rowsum(dO.O)); same decomposition as the forward.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float Ks[KT * LDT];
  __shared__ __attribute__((aligned(16))) float Vs[KT * LDT];
  __shared__ __attribute__((aligned(16))) float Ms[KT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lq = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q = blockIdx.x * 64 + wave * 16 + lq;
  const int T = a.P + a.S;
  const bool qok = q < a.S;
  const float inv_keep = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  const uint32_t drow = (uint32_t)((b * a.NH + h) * a.S + q);

  f32x4 qreg[4], doreg[4];
  float dl = 0.f;
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    qreg[db] = doreg[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (qok) {
      qreg[db] = *reinterpret_cast<const f32x4*>(a.qkv + ((long)b * a.S + q) * 3 * a.H + h * D + 16 * db + 4 * g);
      doreg[db] = *reinterpret_cast<const f32x4*>(a.dctx + ((long)b * a.S + q) * a.H + h * D + 16 * db + 4 * g);
      const f32x4 o = *reinterpret_cast<const f32x4*>(a.ctx + ((long)b * a.S + q) * a.H + h * D + 16 * db + 4 * g);
      dl += o.x * doreg[db].x + o.y * doreg[db].y + o.z * doreg[db].z + o.w * doreg[db].w;
    }
  }
  dl += __shfl_xor(dl, 16, 64);
  dl += __shfl_xor(dl, 32, 64);
  const float lse = qok ? a.lse[((long)b * a.NH + h) * a.S + q] : 1.0e30f;
  if (qok && g == 0) a.delta[((long)b * a.NH + h) * a.S + q] = dl;

  f32x4 dq[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dq[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int t0 = 0; t0 < T; t0 += KT) {
    __syncthreads();
    stage_kv(Ks, a, b, h, t0, 1);
    stage_kv(Vs, a, b, h, t0, 2);
    if (threadIdx.x < KT) Ms[threadIdx.x] = (t0 + (int)threadIdx.x < T) ? a.addmask[(long)b * T + t0 + threadIdx.x] : NEG_BIG;
    __syncthreads();
    const int nsub = min(4, (T - t0 + 15) >> 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j >= nsub) continue;
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + (16 * j + lq) * LDT + 16 * db + 4 * g);
        const f32x4 vf = *reinterpret_cast<const f32x4*>(Vs + (16 * j + lq) * LDT + 16 * db + 4 * g);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          s = MFMA16(kf[t], qreg[db][t], s);
          dp = MFMA16(vf[t], doreg[db][t], dp);
        }
      }
      const int key0 = t0 + 16 * j + 4 * g;
      const f32x4 mv = *reinterpret_cast<const f32x4*>(Ms + 16 * j + 4 * g);
      f32x4 ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = key0 + r;
        float p = 0.f;
        if (mv[r] > -1.0e29f) p = __expf(s[r] * a.scale + mv[r] - lse);
        float dpe = dp[r];
        if (a.p_drop > 0.f) dpe = attn_dropout_keep(a.drop_key, drow, (uint32_t)key, a.drop_thr) ? dpe * inv_keep : 0.f;
        ds[r] = p * (dpe - dl) * a.scale;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float* krow = Ks + (16 * j + 4 * g + t) * LDT + lq;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = MFMA16(krow[16 * dt], ds[t], dq[dt]);
      }
    }
  }
  if (qok) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *reinterpret_cast<f32x4*>(a.dqkv + ((long)b * a.S + q) * 3 * a.H + h * D + 16 * dt + 4 * g) = dq[dt];
  }
}

// ---------------------------------------------------------------------------------------------
// backward, key side: dK, dV for 64 keys of the [prefix ; text] axis per block (prefix slots write
// dpk/dpv -- the gradient that flows on to the prompt generator); loops over query tiles.
// grid (ceil(T/64), NH, B)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float Qs[KT * LDT];
  __shared__ __attribute__((aligned(16))) float dOs[KT * LDT];
  __shared__ __attribute__((aligned(16))) float lse_s[KT];
  __shared__ __attribute__((aligned(16))) float del_s[KT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lk = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const int T = a.P + a.S;
  const int key = blockIdx.x * 64 + wave * 16 + lk;
  const bool kok = key < T;
  const float inv_keep = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  const float mval = kok ? a.addmask[(long)b * T + key] : 0.f;

  const float* krow = nullptr;
  const float* vrow = nullptr;
  if (kok) {
    if (key < a.P) {
      krow = a.pk + ((long)b * a.P * a.NH + (long)h * a.P + key) * D;
      vrow = a.pv + ((long)b * a.P * a.NH + (long)h * a.P + key) * D;
    } else {
      krow = a.qkv + ((long)b * a.S + (key - a.P)) * 3 * a.H + a.H + h * D;
      vrow = krow + a.H;
    }
  }
  f32x4 kreg[4], vreg[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    kreg[db] = vreg[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (kok) {
      kreg[db] = *reinterpret_cast<const f32x4*>(krow + 16 * db + 4 * g);
      vreg[db] = *reinterpret_cast<const f32x4*>(vrow + 16 * db + 4 * g);
    }
  }
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dk[i] = dv[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int q0 = 0; q0 < a.S; q0 += KT) {
    __syncthreads();
    stage_rows(Qs, a.qkv, 3 * a.H, h * D, b, a.S, q0);
    stage_rows(dOs, a.dctx, a.H, h * D, b, a.S, q0);
    if (threadIdx.x < KT) {
      const int qq = q0 + threadIdx.x;
      lse_s[threadIdx.x] = qq < a.S ? a.lse[((long)b * a.NH + h) * a.S + qq] : 1.0e30f;
      del_s[threadIdx.x] = qq < a.S ? a.delta[((long)b * a.NH + h) * a.S + qq] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const f32x4 qf = *reinterpret_cast<const f32x4*>(Qs + (16 * i + lk) * LDT + 16 * db + 4 * g);
        const f32x4 of = *reinterpret_cast<const f32x4*>(dOs + (16 * i + lk) * LDT + 16 * db + 4 * g);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          s = MFMA16(qf[t], kreg[db][t], s);
          dp = MFMA16(of[t], vreg[db][t], dp);
        }
      }
      const f32x4 lse4 = *reinterpret_cast<const f32x4*>(lse_s + 16 * i + 4 * g);
      const f32x4 del4 = *reinterpret_cast<const f32x4*>(del_s + 16 * i + 4 * g);
      f32x4 pd, ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq = q0 + 16 * i + 4 * g + r;
        float p = kok ? __expf(s[r] * a.scale + mval - lse4[r]) : 0.f;
        float dpe = dp[r];
        float pdr = p;
        if (a.p_drop > 0.f) {
          const bool keep = attn_dropout_keep(a.drop_key, (uint32_t)((b * a.NH + h) * a.S + qq), (uint32_t)key, a.drop_thr);
          pdr = keep ? p * inv_keep : 0.f;
          dpe = keep ? dpe * inv_keep : 0.f;
        }
        pd[r] = pdr;
        ds[r] = p * (dpe - del4[r]) * a.scale;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float* orow = dOs + (16 * i + 4 * g + t) * LDT + lk;
        const float* qrow = Qs + (16 * i + 4 * g + t) * LDT + lk;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dv[dt] = MFMA16(orow[16 * dt], pd[t], dv[dt]);
          dk[dt] = MFMA16(qrow[16 * dt], ds[t], dk[dt]);
        }
      }
    }
  }
  if (kok) {
    float* dkrow;
    float* dvrow;
    if (key < a.P) {
      dkrow = a.dpk + ((long)b * a.P * a.NH + (long)h * a.P + key) * D;
      dvrow = a.dpv + ((long)b * a.P * a.NH + (long)h * a.P + key) * D;
    } else {
      dkrow = a.dqkv + ((long)b * a.S + (key - a.P)) * 3 * a.H + a.H + h * D;
      dvrow = dkrow + a.H;
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      *reinterpret_cast<f32x4*>(dkrow + 16 * dt + 4 * g) = dk[dt];
      *reinterpret_cast<f32x4*>(dvrow + 16 * dt + 4 * g) = dv[dt];
    }
  }
}

static int check(const AttnArgs& a) {
  if (a.B <= 0 || a.S <= 0 || a.P < 0 || a.NH <= 0 || a.H != a.NH * D) return MTVAF_ERR_SHAPE;
  if ((long)a.B * a.NH * a.S >= (1L << 32)) return MTVAF_ERR_SHAPE;
  if (a.p_drop < 0.f || a.p_drop >= 1.f) return MTVAF_ERR_ARG;
  if (a.P > 0 && (!a.pk || !a.pv)) return MTVAF_ERR_ARG;
  return MTVAF_OK;
}

}  // namespace mtvaf

using namespace mtvaf;

extern "C" {

// ctx[B*S,H], lse[B,NH,S] <- attention over [prefix ; text] keys.  head_dim must be 64.
int mtvaf_prefix_attn_fwd(const float* qkv, const float* pk, const float* pv, const float* addmask, float* ctx,
                          float* lse, int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed,
                          uint64_t offset, hipStream_t st) {
  if (head_dim != D) return MTVAF_ERR_SHAPE;
  AttnArgs a{};
  a.qkv = qkv; a.pk = pk; a.pv = pv; a.addmask = addmask; a.ctx = ctx; a.lse = lse;
  a.B = B; a.S = S; a.P = P; a.NH = NH; a.H = NH * D;
  a.scale = 0.125f; a.p_drop = p_drop;
  a.drop_thr = p_drop > 0.f ? (uint32_t)fminf(p_drop * 4294967296.0f, 4294967040.0f) : 0u;
  {
    // host-side replica of attn_dropout_key
    auto mix = [](uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; };
    a.drop_key = mix((uint32_t)seed ^ mix((uint32_t)(seed >> 32) ^ mix((uint32_t)offset ^ 0x9E3779B9u)));
  }
  int rc = check(a);
  if (rc) return rc;
  hipLaunchKernelGGL(attn_fwd_kernel, dim3((S + 63) / 64, NH, B), dim3(256), 0, st, a);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

// dqkv[B*S,3H] (all three column blocks overwritten), dpk/dpv[B,P*H] <- gradients; delta[B,NH,S] scratch.
int mtvaf_prefix_attn_bwd(const float* dctx, const float* qkv, const float* pk, const float* pv,
                          const float* addmask, const float* ctx, const float* lse, float* delta, float* dqkv,
                          float* dpk, float* dpv, int B, int S, int P, int NH, int head_dim, float p_drop,
                          uint64_t seed, uint64_t offset, hipStream_t st) {
  if (head_dim != D) return MTVAF_ERR_SHAPE;
  AttnArgs a{};
  a.qkv = qkv; a.pk = pk; a.pv = pv; a.addmask = addmask; a.ctx = const_cast<float*>(ctx);
  a.lse = const_cast<float*>(lse); a.dctx = dctx; a.delta = delta; a.dqkv = dqkv; a.dpk = dpk; a.dpv = dpv;
  a.B = B; a.S = S; a.P = P; a.NH = NH; a.H = NH * D;
  a.scale = 0.125f; a.p_drop = p_drop;
  a.drop_thr = p_drop > 0.f ? (uint32_t)fminf(p_drop * 4294967296.0f, 4294967040.0f) : 0u;
  {
    auto mix = [](uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; };
    a.drop_key = mix((uint32_t)seed ^ mix((uint32_t)(seed >> 32) ^ mix((uint32_t)offset ^ 0x9E3779B9u)));
  }
  int rc = check(a);
  if (rc) return rc;
  if (P > 0 && (!dpk || !dpv)) return MTVAF_ERR_ARG;
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((S + 63) / 64, NH, B), dim3(256), 0, st, a);
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((P + S + 63) / 64, NH, B), dim3(256), 0, st, a);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

}  // extern "C"
