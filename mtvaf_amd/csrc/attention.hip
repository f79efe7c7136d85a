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
#include "attention_args.h"

#include <cstdlib>

namespace mtvaf {

constexpr int LDT = 68;    // LDS row stride (floats) for 64-wide tiles: conflict-free b32 column reads

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// timing-only ablations of the forward kernel for variant builds (MTVAF_EXTRA_FLAGS=-DMTVAF_ATTN_ABL=n; wrong results; the product
// build has 0): 1 no QK^T products, 2 no PV products, 4 no exponentials / dropout, 8 one barrier per key tile (racy)
#ifndef MTVAF_ATTN_ABL
#define MTVAF_ATTN_ABL 0
#endif

// ---------------------------------------------------------------------------------------------
// forward: grid (ceil(S/64), NH, B), 256 threads; wave w owns queries q0+16w .. +15
// ---------------------------------------------------------------------------------------------
constexpr int LDK = 72;  // K tile row stride: conflict-free ds_read_b128 row fragments (slot = 2*row + k-chunk mod 16)

template <int N>
__device__ __forceinline__ void kv_load(f32x4 (&reg)[N], const KvSrc& s, int P, int T, int ld_txt, int t0) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int r = (threadIdx.x >> 4) + 16 * i;
    reg[i] = *reinterpret_cast<const f32x4*>(kv_row_ptr(s, min(t0 + r, T - 1), P, ld_txt));
  }
}
template <int LD, int N>
__device__ __forceinline__ void kv_store(float* dst, const f32x4 (&reg)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i)
    *reinterpret_cast<f32x4*>(dst + ((threadIdx.x >> 4) + 16 * i) * LD + (threadIdx.x & 15) * 4) = reg[i];
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float Ks[KT * LDK];
  __shared__ __attribute__((aligned(16))) float Vs[KT * LDT];
  __shared__ __attribute__((aligned(16))) float Ms[KT];  // additive mask * log2(e) of the tile's keys (-1e30 beyond T)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lq = lane & 15, g = lane >> 4;
  int bx = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  xcd_group(gridDim.x, gridDim.y, a.B, bx, h, b);
  const int q = bx * 64 + wave * 16 + lq;
  if (a.cu && b == a.B) {  // (block-uniform) the rows that pad the packed image: zeros (0 x NaN of an unwritten row would poison dW)
    const int r0 = a.cu[a.B];
    for (int r = bx * 16 + (threadIdx.x >> 4); r < a.pad_rows; r += gridDim.x * 16)
      store_ctx(a, (long)(r0 + r), h * D + (threadIdx.x & 15) * 4, f32x4{0.f, 0.f, 0.f, 0.f});
    return;
  }
  b = slot_sentence(a, b);
  const Sent sn = sentence(a, b);
  const int Sb = sn.n;
  if (bx * 64 >= Sb) return;  // (block-uniform; packed rows: a query tile beyond the sentence)
  const int Tf = a.P + a.S;  // row length of the additive mask
  __shared__ int t_eff_slot;
  const int T = a.cu ? a.P + Sb : effective_keys(a.addmask + (long)b * Tf, a.P, a.S, &t_eff_slot);  // trailing padding keys are skipped
  const bool qok = q < Sb;
  // Round 6: the last query tile of a sentence is rarely full (74 tokens: queries 64 .. 73 live in wave 0 of the second tile);
  // its other waves used to run every product and exponential of the key loop for rows nobody stores -- on the fp32 matrix pipe
  // they share with the live waves of the other blocks on their SIMD.  They now skip the arithmetic (same results: bit-identical).
  const bool wave_live = __builtin_amdgcn_readfirstlane(q - lq) < Sb;
  const float inv_keep = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  const uint32_t rowh = attn_dropout_rowhash(attn_epoch_key(a.drop_key, a.epoch), (uint32_t)((b * a.NH + h) * a.S + q));
  const float sc2 = a.scale * LOG2E;  // scores are kept in the log2 domain: one v_exp_f32 per probability

  const int c4 = (threadIdx.x & 15) * 4;
  KvSrc ksrc, vsrc;
  ksrc.pre = a.pk + ((long)b * a.P * a.NH + (long)h * a.P) * D + c4;
  vsrc.pre = a.pv + ((long)b * a.P * a.NH + (long)h * a.P) * D + c4;
  ksrc.txt = a.qkv + sn.tok0 * 3 * a.H + a.H + h * D + c4;
  vsrc.txt = ksrc.txt + a.H;
  const int ldt = 3 * a.H;

  f32x4 qreg[4];
  {
    const float* qp = a.qkv + (sn.tok0 + min(q, Sb - 1)) * 3 * a.H + h * D + 4 * g;
#pragma unroll
    for (int db = 0; db < 4; ++db) qreg[db] = *reinterpret_cast<const f32x4*>(qp + 16 * db);
  }
  f32x4 oacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = NEG_BIG, l_run = 0.f;
  const float* kfrag = Ks + lq * LDK + 4 * g;
  const float* vcol = Vs + 4 * g * LDT + lq;

  // The next key tile travels global -> registers while the current one is being multiplied: its loads are issued
  // right behind the barrier that publishes the current tile and are first needed at the top of the next iteration.
  f32x4 kreg[4], vreg[4];
  float mreg = NEG_BIG;
  auto fetch = [&](int t0) {
    kv_load(kreg, ksrc, a.P, T, ldt, t0);
    kv_load(vreg, vsrc, a.P, T, ldt, t0);
    if (threadIdx.x < KT) mreg = mask_at(a, b, Tf, min(t0 + (int)threadIdx.x, T - 1));
  };
  fetch(0);
  for (int t0 = 0; t0 < T; t0 += KT) {
    if constexpr (!(MTVAF_ATTN_ABL & 8)) __syncthreads();
    kv_store<LDK>(Ks, kreg);
    kv_store<LDT>(Vs, vreg);
    if (threadIdx.x < KT) Ms[threadIdx.x] = (t0 + (int)threadIdx.x < T) ? mreg * LOG2E : NEG_BIG;
    __syncthreads();
    if (t0 + KT < T) fetch(t0 + KT);
    if (!wave_live) continue;  // (wave-uniform) a wave whose 16 queries all lie beyond the sentence only stages and synchronises
    const int nsub = min(4, (T - t0 + 15) >> 4);  // 16-key sub-tiles of this tile that hold real keys
    f32x4 s[4];
    float tmax = NEG_BIG;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[j] = f32x4{NEG_BIG, NEG_BIG, NEG_BIG, NEG_BIG};
      if (j < nsub) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const f32x4 kf = *reinterpret_cast<const f32x4*>(kfrag + 16 * j * LDK + 16 * db);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if constexpr (!(MTVAF_ATTN_ABL & 1)) acc = MFMA16(kf[t], qreg[db][t], acc);
            else acc[t] += kf[t] * qreg[db][t];
          }
        }
        const f32x4 mv = *reinterpret_cast<const f32x4*>(Ms + 16 * j + 4 * g);
        s[j] = acc * sc2 + mv;
        tmax = fmaxf(tmax, fmaxf(fmaxf(s[j].x, s[j].y), fmaxf(s[j].z, s[j].w)));
      }
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float psum = 0.f;
    const uint32_t cterm0 = (uint32_t)(t0 + 4 * g) * ATTN_DROP_C2;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = s[j][r] - m_new;
        if constexpr (!(MTVAF_ATTN_ABL & 4)) p = __builtin_amdgcn_exp2f(p);
        psum += p;
        float pd = p;
        if constexpr (!(MTVAF_ATTN_ABL & 4)) {
          if (a.p_drop > 0.f)
            pd = attn_dropout_keep2(rowh, cterm0 + (uint32_t)(16 * j + r) * ATTN_DROP_C2, a.drop_thr) ? p * inv_keep : 0.f;
        }
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
      if (j < nsub) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float* vrow = vcol + (16 * j + t) * LDT;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            if constexpr (!(MTVAF_ATTN_ABL & 2)) oacc[dt] = MFMA16(vrow[16 * dt], s[j][t], oacc[dt]);
            else oacc[dt][t] += vrow[16 * dt] * s[j][t];
          }
        }
      }
    }
  }
  if (qok) {
    const float inv_l = 1.f / l_run;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      store_ctx(a, sn.tok0 + q, h * D + 16 * dt + 4 * g, oacc[dt] * inv_l);
    if (g == 0) a.lse[((long)b * a.NH + h) * a.S + q] = (m_run + log2f(l_run)) * LN2;
  }
}

// ---------------------------------------------------------------------------------------------
// backward, query side: dQ (and delta = rowsum(dO.O)); same decomposition as the forward.
// ---------------------------------------------------------------------------------------------
// Ks [KT*LDT]: read as row fragments AND as columns; Vs [KT*LDK]: row fragments only; Ms [KT]
__device__ __forceinline__ void attn_bwd_dq_body(const AttnArgs& a, int qtile, int b, int h, float* Ks, float* Vs, float* Ms, int* t_eff_slot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lq = lane & 15, g = lane >> 4;
  const int q = qtile * 64 + wave * 16 + lq;
  const Sent sn = sentence(a, b);
  const int Sb = sn.n;
  if (qtile * 64 >= Sb) return;  // (block-uniform)
  const int Tf = a.P + a.S;
  const int T = a.cu ? a.P + Sb : effective_keys(a.addmask + (long)b * Tf, a.P, a.S, t_eff_slot);
  const bool qok = q < Sb;
  const bool wave_live = __builtin_amdgcn_readfirstlane(q - lq) < Sb;
  if (a.zero_tail && !a.cu && qtile * 64 >= T - a.P) {  // (block-uniform) a tile of trailing padding: dQ = 0, nothing to read
    if (qok) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) store_dqkv(a, sn.tok0 + q, h * D + 16 * dt + 4 * g, z);
      if (g == 0) a.delta[((long)b * a.NH + h) * a.S + q] = 0.f;
    }
    return;
  }
  const float inv_keep = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  const uint32_t rowh = attn_dropout_rowhash(attn_epoch_key(a.drop_key, a.epoch), (uint32_t)((b * a.NH + h) * a.S + q));
  const float sc2 = a.scale * LOG2E;

  const int c4 = (threadIdx.x & 15) * 4;
  KvSrc ksrc, vsrc;
  ksrc.pre = a.pk + ((long)b * a.P * a.NH + (long)h * a.P) * D + c4;
  vsrc.pre = a.pv + ((long)b * a.P * a.NH + (long)h * a.P) * D + c4;
  ksrc.txt = a.qkv + sn.tok0 * 3 * a.H + a.H + h * D + c4;
  vsrc.txt = ksrc.txt + a.H;
  const int ldt = 3 * a.H;

  f32x4 qreg[4], doreg[4];
  float dl = 0.f;
  {
    const long qrow = sn.tok0 + min(q, Sb - 1);
    const float* qp = a.qkv + qrow * 3 * a.H + h * D + 4 * g;
    const float* dop = a.dctx + qrow * a.H + h * D + 4 * g;
    const float* op = a.ctx + qrow * a.H + h * D + 4 * g;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      qreg[db] = *reinterpret_cast<const f32x4*>(qp + 16 * db);
      doreg[db] = *reinterpret_cast<const f32x4*>(dop + 16 * db);
      const f32x4 o = *reinterpret_cast<const f32x4*>(op + 16 * db);
      dl += o.x * doreg[db].x + o.y * doreg[db].y + o.z * doreg[db].z + o.w * doreg[db].w;
    }
  }
  dl += __shfl_xor(dl, 16, 64);
  dl += __shfl_xor(dl, 32, 64);
  // rows beyond S: lse = +1e30 makes every probability (and with it ds) exactly 0
  const float lse2 = qok ? a.lse[((long)b * a.NH + h) * a.S + q] * LOG2E : 1.0e30f;
  if (qok && g == 0) a.delta[((long)b * a.NH + h) * a.S + q] = dl;

  f32x4 dq[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dq[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    // A query tile whose upstream gradient is all zeros (padded positions: nothing downstream reads them) has dP = dO.V^T = 0
    // and delta = 0, hence dS = 0 and dQ = 0 EXACTLY: write the zeros and skip the key loop.  Detected, not assumed: one
    // LDS flag (Ms[0] is free until the first key tile is staged), no vote primitive.  (The same test on the key side --
    // skipping all-zero query tiles in the dK / dV loop -- measured 3 % SLOWER on the whole step, with a vote primitive
    // and with plain LDS flags alike: the loop is at its register limit.)
    bool nz = false;
#pragma unroll
    for (int db = 0; db < 4; ++db) nz = nz || doreg[db].x != 0.f || doreg[db].y != 0.f || doreg[db].z != 0.f || doreg[db].w != 0.f;
    int* flag = reinterpret_cast<int*>(Ms);
    if (threadIdx.x == 0) *flag = 0;
    __syncthreads();
    if (qok && nz) *flag = 1;  // (benign race: every writer stores the same value)
    __syncthreads();
    const bool live = *flag != 0;
    __syncthreads();  // (the flag word is the mask tile's first entry: nobody may still read it when staging starts)
    if (!live) {
      if (qok) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          store_dqkv(a, sn.tok0 + q, h * D + 16 * dt + 4 * g, dq[dt]);
      }
      return;
    }
  }
  const float* kfrag = Ks + lq * LDT + 4 * g;
  const float* vfrag = Vs + lq * LDK + 4 * g;
  const float* kcol = Ks + 4 * g * LDT + lq;

  f32x4 kreg[4], vreg[4];  // (the next key tile is fetched while the current one is multiplied, as in the forward)
  float mreg = NEG_BIG;
  auto fetch = [&](int t0) {
    kv_load(kreg, ksrc, a.P, T, ldt, t0);
    kv_load(vreg, vsrc, a.P, T, ldt, t0);
    if (threadIdx.x < KT) mreg = mask_at(a, b, Tf, min(t0 + (int)threadIdx.x, T - 1));
  };
  fetch(0);
  for (int t0 = 0; t0 < T; t0 += KT) {
    __syncthreads();
    kv_store<LDT>(Ks, kreg);
    kv_store<LDK>(Vs, vreg);
    if (threadIdx.x < KT) Ms[threadIdx.x] = (t0 + (int)threadIdx.x < T) ? mreg * LOG2E : NEG_BIG;
    __syncthreads();
    if (t0 + KT < T) fetch(t0 + KT);
    if (!wave_live) continue;  // (wave-uniform; as in the forward kernel)
    const int nsub = min(4, (T - t0 + 15) >> 4);
    const uint32_t cterm0 = (uint32_t)(t0 + 4 * g) * ATTN_DROP_C2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < nsub) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const f32x4 kf = *reinterpret_cast<const f32x4*>(kfrag + 16 * j * LDT + 16 * db);
          const f32x4 vf = *reinterpret_cast<const f32x4*>(vfrag + 16 * j * LDK + 16 * db);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            s = MFMA16(kf[t], qreg[db][t], s);
            dp = MFMA16(vf[t], doreg[db][t], dp);
          }
        }
        const f32x4 mv = *reinterpret_cast<const f32x4*>(Ms + 16 * j + 4 * g);
        f32x4 ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(s[r] * sc2 + mv[r] - lse2);
          float dpe = dp[r];
          if (a.p_drop > 0.f)
            dpe = attn_dropout_keep2(rowh, cterm0 + (uint32_t)(16 * j + r) * ATTN_DROP_C2, a.drop_thr) ? dpe * inv_keep : 0.f;
          ds[r] = p * (dpe - dl) * a.scale;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float* krow = kcol + (16 * j + t) * LDT;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) dq[dt] = MFMA16(krow[16 * dt], ds[t], dq[dt]);
        }
      }
    }
  }
  if (qok) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      store_dqkv(a, sn.tok0 + q, h * D + 16 * dt + 4 * g, dq[dt]);
  }
}

// ---------------------------------------------------------------------------------------------
// backward, key side: dK, dV for 64 keys of the [prefix ; text] axis per block (prefix slots write
// dpk/dpv -- the gradient that flows on to the prompt generator); loops over query tiles.
// grid (ceil(T/64), NH, B)
// ---------------------------------------------------------------------------------------------
// Qs, dOs [KT*LDT]; lse_s [KT] = lse * log2(e) (+1e30 for rows beyond S); del_s [KT] = rowsum(dO.O), computed here
// from the staged dO tile and the matching O rows so that this side does not depend on the query side (both run in
// one launch); rh_s [KT] = dropout row hashes of the tile's queries.
__device__ __forceinline__ void attn_bwd_dkv_body(const AttnArgs& a, int ktile, int b, int h, float* Qs, float* dOs, float* lse_s,
                                                  float* del_s, uint32_t* rh_s, int* t_eff_slot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lk = lane & 15, g = lane >> 4;
  const Sent sn = sentence(a, b);
  const int Sb = sn.n;
  const int T = a.cu ? a.P + Sb : effective_keys(a.addmask + (long)b * (a.P + a.S), a.P, a.S, t_eff_slot);  // keys >= T: trailing padding, dK = dV = 0
  const int Tf = a.cu ? T : a.P + a.S;  // (packed rows: keys beyond the sentence do not exist)
  const int key = ktile * 64 + wave * 16 + lk;
  if (ktile * 64 >= T) {  // (block-uniform) a key tile of trailing padding only: exact zeros, no query loop
    if (key < Tf) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        store_dqkv(a, sn.tok0 + (key - a.P), a.H + h * D + 16 * dt + 4 * g, z);
        store_dqkv(a, sn.tok0 + (key - a.P), 2 * a.H + h * D + 16 * dt + 4 * g, z);
      }
    }
    return;
  }
  const bool kok = key < T;
  const bool wave_live = (int)(ktile * 64 + wave * 16) < T;
  const int keyc = min(key, T - 1);
  const float inv_keep = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  // keys beyond T: mask -1e30 makes their probabilities exactly 0
  const float mval2 = kok ? mask_at(a, b, Tf, key) * LOG2E : NEG_BIG;
  const float sc2 = a.scale * LOG2E;
  const uint32_t cterm = (uint32_t)key * ATTN_DROP_C2;

  KvSrc ksrc, vsrc;
  ksrc.pre = a.pk + ((long)b * a.P * a.NH + (long)h * a.P) * D + 4 * g;
  vsrc.pre = a.pv + ((long)b * a.P * a.NH + (long)h * a.P) * D + 4 * g;
  ksrc.txt = a.qkv + sn.tok0 * 3 * a.H + a.H + h * D + 4 * g;
  vsrc.txt = ksrc.txt + a.H;
  f32x4 kreg[4], vreg[4];
  {
    const float* krow = kv_row_ptr(ksrc, keyc, a.P, 3 * a.H);
    const float* vrow = kv_row_ptr(vsrc, keyc, a.P, 3 * a.H);
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      kreg[db] = *reinterpret_cast<const f32x4*>(krow + 16 * db);
      vreg[db] = *reinterpret_cast<const f32x4*>(vrow + 16 * db);
    }
  }
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dk[i] = dv[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int c4 = (threadIdx.x & 15) * 4, r0 = threadIdx.x >> 4;
  const float* qsrc = a.qkv + sn.tok0 * 3 * a.H + h * D + c4;
  const float* dosrc = a.dctx + sn.tok0 * a.H + h * D + c4;
  const float* osrc = a.ctx + sn.tok0 * a.H + h * D + c4;
  const float* qfrag = Qs + lk * LDT + 4 * g;
  const float* ofrag = dOs + lk * LDT + 4 * g;
  const float* qcol = Qs + 4 * g * LDT + lk;
  const float* ocol = dOs + 4 * g * LDT + lk;
  const uint32_t row_base = (uint32_t)((b * a.NH + h) * a.S);

  // the next query tile (Q, dO, O rows, lse) is fetched while the current one is multiplied
  f32x4 qr[4], orr[4], ofw[4];
  float lreg = 1.0e30f;
  auto fetch = [&](int q0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int qq = min(q0 + r0 + 16 * i, Sb - 1);
      qr[i] = *reinterpret_cast<const f32x4*>(qsrc + (long)qq * 3 * a.H);
      orr[i] = *reinterpret_cast<const f32x4*>(dosrc + (long)qq * a.H);
      ofw[i] = *reinterpret_cast<const f32x4*>(osrc + (long)qq * a.H);
    }
    if (threadIdx.x < KT) lreg = a.lse[((long)b * a.NH + h) * a.S + min(q0 + (int)threadIdx.x, Sb - 1)] * LOG2E;
  };
  // (zero_tail: queries from the last unmasked position on have dO = 0 exactly -- no contribution, see AttnArgs)
  const int Sq = (a.zero_tail && !a.cu) ? min(Sb, T - a.P) : Sb;
  if (Sq > 0) fetch(0);
  for (int q0 = 0; q0 < Sq; q0 += KT) {
    float dsum[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      dsum[i] = ofw[i].x * orr[i].x + ofw[i].y * orr[i].y + ofw[i].z * orr[i].z + ofw[i].w * orr[i].w;
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // the 16 threads of a row are 16 consecutive lanes
      dsum[i] += __shfl_xor(dsum[i], 1, 64);
      dsum[i] += __shfl_xor(dsum[i], 2, 64);
      dsum[i] += __shfl_xor(dsum[i], 4, 64);
      dsum[i] += __shfl_xor(dsum[i], 8, 64);
    }
    const float lcur = lreg;
    __syncthreads();
    kv_store<LDT>(Qs, qr);
    kv_store<LDT>(dOs, orr);
    if ((threadIdx.x & 15) == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) del_s[r0 + 16 * i] = dsum[i];
    }
    if (threadIdx.x < KT) {
      const int qq = q0 + threadIdx.x;
      lse_s[threadIdx.x] = qq < Sb ? lcur : 1.0e30f;
      rh_s[threadIdx.x] = attn_dropout_rowhash(attn_epoch_key(a.drop_key, a.epoch), row_base + (uint32_t)qq);
    }
    __syncthreads();
    if (q0 + KT < Sq) fetch(q0 + KT);
    // a wave whose 16 keys all lie beyond T (last key tile) only takes part in the staging and the barriers
    const int nsub = wave_live ? min(4, (Sq - q0 + 15) >> 4) : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < nsub) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const f32x4 qf = *reinterpret_cast<const f32x4*>(qfrag + 16 * i * LDT + 16 * db);
          const f32x4 of = *reinterpret_cast<const f32x4*>(ofrag + 16 * i * LDT + 16 * db);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            s = MFMA16(qf[t], kreg[db][t], s);
            dp = MFMA16(of[t], vreg[db][t], dp);
          }
        }
        const f32x4 lse4 = *reinterpret_cast<const f32x4*>(lse_s + 16 * i + 4 * g);
        const f32x4 del4 = *reinterpret_cast<const f32x4*>(del_s + 16 * i + 4 * g);
        const uint4 rh4 = *reinterpret_cast<const uint4*>(rh_s + 16 * i + 4 * g);
        const uint32_t rh[4] = {rh4.x, rh4.y, rh4.z, rh4.w};
        f32x4 pd, ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(s[r] * sc2 + mval2 - lse4[r]);
          float dpe = dp[r], pdr = p;
          if (a.p_drop > 0.f) {
            const bool keep = attn_dropout_keep2(rh[r], cterm, a.drop_thr);
            pdr = keep ? p * inv_keep : 0.f;
            dpe = keep ? dpe * inv_keep : 0.f;
          }
          pd[r] = pdr;
          ds[r] = p * (dpe - del4[r]) * a.scale;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float* orow = ocol + (16 * i + t) * LDT;
          const float* qrow = qcol + (16 * i + t) * LDT;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            dv[dt] = MFMA16(orow[16 * dt], pd[t], dv[dt]);
            dk[dt] = MFMA16(qrow[16 * dt], ds[t], dk[dt]);
          }
        }
      }
    }
  }
  if (!kok && key < Tf) {  // trailing padding inside a partially valid tile
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      store_dqkv(a, sn.tok0 + (key - a.P), a.H + h * D + 16 * dt + 4 * g, z);
      store_dqkv(a, sn.tok0 + (key - a.P), 2 * a.H + h * D + 16 * dt + 4 * g, z);
    }
  }
  if (kok) {
    if (key < a.P) {
      float* dkrow = a.dpk + ((long)b * a.P * a.NH + (long)h * a.P + key) * D;
      float* dvrow = a.dpv + ((long)b * a.P * a.NH + (long)h * a.P + key) * D;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        *reinterpret_cast<f32x4*>(dkrow + 16 * dt + 4 * g) = dk[dt];
        *reinterpret_cast<f32x4*>(dvrow + 16 * dt + 4 * g) = dv[dt];
      }
    } else {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        store_dqkv(a, sn.tok0 + (key - a.P), a.H + h * D + 16 * dt + 4 * g, dk[dt]);
        store_dqkv(a, sn.tok0 + (key - a.P), 2 * a.H + h * D + 16 * dt + 4 * g, dv[dt]);
      }
    }
  }
}

// One launch for the whole attention backward: blocks [0, nq) of x are query tiles (dQ), the rest key tiles
// (dK, dV).  The two sides are independent (the key side recomputes delta), so they share the machine -- each is
// VALU / issue bound at ~40 % MFMA utilisation on its own -- and need neither atomics nor a second stream.
__global__ __launch_bounds__(256, 3) void attn_bwd_kernel(AttnArgs a, int nq) {  // 4 per SIMD spills 10 VGPRs

  __shared__ __attribute__((aligned(16))) float tile0[KT * LDK];
  __shared__ __attribute__((aligned(16))) float tile1[KT * LDK];
  __shared__ __attribute__((aligned(16))) float small[3 * KT];
  __shared__ int t_eff_slot;
  if (a.cu && (int)blockIdx.z == a.B) {  // (block-uniform) zero dQ | dK | dV of the rows that pad the packed image
    const int r0 = a.cu[a.B], h = blockIdx.y;
    for (int r = blockIdx.x * 16 + (threadIdx.x >> 4); r < a.pad_rows; r += gridDim.x * 16)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        store_dqkv(a, (long)(r0 + r), c * a.H + h * D + (threadIdx.x & 15) * 4, f32x4{0.f, 0.f, 0.f, 0.f});
    return;
  }
  int bx = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  xcd_group(gridDim.x, gridDim.y, a.B, bx, h, b);
  b = slot_sentence(a, b);
  if (bx < nq) {
    attn_bwd_dq_body(a, bx, b, h, tile0, tile1, small, &t_eff_slot);
  } else {
    attn_bwd_dkv_body(a, bx - nq, b, h, tile0, tile1, small, small + KT, reinterpret_cast<uint32_t*>(small + 2 * KT), &t_eff_slot);
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

int mtvaf_f32_split(int on);  // (gemm.hip)

// Round 6: under the split arithmetic (the library default: mtvaf_f32_split) the attention products are split bf16 products too
// (csrc/attention_f32s.hip: same interface, geometry and outputs); the fp32 MFMA pipe keeps the kernels of this file.
// MTVAF_ATTN_SPLIT=0: the kernels of this file in either arithmetic.
static bool attn_split_on() {
  static const int env = [] { const char* e = getenv("MTVAF_ATTN_SPLIT"); return e ? atoi(e) : 1; }();
  return env != 0 && mtvaf_f32_split(-1) != 0;
}

static void fill_common(AttnArgs& a, int B, int S, int P, int NH, float p_drop, uint64_t seed, uint64_t offset) {
  a.B = B; a.S = S; a.P = P; a.NH = NH; a.H = NH * D;
  a.scale = 0.125f; a.p_drop = p_drop;
  a.drop_thr = p_drop > 0.f ? (uint32_t)fminf(p_drop * 4294967296.0f, 4294967040.0f) : 0u;
  // host-side replica of attn_dropout_key
  auto mix = [](uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; };
  a.drop_key = mix((uint32_t)seed ^ mix((uint32_t)(seed >> 32) ^ mix((uint32_t)offset ^ 0x9E3779B9u)));
  a.epoch = rng_epoch_ptr();
}

static int attn_fwd_launch(const float* qkv, const float* pk, const float* pv, const float* addmask, const int* cu, int pad_rows,
                           float* ctx, float* lse, int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed,
                           uint64_t offset, hipStream_t st, void* ctx_planes = nullptr, long rows = 0) {
  if (head_dim != D) return MTVAF_ERR_SHAPE;
  if (!cu && !addmask) return MTVAF_ERR_ARG;
  AttnArgs a{};
  a.qkv = qkv; a.pk = pk; a.pv = pv; a.addmask = addmask; a.cu = cu; a.pad_rows = pad_rows; a.ctx = ctx; a.lse = lse;
  a.ctx_p = static_cast<unsigned char*>(ctx_planes); a.Mrows = rows;
  if (ctx_planes && (!cu || rows <= 0 || (((uintptr_t)ctx_planes) & 15))) return MTVAF_ERR_ARG;
  fill_common(a, B, S, P, NH, p_drop, seed, offset);
  int rc = check(a);
  if (rc) return rc;
  if (pad_rows < 0 || (pad_rows && !cu)) return MTVAF_ERR_ARG;
  const dim3 grid((S + 63) / 64, NH, B + (pad_rows > 0 ? 1 : 0));
  if (attn_split_on()) return launch_attn_f32s_fwd(a, grid, st);
  hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(256), 0, st, a);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

static int attn_bwd_launch(const float* dctx, const float* qkv, const float* pk, const float* pv, const float* addmask,
                           const int* cu, int pad_rows, const float* ctx, const float* lse, float* delta, float* dqkv, float* dpk, float* dpv,
                           int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed, uint64_t offset, hipStream_t st,
                           int zero_tail = 0, void* dqkv_planes = nullptr, long rows = 0) {
  if (head_dim != D) return MTVAF_ERR_SHAPE;
  if (!cu && !addmask) return MTVAF_ERR_ARG;
  AttnArgs a{};
  a.qkv = qkv; a.pk = pk; a.pv = pv; a.addmask = addmask; a.cu = cu; a.pad_rows = pad_rows; a.ctx = const_cast<float*>(ctx);
  a.lse = const_cast<float*>(lse); a.dctx = dctx; a.delta = delta; a.dqkv = dqkv; a.dpk = dpk; a.dpv = dpv;
  a.zero_tail = zero_tail;
  a.dqkv_p = static_cast<unsigned char*>(dqkv_planes); a.Mrows = rows;
  if (dqkv_planes && (!cu || rows <= 0 || (((uintptr_t)dqkv_planes) & 15))) return MTVAF_ERR_ARG;
  fill_common(a, B, S, P, NH, p_drop, seed, offset);
  int rc = check(a);
  if (rc) return rc;
  if (P > 0 && (!dpk || !dpv)) return MTVAF_ERR_ARG;
  if (pad_rows < 0 || (pad_rows && !cu)) return MTVAF_ERR_ARG;
  const int nq = (S + 63) / 64;
  const dim3 grid(nq + (P + S + 63) / 64, NH, B + (pad_rows > 0 ? 1 : 0));
  if (attn_split_on()) return launch_attn_f32s_bwd(a, nq, grid, st);
  hipLaunchKernelGGL(attn_bwd_kernel, grid, dim3(256), 0, st, a, nq);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

// ctx[B*S,H], lse[B,NH,S] <- attention over [prefix ; text] keys.  head_dim must be 64.
int mtvaf_prefix_attn_fwd(const float* qkv, const float* pk, const float* pv, const float* addmask, float* ctx,
                          float* lse, int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed,
                          uint64_t offset, hipStream_t st) {
  return attn_fwd_launch(qkv, pk, pv, addmask, nullptr, 0, ctx, lse, B, S, P, NH, head_dim, p_drop, seed, offset, st);
}

// dqkv[B*S,3H] (all three column blocks overwritten), dpk/dpv[B,P*H] <- gradients; delta[B,NH,S] scratch.
int mtvaf_prefix_attn_bwd(const float* dctx, const float* qkv, const float* pk, const float* pv,
                          const float* addmask, const float* ctx, const float* lse, float* delta, float* dqkv,
                          float* dpk, float* dpv, int B, int S, int P, int NH, int head_dim, float p_drop,
                          uint64_t seed, uint64_t offset, hipStream_t st) {
  return attn_bwd_launch(dctx, qkv, pk, pv, addmask, nullptr, 0, ctx, lse, delta, dqkv, dpk, dpv, B, S, P, NH, head_dim, p_drop, seed,
                         offset, st);
}

// mtvaf_prefix_attn_bwd for callers that vouch that dctx is exactly zero for the queries behind a sentence's last unmasked text
// position (zero_tail != 0; see AttnArgs): same results bit for bit under that contract, the query loops stop there.
int mtvaf_prefix_attn_bwd_tail(const float* dctx, const float* qkv, const float* pk, const float* pv, const float* addmask,
                               const float* ctx, const float* lse, float* delta, float* dqkv, float* dpk, float* dpv, int B, int S,
                               int P, int NH, int head_dim, float p_drop, uint64_t seed, uint64_t offset, int zero_tail,
                               hipStream_t st) {
  return attn_bwd_launch(dctx, qkv, pk, pv, addmask, nullptr, 0, ctx, lse, delta, dqkv, dpk, dpv, B, S, P, NH, head_dim, p_drop, seed,
                         offset, st, zero_tail);
}

// The same attention over PACKED token rows (padding-free execution): cu [B+1] int32 row offsets -- sentence b owns rows
// cu[b] .. cu[b+1]-1 of qkv / ctx / dctx / dqkv (its unmasked tokens, at most S of them).  Every kept key is unmasked, so
// no additive mask is read; lse / delta stay [B,NH,S].  The pad_rows rows behind the last sentence (they pad the packed
// image to whole tiles) are zero-filled in ctx / dqkv by an extra slice of the same launch.
int mtvaf_prefix_attn_varlen_fwd(const float* qkv, const float* pk, const float* pv, const int* cu, int pad_rows, float* ctx,
                                 float* lse, int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed, uint64_t offset,
                                 hipStream_t st) {
  if (!cu) return MTVAF_ERR_ARG;
  return attn_fwd_launch(qkv, pk, pv, nullptr, cu, pad_rows, ctx, lse, B, S, P, NH, head_dim, p_drop, seed, offset, st);
}

int mtvaf_prefix_attn_varlen_bwd(const float* dctx, const float* qkv, const float* pk, const float* pv, const int* cu, int pad_rows,
                                 const float* ctx, const float* lse, float* delta, float* dqkv, float* dpk, float* dpv, int B, int S,
                                 int P, int NH, int head_dim, float p_drop, uint64_t seed, uint64_t offset, hipStream_t st) {
  if (!cu) return MTVAF_ERR_ARG;
  return attn_bwd_launch(dctx, qkv, pk, pv, nullptr, cu, pad_rows, ctx, lse, delta, dqkv, dpk, dpv, B, S, P, NH, head_dim, p_drop, seed,
                         offset, st);
}

// The packed-row attention that ALSO writes the tile-blocked plane image of its GEMM-operand result (round 5, pre-split operands:
// csrc/gemm_f32p.hip): the context as [H / 32][3][rows][32] -- the operand of the Wo product and of its weight gradient -- / dQ | dK |
// dV as [3H / 32][3][rows][32] -- the operand of the QKV dX product and of its weight gradient; rows = the packed image's row count
// (whole 128-row tiles).  Bit for bit what mtvaf_f32_split_planes writes over the fp32 result, which is written too.
int mtvaf_prefix_attn_varlen_fwd_planes(const float* qkv, const float* pk, const float* pv, const int* cu, int pad_rows, float* ctx,
                                        float* lse, int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed,
                                        uint64_t offset, void* ctx_planes, int rows, hipStream_t st) {
  if (!cu || !ctx_planes) return MTVAF_ERR_ARG;
  return attn_fwd_launch(qkv, pk, pv, nullptr, cu, pad_rows, ctx, lse, B, S, P, NH, head_dim, p_drop, seed, offset, st, ctx_planes, rows);
}
int mtvaf_prefix_attn_varlen_bwd_planes(const float* dctx, const float* qkv, const float* pk, const float* pv, const int* cu, int pad_rows,
                                        const float* ctx, const float* lse, float* delta, float* dqkv, float* dpk, float* dpv, int B,
                                        int S, int P, int NH, int head_dim, float p_drop, uint64_t seed, uint64_t offset,
                                        void* dqkv_planes, int rows, hipStream_t st) {
  if (!cu || !dqkv_planes) return MTVAF_ERR_ARG;
  return attn_bwd_launch(dctx, qkv, pk, pv, nullptr, cu, pad_rows, ctx, lse, delta, dqkv, dpk, dpv, B, S, P, NH, head_dim, p_drop, seed,
                         offset, st, 0, dqkv_planes, rows);
}

}  // extern "C"
