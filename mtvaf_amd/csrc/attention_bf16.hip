// Fused prefix self-attention of the mixed-precision mode (bf16 operands, v_mfma_f32_16x16x32_bf16, fp32 softmax
// statistics and accumulation): the same algorithm, key order, masking and dropout hash as attention.hip
//   softmax(Q.[Kp;K]^T / sqrt(D) + mask) . [Vp;V]     models/modeling_bert.py:282-286, 303, 320-337
// reading the bf16 Q|K|V written by the QKV projection's epilogue and writing the bf16 context / bf16 dQ|dK|dV that the
// next projections consume -- no fp32 copies of these tensors exist in this mode, and the per-head column sums of
// dQ|dK|dV (the QKV bias gradient) leave the backward kernel as per-block partials.
//
// Data layout:
//   qkv16   [B*S, 3H] bf16  token-major (Q | K | V column blocks)
//   pk16,pv16 [B, P*H] bf16 one layer's prefix slab (head h, slot p, dim d at h*(P*64) + p*64 + d), cast once per step
//   addmask [B, T] fp32     additive mask, T = P + S
//   ctx16   [B*S, H] bf16   merged heads;  lse [B, NH, S] fp32 (natural log)
//
// MFMA mapping (as attention.hip, 8x the k-depth): scores are produced TRANSPOSED, S^T[key][q] = K.Q^T, so a query
// lives on a lane (A = K rows from LDS by ds_read_b128, B = Q straight from global memory: 8 consecutive d per lane);
// the C layout hands lane (q, g) the keys 16*kb + 4g + r, which -- rounded to bf16 -- ARE the B operand of
// O^T[d][q] = V^T.P^T when the k-slot (g, j) of that product is defined as key 16*(2u + (j >> 2)) + 4g + (j & 3); the
// matching A operand V^T comes from the row-major V tile by two transposing reads (ds_read_b64_tr_b16).  All [64][64]
// bf16 tiles (128-byte rows) use ONE LDS image that is conflict-free for both the row reads and the transposing reads:
// 16-byte chunk c of row r at chunk c ^ (((r >> 1) & 3) << 1).
#include "common.h"

namespace mtvaf {

namespace ab {

constexpr int D = 64;
constexpr int KT = 64;
constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define MFMA_BF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

struct Args {
  const __bf16* qkv;
  const __bf16* pk;
  const __bf16* pv;
  const float* addmask;
  __bf16* ctx;
  float* lse;
  // backward
  const __bf16* dctx;
  __bf16* dqkv;
  float* dpk;
  float* dpv;
  float* partq;   // [B * nqt][H]   column sums of dQ per query-tile block
  float* partkv;  // [B * nkt][2H]  column sums of dK | dV (text keys only) per key-tile block
  int B, S, P, NH, H;
  float scale, p_drop;
  uint32_t drop_key, drop_thr;
  const uint64_t* epoch;  // device-side dropout epoch (captured launches), or NULL
  const int* cu;          // PACKED token rows (see attention.hip): [B+1] row offsets of the sentences, or NULL
  int pad_rows;           // rows behind the last sentence that pad the packed image: zero-filled by the z-slice b == B
  int zero_tail;          // backward, padded layout: dctx is exactly zero behind a sentence's last unmasked position (the caller's
                          // word, as in attention.hip): the key side's query loop stops there
};

struct Sent {
  long tok0;
  int n;
};
__device__ __forceinline__ Sent sentence(const Args& a, int b) {
  if (a.cu) {
    const int c0 = b ? a.cu[b] : 0;  // (cu[0] = -1 marks a launch order behind the offsets: slot_sentence)
    return Sent{(long)c0, a.cu[b + 1] - c0};
  }
  return Sent{(long)b * a.S, a.S};
}
// the sentence of grid slot z (mtvaf_build_packing_ordered: longest first; identity without the list)
__device__ __forceinline__ int slot_sentence(const Args& a, int z) { return (a.cu && a.cu[0] < 0) ? a.cu[a.B + 1 + z] : z; }
__device__ __forceinline__ float mask_at(const Args& a, int b, int Tf, int t) {
  return a.cu ? 0.f : a.addmask[(long)b * Tf + t];
}

// byte offset of 16-byte chunk c (0..7) of row r in a [64][64] bf16 tile image
__device__ __forceinline__ int tile_off(int r, int c) { return r * 128 + ((c ^ (((r >> 1) & 3) << 1)) << 4); }

// [prefix ; text] row pointer selected with bit arithmetic (no divergent branches: see attention.hip)
struct KvSrc {
  const __bf16* pre;
  const __bf16* txt;
};
__device__ __forceinline__ const __bf16* kv_row_ptr(const KvSrc& s, int t, int P, int ld_txt) {
  const bool ispre = t < P;
  const uint64_t m = ispre ? ~0ull : 0ull;
  const uint64_t base = (uint64_t)s.txt ^ (((uint64_t)s.txt ^ (uint64_t)s.pre) & m);
  const int off = ispre ? t * D : (t - P) * ld_txt;
  return reinterpret_cast<const __bf16*>(base) + off;
}

// stage a [64][64] bf16 tile: 512 chunks of 16 B, two per thread (rows r and r + 32, chunk c = tid & 7)
__device__ __forceinline__ void tile_load_kv(bf16x8 (&reg)[2], const KvSrc& s, int P, int T, int ld_txt, int t0) {
  const int c = threadIdx.x & 7, r = threadIdx.x >> 3;
#pragma unroll
  for (int i = 0; i < 2; ++i)
    reg[i] = *reinterpret_cast<const bf16x8*>(kv_row_ptr(s, min(t0 + r + 32 * i, T - 1), P, ld_txt) + c * 8);
}
__device__ __forceinline__ void tile_store(unsigned char* dst, const bf16x8 (&reg)[2]) {
  const int c = threadIdx.x & 7, r = threadIdx.x >> 3;
#pragma unroll
  for (int i = 0; i < 2; ++i) *reinterpret_cast<bf16x8*>(dst + tile_off(r + 32 * i, c)) = reg[i];
}

// A operand of a "rows x d" product: row (16*blk + lane&15), the 8 d values 32*ks + 8g .. +7
__device__ __forceinline__ bf16x8 row_frag(const unsigned char* tile, int blk, int ks, int lr, int g) {
  return *reinterpret_cast<const bf16x8*>(tile + tile_off(16 * blk + lr, 4 * ks + g));
}
// A operand of a "d x rows" product (transposed read): d = 16*dt + lane&15, k-slot (g, j) = tile row
// 16*(2u + (j >> 2)) + 4g + (j & 3).  Lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of the 4 x 16 block.
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* tile, int u, int dt, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int r0 = 32 * u + 4 * g + q, r1 = r0 + 16;
  const int c = 2 * dt + (p >> 1), e = 8 * (p & 1);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + tile_off(r0, c) + e));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + tile_off(r1, c) + e));
  return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  return bf16x8{(__bf16)a.x, (__bf16)a.y, (__bf16)a.z, (__bf16)a.w, (__bf16)b.x, (__bf16)b.y, (__bf16)b.z, (__bf16)b.w};
}
__device__ __forceinline__ float dot8(const bf16x8& a, const bf16x8& b) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += (float)a[i] * (float)b[i];
  return s;
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

// ---------------------------------------------------------------------------------------------
// forward: grid (ceil(S/64), NH, B), 256 threads; wave w owns queries q0+16w .. +15
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bf16_fwd_kernel(Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char Ks[KT * 128];
  __shared__ __attribute__((aligned(16))) unsigned char Vs[KT * 128];
  __shared__ __attribute__((aligned(16))) float Ms[KT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lq = lane & 15, g = lane >> 4;
  int b = blockIdx.z;
  const int h = blockIdx.y;
  const int q = blockIdx.x * 64 + wave * 16 + lq;
  if (a.cu && b == a.B) {  // (block-uniform) the rows that pad the packed image: zeros
    const int r0 = a.cu[a.B];
    const bf16x8 z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
    for (int r = blockIdx.x * 32 + (threadIdx.x >> 3); r < a.pad_rows; r += gridDim.x * 32)
      *reinterpret_cast<bf16x8*>(a.ctx + (long)(r0 + r) * a.H + h * D + (threadIdx.x & 7) * 8) = z;
    return;
  }
  b = slot_sentence(a, b);
  const Sent sn = sentence(a, b);
  const int Sb = sn.n;
  if ((int)blockIdx.x * 64 >= Sb) return;  // (block-uniform; packed rows: a query tile beyond the sentence)
  const int Tf = a.P + a.S;  // row length of the additive mask
  __shared__ int t_eff_slot;
  const int T = a.cu ? a.P + Sb : effective_keys(a.addmask + (long)b * Tf, a.P, a.S, &t_eff_slot);  // trailing padding keys are skipped
  const bool qok = q < Sb;
  const bool wave_live = __builtin_amdgcn_readfirstlane(q - lq) < Sb;  // (round 6, as attention.hip: dead waves skip the arithmetic)
  const float inv_keep = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  const uint32_t rowh = attn_dropout_rowhash(attn_epoch_key(a.drop_key, a.epoch), (uint32_t)((b * a.NH + h) * a.S + q));
  const float sc2 = a.scale * LOG2E;

  KvSrc ksrc, vsrc;
  ksrc.pre = a.pk + ((long)b * a.P * a.NH + (long)h * a.P) * D;
  vsrc.pre = a.pv + ((long)b * a.P * a.NH + (long)h * a.P) * D;
  ksrc.txt = a.qkv + sn.tok0 * 3 * a.H + a.H + h * D;
  vsrc.txt = ksrc.txt + a.H;
  const int ldt = 3 * a.H;

  bf16x8 qf[2];
  {
    const __bf16* qp = a.qkv + (sn.tok0 + min(q, Sb - 1)) * 3 * a.H + h * D + 8 * g;
    qf[0] = *reinterpret_cast<const bf16x8*>(qp);
    qf[1] = *reinterpret_cast<const bf16x8*>(qp + 32);
  }
  f32x4 oacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = NEG_BIG, l_run = 0.f;

  // (prefetching the next key tile into registers, as the backward kernels do, measured 6 % SLOWER here: 13.6 -> 14.5 us)
  for (int t0 = 0; t0 < T; t0 += KT) {
    bf16x8 kreg[2], vreg[2];
    tile_load_kv(kreg, ksrc, a.P, T, ldt, t0);
    tile_load_kv(vreg, vsrc, a.P, T, ldt, t0);
    float mreg = NEG_BIG;
    if (threadIdx.x < KT) mreg = mask_at(a, b, Tf, min(t0 + (int)threadIdx.x, T - 1));
    __syncthreads();
    tile_store(Ks, kreg);
    tile_store(Vs, vreg);
    if (threadIdx.x < KT) Ms[threadIdx.x] = (t0 + (int)threadIdx.x < T) ? mreg * LOG2E : NEG_BIG;
    __syncthreads();
    if (!wave_live) continue;  // (wave-uniform) a wave whose 16 queries all lie beyond the sentence only stages and synchronises
    const int nsub = min(4, (T - t0 + 15) >> 4);  // 16-key blocks of this tile that hold real keys
    f32x4 s[4];
    float tmax = NEG_BIG;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[j] = f32x4{NEG_BIG, NEG_BIG, NEG_BIG, NEG_BIG};
      if (j < nsub) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = MFMA_BF(row_frag(Ks, j, 0, lq, g), qf[0], acc);
        acc = MFMA_BF(row_frag(Ks, j, 1, lq, g), qf[1], acc);
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
        const float p = __builtin_amdgcn_exp2f(s[j][r] - m_new);
        psum += p;
        float pd = p;
        if (a.p_drop > 0.f)
          pd = attn_dropout_keep2(rowh, cterm0 + (uint32_t)(16 * j + r) * ATTN_DROP_C2, a.drop_thr) ? p * inv_keep : 0.f;
        s[j][r] = pd;
      }
    psum += __shfl_xor(psum, 16, 64);
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] *= alpha;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (2 * u < nsub) {
        const bf16x8 pb = pack8(s[2 * u], s[2 * u + 1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oacc[dt] = MFMA_BF(tr_frag(Vs, u, dt, lane), pb, oacc[dt]);
      }
    }
  }
  if (qok) {
    const float inv_l = 1.f / l_run;
    __bf16* op = a.ctx + (sn.tok0 + q) * a.H + h * D + 4 * g;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const f32x4 o = oacc[dt] * inv_l;
      *reinterpret_cast<bf16x4*>(op + 16 * dt) = bf16x4{(__bf16)o.x, (__bf16)o.y, (__bf16)o.z, (__bf16)o.w};
    }
    if (g == 0) a.lse[((long)b * a.NH + h) * a.S + q] = (m_run + log2f(l_run)) * LN2;
  }
}

// ---------------------------------------------------------------------------------------------
// backward, query side: dQ for 64 queries per block (wave w: 16 of them), loop over key tiles
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void bwd_dq_body(const Args& a, int qtile, unsigned char* Ks, unsigned char* Vs, float* Ms, float* red,
                                            int* t_eff_slot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lq = lane & 15, g = lane >> 4;
  const int b = slot_sentence(a, blockIdx.z), h = blockIdx.y;
  const int q = qtile * 64 + wave * 16 + lq;
  const Sent sn = sentence(a, b);
  const int Sb = sn.n;
  if (qtile * 64 >= Sb) {  // (block-uniform) a query tile beyond the sentence: its column-sum partial is zero
    if (threadIdx.x < 64) a.partq[((long)b * ((a.S + 63) / 64) + qtile) * a.H + h * D + threadIdx.x] = 0.f;
    return;
  }
  const int Tf = a.P + a.S;
  const int T = a.cu ? a.P + Sb : effective_keys(a.addmask + (long)b * Tf, a.P, a.S, t_eff_slot);
  const bool qok = q < Sb;
  const bool wave_live = __builtin_amdgcn_readfirstlane(q - lq) < Sb;
  const float inv_keep = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  const uint32_t rowh = attn_dropout_rowhash(attn_epoch_key(a.drop_key, a.epoch), (uint32_t)((b * a.NH + h) * a.S + q));
  const float sc2 = a.scale * LOG2E;

  KvSrc ksrc, vsrc;
  ksrc.pre = a.pk + ((long)b * a.P * a.NH + (long)h * a.P) * D;
  vsrc.pre = a.pv + ((long)b * a.P * a.NH + (long)h * a.P) * D;
  ksrc.txt = a.qkv + sn.tok0 * 3 * a.H + a.H + h * D;
  vsrc.txt = ksrc.txt + a.H;
  const int ldt = 3 * a.H;

  bf16x8 qf[2], dof[2];
  float dl = 0.f;
  {
    const long qrow = sn.tok0 + min(q, Sb - 1);
    const __bf16* qp = a.qkv + qrow * 3 * a.H + h * D + 8 * g;
    const __bf16* dop = a.dctx + qrow * a.H + h * D + 8 * g;
    const __bf16* op = a.ctx + qrow * a.H + h * D + 8 * g;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 32 * ks);
      dof[ks] = *reinterpret_cast<const bf16x8*>(dop + 32 * ks);
      dl += dot8(dof[ks], *reinterpret_cast<const bf16x8*>(op + 32 * ks));
    }
  }
  dl += __shfl_xor(dl, 16, 64);
  dl += __shfl_xor(dl, 32, 64);
  // rows beyond S: lse = +1e30 makes every probability (and with it ds) exactly 0
  const float lse2 = qok ? a.lse[((long)b * a.NH + h) * a.S + q] * LOG2E : 1.0e30f;

  f32x4 dq[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dq[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // the next key tile travels global -> registers while the current one is multiplied (first needed at the top of the
  // next iteration)
  bf16x8 kreg[2], vreg[2];
  float mreg = NEG_BIG;
  auto fetch = [&](int t0) {
    tile_load_kv(kreg, ksrc, a.P, T, ldt, t0);
    tile_load_kv(vreg, vsrc, a.P, T, ldt, t0);
    if (threadIdx.x < KT) mreg = mask_at(a, b, Tf, min(t0 + (int)threadIdx.x, T - 1));
  };
  fetch(0);
  for (int t0 = 0; t0 < T; t0 += KT) {
    __syncthreads();
    tile_store(Ks, kreg);
    tile_store(Vs, vreg);
    if (threadIdx.x < KT) Ms[threadIdx.x] = (t0 + (int)threadIdx.x < T) ? mreg * LOG2E : NEG_BIG;
    __syncthreads();
    if (t0 + KT < T) fetch(t0 + KT);
    if (!wave_live) continue;  // (wave-uniform; dq stays zero: the column sums below read it through qok)
    const int nsub = min(4, (T - t0 + 15) >> 4);
    const uint32_t cterm0 = (uint32_t)(t0 + 4 * g) * ATTN_DROP_C2;
    f32x4 ds[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ds[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (j < nsub) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          s = MFMA_BF(row_frag(Ks, j, ks, lq, g), qf[ks], s);
          dp = MFMA_BF(row_frag(Vs, j, ks, lq, g), dof[ks], dp);
        }
        const f32x4 mv = *reinterpret_cast<const f32x4*>(Ms + 16 * j + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(s[r] * sc2 + mv[r] - lse2);
          float dpe = dp[r];
          if (a.p_drop > 0.f)
            dpe = attn_dropout_keep2(rowh, cterm0 + (uint32_t)(16 * j + r) * ATTN_DROP_C2, a.drop_thr) ? dpe * inv_keep : 0.f;
          ds[j][r] = p * (dpe - dl) * a.scale;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (2 * u < nsub) {
        const bf16x8 dsb = pack8(ds[2 * u], ds[2 * u + 1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = MFMA_BF(tr_frag(Ks, u, dt, lane), dsb, dq[dt]);
      }
    }
  }
  if (qok) {
    __bf16* dqp = a.dqkv + (sn.tok0 + q) * 3 * a.H + h * D + 4 * g;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *reinterpret_cast<bf16x4*>(dqp + 16 * dt) = bf16x4{(__bf16)dq[dt].x, (__bf16)dq[dt].y, (__bf16)dq[dt].z, (__bf16)dq[dt].w};
  }
  // column sums of this block's dQ rows (the query-bias gradient): over the 16 query lanes, then the 4 waves
  __syncthreads();
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = qok ? dq[dt][r] : 0.f;
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
      if (lq == 0) red[wave * 64 + 16 * dt + 4 * g + r] = v;
    }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int nqt = (a.S + 63) / 64;
    a.partq[((long)b * nqt + qtile) * a.H + h * D + threadIdx.x] =
        red[threadIdx.x] + red[64 + threadIdx.x] + red[128 + threadIdx.x] + red[192 + threadIdx.x];
  }
}

// ---------------------------------------------------------------------------------------------
// backward, key side: dK, dV for 64 keys of the [prefix ; text] axis per block; loops over query tiles.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void bwd_dkv_body(const Args& a, int ktile, unsigned char* Qs, unsigned char* dOs, float* lse_s,
                                             float* del_s, uint32_t* rh_s, float* red, int* t_eff_slot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lk = lane & 15, g = lane >> 4;
  const int b = slot_sentence(a, blockIdx.z), h = blockIdx.y;
  const Sent sn = sentence(a, b);
  const int Sb = sn.n;
  const int nkt = (a.P + a.S + 63) / 64;  // rows of partkv per sentence (allocation: the padded key count)
  const int T = a.cu ? a.P + Sb : effective_keys(a.addmask + (long)b * (a.P + a.S), a.P, a.S, t_eff_slot);  // keys >= T: trailing padding, dK = dV = 0
  const int Tf = a.cu ? T : a.P + a.S;  // (packed rows: keys beyond the sentence do not exist)
  const int key = ktile * 64 + wave * 16 + lk;
  if (ktile * 64 >= T) {  // (block-uniform) a key tile of trailing padding only: exact zeros, no query loop
    if (key < Tf) {
      __bf16* dkrow = a.dqkv + (sn.tok0 + (key - a.P)) * 3 * a.H + a.H + h * D + 4 * g;
      const bf16x4 z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        *reinterpret_cast<bf16x4*>(dkrow + 16 * dt) = z;
        *reinterpret_cast<bf16x4*>(dkrow + a.H + 16 * dt) = z;
      }
    }
    if (threadIdx.x < 128)
      a.partkv[((long)b * nkt + ktile) * 2 * a.H + (threadIdx.x >> 6) * a.H + h * D + (threadIdx.x & 63)] = 0.f;
    return;
  }
  const bool kok = key < T;
  const bool wave_live = (int)(ktile * 64 + wave * 16) < T;
  const int keyc = min(key, T - 1);
  const float inv_keep = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  const float mval2 = kok ? mask_at(a, b, Tf, key) * LOG2E : NEG_BIG;  // keys beyond T: probability exactly 0
  const float sc2 = a.scale * LOG2E;
  const uint32_t cterm = (uint32_t)key * ATTN_DROP_C2;

  bf16x8 kf[2], vf[2];
  {
    KvSrc ksrc, vsrc;
    ksrc.pre = a.pk + ((long)b * a.P * a.NH + (long)h * a.P) * D;
    vsrc.pre = a.pv + ((long)b * a.P * a.NH + (long)h * a.P) * D;
    ksrc.txt = a.qkv + sn.tok0 * 3 * a.H + a.H + h * D;
    vsrc.txt = ksrc.txt + a.H;
    const __bf16* krow = kv_row_ptr(ksrc, keyc, a.P, 3 * a.H) + 8 * g;
    const __bf16* vrow = kv_row_ptr(vsrc, keyc, a.P, 3 * a.H) + 8 * g;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[ks] = *reinterpret_cast<const bf16x8*>(krow + 32 * ks);
      vf[ks] = *reinterpret_cast<const bf16x8*>(vrow + 32 * ks);
    }
  }
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dk[i] = dv[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int c8 = threadIdx.x & 7, r0 = threadIdx.x >> 3;  // staging: rows r0, r0 + 32; 16-byte chunk c8
  const __bf16* qsrc = a.qkv + sn.tok0 * 3 * a.H + h * D + c8 * 8;
  const __bf16* dosrc = a.dctx + sn.tok0 * a.H + h * D + c8 * 8;
  const __bf16* osrc = a.ctx + sn.tok0 * a.H + h * D + c8 * 8;
  const uint32_t row_base = (uint32_t)((b * a.NH + h) * a.S);

  // the next query tile (Q, dO, O rows, lse) is fetched while the current one is multiplied
  bf16x8 qr[2], orr[2], ofw[2];
  float lreg = 1.0e30f;
  auto fetch = [&](int q0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int qq = min(q0 + r0 + 32 * i, Sb - 1);
      qr[i] = *reinterpret_cast<const bf16x8*>(qsrc + (long)qq * 3 * a.H);
      orr[i] = *reinterpret_cast<const bf16x8*>(dosrc + (long)qq * a.H);
      ofw[i] = *reinterpret_cast<const bf16x8*>(osrc + (long)qq * a.H);
    }
    if (threadIdx.x < KT) lreg = a.lse[((long)b * a.NH + h) * a.S + min(q0 + (int)threadIdx.x, Sb - 1)] * LOG2E;
  };
  const int Sq = (a.zero_tail && !a.cu) ? min(Sb, T - a.P) : Sb;  // (queries behind it have dO = 0: they add exactly nothing)
  if (Sq > 0) fetch(0);
  for (int q0 = 0; q0 < Sq; q0 += KT) {
    float dsum[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) dsum[i] = dot8(orr[i], ofw[i]);
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // the 8 threads of a row are 8 consecutive lanes
      dsum[i] += __shfl_xor(dsum[i], 1, 64);
      dsum[i] += __shfl_xor(dsum[i], 2, 64);
      dsum[i] += __shfl_xor(dsum[i], 4, 64);
    }
    const float lcur = lreg;
    __syncthreads();
    tile_store(Qs, qr);
    tile_store(dOs, orr);
    if (c8 == 0) {
      del_s[r0] = dsum[0];
      del_s[r0 + 32] = dsum[1];
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
    f32x4 pd[4], ds[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      pd[i] = ds[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < nsub) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          s = MFMA_BF(row_frag(Qs, i, ks, lk, g), kf[ks], s);     // S[q][key]: lane = key, rows q = 16i + 4g + r
          dp = MFMA_BF(row_frag(dOs, i, ks, lk, g), vf[ks], dp);  // dP[q][key]
        }
        const f32x4 lse4 = *reinterpret_cast<const f32x4*>(lse_s + 16 * i + 4 * g);
        const f32x4 del4 = *reinterpret_cast<const f32x4*>(del_s + 16 * i + 4 * g);
        const uint4 rh4 = *reinterpret_cast<const uint4*>(rh_s + 16 * i + 4 * g);
        const uint32_t rh[4] = {rh4.x, rh4.y, rh4.z, rh4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(s[r] * sc2 + mval2 - lse4[r]);
          float dpe = dp[r], pdr = p;
          if (a.p_drop > 0.f) {
            const bool keep = attn_dropout_keep2(rh[r], cterm, a.drop_thr);
            pdr = keep ? p * inv_keep : 0.f;
            dpe = keep ? dpe * inv_keep : 0.f;
          }
          pd[i][r] = pdr;
          ds[i][r] = p * (dpe - del4[r]) * a.scale;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (2 * u < nsub) {
        const bf16x8 pb = pack8(pd[2 * u], pd[2 * u + 1]), dsb = pack8(ds[2 * u], ds[2 * u + 1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dv[dt] = MFMA_BF(tr_frag(dOs, u, dt, lane), pb, dv[dt]);  // dV^T[d][key] += dO^T[d][q] Pd[q][key]
          dk[dt] = MFMA_BF(tr_frag(Qs, u, dt, lane), dsb, dk[dt]);  // dK^T[d][key] += Q^T[d][q] dS[q][key]
        }
      }
    }
  }
  const bool is_text = kok && key >= a.P;
  if (!kok && key < Tf) {  // trailing padding inside a partially valid tile
    __bf16* dkrow = a.dqkv + (sn.tok0 + (key - a.P)) * 3 * a.H + a.H + h * D + 4 * g;
    const bf16x4 z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      *reinterpret_cast<bf16x4*>(dkrow + 16 * dt) = z;
      *reinterpret_cast<bf16x4*>(dkrow + a.H + 16 * dt) = z;
    }
  }
  if (kok) {
    if (key < a.P) {
      float* dkrow = a.dpk + ((long)b * a.P * a.NH + (long)h * a.P + key) * D + 4 * g;
      float* dvrow = a.dpv + ((long)b * a.P * a.NH + (long)h * a.P + key) * D + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        *reinterpret_cast<f32x4*>(dkrow + 16 * dt) = dk[dt];
        *reinterpret_cast<f32x4*>(dvrow + 16 * dt) = dv[dt];
      }
    } else {
      __bf16* dkrow = a.dqkv + (sn.tok0 + (key - a.P)) * 3 * a.H + a.H + h * D + 4 * g;
      __bf16* dvrow = dkrow + a.H;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        *reinterpret_cast<bf16x4*>(dkrow + 16 * dt) = bf16x4{(__bf16)dk[dt].x, (__bf16)dk[dt].y, (__bf16)dk[dt].z, (__bf16)dk[dt].w};
        *reinterpret_cast<bf16x4*>(dvrow + 16 * dt) = bf16x4{(__bf16)dv[dt].x, (__bf16)dv[dt].y, (__bf16)dv[dt].z, (__bf16)dv[dt].w};
      }
    }
  }
  // column sums over this block's TEXT keys (the key / value bias gradients)
  __syncthreads();
#pragma unroll
  for (int which = 0; which < 2; ++which)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = is_text ? (which ? dv[dt][r] : dk[dt][r]) : 0.f;
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
        if (lk == 0) red[(which * 4 + wave) * 64 + 16 * dt + 4 * g + r] = v;
      }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int which = threadIdx.x >> 6, d = threadIdx.x & 63;
    const float* rr = red + which * 256 + d;
    a.partkv[((long)b * nkt + ktile) * 2 * a.H + which * a.H + h * D + d] = rr[0] + rr[64] + rr[128] + rr[192];
  }
}

// One launch for the whole attention backward: blocks [0, nq) of x are query tiles (dQ), the rest key tiles (dK, dV).
__global__ __launch_bounds__(256, 2) void attn_bf16_bwd_kernel(Args a, int nq) {
  __shared__ __attribute__((aligned(16))) unsigned char tile0[KT * 128];
  __shared__ __attribute__((aligned(16))) unsigned char tile1[KT * 128];
  __shared__ __attribute__((aligned(16))) float small[3 * KT];
  __shared__ __attribute__((aligned(16))) float red[8 * 64];
  __shared__ int t_eff_slot;
  if (a.cu && (int)blockIdx.z == a.B) {  // (block-uniform) zero dQ | dK | dV of the rows that pad the packed image
    const int r0 = a.cu[a.B], h = blockIdx.y;
    const bf16x8 z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
    for (int r = blockIdx.x * 32 + (threadIdx.x >> 3); r < a.pad_rows; r += gridDim.x * 32)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        *reinterpret_cast<bf16x8*>(a.dqkv + (long)(r0 + r) * 3 * a.H + c * a.H + h * D + (threadIdx.x & 7) * 8) = z;
    return;
  }
  if ((int)blockIdx.x < nq) {
    bwd_dq_body(a, blockIdx.x, tile0, tile1, small, red, &t_eff_slot);
  } else {
    bwd_dkv_body(a, blockIdx.x - nq, tile0, tile1, small, small + KT, reinterpret_cast<uint32_t*>(small + 2 * KT), red, &t_eff_slot);
  }
}

static int check(const Args& a) {
  if (a.B <= 0 || a.S <= 0 || a.P < 0 || a.NH <= 0 || a.H != a.NH * D) return MTVAF_ERR_SHAPE;
  if ((long)a.B * a.NH * a.S >= (1L << 32)) return MTVAF_ERR_SHAPE;
  if (a.p_drop < 0.f || a.p_drop >= 1.f) return MTVAF_ERR_ARG;
  if (a.P > 0 && (!a.pk || !a.pv)) return MTVAF_ERR_ARG;
  if (((uintptr_t)a.qkv | (uintptr_t)a.pk | (uintptr_t)a.pv) & 15) return MTVAF_ERR_ALIGN;
  return MTVAF_OK;
}
static uint32_t host_drop_key(uint64_t seed, uint64_t offset) {  // host-side replica of attn_dropout_key
  auto mix = [](uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; };
  return mix((uint32_t)seed ^ mix((uint32_t)(seed >> 32) ^ mix((uint32_t)offset ^ 0x9E3779B9u)));
}

}  // namespace ab
}  // namespace mtvaf

using namespace mtvaf;

extern "C" {

static int attn16_fwd_launch(const void* qkv16, const void* pk16, const void* pv16, const float* addmask, const int* cu, int pad_rows,
                             void* ctx16, float* lse, int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed, uint64_t offset,
                             hipStream_t st) {
  if (head_dim != ab::D) return MTVAF_ERR_SHAPE;
  ab::Args a{};
  a.qkv = static_cast<const __bf16*>(qkv16); a.pk = static_cast<const __bf16*>(pk16); a.pv = static_cast<const __bf16*>(pv16);
  a.addmask = addmask; a.cu = cu; a.pad_rows = pad_rows; a.ctx = static_cast<__bf16*>(ctx16); a.lse = lse;
  a.B = B; a.S = S; a.P = P; a.NH = NH; a.H = NH * ab::D;
  a.scale = 0.125f; a.p_drop = p_drop;
  a.drop_thr = p_drop > 0.f ? (uint32_t)fminf(p_drop * 4294967296.0f, 4294967040.0f) : 0u;
  a.drop_key = ab::host_drop_key(seed, offset);
  a.epoch = rng_epoch_ptr();
  int rc = ab::check(a);
  if (rc) return rc;
  if (!ctx16 || !lse || (!addmask && !cu) || pad_rows < 0 || (pad_rows && !cu)) return MTVAF_ERR_ARG;
  hipLaunchKernelGGL(ab::attn_bf16_fwd_kernel, dim3((S + 63) / 64, NH, B + (pad_rows > 0 ? 1 : 0)), dim3(256), 0, st, a);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

static int attn16_bwd_launch(const void* dctx16, const void* qkv16, const void* pk16, const void* pv16, const float* addmask,
                             const int* cu, int pad_rows, const void* ctx16, const float* lse, void* dqkv16, float* dpk, float* dpv, float* partq,
                             float* partkv, int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed, uint64_t offset,
                             hipStream_t st, int zero_tail = 0) {
  if (head_dim != ab::D) return MTVAF_ERR_SHAPE;
  ab::Args a{};
  a.zero_tail = zero_tail;
  a.qkv = static_cast<const __bf16*>(qkv16); a.pk = static_cast<const __bf16*>(pk16); a.pv = static_cast<const __bf16*>(pv16);
  a.addmask = addmask; a.cu = cu; a.pad_rows = pad_rows; a.ctx = static_cast<__bf16*>(const_cast<void*>(ctx16)); a.lse = const_cast<float*>(lse);
  a.dctx = static_cast<const __bf16*>(dctx16); a.dqkv = static_cast<__bf16*>(dqkv16); a.dpk = dpk; a.dpv = dpv;
  a.partq = partq; a.partkv = partkv;
  a.B = B; a.S = S; a.P = P; a.NH = NH; a.H = NH * ab::D;
  a.scale = 0.125f; a.p_drop = p_drop;
  a.drop_thr = p_drop > 0.f ? (uint32_t)fminf(p_drop * 4294967296.0f, 4294967040.0f) : 0u;
  a.drop_key = ab::host_drop_key(seed, offset);
  a.epoch = rng_epoch_ptr();
  int rc = ab::check(a);
  if (rc) return rc;
  if (!dctx16 || !ctx16 || !lse || !dqkv16 || !partq || !partkv || (!addmask && !cu)) return MTVAF_ERR_ARG;
  if (P > 0 && (!dpk || !dpv)) return MTVAF_ERR_ARG;
  if (((uintptr_t)dctx16 | (uintptr_t)ctx16 | (uintptr_t)dqkv16) & 15) return MTVAF_ERR_ALIGN;
  if (pad_rows < 0 || (pad_rows && !cu)) return MTVAF_ERR_ARG;
  const int nq = (S + 63) / 64;
  hipLaunchKernelGGL(ab::attn_bf16_bwd_kernel, dim3(nq + (P + S + 63) / 64, NH, B + (pad_rows > 0 ? 1 : 0)), dim3(256), 0, st, a, nq);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

// ctx16 [B*S,H] bf16, lse [B,NH,S] <- attention over [prefix ; text] keys; qkv16 [B*S,3H] / pk16, pv16 [B,P*H] bf16.
int mtvaf_prefix_attn_bf16_fwd(const void* qkv16, const void* pk16, const void* pv16, const float* addmask, void* ctx16,
                               float* lse, int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed,
                               uint64_t offset, hipStream_t st) {
  if (!addmask) return MTVAF_ERR_ARG;
  return attn16_fwd_launch(qkv16, pk16, pv16, addmask, nullptr, 0, ctx16, lse, B, S, P, NH, head_dim, p_drop, seed, offset, st);
}

// dqkv16 [B*S,3H] bf16 (all three column blocks overwritten), dpk / dpv [B,P*H] fp32 <- gradients.
// partq [B*ceil(S/64), H] and partkv [B*ceil((P+S)/64), 2H] fp32: per-block column sums of dQ and of dK | dV (text keys):
// summed over their rows they are the Q / K / V bias gradients.
int mtvaf_prefix_attn_bf16_bwd(const void* dctx16, const void* qkv16, const void* pk16, const void* pv16,
                               const float* addmask, const void* ctx16, const float* lse, void* dqkv16, float* dpk,
                               float* dpv, float* partq, float* partkv, int B, int S, int P, int NH, int head_dim,
                               float p_drop, uint64_t seed, uint64_t offset, hipStream_t st) {
  if (!addmask) return MTVAF_ERR_ARG;
  return attn16_bwd_launch(dctx16, qkv16, pk16, pv16, addmask, nullptr, 0, ctx16, lse, dqkv16, dpk, dpv, partq, partkv, B, S, P, NH,
                           head_dim, p_drop, seed, offset, st);
}

// mtvaf_prefix_attn_bf16_bwd for callers that vouch (zero_tail != 0) that dctx is exactly zero behind each sentence's last
// unmasked position (see mtvaf_prefix_attn_bwd_tail): same bits, the key side's query loop stops there.
int mtvaf_prefix_attn_bf16_bwd_tail(const void* dctx16, const void* qkv16, const void* pk16, const void* pv16, const float* addmask,
                                    const void* ctx16, const float* lse, void* dqkv16, float* dpk, float* dpv, float* partq,
                                    float* partkv, int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed,
                                    uint64_t offset, int zero_tail, hipStream_t st) {
  if (!addmask) return MTVAF_ERR_ARG;
  return attn16_bwd_launch(dctx16, qkv16, pk16, pv16, addmask, nullptr, 0, ctx16, lse, dqkv16, dpk, dpv, partq, partkv, B, S, P, NH,
                           head_dim, p_drop, seed, offset, st, zero_tail);
}

// PACKED token rows (padding-free execution; see mtvaf_prefix_attn_varlen_fwd): cu [B+1] int32 row offsets, no mask read;
// partq / partkv keep their padded row counts (blocks beyond a sentence write zeros).
int mtvaf_prefix_attn_bf16_varlen_fwd(const void* qkv16, const void* pk16, const void* pv16, const int* cu, int pad_rows, void* ctx16,
                                      float* lse, int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed,
                                      uint64_t offset, hipStream_t st) {
  if (!cu) return MTVAF_ERR_ARG;
  return attn16_fwd_launch(qkv16, pk16, pv16, nullptr, cu, pad_rows, ctx16, lse, B, S, P, NH, head_dim, p_drop, seed, offset, st);
}

int mtvaf_prefix_attn_bf16_varlen_bwd(const void* dctx16, const void* qkv16, const void* pk16, const void* pv16, const int* cu,
                                      int pad_rows, const void* ctx16, const float* lse, void* dqkv16, float* dpk, float* dpv, float* partq,
                                      float* partkv, int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed,
                                      uint64_t offset, hipStream_t st) {
  if (!cu) return MTVAF_ERR_ARG;
  return attn16_bwd_launch(dctx16, qkv16, pk16, pv16, nullptr, cu, pad_rows, ctx16, lse, dqkv16, dpk, dpv, partq, partkv, B, S, P, NH,
                           head_dim, p_drop, seed, offset, st);
}

}  // extern "C"
