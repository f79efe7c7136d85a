// fp32 prefix self-attention with its four (forward) / ten (backward) matrix products formed as SPLIT bf16 products -- the fp32
// arithmetic of csrc/gemm_f32x3.hip / gemm_f32p.hip (every fp32 operand value x = x1 + x2 + x3, three RNE bf16 planes; a product
// a.b = a3 b1 + a1 b3 + a2 b2 + a2 b1 + a1 b2 + a1 b1 on v_mfma_f32_16x16x32_bf16, fp32 accumulation, smallest terms first; the
// dropped terms are below 2^-26 |a||b|) -- instead of v_mfma_f32_16x16x4_f32, which runs at 1/16 of the bf16 rate:
//   softmax(Q.[Kp;K]^T / sqrt(D) + mask) . [Vp;V]     models/modeling_bert.py:282-286, 303, 320-337
// Round 6.  Why: on the CUs that hold a launch's long sentences the fp32 matrix pipe is the critical path of the fp32-pipe kernels
// (timing-only ablations of attention.hip, profiles/r06_attention_ablation.txt: 44 % of the forward launch is its QK^T and PV
// products); six bf16 products per fp32 product need 2.7 x fewer matrix cycles.
//
// Same interface (AttnArgs), launch geometry, key order, masking, log2-domain softmax, dropout hash, outputs (fp32 + optional plane
// images) and contracts as attention.hip; the structure is that of attention_bf16.hip (scores TRANSPOSED, S^T[key][q] = K.Q^T: a
// query lives on a lane; the C layout of the score block IS the B operand of O^T[d][q] = V^T.P^T under the k-slot definition
// (g, j) -> key 16 (2u + (j >> 2)) + 4g + (j & 3); V^T by transposing LDS reads) with every operand as three planes:
//   * K / V (Q / dO on the key side of the backward pass) tiles: fp32 rows from global memory, split while they are staged -- three
//     [64][64] bf16 images per tile (the image of attention_bf16.hip: 16-byte chunk c of row r at c ^ (((r >> 1) & 3) << 1));
//   * Q / dO (K / V on the key side) fragments: 8 consecutive d per lane, split once per block into registers;
//   * probabilities, dS: split in registers (their fp32 values are the MFMA result layout already).
#include "attention_args.h"

namespace mtvaf {
namespace as3 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define MFMA_BF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

constexpr int PLANE_B = KT * 128;  // one [64][64] bf16 plane image: 8 KiB
constexpr int TILE_B = 3 * PLANE_B;

struct P3 {  // the three planes of 8 fp32 values
  bf16x8 p[3];
};

// 8 fp32 values -> their three bf16 planes (RNE at each level: the split of csrc/planes.h, bit for bit)
__device__ __forceinline__ P3 split8(f32x4 a, f32x4 b) {
  // (as ROUNDED fp32 values: under -ffp-contract=fast a residual must not fuse with the arithmetic that produced its operand)
  asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));
  unsigned h[4], m[4], l[4];
  pl_split(pl_f32x2{a.x, a.y}, h[0], m[0], l[0]);
  pl_split(pl_f32x2{a.z, a.w}, h[1], m[1], l[1]);
  pl_split(pl_f32x2{b.x, b.y}, h[2], m[2], l[2]);
  pl_split(pl_f32x2{b.z, b.w}, h[3], m[3], l[3]);
  P3 r;
  r.p[0] = __builtin_bit_cast(bf16x8, uint4{h[0], h[1], h[2], h[3]});
  r.p[1] = __builtin_bit_cast(bf16x8, uint4{m[0], m[1], m[2], m[3]});
  r.p[2] = __builtin_bit_cast(bf16x8, uint4{l[0], l[1], l[2], l[3]});
  return r;
}

// byte offset of 16-byte chunk c (0..7) of row r in a [64][64] bf16 plane image
__device__ __forceinline__ int tile_off(int r, int c) { return r * 128 + ((c ^ (((r >> 1) & 3) << 1)) << 4); }

// staging of a [64][64] fp32 tile as three plane images: thread -> rows r, r + 32 (r = tid >> 3), the 8 values of chunk c = tid & 7
struct Stage {
  f32x4 v[2][2];
};
__device__ __forceinline__ void stage_store(unsigned char* tile, const Stage& s) {
  const int c = threadIdx.x & 7, r = threadIdx.x >> 3;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const P3 q = split8(s.v[i][0], s.v[i][1]);
    const int o = tile_off(r + 32 * i, c);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<bf16x8*>(tile + pl * PLANE_B + o) = q.p[pl];
  }
}
// ... of the [prefix ; text] key axis (rows beyond T re-read row T - 1: finite values, probability exactly 0 through the mask tile)
__device__ __forceinline__ void kv_fetch(Stage& s, const KvSrc& src, int P, int T, int ld_txt, int t0) {
  const int c = threadIdx.x & 7, r = threadIdx.x >> 3;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float* row = kv_row_ptr(src, min(t0 + r + 32 * i, T - 1), P, ld_txt) + 8 * c;
    s.v[i][0] = *reinterpret_cast<const f32x4*>(row);
    s.v[i][1] = *reinterpret_cast<const f32x4*>(row + 4);
  }
}

// A operand of a "rows x d" product: row 16 blk + (lane & 15), the 8 d values 32 ks + 8 g .. + 7 of plane pl
__device__ __forceinline__ bf16x8 row_frag(const unsigned char* tile, int pl, int blk, int ks, int lr, int g) {
  return *reinterpret_cast<const bf16x8*>(tile + pl * PLANE_B + tile_off(16 * blk + lr, 4 * ks + g));
}
// A operand of a "d x rows" product (transposing read): d = 16 dt + (lane & 15), k-slot (g, j) = tile row 16 (2u + (j >> 2)) + 4g + (j & 3)
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* tile, int pl, int u, int dt, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int r0 = 32 * u + 4 * g + q, r1 = r0 + 16;
  const int c = 2 * dt + (p >> 1), e = 8 * (p & 1);
  const unsigned char* t = tile + pl * PLANE_B;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(t + tile_off(r0, c) + e));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(t + tile_off(r1, c) + e));
  return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// the six plane pairs of a split product (A plane, B plane), smallest terms first: a3 b1, a1 b3, a2 b2, a2 b1, a1 b2, a1 b1
#define AS3_PAIRS constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0}

// acc += rows(tile, blk) . B^T over d = 64: B as two register fragments per plane (ks = 0, 1)
__device__ __forceinline__ f32x4 rows_dot(const unsigned char* tile, int blk, const P3 (&b)[2], int lr, int g, f32x4 acc) {
  bf16x8 a[3][2];
#pragma unroll
  for (int pl = 0; pl < 3; ++pl)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) a[pl][ks] = row_frag(tile, pl, blk, ks, lr, g);
  AS3_PAIRS;
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) acc = MFMA_BF(a[PA[t]][ks], b[ks].p[PB[t]], acc);
  return acc;
}
// acc[dt] += tile^T(d, rows of slot u) . b over the 32 tile rows of slot u: b = the split of a score-layout register pair
__device__ __forceinline__ void cols_dot(const unsigned char* tile, int u, const P3& b, int lane, f32x4 (&acc)[4]) {
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    bf16x8 a[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) a[pl] = tr_frag(tile, pl, u, dt, lane);
    AS3_PAIRS;
#pragma unroll
    for (int t = 0; t < 6; ++t) acc[dt] = MFMA_BF(a[PA[t]], b.p[PB[t]], acc[dt]);
  }
}

// 8 consecutive fp32 values at p (16-byte aligned) -> planes
__device__ __forceinline__ P3 load_split8(const float* p) {
  return split8(*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4));
}

// ---------------------------------------------------------------------------------------------
// forward: grid (ceil(S/64), NH, B [+ 1]), 256 threads; wave w owns queries q0 + 16 w .. + 15
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_f32s_fwd_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char Ks[TILE_B];
  __shared__ __attribute__((aligned(16))) unsigned char Vs[TILE_B];
  __shared__ __attribute__((aligned(16))) float Ms[KT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lq = lane & 15, g = lane >> 4;
  int bx = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  xcd_group(gridDim.x, gridDim.y, a.B, bx, h, b);
  const int q = bx * 64 + wave * 16 + lq;
  if (a.cu && b == a.B) {  // (block-uniform) the rows that pad the packed image: zeros
    const int r0 = a.cu[a.B];
    for (int r = bx * 16 + (threadIdx.x >> 4); r < a.pad_rows; r += gridDim.x * 16)
      store_ctx(a, (long)(r0 + r), h * D + (threadIdx.x & 15) * 4, f32x4{0.f, 0.f, 0.f, 0.f});
    return;
  }
  b = slot_sentence(a, b);
  const Sent sn = sentence(a, b);
  const int Sb = sn.n;
  if (bx * 64 >= Sb) return;  // (block-uniform; packed rows: a query tile beyond the sentence)
  const int Tf = a.P + a.S;
  __shared__ int t_eff_slot;
  const int T = a.cu ? a.P + Sb : effective_keys(a.addmask + (long)b * Tf, a.P, a.S, &t_eff_slot);
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

  P3 qf[2];
  {
    const float* qp = a.qkv + (sn.tok0 + min(q, Sb - 1)) * 3 * a.H + h * D + 8 * g;
    qf[0] = load_split8(qp);
    qf[1] = load_split8(qp + 32);
  }
  f32x4 oacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = NEG_BIG, l_run = 0.f;

  Stage kst, vst;  // the next key tile travels global -> registers while the current one is multiplied
  float mreg = NEG_BIG;
  auto fetch = [&](int t0) {
    kv_fetch(kst, ksrc, a.P, T, ldt, t0);
    kv_fetch(vst, vsrc, a.P, T, ldt, t0);
    if (threadIdx.x < KT) mreg = mask_at(a, b, Tf, min(t0 + (int)threadIdx.x, T - 1));
  };
  fetch(0);
  for (int t0 = 0; t0 < T; t0 += KT) {
    __syncthreads();
    stage_store(Ks, kst);
    stage_store(Vs, vst);
    if (threadIdx.x < KT) Ms[threadIdx.x] = (t0 + (int)threadIdx.x < T) ? mreg * LOG2E : NEG_BIG;
    __syncthreads();
    if (t0 + KT < T) fetch(t0 + KT);
    if (!wave_live) continue;  // (wave-uniform) no live query in this wave: it only stages and synchronises
    const int nsub = min(4, (T - t0 + 15) >> 4);  // 16-key blocks of this tile that hold real keys
    f32x4 s[4];
    float tmax = NEG_BIG;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[j] = f32x4{NEG_BIG, NEG_BIG, NEG_BIG, NEG_BIG};
      if (j < nsub) {
        const f32x4 acc = rows_dot(Ks, j, qf, lq, g, f32x4{0.f, 0.f, 0.f, 0.f});
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
      if (2 * u < nsub) cols_dot(Vs, u, split8(s[2 * u], s[2 * u + 1]), lane, oacc);
    }
  }
  if (qok) {
    const float inv_l = 1.f / l_run;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) store_ctx(a, sn.tok0 + q, h * D + 16 * dt + 4 * g, oacc[dt] * inv_l);
    if (g == 0) a.lse[((long)b * a.NH + h) * a.S + q] = (m_run + log2f(l_run)) * LN2;
  }
}

// ---------------------------------------------------------------------------------------------
// backward, query side: dQ (and delta = rowsum(dO.O)) for 64 queries per block; loop over key tiles
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void bwd_dq_body(const AttnArgs& a, int qtile, int b, int h, unsigned char* Ks, unsigned char* Vs, float* Ms,
                                            int* t_eff_slot) {
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

  KvSrc ksrc, vsrc;
  ksrc.pre = a.pk + ((long)b * a.P * a.NH + (long)h * a.P) * D;
  vsrc.pre = a.pv + ((long)b * a.P * a.NH + (long)h * a.P) * D;
  ksrc.txt = a.qkv + sn.tok0 * 3 * a.H + a.H + h * D;
  vsrc.txt = ksrc.txt + a.H;
  const int ldt = 3 * a.H;

  P3 qf[2], dof[2];
  float dl = 0.f;
  {
    const long qrow = sn.tok0 + min(q, Sb - 1);
    const float* qp = a.qkv + qrow * 3 * a.H + h * D + 8 * g;
    const float* dop = a.dctx + qrow * a.H + h * D + 8 * g;
    const float* op = a.ctx + qrow * a.H + h * D + 8 * g;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[ks] = load_split8(qp + 32 * ks);
      const f32x4 d0 = *reinterpret_cast<const f32x4*>(dop + 32 * ks), d1 = *reinterpret_cast<const f32x4*>(dop + 32 * ks + 4);
      const f32x4 o0 = *reinterpret_cast<const f32x4*>(op + 32 * ks), o1 = *reinterpret_cast<const f32x4*>(op + 32 * ks + 4);
      dl += o0.x * d0.x + o0.y * d0.y + o0.z * d0.z + o0.w * d0.w + o1.x * d1.x + o1.y * d1.y + o1.z * d1.z + o1.w * d1.w;
      dof[ks] = split8(d0, d1);
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

  Stage kst, vst;
  float mreg = NEG_BIG;
  auto fetch = [&](int t0) {
    kv_fetch(kst, ksrc, a.P, T, ldt, t0);
    kv_fetch(vst, vsrc, a.P, T, ldt, t0);
    if (threadIdx.x < KT) mreg = mask_at(a, b, Tf, min(t0 + (int)threadIdx.x, T - 1));
  };
  fetch(0);
  for (int t0 = 0; t0 < T; t0 += KT) {
    __syncthreads();
    stage_store(Ks, kst);
    stage_store(Vs, vst);
    if (threadIdx.x < KT) Ms[threadIdx.x] = (t0 + (int)threadIdx.x < T) ? mreg * LOG2E : NEG_BIG;
    __syncthreads();
    if (t0 + KT < T) fetch(t0 + KT);
    if (!wave_live) continue;
    const int nsub = min(4, (T - t0 + 15) >> 4);
    const uint32_t cterm0 = (uint32_t)(t0 + 4 * g) * ATTN_DROP_C2;
    f32x4 ds[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ds[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (j < nsub) {
        const f32x4 s = rows_dot(Ks, j, qf, lq, g, f32x4{0.f, 0.f, 0.f, 0.f});
        const f32x4 dp = rows_dot(Vs, j, dof, lq, g, f32x4{0.f, 0.f, 0.f, 0.f});
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
      if (2 * u < nsub) cols_dot(Ks, u, split8(ds[2 * u], ds[2 * u + 1]), lane, dq);  // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
    }
  }
  if (qok) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) store_dqkv(a, sn.tok0 + q, h * D + 16 * dt + 4 * g, dq[dt]);
  }
}

// ---------------------------------------------------------------------------------------------
// backward, key side: dK, dV for 64 keys of the [prefix ; text] axis per block (prefix slots write dpk / dpv); loop over query tiles
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void bwd_dkv_body(const AttnArgs& a, int ktile, int b, int h, unsigned char* Qs, unsigned char* dOs, float* lse_s,
                                             float* del_s, uint32_t* rh_s, int* t_eff_slot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lk = lane & 15, g = lane >> 4;
  const Sent sn = sentence(a, b);
  const int Sb = sn.n;
  const int T = a.cu ? a.P + Sb : effective_keys(a.addmask + (long)b * (a.P + a.S), a.P, a.S, t_eff_slot);  // keys >= T: trailing padding
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
  const float mval2 = kok ? mask_at(a, b, Tf, key) * LOG2E : NEG_BIG;  // keys beyond T: probability exactly 0
  const float sc2 = a.scale * LOG2E;
  const uint32_t cterm = (uint32_t)key * ATTN_DROP_C2;

  P3 kf[2], vf[2];
  {
    KvSrc ksrc, vsrc;
    ksrc.pre = a.pk + ((long)b * a.P * a.NH + (long)h * a.P) * D;
    vsrc.pre = a.pv + ((long)b * a.P * a.NH + (long)h * a.P) * D;
    ksrc.txt = a.qkv + sn.tok0 * 3 * a.H + a.H + h * D;
    vsrc.txt = ksrc.txt + a.H;
    const float* krow = kv_row_ptr(ksrc, keyc, a.P, 3 * a.H) + 8 * g;
    const float* vrow = kv_row_ptr(vsrc, keyc, a.P, 3 * a.H) + 8 * g;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[ks] = load_split8(krow + 32 * ks);
      vf[ks] = load_split8(vrow + 32 * ks);
    }
  }
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dk[i] = dv[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int c8 = threadIdx.x & 7, r0 = threadIdx.x >> 3;  // staging: rows r0, r0 + 32; the 8 values of chunk c8
  const float* qsrc = a.qkv + sn.tok0 * 3 * a.H + h * D + c8 * 8;
  const float* dosrc = a.dctx + sn.tok0 * a.H + h * D + c8 * 8;
  const float* osrc = a.ctx + sn.tok0 * a.H + h * D + c8 * 8;
  const uint32_t row_base = (uint32_t)((b * a.NH + h) * a.S);

  // the next query tile (Q, dO, O rows, lse) is fetched while the current one is multiplied
  Stage qst, dst;
  f32x4 ofw[2][2];
  float lreg = 1.0e30f;
  auto fetch = [&](int q0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int qq = min(q0 + r0 + 32 * i, Sb - 1);
      qst.v[i][0] = *reinterpret_cast<const f32x4*>(qsrc + (long)qq * 3 * a.H);
      qst.v[i][1] = *reinterpret_cast<const f32x4*>(qsrc + (long)qq * 3 * a.H + 4);
      dst.v[i][0] = *reinterpret_cast<const f32x4*>(dosrc + (long)qq * a.H);
      dst.v[i][1] = *reinterpret_cast<const f32x4*>(dosrc + (long)qq * a.H + 4);
      ofw[i][0] = *reinterpret_cast<const f32x4*>(osrc + (long)qq * a.H);
      ofw[i][1] = *reinterpret_cast<const f32x4*>(osrc + (long)qq * a.H + 4);
    }
    if (threadIdx.x < KT) lreg = a.lse[((long)b * a.NH + h) * a.S + min(q0 + (int)threadIdx.x, Sb - 1)] * LOG2E;
  };
  const int Sq = (a.zero_tail && !a.cu) ? min(Sb, T - a.P) : Sb;  // (queries behind it have dO = 0: they add exactly nothing)
  if (Sq > 0) fetch(0);
  for (int q0 = 0; q0 < Sq; q0 += KT) {
    float dsum[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const f32x4 x0 = ofw[i][0] * dst.v[i][0], x1 = ofw[i][1] * dst.v[i][1];
      dsum[i] = (x0.x + x0.y + x0.z + x0.w) + (x1.x + x1.y + x1.z + x1.w);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // the 8 threads of a row are 8 consecutive lanes
      dsum[i] += __shfl_xor(dsum[i], 1, 64);
      dsum[i] += __shfl_xor(dsum[i], 2, 64);
      dsum[i] += __shfl_xor(dsum[i], 4, 64);
    }
    const float lcur = lreg;
    __syncthreads();
    stage_store(Qs, qst);
    stage_store(dOs, dst);
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
        const f32x4 s = rows_dot(Qs, i, kf, lk, g, f32x4{0.f, 0.f, 0.f, 0.f});    // S[q][key]: lane = key, rows q = 16 i + 4 g + r
        const f32x4 dp = rows_dot(dOs, i, vf, lk, g, f32x4{0.f, 0.f, 0.f, 0.f});  // dP[q][key]
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
        cols_dot(dOs, u, split8(pd[2 * u], pd[2 * u + 1]), lane, dv);  // dV^T[d][key] += dO^T[d][q] Pd[q][key]
        cols_dot(Qs, u, split8(ds[2 * u], ds[2 * u + 1]), lane, dk);   // dK^T[d][key] += Q^T[d][q] dS[q][key]
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

// One launch for the whole attention backward: blocks [0, nq) of x are query tiles (dQ), the rest key tiles (dK, dV)
__global__ __launch_bounds__(256, 2) void attn_f32s_bwd_kernel(AttnArgs a, int nq) {
  __shared__ __attribute__((aligned(16))) unsigned char tile0[TILE_B];
  __shared__ __attribute__((aligned(16))) unsigned char tile1[TILE_B];
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
    bwd_dq_body(a, bx, b, h, tile0, tile1, small, &t_eff_slot);
  } else {
    bwd_dkv_body(a, bx - nq, b, h, tile0, tile1, small, small + KT, reinterpret_cast<uint32_t*>(small + 2 * KT), &t_eff_slot);
  }
}

}  // namespace as3

int launch_attn_f32s_fwd(const AttnArgs& a, dim3 grid, hipStream_t st) {
  hipLaunchKernelGGL(as3::attn_f32s_fwd_kernel, grid, dim3(256), 0, st, a);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}
int launch_attn_f32s_bwd(const AttnArgs& a, int nq, dim3 grid, hipStream_t st) {
  hipLaunchKernelGGL(as3::attn_f32s_bwd_kernel, grid, dim3(256), 0, st, a, nq);
  MTVAF_LAUNCH_CHECK();
  return MTVAF_OK;
}

}  // namespace mtvaf
