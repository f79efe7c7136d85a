"""Parity of every HIP kernel (through the C-ABI) against the CPU oracle / plain torch fp32 math.
Runs only on the MI355X box (`-m gpu`)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import params as P
from x3_cases import CASES as X3_CASES, operands as x3_operands
from oracle import mtvaf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    from mtvaf_amd import hip as h
    h.lib()
    return h


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def close(got, ref, rtol=2e-4, atol=None, name=""):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    if atol is None:
        atol = rtol * float(ref.abs().max()) + 1e-7
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert not bool(bad.any()), f"{name}: max err {float(err.max()):.3e} (ref max {float(ref.abs().max()):.3e}), " \
                                f"{int(bad.sum())}/{bad.numel()} bad"


# ---------------------------------------------------------------------------------------------
# GEMM
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("M,N,K", [(256, 288, 64), (130, 100, 72), (48, 11, 128), (512, 768, 768), (384, 576, 96)])
def test_gemm_nt(hip, cfg, M, N, K):
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    out = torch.empty(M, N, device=DEV)
    hip.gemm(x.to(DEV), hip.KC, w.to(DEV), hip.KC, out, M, N, K, bias=b.to(DEV), cfg=cfg)
    close(out, F.linear(x.double(), w.double(), b.double()), name="nt")


@pytest.mark.parametrize("cfg", [0, 1, 3, 5, 6, 7, 8])
@pytest.mark.parametrize("M,N,K", [(256, 96, 160), (77, 130, 11), (48, 128, 256), (384, 768, 384)])
def test_gemm_nn_and_accumulate(hip, cfg, M, N, K):
    dy, w = rnd(M, K, seed=4), rnd(K, N, seed=5)  # out[M,N] = dy[M,K] . w[K,N]
    out = rnd(M, N, seed=6).to(DEV)
    ref = out.cpu().double() + dy.double() @ w.double()
    hip.gemm(dy.to(DEV), hip.KC, w.to(DEV), hip.KM, out, M, N, K, accumulate=True, cfg=cfg)
    close(out, ref, name="nn+acc")


@pytest.mark.parametrize("cfg,splits", [(0, 1), (3, 4), (1, 3), (-1, -1), (5, 2), (6, 4), (8, 2)])
@pytest.mark.parametrize("M,N,K", [(768, 768, 4096), (11, 768, 1000), (100, 60, 48), (128, 256, 2051)])
def test_gemm_tn_splitk(hip, cfg, splits, M, N, K):
    dy, x = rnd(K, M, seed=7), rnd(K, N, seed=8)  # out[M,N] = dy^T . x (reduction over rows)
    out = torch.empty(M, N, device=DEV)
    hip.gemm(dy.to(DEV), hip.KM, x.to(DEV), hip.KM, out, M, N, K, allow_split=True, cfg=cfg, splits=splits)
    close(out, dy.double().t() @ x.double(), rtol=3e-4, name="tn")
    base = rnd(M, N, seed=9)
    out2 = base.clone().to(DEV)
    hip.gemm(dy.to(DEV), hip.KM, x.to(DEV), hip.KM, out2, M, N, K, allow_split=True, accumulate=True, cfg=cfg,
             splits=splits)
    close(out2, base.double() + dy.double().t() @ x.double(), rtol=3e-4, name="tn+acc")


def test_gemm_epilogues(hip):
    M, N, K = 192, 160, 96
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.2), rnd(N, seed=3)
    pre = F.linear(x.double(), w.double(), b.double())
    out, aux = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    hip.gemm(x.to(DEV), hip.KC, w.to(DEV), hip.KC, out, M, N, K, bias=b.to(DEV), epi=hip.EPI_GELU, aux=aux)
    close(aux, pre, name="gelu-pre")
    close(out, F.gelu(pre), name="gelu")
    hip.gemm(x.to(DEV), hip.KC, w.to(DEV), hip.KC, out, M, N, K, bias=b.to(DEV), epi=hip.EPI_TANH)
    close(out, torch.tanh(pre), name="tanh")
    # dgelu: out = (x.w^T) * gelu'(aux)
    a = rnd(M, N, seed=5)
    hip.gemm(x.to(DEV), hip.KC, w.to(DEV), hip.KC, out, M, N, K, epi=hip.EPI_DGELU, aux=a.to(DEV))
    ad = a.double().requires_grad_(True)
    F.gelu(ad).sum().backward()
    close(out, F.linear(x.double(), w.double()) * ad.grad, name="dgelu")
    t = torch.tanh(a)
    hip.gemm(x.to(DEV), hip.KC, w.to(DEV), hip.KC, out, M, N, K, epi=hip.EPI_DTANH, aux=t.to(DEV))
    close(out, F.linear(x.double(), w.double()) * (1 - t.double() ** 2), name="dtanh")


def test_gemm_deterministic(hip):
    M, N, K = 768, 768, 4096
    dy, x = rnd(K, M, seed=7).to(DEV), rnd(K, N, seed=8).to(DEV)
    o1, o2 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    hip.gemm(dy, hip.KM, x, hip.KM, o1, M, N, K, allow_split=True)
    hip.gemm(dy, hip.KM, x, hip.KM, o2, M, N, K, allow_split=True)
    assert torch.equal(o1, o2)


# ---------------------------------------------------------------------------------------------
# row ops
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,H", [(48, 128), (4096, 768), (37, 1024)])
def test_dropout_res_ln(hip, M, H):
    x, r = rnd(M, H, seed=1), rnd(M, H, seed=2)
    g, b = 1 + 0.1 * rnd(H, seed=3), 0.1 * rnd(H, seed=4)
    dout = rnd(M, H, seed=5)
    xd, rd, gd, bd = (t.double().requires_grad_(True) for t in (x, r, g, b))
    y = F.layer_norm(xd + rd, (H,), gd, bd, 1e-12)
    (y * dout.double()).sum().backward()
    out, mean, rstd = torch.empty(M, H, device=DEV), torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    hip.dropout_res_ln_fwd(x.to(DEV), r.to(DEV), g.to(DEV), b.to(DEV), out, mean, rstd, 1e-12, 0.0, 1, 2)
    close(out, y, name="ln fwd")
    dx, dres = torch.empty(M, H, device=DEV), torch.empty(M, H, device=DEV)
    dg, db = torch.empty(H, device=DEV), torch.empty(H, device=DEV)
    dbx = torch.empty(H, device=DEV)
    hip.dropout_res_ln_bwd(dout.to(DEV), x.to(DEV), r.to(DEV), g.to(DEV), mean, rstd, dx, dres, False, dg, db, False, 0.0,
                           1, 2, dbias_x=dbx)
    close(dbx, xd.grad.sum(0), rtol=5e-4, name="ln dbias_x")
    close(dx, xd.grad, name="ln dx")
    close(dres, rd.grad, name="ln dres")
    close(dg, gd.grad, rtol=5e-4, name="ln dgamma")
    close(db, bd.grad, rtol=5e-4, name="ln dbeta")
    # accumulate flags
    dres2 = torch.ones(M, H, device=DEV)
    dg2, db2 = torch.ones(H, device=DEV), torch.ones(H, device=DEV)
    hip.dropout_res_ln_bwd(dout.to(DEV), x.to(DEV), r.to(DEV), g.to(DEV), mean, rstd, dx, dres2, True, dg2, db2, True, 0.0,
                           1, 2)
    close(dres2, rd.grad + 1, name="ln dres acc")
    close(dg2, gd.grad + 1, rtol=5e-4, name="ln dgamma acc")


def test_dropout_res_ln_with_dropout(hip):
    M, H, p = 512, 768, 0.1
    x, r = rnd(M, H, seed=1), rnd(M, H, seed=2)
    g, b = 1 + 0.1 * rnd(H, seed=3), 0.1 * rnd(H, seed=4)
    # recover the mask with the plain dropout kernel semantics: same (seed, offset, index) -> same mask
    ones = torch.ones(M, H, device=DEV)
    keep = torch.empty(M, H, device=DEV)
    hip.dropout(ones, keep, p, 11, 22)
    keep = keep.cpu()
    frac = float((keep > 0).float().mean())
    assert abs(frac - (1 - p)) < 0.01, frac
    assert torch.allclose(keep[keep > 0], torch.tensor(1 / (1 - p)))
    out, mean, rstd = torch.empty(M, H, device=DEV), torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    hip.dropout_res_ln_fwd(x.to(DEV), r.to(DEV), g.to(DEV), b.to(DEV), out, mean, rstd, 1e-12, p, 11, 22)
    xd, rd = x.double().requires_grad_(True), r.double().requires_grad_(True)
    y = F.layer_norm(xd * keep.double() + rd, (H,), g.double(), b.double(), 1e-12)
    close(out, y, name="ln+dropout fwd")
    dout = rnd(M, H, seed=5)
    (y * dout.double()).sum().backward()
    dx, dres = torch.empty(M, H, device=DEV), torch.empty(M, H, device=DEV)
    dg, db = torch.empty(H, device=DEV), torch.empty(H, device=DEV)
    hip.dropout_res_ln_bwd(dout.to(DEV), x.to(DEV), r.to(DEV), g.to(DEV), mean, rstd, dx, dres, False, dg, db, False, p, 11,
                           22)
    close(dx, xd.grad, name="ln+dropout dx")
    close(dres, rd.grad, name="ln+dropout dres")
    # a different offset gives a different mask
    keep2 = torch.empty(M, H, device=DEV)
    hip.dropout(ones, keep2, p, 11, 23)
    assert not torch.equal(keep2.cpu(), keep)


@pytest.mark.parametrize("cfg", [P.TINY_BERT, P.TINY_ROBERTA, P.EncCfg(vocab_size=1000, hidden=768, max_pos=512),
                                 # vocabularies beyond the owner scheme's 65536-row scan (bert-base-multilingual, xlm-roberta: the
                                 # reference takes any `bert_name`) keep the atomic scatter
                                 P.EncCfg(vocab_size=119547, hidden=128, max_pos=64),
                                 P.EncCfg(vocab_size=250002, hidden=128, type_vocab=1, eps=1e-5, roberta=True, pad_idx=1, max_pos=66)])
def test_embed_ln(hip, cfg):
    B, S = 5, 24
    sd = {k: v for k, v in P.encoder_params(cfg, 5).items() if k.startswith("embeddings.")}
    ids, mask, tt, _ = P.text_batch(cfg, 6, B, S)
    if cfg.roberta:
        ids[1, 3] = 1
        ids[2, 0] = 1
    else:
        tt[:, S // 2:] = 1 if cfg.type_vocab > 1 else 0
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.embeddings(sdg, "embeddings.", ids, tt, cfg.eps, cfg.roberta, cfg.pad_idx)
    dout = rnd(B, S, cfg.hidden, seed=9)
    (ref * dout).sum().backward()
    d = {k: v.to(DEV) for k, v in sd.items()}
    H = cfg.hidden
    out, mean, rstd = torch.empty(B * S, H, device=DEV), torch.empty(B * S, device=DEV), torch.empty(B * S, device=DEV)
    pos_ids = None
    if cfg.roberta:
        pos_ids = torch.empty(B, S, dtype=torch.int32, device=DEV)
        hip.roberta_position_ids(ids.to(DEV), pos_ids, cfg.pad_idx)
        assert torch.equal(pos_ids.cpu().long(), O.roberta_position_ids(ids, cfg.pad_idx))
    args = (ids.to(DEV), tt.to(DEV), pos_ids, d["embeddings.word_embeddings.weight"],
            d["embeddings.position_embeddings.weight"], d["embeddings.token_type_embeddings.weight"],
            d["embeddings.LayerNorm.weight"])
    hip.embed_ln_fwd(*args, d["embeddings.LayerNorm.bias"], out, mean, rstd, cfg.eps, 0.0, 1, 2)
    close(out.view(B, S, H), ref, name="embed fwd")
    gw, gp, gt = (torch.full_like(d[f"embeddings.{n}_embeddings.weight"], 7.0) for n in ("word", "position", "token_type"))
    gg, gb = torch.empty(H, device=DEV), torch.empty(H, device=DEV)
    dz = torch.empty(B * S, H, device=DEV)
    hip.embed_ln_bwd(dout.view(B * S, H).to(DEV), *args, mean, rstd, gw, gp, gt, gg, gb, False, cfg.pad_idx,
                     cfg.pad_idx if cfg.roberta else -1, 0.0, 1, 2, dz)
    close(gw, sdg["embeddings.word_embeddings.weight"].grad, rtol=5e-4, name="dword")
    close(gp, sdg["embeddings.position_embeddings.weight"].grad, rtol=5e-4, name="dpos")
    close(gt, sdg["embeddings.token_type_embeddings.weight"].grad, rtol=5e-4, name="dtype")
    close(gg, sdg["embeddings.LayerNorm.weight"].grad, rtol=5e-4, name="dgamma")
    close(gb, sdg["embeddings.LayerNorm.bias"].grad, rtol=5e-4, name="dbeta")


def test_colsum(hip):
    for rows, cols in [(4096, 768), (100, 11), (5000, 2304)]:
        x = rnd(rows, cols, seed=rows)
        out = torch.ones(cols, device=DEV)
        hip.colsum(x.to(DEV), out, accumulate=True)
        close(out, x.double().sum(0) + 1, rtol=5e-4, name="colsum")


# ---------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------
def attn_ref(qkv, pk, pv, addmask, B, S, Pn, NH):
    H = NH * 64
    q, k, v = (qkv.view(B, S, 3, NH, 64)[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    if Pn:
        k = torch.cat([pk.view(B, NH, Pn, 64), k], 2)
        v = torch.cat([pv.view(B, NH, Pn, 64), v], 2)
    s = q @ k.transpose(-1, -2) / 8.0 + addmask[:, None, None, :]
    p = torch.softmax(s, -1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B * S, H), p


@pytest.mark.parametrize("B,S,Pn,NH", [(3, 16, 0, 2), (3, 16, 4, 2), (2, 128, 36, 12), (2, 100, 16, 3), (1, 200, 36, 2),
                                       (2, 64, 100, 1)])
def test_prefix_attention(hip, B, S, Pn, NH):
    H, T = NH * 64, Pn + S
    qkv = rnd(B * S, 3 * H, seed=1)
    pk, pv = rnd(B, max(Pn, 1) * H, seed=2), rnd(B, max(Pn, 1) * H, seed=3)
    lens = [S] + [max(1, S // (i + 2)) for i in range(B - 1)]
    mask = torch.zeros(B, T)
    for b, Lb in enumerate(lens):
        mask[b, : Pn + Lb] = 1
    addmask = (1 - mask) * -10000.0
    dctx = rnd(B * S, H, seed=4)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (qkv, pk, pv))
    ref, _ = attn_ref(qd, kd, vd, addmask.double(), B, S, Pn, NH)
    (ref * dctx.double()).sum().backward()
    g = lambda t: t.to(DEV)
    ctx, lse = torch.empty(B * S, H, device=DEV), torch.empty(B, NH, S, device=DEV)
    gq, gk, gv, gm = g(qkv), g(pk) if Pn else None, g(pv) if Pn else None, g(addmask)
    hip.prefix_attn_fwd(gq, gk, gv, gm, ctx, lse, B, S, Pn, NH, 0.0, 0, 0)
    close(ctx, ref, name="attn fwd")
    delta = torch.empty(B, NH, S, device=DEV)
    dqkv = torch.full((B * S, 3 * H), float("nan"), device=DEV)
    dpk = torch.full((B, max(Pn, 1) * H), float("nan"), device=DEV) if Pn else None
    dpv = torch.full((B, max(Pn, 1) * H), float("nan"), device=DEV) if Pn else None
    hip.prefix_attn_bwd(g(dctx), gq, gk, gv, gm, ctx, lse, delta, dqkv, dpk, dpv, B, S, Pn, NH, 0.0, 0, 0)
    close(dqkv, qd.grad, rtol=5e-4, name="attn dqkv")
    if Pn:
        close(dpk, kd.grad, rtol=5e-4, name="attn dpk")
        close(dpv, vd.grad, rtol=5e-4, name="attn dpv")


@pytest.mark.parametrize("B,S,Pn,NH,p", [(4, 128, 36, 12, 0.0), (4, 128, 36, 3, 0.1), (3, 200, 16, 2, 0.1), (3, 64, 0, 2, 0.0)])
def test_prefix_attention_backward_zero_tail_contract(hip, B, S, Pn, NH, p):
    """mtvaf_prefix_attn_bwd_tail: when the upstream gradient is exactly zero for the queries behind a sentence's last unmasked
    position (what the masked CRF / the k-tile-list contract guarantees), stopping the query loops there gives the same bits
    as the full loops -- ragged lengths incl. a full and a one-token sentence, holes in the mask, dropout live."""
    H, T = NH * 64, Pn + S
    qkv, pk, pv = rnd(B * S, 3 * H, seed=11), rnd(B, max(Pn, 1) * H, seed=12), rnd(B, max(Pn, 1) * H, seed=13)
    lens = [S, 1, S // 2 + 3, max(2, S // 5)][:B]
    mask = torch.zeros(B, T)
    for b, Lb in enumerate(lens):
        mask[b, : Pn + Lb] = 1
    if B > 2:
        mask[2, Pn + 4] = 0  # a hole before the last unmasked position stays inside the loops
    addmask = (1 - mask) * -10000.0
    dctx = rnd(B * S, H, seed=14).view(B, S, H)
    for b, Lb in enumerate(lens):
        dctx[b, Lb:] = 0.0  # the contract
    g = lambda t: t.to(DEV)
    gq, gk, gv, gm = g(qkv), g(pk) if Pn else None, g(pv) if Pn else None, g(addmask)
    ctx, lse = torch.empty(B * S, H, device=DEV), torch.empty(B, NH, S, device=DEV)
    hip.prefix_attn_fwd(gq, gk, gv, gm, ctx, lse, B, S, Pn, NH, p, 5, 9)
    outs = []
    for tail in (False, True):
        delta = torch.full((B, NH, S), float("nan"), device=DEV)
        dqkv = torch.full((B * S, 3 * H), float("nan"), device=DEV)
        dpk = torch.full((B, max(Pn, 1) * H), float("nan"), device=DEV) if Pn else None
        dpv = torch.full((B, max(Pn, 1) * H), float("nan"), device=DEV) if Pn else None
        hip.prefix_attn_bwd(g(dctx.view(B * S, H)), gq, gk, gv, gm, ctx, lse, delta, dqkv, dpk, dpv, B, S, Pn, NH, p, 5, 9, zero_tail=tail)
        assert bool(torch.isfinite(dqkv).all())
        outs.append((dqkv, dpk, dpv))
    assert torch.equal(outs[0][0], outs[1][0])
    if Pn:
        assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


@pytest.mark.parametrize("B,S,Pn,NH,p", [(4, 128, 36, 12, 0.0), (4, 128, 36, 3, 0.1), (3, 200, 16, 2, 0.1)])
def test_prefix_attention_bf16_backward_zero_tail_contract(hip, B, S, Pn, NH, p):
    """The bf16 attention backward under the zero-tail contract (mtvaf_prefix_attn_bf16_bwd_tail): with dctx exactly zero behind
    each sentence's last unmasked position, the key side's shortened query loop gives the same bits as the full one."""
    H, T = NH * 64, Pn + S
    bf = lambda t: t.to(DEV).to(torch.bfloat16)
    qkv, pk, pv = bf(rnd(B * S, 3 * H, seed=21)), bf(rnd(B, max(Pn, 1) * H, seed=22)), bf(rnd(B, max(Pn, 1) * H, seed=23))
    lens = [S, 1, S // 2 + 3, max(2, S // 5)][:B]
    mask = torch.zeros(B, T)
    for b, Lb in enumerate(lens):
        mask[b, : Pn + Lb] = 1
    if B > 2:
        mask[2, Pn + 4] = 0
    addmask = ((1 - mask) * -10000.0).to(DEV)
    dctx = rnd(B * S, H, seed=24).view(B, S, H)
    for b, Lb in enumerate(lens):
        dctx[b, Lb:] = 0.0
    dctx = bf(dctx.view(B * S, H))
    ctx, lse = torch.empty(B * S, H, device=DEV, dtype=torch.bfloat16), torch.empty(B, NH, S, device=DEV)
    hip.prefix_attn_bf16_fwd(qkv, pk if Pn else None, pv if Pn else None, addmask, ctx, lse, B, S, Pn, NH, p, 5, 9)
    nqt, nkt = (S + 63) // 64, (Pn + S + 63) // 64
    outs = []
    for tail in (False, True):
        dqkv = torch.full((B * S, 3 * H), float("nan"), device=DEV, dtype=torch.bfloat16)
        dpk = torch.full((B, max(Pn, 1) * H), float("nan"), device=DEV) if Pn else None
        dpv = torch.full((B, max(Pn, 1) * H), float("nan"), device=DEV) if Pn else None
        partq, partkv = torch.full((B * nqt, H), float("nan"), device=DEV), torch.full((B * nkt, 2 * H), float("nan"), device=DEV)
        hip.prefix_attn_bf16_bwd(dctx, qkv, pk if Pn else None, pv if Pn else None, addmask, ctx, lse, dqkv, dpk, dpv, partq, partkv,
                                 B, S, Pn, NH, p, 5, 9, zero_tail=tail)
        assert bool(torch.isfinite(dqkv.float()).all()) and bool(torch.isfinite(partq).all()) and bool(torch.isfinite(partkv).all())
        outs.append((dqkv, dpk, dpv, partq, partkv))
    for x, y in zip(outs[0], outs[1]):
        if x is not None:
            assert torch.equal(x, y)


@pytest.mark.parametrize("B,S,Pn,NH", [(3, 16, 0, 2), (3, 16, 4, 2), (2, 128, 36, 12), (2, 100, 16, 3), (1, 200, 36, 2),
                                       (2, 64, 100, 1)])
def test_prefix_attention_bf16(hip, B, S, Pn, NH):
    """bf16 attention kernels (mixed-precision mode) against fp64 math on the SAME bf16-rounded inputs: context and
    gradients to bf16 accuracy (the probabilities are rounded to bf16 before the second product, results are stored as
    bf16), log-sum-exp to fp32 accuracy, and the per-block column sums against the stored gradients."""
    H, T = NH * 64, Pn + S
    bf = lambda t: t.to(torch.bfloat16)
    qkv, pk, pv = bf(rnd(B * S, 3 * H, seed=1)), bf(rnd(B, max(Pn, 1) * H, seed=2)), bf(rnd(B, max(Pn, 1) * H, seed=3))
    lens = [S] + [max(1, S // (i + 2)) for i in range(B - 1)]
    mask = torch.zeros(B, T)
    for b, Lb in enumerate(lens):
        mask[b, : Pn + Lb] = 1
    addmask = (1 - mask) * -10000.0
    dctx = bf(rnd(B * S, H, seed=4))
    qd, kd, vd = (t.double().requires_grad_(True) for t in (qkv, pk, pv))
    ref, probs = attn_ref(qd, kd, vd, addmask.double(), B, S, Pn, NH)
    (ref * dctx.double()).sum().backward()
    g = lambda t: t.to(DEV)
    ctx = torch.empty(B * S, H, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B, NH, S, device=DEV)
    gq, gk, gv, gm = g(qkv), g(pk) if Pn else None, g(pv) if Pn else None, g(addmask)
    hip.prefix_attn_bf16_fwd(gq, gk, gv, gm, ctx, lse, B, S, Pn, NH, 0.0, 0, 0)
    relerr = lambda got, want: float((got.double().cpu() - want).norm() / want.norm())
    assert relerr(ctx, ref.detach()) < 6e-3, relerr(ctx, ref.detach())
    close(ctx.float(), ref.detach(), rtol=2e-2, name="attn bf16 fwd")
    q_, k_ = qd.detach().view(B, S, 3, NH, 64)[:, :, 0].permute(0, 2, 1, 3), qd.detach().view(B, S, 3, NH, 64)[:, :, 1].permute(0, 2, 1, 3)
    if Pn:
        k_ = torch.cat([kd.detach().view(B, NH, Pn, 64), k_], 2)
    lse_ref = torch.logsumexp(q_ @ k_.transpose(-1, -2) / 8.0 + addmask.double()[:, None, None, :], -1)
    close(lse, lse_ref, rtol=1e-5, name="lse")
    dqkv = torch.full((B * S, 3 * H), float("nan"), dtype=torch.bfloat16, device=DEV)
    dpk = torch.full((B, max(Pn, 1) * H), float("nan"), device=DEV) if Pn else None
    dpv = torch.full((B, max(Pn, 1) * H), float("nan"), device=DEV) if Pn else None
    nqt, nkt = (S + 63) // 64, (T + 63) // 64
    partq, partkv = torch.full((B * nqt, H), float("nan"), device=DEV), torch.full((B * nkt, 2 * H), float("nan"), device=DEV)
    hip.prefix_attn_bf16_bwd(g(dctx), gq, gk, gv, gm, ctx, lse, dqkv, dpk, dpv, partq, partkv, B, S, Pn, NH, 0.0, 0, 0)
    assert torch.isfinite(dqkv.float()).all() and torch.isfinite(partq).all() and torch.isfinite(partkv).all()
    assert relerr(dqkv, qd.grad) < 1.2e-2, relerr(dqkv, qd.grad)
    if Pn:
        assert relerr(dpk, kd.grad) < 1.2e-2 and relerr(dpv, vd.grad) < 1.2e-2, (relerr(dpk, kd.grad), relerr(dpv, vd.grad))
    # bias-gradient partials: fp32 sums of the unrounded gradients = column sums of the exact gradient to bf16-input accuracy
    bsum = torch.cat([partq.sum(0), partkv.sum(0)]).cpu().double()
    want = qd.grad.sum(0)
    assert float((bsum - want).norm() / want.norm()) < 1.2e-2
    close(bsum, dqkv.double().cpu().sum(0), rtol=2e-2, atol=2e-2 * float(dqkv.double().cpu().sum(0).abs().max()), name="partials vs stored")


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("B,S,Pn,NH,p", [(5, 128, 36, 12, 0.0), (4, 200, 16, 3, 0.0), (3, 64, 0, 2, 0.0), (6, 128, 36, 4, 0.1)])
def test_varlen_attention_is_the_padded_attention_on_packed_rows(hip, dtype, B, S, Pn, NH, p):
    """Padding-free execution at kernel level: the varlen launch on PACKED rows against the padded launch on the same
    sentences with trailing padding.  Same keys in the same order through the same tiles, and the dropout hash is
    indexed by (sentence, head, query, key) in both layouts: every kept row must come out BIT-identical (context, lse,
    dQ|dK|dV), the prefix gradients too; the rows that pad the packed image are zeros; the bf16 column-sum partials add
    up to the same bias gradients."""
    H, T = NH * 64, Pn + S
    bf16 = dtype == "bf16"
    cast = (lambda t: t.to(torch.bfloat16)) if bf16 else (lambda t: t)
    lens = [S] + [max(1, (S * (i + 1)) // (B + 1)) for i in range(B - 1)]
    Mv = sum(lens)
    Mp = (Mv + 127) // 128 * 128 + 128  # (one whole tile of padding rows as well)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    qkv = cast(rnd(B * S, 3 * H, seed=1))
    pk, pv = cast(rnd(B, max(Pn, 1) * H, seed=2)), cast(rnd(B, max(Pn, 1) * H, seed=3))
    dctx = cast(rnd(B * S, H, seed=4))
    mask = torch.zeros(B, T)
    rows = []
    for b, Lb in enumerate(lens):
        mask[b, : Pn + Lb] = 1
        rows += [b * S + s_ for s_ in range(Lb)]
    rows = torch.tensor(rows)
    keep = torch.zeros(B * S, dtype=torch.bool)
    keep[rows] = True
    dctx = dctx * keep[:, None].to(dctx.dtype)  # (nothing downstream reads a padded query: its upstream gradient is zero)
    addmask = (1 - mask) * -10000.0
    g = lambda t: t.to(DEV)
    gk, gv = (g(pk), g(pv)) if Pn else (None, None)
    odt = torch.bfloat16 if bf16 else torch.float32
    # ---- padded launch
    ctx0, lse0 = torch.zeros(B * S, H, dtype=odt, device=DEV), torch.zeros(B, NH, S, device=DEV)
    dq0 = torch.zeros(B * S, 3 * H, dtype=odt, device=DEV)
    dpk0, dpv0 = (torch.zeros(B, Pn * H, device=DEV), torch.zeros(B, Pn * H, device=DEV)) if Pn else (None, None)
    nqt, nkt = (S + 63) // 64, (T + 63) // 64
    if bf16:
        pq0, pkv0 = torch.zeros(B * nqt, H, device=DEV), torch.zeros(B * nkt, 2 * H, device=DEV)
        hip.prefix_attn_bf16_fwd(g(qkv), gk, gv, g(addmask), ctx0, lse0, B, S, Pn, NH, p, 11, 5)
        hip.prefix_attn_bf16_bwd(g(dctx), g(qkv), gk, gv, g(addmask), ctx0, lse0, dq0, dpk0, dpv0, pq0, pkv0, B, S, Pn, NH, p, 11, 5)
    else:
        delta0 = torch.zeros(B, NH, S, device=DEV)
        hip.prefix_attn_fwd(g(qkv), gk, gv, g(addmask), ctx0, lse0, B, S, Pn, NH, p, 11, 5)
        hip.prefix_attn_bwd(g(dctx), g(qkv), gk, gv, g(addmask), ctx0, lse0, delta0, dq0, dpk0, dpv0, B, S, Pn, NH, p, 11, 5)
    # ---- packed launch
    def packed(t, width):
        out = torch.zeros(Mp, width, dtype=t.dtype)
        out[:Mv] = t[rows]
        return g(out)
    qp, dcp = packed(qkv, 3 * H), packed(dctx, H)
    ctx1 = torch.full((Mp, H), float("nan"), dtype=odt, device=DEV)
    lse1 = torch.zeros(B, NH, S, device=DEV)
    dq1 = torch.full((Mp, 3 * H), float("nan"), dtype=odt, device=DEV)
    dpk1, dpv1 = (torch.zeros(B, Pn * H, device=DEV), torch.zeros(B, Pn * H, device=DEV)) if Pn else (None, None)
    if bf16:
        pq1, pkv1 = torch.full((B * nqt, H), float("nan"), device=DEV), torch.full((B * nkt, 2 * H), float("nan"), device=DEV)
        hip.prefix_attn_bf16_varlen_fwd(qp, gk, gv, g(cu), Mp - Mv, ctx1, lse1, B, S, Pn, NH, p, 11, 5)
        hip.prefix_attn_bf16_varlen_bwd(dcp, qp, gk, gv, g(cu), Mp - Mv, ctx1, lse1, dq1, dpk1, dpv1, pq1, pkv1, B, S, Pn, NH, p, 11, 5)
    else:
        delta1 = torch.zeros(B, NH, S, device=DEV)
        hip.prefix_attn_varlen_fwd(qp, gk, gv, g(cu), Mp - Mv, ctx1, lse1, B, S, Pn, NH, p, 11, 5)
        hip.prefix_attn_varlen_bwd(dcp, qp, gk, gv, g(cu), Mp - Mv, ctx1, lse1, delta1, dq1, dpk1, dpv1, B, S, Pn, NH, p, 11, 5)
    torch.cuda.synchronize()
    rd = rows.to(DEV)
    assert torch.equal(ctx1[:Mv], ctx0[rd]), "context rows"
    assert torch.equal(dq1[:Mv], dq0[rd]), "dQ | dK | dV rows"
    assert float(ctx1[Mv:].float().abs().max()) == 0.0 and float(dq1[Mv:].float().abs().max()) == 0.0, "padding rows of the image"
    for b, Lb in enumerate(lens):
        assert torch.equal(lse1[b, :, :Lb], lse0[b, :, :Lb]), "lse"
    if Pn:
        assert torch.equal(dpk1, dpk0) and torch.equal(dpv1, dpv0), "prefix gradients"
    if bf16:
        assert torch.isfinite(pq1).all() and torch.isfinite(pkv1).all()
        # the padded launch also sums the (exactly zero) gradients of padded keys / the garbage-free padded queries:
        close(pkv1.sum(0), pkv0.sum(0), rtol=1e-5, atol=1e-5 * float(pkv0.sum(0).abs().max()), name="dK | dV column sums")
        valid_q = torch.zeros(B * S, dtype=torch.bool, device=DEV)
        valid_q[rd] = True
        want_q = dq0[:, :H].float()[valid_q].sum(0)
        close(pq1.sum(0), want_q, rtol=2e-2, atol=2e-2 * float(want_q.abs().max()), name="dQ column sums over the kept rows")


def test_prefix_attention_bf16_dropout(hip):
    """Dropout in the bf16 kernels: keep fraction, the forward mask regenerated identically by both backward sides
    (directional derivative by central differences under the same mask)."""
    B, S, Pn, NH, p = 2, 64, 16, 2, 0.3
    H, T = NH * 64, Pn + S
    g = lambda t: t.to(DEV)
    bf = lambda t: t.to(torch.bfloat16)
    qkv, pk, pv = rnd(B * S, 3 * H, seed=1, scale=0.5), rnd(B, Pn * H, seed=2, scale=0.5), rnd(B, Pn * H, seed=3)
    addmask = torch.zeros(B, T)
    pv2 = torch.zeros(B, NH, Pn, 64)
    vt = torch.zeros(B, S, NH, 64)
    for t in range(64):
        if t < Pn:
            pv2[:, :, t, t] = 1
        else:
            vt[:, t - Pn, :, t] = 1
    qkv3 = qkv.view(B * S, 3, H).clone()
    qkv3[:, 2] = vt.reshape(B * S, H)
    qkv3 = qkv3.view(B * S, 3 * H)
    ctx, lse = torch.empty(B * S, H, dtype=torch.bfloat16, device=DEV), torch.empty(B, NH, S, device=DEV)
    hip.prefix_attn_bf16_fwd(g(bf(qkv3)), g(bf(pk)), g(bf(pv2.reshape(B, Pn * H))), g(addmask), ctx, lse, B, S, Pn, NH, p, 5, 6)
    _, probs = attn_ref(bf(qkv3).double(), bf(pk).double(), pv2.reshape(B, Pn * H).double(), addmask.double(), B, S, Pn, NH)
    pt = ctx.float().cpu().view(B, S, NH, 64).permute(0, 2, 1, 3).double()
    kept = pt > 0
    frac = float(kept.float().mean())
    assert abs(frac - (1 - p)) < 0.02, frac
    assert torch.allclose(pt[kept], (probs[..., :64] / (1 - p))[kept], rtol=1.5e-2, atol=1e-5)
    # directional derivative under a fixed mask (fp32 perturbations rounded to bf16: use a coarse step)
    dctx = bf(rnd(B * S, H, seed=4))
    dqkv = torch.empty(B * S, 3 * H, dtype=torch.bfloat16, device=DEV)
    dpk, dpv = torch.empty(B, Pn * H, device=DEV), torch.empty(B, Pn * H, device=DEV)
    partq, partkv = torch.empty(B * 1, H, device=DEV), torch.empty(B * 2, 2 * H, device=DEV)
    gpv = g(bf(pv))
    hip.prefix_attn_bf16_fwd(g(bf(qkv)), g(bf(pk)), gpv, g(addmask), ctx, lse, B, S, Pn, NH, p, 5, 6)
    hip.prefix_attn_bf16_bwd(g(dctx), g(bf(qkv)), g(bf(pk)), gpv, g(addmask), ctx, lse, dqkv, dpk, dpv, partq, partkv, B, S, Pn,
                             NH, p, 5, 6)
    # reference gradient with the SAME mask, recovered from the kernel itself: ctx is linear in V for fixed probabilities
    qd, kd, vd = (t.double().requires_grad_(True) for t in (bf(qkv), bf(pk), bf(pv)))
    ref, pr = attn_ref(qd, kd, vd, addmask.double(), B, S, Pn, NH)
    hip.prefix_attn_bf16_fwd(g(bf(qkv3)), g(bf(pk)), g(bf(pv2.reshape(B, Pn * H))), g(addmask), ctx, lse, B, S, Pn, NH, p, 5, 6)
    # (qkv3 shares Q and K with qkv, so its one-hot-V context exposes exactly this call's keep mask for the first 64 keys)
    keep64 = (ctx.float().cpu().view(B, S, NH, 64).permute(0, 2, 1, 3) > 0).double()
    keep = torch.ones(B, NH, S, T, dtype=torch.double)
    keep[..., :64] = keep64
    # keys >= 64 keep an unknown mask: zero their contribution on both sides by comparing only dV of the first 64 keys
    v_all = torch.cat([vd.view(B, NH, Pn, 64), qd.view(B, S, 3, NH, 64)[:, :, 2].permute(0, 2, 1, 3)], 2)
    o = ((pr * keep / (1 - p))[..., :64] @ v_all[:, :, :64]).permute(0, 2, 1, 3).reshape(B * S, H)
    (o * dctx.double()).sum().backward()
    dv_first = torch.cat([dpv.double().cpu().view(B, NH, Pn, 64),
                          dqkv.double().cpu().view(B, S, 3, NH, 64)[:, :, 2].permute(0, 2, 1, 3)[:, :, :64 - Pn]], 2)
    dv_ref = torch.cat([vd.grad.view(B, NH, Pn, 64), qd.grad.view(B, S, 3, NH, 64)[:, :, 2].permute(0, 2, 1, 3)[:, :, :64 - Pn]], 2)
    assert float((dv_first - dv_ref).norm() / dv_ref.norm()) < 1.5e-2


def test_prefix_attention_dropout(hip):
    B, S, Pn, NH, p = 2, 64, 16, 2, 0.3
    H, T = NH * 64, Pn + S
    g = lambda t: t.to(DEV)
    qkv, pk, pv = rnd(B * S, 3 * H, seed=1, scale=0.5), rnd(B, Pn * H, seed=2, scale=0.5), rnd(B, Pn * H, seed=3)
    addmask = torch.zeros(B, T)
    # V = one-hot rows (first 64 keys) exposes the dropped probabilities directly: ctx[:, d] = P~[:, key d]
    pv2 = torch.zeros(B, NH, Pn, 64)
    vt = torch.zeros(B, S, NH, 64)
    for t in range(64):
        if t < Pn:
            pv2[:, :, t, t] = 1
        else:
            vt[:, t - Pn, :, t] = 1
    qkv3 = qkv.view(B * S, 3, H).clone()
    qkv3[:, 2] = vt.reshape(B * S, H)
    qkv3 = qkv3.view(B * S, 3 * H)
    ctx, lse = torch.empty(B * S, H, device=DEV), torch.empty(B, NH, S, device=DEV)
    hip.prefix_attn_fwd(g(qkv3), g(pk), g(pv2.reshape(B, Pn * H)), g(addmask), ctx, lse, B, S, Pn, NH, p, 5, 6)
    _, probs = attn_ref(qkv3.double(), pk.double(), pv2.reshape(B, Pn * H).double(), addmask.double(), B, S, Pn, NH)
    pt = ctx.cpu().view(B, S, NH, 64).permute(0, 2, 1, 3).double()  # P~[b,h,q,key<64]
    pref = probs[..., :64]
    kept = pt > 0
    frac = float(kept.float().mean())
    assert abs(frac - (1 - p)) < 0.02, frac
    assert torch.allclose(pt[kept], (pref / (1 - p))[kept], rtol=1e-3, atol=1e-6)
    # lse is the log-sum-exp of the UNdropped scores
    q, k = qkv3.view(B, S, 3, NH, 64)[:, :, 0].permute(0, 2, 1, 3), None
    # backward consistency under dropout: directional derivative by central differences (same mask)
    dctx = rnd(B * S, H, seed=4)
    delta = torch.empty(B, NH, S, device=DEV)
    dqkv, dpk, dpv = torch.empty(B * S, 3 * H, device=DEV), torch.empty(B, Pn * H, device=DEV), torch.empty(B, Pn * H, device=DEV)
    gpv = g(pv)
    hip.prefix_attn_fwd(g(qkv), g(pk), gpv, g(addmask), ctx, lse, B, S, Pn, NH, p, 5, 6)
    hip.prefix_attn_bwd(g(dctx), g(qkv), g(pk), gpv, g(addmask), ctx, lse, delta, dqkv, dpk, dpv, B, S, Pn, NH, p, 5, 6)
    dirq, dirk = rnd(B * S, 3 * H, seed=7), rnd(B, Pn * H, seed=8)
    eps = 1e-2

    def f(sign):
        c = torch.empty(B * S, H, device=DEV)
        l2 = torch.empty(B, NH, S, device=DEV)
        hip.prefix_attn_fwd(g(qkv + sign * eps * dirq), g(pk + sign * eps * dirk), gpv, g(addmask), c, l2, B, S, Pn, NH, p, 5, 6)
        return float((c.double().cpu() * dctx.double()).sum())

    num = (f(+1) - f(-1)) / (2 * eps)
    ana = float((dqkv.cpu().double() * dirq.double()).sum() + (dpk.cpu().double() * dirk.double()).sum())
    assert abs(num - ana) < 2e-2 * max(1.0, abs(ana)), (num, ana)


# ---------------------------------------------------------------------------------------------
# CRF
# ---------------------------------------------------------------------------------------------
def _crf_inputs(B, S, C, seed, lengths=None):
    gnr = torch.Generator().manual_seed(seed)
    em = torch.randn(B, S, C, generator=gnr)
    start, end = torch.rand(C, generator=gnr) - 0.5, torch.rand(C, generator=gnr) - 0.5
    trans = torch.rand(C, C, generator=gnr) - 0.5
    if lengths is None:
        lengths = [S] + [int(x) for x in torch.randint(1, S + 1, (B - 1,), generator=gnr)]
    mask = torch.zeros(B, S, dtype=torch.uint8)
    for b, Lb in enumerate(lengths):
        mask[b, :Lb] = 1
    tags = torch.randint(0, C, (B, S), generator=gnr)
    return em, tags, mask, start, end, trans


@pytest.mark.parametrize("B,S,C,scale", [(4, 5, 11, 1), (32, 128, 11, 1), (3, 70, 5, 1), (2, 1, 11, 1), (3, 512, 11, 1),
                                          (5, 131, 16, 1), (4, 9, 1, 1), (6, 128, 11, 6), (4, 66, 11, -1)])
def test_crf(hip, B, S, C, scale):
    """scale > 1: emissions and transitions spread over tens of nats (confident model, near-forbidden transitions) -- the
    scaled linear-domain recursion must stay in range; scale < 0: masks with holes (pytorch-crf carries the score over
    a masked step and scores the edge from the literal previous position)."""
    em, tags, mask, start, end, trans = _crf_inputs(B, S, C, 3 + S)
    if scale > 1:
        em, trans = em * scale, trans * 4 * scale
    if scale < 0:
        holes = torch.rand(B, S, generator=torch.Generator().manual_seed(5)) < 0.2
        holes[:, 0] = False
        mask = mask * (~holes).to(mask.dtype)
    g = lambda t: t.to(DEV)
    emd, sd_, ed, td = (t.double().requires_grad_(True) for t in (em, start, end, trans))
    ref = -O.crf_log_likelihood(emd, tags, mask, sd_, ed, td, "mean")
    (ref * 1.7).backward()
    ws, wsb = hip.crf_workspace(B, S, C, DEV)
    loss = torch.empty(1, device=DEV)
    args = (g(em), g(tags), g(mask), g(start), g(end), g(trans))
    hip.crf_nll_fwd(*args, loss, ws, wsb)
    close(loss, ref.reshape(1), rtol=1e-5, name="crf loss")
    dem = torch.empty(B, S, C, device=DEV)
    ds, de, dt = torch.ones(C, device=DEV), torch.ones(C, device=DEV), torch.ones(C, C, device=DEV)
    gout = torch.tensor([1.7], device=DEV)
    hip.crf_nll_bwd(gout, *args, dem, ds, de, dt, True, ws, wsb)
    close(dem, emd.grad, rtol=1e-4, atol=1e-6, name="crf dem")
    close(ds, sd_.grad + 1, rtol=1e-4, name="crf dstart")
    close(de, ed.grad + 1, rtol=1e-4, name="crf dend")
    tgrad = td.grad if td.grad is not None else torch.zeros_like(td)  # S == 1: no transition is used
    close(dt, tgrad + 1, rtol=1e-4, name="crf dtrans")
    tg, ln = torch.empty(B, S, dtype=torch.int32, device=DEV), torch.empty(B, dtype=torch.int32, device=DEV)
    hip.crf_viterbi(g(em), g(mask), g(start), g(end), g(trans), tg, ln)
    want = O.crf_decode(em, mask, start, end, trans)
    got = [[int(t) for t in row[: int(n)]] for row, n in zip(tg.cpu(), ln.cpu())]
    assert got == want
    assert all(int((row[int(n):] != -1).sum()) == 0 for row, n in zip(tg.cpu(), ln.cpu()))


def test_crf_random_shapes(hip):
    """Thirty random (B, S, C, lengths, holes) draws: loss, all gradients and Viterbi paths of the hand-scheduled kernels
    against the oracle -- S crosses the 8-step unroll groups and the 64-step mask words at arbitrary offsets."""
    gnr = torch.Generator().manual_seed(2024)
    g = lambda t: t.to(DEV)
    for it in range(30):
        B = int(torch.randint(1, 7, (1,), generator=gnr))
        S = int(torch.randint(1, 200, (1,), generator=gnr))
        C = int(torch.randint(2, 17, (1,), generator=gnr))
        em, tags, mask, start, end, trans = _crf_inputs(B, S, C, 1000 + it)
        if it % 3 == 0 and S > 2:
            holes = torch.rand(B, S, generator=gnr) < 0.15
            holes[:, 0] = False
            mask = mask * (~holes).to(mask.dtype)
        emd, sd_, ed, td = (t.double().requires_grad_(True) for t in (em, start, end, trans))
        ref = -O.crf_log_likelihood(emd, tags, mask, sd_, ed, td, "mean")
        ref.backward()
        ws, wsb = hip.crf_workspace(B, S, C, DEV)
        loss = torch.empty(1, device=DEV)
        args = (g(em), g(tags), g(mask), g(start), g(end), g(trans))
        hip.crf_nll_fwd(*args, loss, ws, wsb)
        tag = f"draw {it}: B={B} S={S} C={C}"
        close(loss, ref.reshape(1), rtol=2e-5, name="crf loss " + tag)
        dem = torch.empty(B, S, C, device=DEV)
        ds, de, dt = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty(C, C, device=DEV)
        hip.crf_nll_bwd(None, *args, dem, ds, de, dt, False, ws, wsb)
        close(dem, emd.grad, rtol=1e-4, atol=2e-6, name="crf dem " + tag)
        close(ds, sd_.grad, rtol=1e-4, atol=2e-6, name="crf dstart " + tag)
        close(de, ed.grad, rtol=1e-4, atol=2e-6, name="crf dend " + tag)
        close(dt, td.grad if td.grad is not None else torch.zeros_like(td), rtol=1e-4, atol=2e-6, name="crf dtrans " + tag)
        tg, ln = torch.empty(B, S, dtype=torch.int32, device=DEV), torch.empty(B, dtype=torch.int32, device=DEV)
        hip.crf_viterbi(g(em), g(mask), g(start), g(end), g(trans), tg, ln)
        want = O.crf_decode(em, mask, start, end, trans)
        got = [[int(t) for t in row[: int(n)]] for row, n in zip(tg.cpu(), ln.cpu())]
        assert got == want, tag


def test_crf_bruteforce_known_answer(hip):
    em, tags, mask, start, end, trans = _crf_inputs(3, 5, 11, 77, lengths=[4, 3, 1])
    logZ, best = O.crf_bruteforce(em, mask, start, end, trans)
    g = lambda t: t.to(DEV)
    tg, ln = torch.empty(3, 5, dtype=torch.int32, device=DEV), torch.empty(3, dtype=torch.int32, device=DEV)
    hip.crf_viterbi(g(em), g(mask), g(start), g(end), g(trans), tg, ln)
    got = [[int(t) for t in row[: int(n)]] for row, n in zip(tg.cpu(), ln.cpu())]
    assert got == best
    ws, wsb = hip.crf_workspace(3, 5, 11, DEV)
    loss = torch.empty(1, device=DEV)
    hip.crf_nll_fwd(g(em), g(tags), g(mask), g(start), g(end), g(trans), loss, ws, wsb)
    sc = O.crf_sequence_score(em, tags, mask, start, end, trans)
    want = -float((sc.double() - torch.tensor(logZ)).mean())
    assert abs(float(loss) - want) < 1e-4 * max(1, abs(want))


# ---------------------------------------------------------------------------------------------
# prompt generator pieces + KL
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("NL,W", [(12, 1536), (2, 256)])
def test_prompt_mix(hip, NL, W):
    NI, B, Lp = 4, 3, 4
    hid = W // 2
    enc = rnd(NI, B, Lp, 4 * W, seed=1)
    wp, bp = rnd(NL * 4, Lp * W, seed=2, scale=0.05), rnd(NL * 4, seed=3, scale=0.1)
    gk = rnd(NL, 2, B, NI * Lp * hid, seed=4)
    encd, wpd, bpd = (t.double().requires_grad_(True) for t in (enc, wp, bp))

    def ref_fn(e):
        outs = []
        sm = e.view(NI, B, Lp, 4, W).mean(3).reshape(NI * B, Lp * W)
        gate = torch.softmax(F.leaky_relu(F.linear(sm, wpd, bpd)).view(NI * B, NL, 4), -1)
        kv = torch.einsum("nik,nlkc->nilc", gate, e.view(NI * B, Lp, 4, W))  # [n, NL, L, W]
        kv = kv.view(NI, B, NL, Lp, W).permute(2, 1, 0, 3, 4).reshape(NL, B, NI * Lp, W)
        return torch.stack([kv[..., :hid].reshape(NL, B, -1), kv[..., hid:].reshape(NL, B, -1)], 1)

    ref = ref_fn(encd)
    (ref * gk.double()).sum().backward()
    g = lambda t: t.to(DEV)
    n = NI * B
    sm = torch.empty(n, Lp * W, device=DEV)
    from mtvaf_amd.hip import _ck, _p, _st, lib
    _ck(lib().mtvaf_split_mean(_p(g(enc)), _p(sm), n * Lp, W, _st()), "split_mean")
    close(sm, enc.view(NI, B, Lp, 4, W).mean(3).reshape(n, Lp * W), name="split mean")
    logits = torch.empty(n, NL * 4, device=DEV)
    hip.linear_fwd(sm, g(wp), g(bp), logits)
    gate = torch.empty_like(logits)
    _ck(lib().mtvaf_gate_fwd(_p(logits), _p(gate), n * NL, _st()), "gate")
    pkv = torch.empty(NL, 2, B, NI * Lp * hid, device=DEV)
    genc = g(enc)
    _ck(lib().mtvaf_prompt_mix_fwd(_p(genc), _p(gate), _p(pkv), NI, B, Lp, W, NL, _st()), "mix")
    close(pkv, ref, name="mix fwd")
    dpart = torch.empty(n * Lp, NL * 4, device=DEV)
    dlog = torch.empty(n, NL * 4, device=DEV)
    ggk = g(gk)
    _ck(lib().mtvaf_prompt_mix_bwd_gate(_p(genc), _p(ggk), _p(logits), _p(gate), _p(dpart), _p(dlog), NI, B, Lp, W, NL, _st()),
        "mix bwd gate")
    dwp = torch.empty(NL * 4, Lp * W, device=DEV)
    hip.linear_bwd_weight(dlog, sm, dwp)
    close(dwp, wpd.grad, rtol=1e-3, name="mix dWp")
    dbp = torch.empty(NL * 4, device=DEV)
    hip.colsum(dlog, dbp)
    close(dbp, bpd.grad, rtol=1e-3, name="mix dbp")
    dsm = torch.empty(n, Lp * W, device=DEV)
    hip.linear_bwd_input(dlog, g(wp), dsm)
    denc = torch.empty(NI, B, Lp, 4 * W, device=DEV)
    _ck(lib().mtvaf_prompt_mix_bwd_enc(_p(gate), _p(ggk), _p(dsm), _p(denc), NI, B, Lp, W, NL, _st()), "mix bwd enc")
    close(denc, encd.grad, rtol=1e-3, name="mix denc")


def test_kl_logsoftmax(hip):
    B, N = 5, 2089
    z = rnd(B, N, seed=1, scale=2.0)
    t = torch.softmax(rnd(B, N, seed=2), -1)
    t[0, :10] = 0
    zd = z.double().requires_grad_(True)
    ref = O.kl_batchmean_log_softmax(zd, t.double())
    (ref * 0.7 * 2.0).backward()
    from mtvaf_amd.hip import _ck, _p, _st, lib
    g = lambda x: x.to(DEV)
    loss, row = torch.empty(1, device=DEV), torch.empty(B, device=DEV)
    gz, gt = g(z), g(t)
    _ck(lib().mtvaf_kl_logsoftmax_fwd(_p(gz), _p(gt), _p(loss), _p(row), B, N, _st()), "kl fwd")
    close(loss, ref.reshape(1), rtol=1e-5, name="kl")
    dz = torch.empty(B, N, device=DEV)
    gout = torch.tensor([0.7], device=DEV)
    _ck(lib().mtvaf_kl_logsoftmax_bwd(_p(gout), 2.0, _p(gz), _p(gt), _p(dz), B, N, _st()), "kl bwd")
    close(dz, zd.grad, rtol=1e-4, atol=1e-8, name="kl dz")


@pytest.mark.parametrize("cfg", [9, 10, 11, 12, 13, 14, 15, 16, 17])
def test_gemm_dma_pipeline(hip, cfg):
    """LDS-DMA pipelined kernels (aligned shapes only): all three operand-layout combinations, short and
    long K (1, 2, 3 and many k-tiles exercise the 3-stage ring prologue/drain), split-K, epilogues."""
    bn = {9: 96, 10: 128, 11: 192, 12: 96, 13: 128, 14: 64, 15: 64, 16: 64, 17: 64}[cfg]
    for K in (32, 64, 96, 768):
        M, N = 256, 2 * bn
        x, w, b = rnd(M, K, seed=K), rnd(N, K, seed=K + 1), rnd(N, seed=3)
        out = torch.empty(M, N, device=DEV)
        hip.gemm(x.to(DEV), hip.KC, w.to(DEV), hip.KC, out, M, N, K, bias=b.to(DEV), cfg=cfg)
        close(out, F.linear(x.double(), w.double(), b.double()), name=f"dma nt K={K}")
    M, N, K = 384, 3 * bn, 160
    dy, w = rnd(M, K, seed=4), rnd(K, N, seed=5)
    out = rnd(M, N, seed=6).to(DEV)
    ref = out.cpu().double() + dy.double() @ w.double()
    hip.gemm(dy.to(DEV), hip.KC, w.to(DEV), hip.KM, out, M, N, K, accumulate=True, cfg=cfg)
    close(out, ref, name="dma nn+acc")
    M, N, K = 256, 2 * bn, 2048
    dy, x = rnd(K, M, seed=7), rnd(K, N, seed=8)
    for splits in (1, 4):
        out = torch.empty(M, N, device=DEV)
        hip.gemm(dy.to(DEV), hip.KM, x.to(DEV), hip.KM, out, M, N, K, allow_split=True, cfg=cfg, splits=splits)
        close(out, dy.double().t() @ x.double(), rtol=3e-4, name=f"dma tn s={splits}")
    M, N, K = 128, bn, 96
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.2), rnd(N, seed=3)
    pre = F.linear(x.double(), w.double(), b.double())
    out, aux = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    hip.gemm(x.to(DEV), hip.KC, w.to(DEV), hip.KC, out, M, N, K, bias=b.to(DEV), epi=hip.EPI_GELU, aux=aux, cfg=cfg)
    close(aux, pre, name="dma gelu-pre")
    close(out, F.gelu(pre), name="dma gelu")
    with pytest.raises(RuntimeError):
        hip.gemm(x.to(DEV), hip.KC, w.to(DEV), hip.KC, out, M - 1, N, K, cfg=cfg)  # unaligned shape is refused


# ---------------------------------------------------------------------------------------------
# bf16-compute GEMM (mixed-precision configurations): fp32 buffers, bf16 MFMA, fp32 accumulation
# ---------------------------------------------------------------------------------------------
def _bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float64)


@pytest.mark.parametrize("cfg", [-1, 6, 5, 3])
def test_gemm_bf16_compute(hip, cfg):
    """Exact model of the kernel: operands rounded to bf16 (RNE), products and sums in >= fp32."""
    for (M, N, K) in [(256, 192, 96), (130, 100, 64), (512, 768, 768)]:
        x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
        out = torch.empty(M, N, device=DEV)
        hip.gemm(x.to(DEV), hip.KC, w.to(DEV), hip.KC, out, M, N, K, bias=b.to(DEV), cfg=cfg, compute="bf16")
        ref = _bf16_round(x) @ _bf16_round(w).t() + b.double()
        close(out, ref, rtol=2e-5, name=f"bf16 nt {M}x{N}x{K}")
        # and it stays within bf16 rounding of the fp32 result
        close(out, F.linear(x.double(), w.double(), b.double()), rtol=2e-2, name="bf16 vs fp32")
    M, N, K = 384, 288, 160
    dy, w = rnd(M, K, seed=4), rnd(K, N, seed=5)
    out = rnd(M, N, seed=6).to(DEV)
    ref = out.cpu().double() + _bf16_round(dy) @ _bf16_round(w)
    hip.gemm(dy.to(DEV), hip.KC, w.to(DEV), hip.KM, out, M, N, K, accumulate=True, cfg=cfg, compute="bf16")
    close(out, ref, rtol=2e-5, name="bf16 nn+acc")
    for (M, N, K, splits) in [(256, 192, 2048, 4), (100, 60, 64, 1), (768, 768, 4096, -1)]:
        dy, x = rnd(K, M, seed=7), rnd(K, N, seed=8)
        out = torch.empty(M, N, device=DEV)
        hip.gemm(dy.to(DEV), hip.KM, x.to(DEV), hip.KM, out, M, N, K, allow_split=True, cfg=cfg, splits=splits,
                 compute="bf16")
        close(out, _bf16_round(dy).t() @ _bf16_round(x), rtol=5e-5, name=f"bf16 tn {M}x{N}x{K}")
    # epilogue + unaligned K falls back to the fp32 kernels (exact fp32 result)
    M, N, K = 128, 96, 72
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2)
    out = torch.empty(M, N, device=DEV)
    hip.gemm(x.to(DEV), hip.KC, w.to(DEV), hip.KC, out, M, N, K, compute="bf16")
    close(out, x.double() @ w.double().t(), name="bf16 request on K%32!=0 -> fp32 kernels")
    M, N, K = 128, 192, 64
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.2), rnd(N, seed=3)
    pre = _bf16_round(x) @ _bf16_round(w).t() + b.double()
    out, aux = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    hip.gemm(x.to(DEV), hip.KC, w.to(DEV), hip.KC, out, M, N, K, bias=b.to(DEV), epi=hip.EPI_GELU, aux=aux, cfg=cfg,
             compute="bf16")
    close(aux, pre, rtol=2e-5, name="bf16 gelu-pre")
    close(out, F.gelu(pre), rtol=1e-4, name="bf16 gelu")


def test_gemm_splitk_with_tanh_epilogues(hip):
    """Few output tiles + long K (the prompt generator's shapes): the split-K path applies bias+tanh / dtanh in
    the ordered slab reduction."""
    M, N, K = 288, 800, 3840
    x, w, b = rnd(M, K, seed=1, scale=0.1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3)
    out = torch.empty(M, N, device=DEV)
    hip.gemm(x.to(DEV), hip.KC, w.to(DEV), hip.KC, out, M, N, K, bias=b.to(DEV), epi=hip.EPI_TANH, allow_split=True,
             splits=4)
    close(out, torch.tanh(F.linear(x.double(), w.double(), b.double())), rtol=3e-4, name="split tanh")
    t = torch.tanh(rnd(M, N, seed=5))
    hip.gemm(x.to(DEV), hip.KC, w.to(DEV), hip.KC, out, M, N, K, epi=hip.EPI_DTANH, aux=t.to(DEV), allow_split=True,
             splits=3)
    close(out, F.linear(x.double(), w.double()) * (1 - t.double() ** 2), rtol=3e-4, name="split dtanh")
    out2 = torch.empty(48, 288, device=DEV).t()  # non-contiguous guard is the wrapper's job: use contiguous
    lg = torch.empty(288, 48, device=DEV)
    xs, ws, bs = rnd(288, 6144, seed=7, scale=0.1), rnd(48, 6144, seed=8, scale=0.1), rnd(48, seed=9)
    hip.linear_fwd(xs.to(DEV), ws.to(DEV), bs.to(DEV), lg)
    close(lg, F.linear(xs.double(), ws.double(), bs.double()), rtol=3e-4, name="auto split skinny")


@pytest.mark.parametrize("B,S,Pn", [(32, 128, 36), (8, 96, 4), (5, 64, 0)])
def test_weight_gradient_over_listed_k_tiles(hip, B, S, Pn):
    """mtvaf_build_ktiles + mtvaf_gemm_f32_ktiles: dW = dY^T . X over the 32-row k-tiles that hold an unmasked token, dY exactly
    zero at masked rows -- against the full fp64 product; every DMA tile, forced split counts (more splits than listed
    tiles included), an EMPTY-but-for-one list, accumulate; a plan that cannot use the list gives the same result."""
    T, Mtok = Pn + S, B * S
    gnr = torch.Generator().manual_seed(7)
    lens = [S] + [int(x) for x in torch.randint(1, S + 1, (B - 1,), generator=gnr)]
    mask = torch.zeros(B, T)
    for b, Lb in enumerate(lens):
        mask[b, : Pn + Lb] = 1
    mask[1, Pn + 3] = 0  # (a hole)
    addmask = ((1 - mask) * -10000.0).to(DEV)
    valid = mask[:, Pn:].reshape(-1).bool()
    NO, KI = 768, 384
    dy = rnd(Mtok, NO, seed=1) * valid[:, None]
    x = rnd(Mtok, KI, seed=2)
    ref = dy.double().t() @ x.double()
    klist, kcnt = hip.build_ktiles(addmask, Pn, S)
    n = int(kcnt.item())
    tiles = sorted(set(int(r) // 32 for r in torch.nonzero(valid).flatten()))
    assert klist[:n].tolist() == tiles
    out = torch.empty(NO, KI, device=DEV)
    dyd, xd = dy.to(DEV), x.to(DEV)
    for cfg in (-1, 9, 12, 14, 10, 6):  # (6: a register-staged plan -- ignores the list)
        for sp in (-1, 1, 3, 16):
            out.fill_(float("nan"))
            hip.gemm_ktiles(dyd, xd, out, NO, KI, Mtok, klist, kcnt, cfg=cfg, splits=sp)
            close(out, ref, rtol=2e-5, name=f"dW over listed k-tiles, cfg {cfg} splits {sp}")
    acc0 = rnd(NO, KI, seed=3)
    out.copy_(acc0)
    hip.gemm_ktiles(dyd, xd, out, NO, KI, Mtok, klist, kcnt, accumulate=True)
    close(out, ref + acc0.double(), rtol=2e-5, name="accumulate")
    one = torch.tensor([tiles[-1]], dtype=torch.int32, device=DEV)
    cnt1 = torch.ones(1, dtype=torch.int32, device=DEV)
    dy1 = torch.zeros_like(dy)
    dy1[tiles[-1] * 32:(tiles[-1] + 1) * 32] = dy[tiles[-1] * 32:(tiles[-1] + 1) * 32]
    hip.gemm_ktiles(dy1.to(DEV), xd, out, NO, KI, Mtok, one, cnt1, splits=4)
    close(out, dy1.double().t() @ x.double(), rtol=2e-5, name="one listed tile, four splits")


@pytest.mark.parametrize("M,N,K", [(128, 96, 64), (256, 384, 192), (512, 768, 768), (384, 128, 4096), (128, 128, 64),
                                   (512, 192, 128)])
def test_gemm_bf16_operands(hip, M, N, K):
    """bf16-OPERAND kernel (gemm_bf16x.hip): exact model = fp64 products of the bf16-rounded operands.  All three operand
    layout pairs of the path (forward KCxKC, dX KCxKM, dW KMxKM: the SAME row-major tensors read in both roles through
    the transposing LDS read), both ring depths, forced tiles, every epilogue, fp32 / bf16 results, per-tile column
    sums, accumulate, split-K; the cast kernel (row-major + transposed copies) is exact."""
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    xh, xt = torch.empty(M, K, dtype=torch.bfloat16, device=DEV), torch.empty(K, M, dtype=torch.bfloat16, device=DEV)
    hip.cast_bf16(x.to(DEV), out=xh, out_t=xt)
    assert torch.equal(xh.cpu(), x.to(torch.bfloat16)) and torch.equal(xt.cpu(), x.t().contiguous().to(torch.bfloat16))
    wh = w.to(torch.bfloat16).to(DEV)
    wt = wh.t().contiguous()  # [K, N]: the KM image of the same matrix
    ref = xh.double().cpu() @ wh.double().cpu().t()
    out = torch.empty(M, N, device=DEV)
    bd = b.to(DEV)
    layouts = [("KCxKC", xh, hip.KC, wh, hip.KC)]
    layouts += [("KCxKM", xh, hip.KC, wt, hip.KM), ("KMxKM", xt, hip.KM, wt, hip.KM)]
    for name, a_, la, b_, lb in layouts:
        for stages in (2, 3):
            for tile in ([0] + ([1] if N % 96 == 0 else []) + ([2] if N % 128 == 0 else []) +
                         ([3] if (N % 128 == 0 and M % 256 == 0 and la == hip.KC) else []) +
                         ([4] if (N % 192 == 0 and M % 256 == 0 and la == hip.KC and stages == 2) else [])):
                out.fill_(float("nan"))
                hip.gemm_bf16x(a_, la, b_, lb, M, N, K, out32=out, bias=bd, tile=tile, stages=stages)
                close(out, ref + b.double(), rtol=2e-5, name=f"{name} bias tile {tile} stages {stages}")
    a_, la, b_, lb = layouts[-1][1:]
    # bf16 + fp32 results together; bf16 result = RNE of the fp32 one
    o16 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    hip.gemm_bf16x(a_, la, b_, lb, M, N, K, out32=out, out16=o16, bias=bd)
    assert torch.equal(o16, out.to(torch.bfloat16))
    hip.gemm_bf16x(a_, la, b_, lb, M, N, K, out16=o16.zero_(), bias=bd)
    assert torch.equal(o16, out.to(torch.bfloat16))
    # GELU: pre-activation saved as bf16, activation evaluated on the saved value
    pre16 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    hip.gemm_bf16x(xh, hip.KC, wh, hip.KC, M, N, K, out32=out, out16=o16, bias=bd, epi=hip.EPI_GELU, aux16=pre16)
    pre_ref = (ref + b.double()).float().to(torch.bfloat16)
    mism = (pre16.cpu() != pre_ref)
    assert mism.float().mean() < 2e-3  # fp32-sum rounding at a bf16 tie flips a few
    close(out, F.gelu(pre16.double().cpu()), rtol=2e-5, atol=2e-6 * float(ref.abs().max()), name="gelu")
    assert torch.equal(o16, out.to(torch.bfloat16))
    # dGELU with the bf16 pre-activation + per-tile column sums
    pre = rnd(M, N, seed=5).to(torch.bfloat16)
    pd = pre.double().requires_grad_(True)
    F.gelu(pd).sum().backward()
    for tile in ([0] if N % 128 == 0 else []) + ([4] if (N % 192 == 0 and M % 256 == 0) else []):
        part = torch.full((M // 128, N), float("nan"), device=DEV)
        hip.gemm_bf16x(xh, hip.KC, wt, hip.KM, M, N, K, out32=out, out16=o16, epi=hip.EPI_DGELU, aux16=pre.to(DEV), colpart=part,
                       tile=tile)
        close(out, ref * pd.grad, rtol=3e-5, atol=3e-6 * float(ref.abs().max()), name=f"dgelu tile {tile}")
        assert torch.equal(o16, out.to(torch.bfloat16))
        cs = torch.empty(N, device=DEV)
        hip.colsum_small(part, cs)
        close(cs, out.double().sum(0), rtol=1e-5, atol=1e-5 * float(out.abs().sum(0).max()), name="colsum")
        hip.colsum_small(part, cs, accumulate=True)
        close(cs, 2 * out.double().sum(0), rtol=1e-5, atol=2e-5 * float(out.abs().sum(0).max()), name="colsum acc")
    if N % 192 == 0 and M % 256 == 0:  # the 256x192 tile with the GELU epilogue (two 128-row epilogue passes)
        hip.gemm_bf16x(xh, hip.KC, wh, hip.KC, M, N, K, out32=out, out16=o16, bias=bd, epi=hip.EPI_GELU, aux16=pre16, tile=4)
        assert (pre16.cpu() != pre_ref).float().mean() < 2e-3
        close(out, F.gelu(pre16.double().cpu()), rtol=2e-5, atol=2e-6 * float(ref.abs().max()), name="gelu tile 4")
        assert torch.equal(o16, out.to(torch.bfloat16))
    acc0 = rnd(M, N, seed=6)
    out.copy_(acc0)
    hip.gemm_bf16x(a_, la, b_, lb, M, N, K, out32=out, accumulate=True)
    close(out, ref + acc0.double(), rtol=2e-5, name="accumulate")
    if K >= 512:
        for name, a2, la2, b2, lb2 in layouts:
            for sp in (2, 3):
                hip.gemm_bf16x(a2, la2, b2, lb2, M, N, K, out32=out, allow_split=True, splits=sp)
                close(out, ref, rtol=2e-5, name=f"{name} split {sp}")
    with pytest.raises(RuntimeError):
        hip.gemm_bf16x(xh[:100], hip.KC, wh, hip.KC, 100, N, K, out32=out[:100])  # ragged M: no fallback inside the library


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (256, 256, 128), (512, 256, 192), (256, 768, 768), (768, 512, 3072),
                                   (1024, 768, 4096)])
def test_gemm_bf16_eight_phase_tile(hip, M, N, K):
    """The 256x256 eight-wave / eight-phase kernel (gemm_bf16p.hip, tile 5) against the exact model (fp64 products of the
    bf16-rounded operands): all three operand-layout pairs, one / two / three / many k-tiles per block (the pipeline's
    prologue and drain paths), split-K (1 .. 6 k-tiles per split), every epilogue, fp32 / bf16 results, column sums,
    accumulate.  Operands differ per row AND per column (random): a swapped or transposed fragment map cannot pass."""
    x, w, b = rnd(M, K, seed=11), rnd(N, K, seed=12), rnd(N, seed=13)
    xh, wh = x.to(torch.bfloat16).to(DEV), w.to(torch.bfloat16).to(DEV)
    xt, wt = xh.t().contiguous(), wh.t().contiguous()
    ref = xh.double().cpu() @ wh.double().cpu().t()
    out = torch.empty(M, N, device=DEV)
    bd = b.to(DEV)
    layouts = [("KCxKC", xh, hip.KC, wh, hip.KC), ("KCxKM", xh, hip.KC, wt, hip.KM), ("KMxKM", xt, hip.KM, wt, hip.KM)]
    for name, a_, la, b_, lb in layouts:
        out.fill_(float("nan"))
        hip.gemm_bf16x(a_, la, b_, lb, M, N, K, out32=out, bias=bd, tile=5)
        close(out, ref + b.double(), rtol=2e-5, name=f"{name} bias")
        for _ in range(3):  # the same launch again: a race between DMA and reads would come and go
            o2 = torch.full_like(out, float("nan"))
            hip.gemm_bf16x(a_, la, b_, lb, M, N, K, out32=o2, bias=bd, tile=5)
            assert torch.equal(o2, out), f"{name}: repeated launch differs"
        for sp in (2, 3, 8):
            if K // 64 >= sp:
                out.fill_(float("nan"))
                hip.gemm_bf16x(a_, la, b_, lb, M, N, K, out32=out, allow_split=True, splits=sp, tile=5)
                close(out, ref, rtol=2e-5, name=f"{name} split {sp}")
    o16 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    hip.gemm_bf16x(xt, hip.KM, wt, hip.KM, M, N, K, out32=out, out16=o16, bias=bd, tile=5)
    assert torch.equal(o16, out.to(torch.bfloat16))
    hip.gemm_bf16x(xh, hip.KC, wt, hip.KM, M, N, K, out16=o16.zero_(), bias=bd, tile=5)
    assert torch.equal(o16, out.to(torch.bfloat16))
    pre16 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    hip.gemm_bf16x(xh, hip.KC, wh, hip.KC, M, N, K, out32=out, out16=o16, bias=bd, epi=hip.EPI_GELU, aux16=pre16, tile=5)
    pre_ref = (ref + b.double()).float().to(torch.bfloat16)
    assert (pre16.cpu() != pre_ref).float().mean() < 2e-3  # fp32-sum rounding at a bf16 tie flips a few
    close(out, F.gelu(pre16.double().cpu()), rtol=2e-5, atol=2e-6 * float(ref.abs().max()), name="gelu")
    assert torch.equal(o16, out.to(torch.bfloat16))
    pre = rnd(M, N, seed=15).to(torch.bfloat16)
    pd = pre.double().requires_grad_(True)
    F.gelu(pd).sum().backward()
    part = torch.full((M // 128, N), float("nan"), device=DEV)
    hip.gemm_bf16x(xh, hip.KC, wt, hip.KM, M, N, K, out32=out, out16=o16, epi=hip.EPI_DGELU, aux16=pre.to(DEV), colpart=part, tile=5)
    close(out, ref * pd.grad, rtol=3e-5, atol=3e-6 * float(ref.abs().max()), name="dgelu")
    assert torch.equal(o16, out.to(torch.bfloat16))
    cs = torch.empty(N, device=DEV)
    hip.colsum_small(part, cs)
    close(cs, out.double().sum(0), rtol=1e-5, atol=1e-5 * float(out.abs().sum(0).max()), name="colsum")
    acc0 = rnd(M, N, seed=16)
    out.copy_(acc0)
    hip.gemm_bf16x(xt, hip.KM, wt, hip.KM, M, N, K, out32=out, accumulate=True, tile=5)
    close(out, ref + acc0.double(), rtol=2e-5, name="accumulate")
    # strided operands (a column block of a wider tensor, as the QKV / FFN buffers are read): leading dimension != width
    if K >= 128:
        wide = torch.zeros(M, K + 64, dtype=torch.bfloat16, device=DEV)
        wide[:, 64:] = xh
        hip.gemm_bf16x(wide[:, 64:], hip.KC, wh, hip.KC, M, N, K, out32=out, tile=5)
        close(out, ref, rtol=2e-5, name="strided A")
    with pytest.raises(RuntimeError):
        hip.gemm_bf16x(xh, hip.KC, wh[:N - 128], hip.KC, M, N - 128, K, out32=out[:, :N - 128], tile=5)  # N % 256 != 0


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (256, 256, 16384), (512, 768, 1216), (768, 512, 3072), (2048, 2304, 768),
                                   (1024, 768, 4096)])
def test_gemm_bf16_stream_k(hip, M, N, K):
    """Stream-K form of the 256x256 kernel (tile 6): one block per CU walks an equal run of k-tile steps; tiles shared by
    several blocks are combined in the launch (contribution slabs behind an agent-scope release / acquire).  Shapes: fewer
    steps than CUs, ONE tile cut over all 256 blocks (255 contributions to one finisher), runs that start and end inside
    tiles, 3+ blocks per tile.  Against the exact model, all three layouts, every epilogue; repeated launches bit-identical
    (the combine order is the block order, not the arrival order); the flag words are zero after every launch."""
    assert hip.streamk_ensure(DEV)
    x, w, b = rnd(M, K, seed=21), rnd(N, K, seed=22), rnd(N, seed=23)
    xh, wh = x.to(torch.bfloat16).to(DEV), w.to(torch.bfloat16).to(DEV)
    xt, wt = xh.t().contiguous(), wh.t().contiguous()
    ref = xh.double().cpu() @ wh.double().cpu().t()
    out = torch.empty(M, N, device=DEV)
    bd = b.to(DEV)
    scratch = hip._sk_scratch[(0, hip._st())]
    for name, a_, la, b_, lb in [("KCxKC", xh, hip.KC, wh, hip.KC), ("KCxKM", xh, hip.KC, wt, hip.KM), ("KMxKM", xt, hip.KM, wt, hip.KM)]:
        out.fill_(float("nan"))
        hip.gemm_bf16x(a_, la, b_, lb, M, N, K, out32=out, bias=bd, tile=6)
        close(out, ref + b.double(), rtol=2e-5, name=f"{name} bias")
        for _ in range(3):
            o2 = torch.full_like(out, float("nan"))
            hip.gemm_bf16x(a_, la, b_, lb, M, N, K, out32=o2, bias=bd, tile=6)
            assert torch.equal(o2, out), f"{name}: repeated launch differs"
        torch.cuda.synchronize()
        assert not scratch[:4096].any(), "flag / error words must be zero between launches"
    o16 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    pre16 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    hip.gemm_bf16x(xh, hip.KC, wh, hip.KC, M, N, K, out32=out, out16=o16, bias=bd, epi=hip.EPI_GELU, aux16=pre16, tile=6)
    pre_ref = (ref + b.double()).float().to(torch.bfloat16)
    assert (pre16.cpu() != pre_ref).float().mean() < 2e-3
    close(out, F.gelu(pre16.double().cpu()), rtol=2e-5, atol=2e-6 * float(ref.abs().max()), name="gelu")
    assert torch.equal(o16, out.to(torch.bfloat16))
    pre = rnd(M, N, seed=25).to(torch.bfloat16)
    pd = pre.double().requires_grad_(True)
    F.gelu(pd).sum().backward()
    part = torch.full((M // 128, N), float("nan"), device=DEV)
    hip.gemm_bf16x(xh, hip.KC, wt, hip.KM, M, N, K, out32=out, out16=o16, epi=hip.EPI_DGELU, aux16=pre.to(DEV), colpart=part, tile=6)
    close(out, ref * pd.grad, rtol=3e-5, atol=3e-6 * float(ref.abs().max()), name="dgelu")
    cs = torch.empty(N, device=DEV)
    hip.colsum_small(part, cs)
    close(cs, out.double().sum(0), rtol=1e-5, atol=1e-5 * float(out.abs().sum(0).max()), name="colsum")
    acc0 = rnd(M, N, seed=26)
    out.copy_(acc0)
    hip.gemm_bf16x(xt, hip.KM, wt, hip.KM, M, N, K, out32=out, accumulate=True, tile=6)
    close(out, ref + acc0.double(), rtol=2e-5, name="accumulate")
    torch.cuda.synchronize()
    assert not scratch[:4096].any()


def test_gemm_bf16_dw_group_stream_k(hip):
    """The four weight-gradient products of an encoder layer in ONE stream-K launch (mtvaf_gemm_bf16x_dw_group): shapes of
    BERT-base at 1024 and 4096 token rows, against the exact model; under load from another stream (uneven arrival of the
    blocks) and with the slabs' lines pre-read by the consumer's CU set (an L1-warm consumer), bit-identical every time."""
    assert hip.streamk_ensure(DEV)
    H, I = 768, 3072
    for T in (1024, 4096):
        dy, dy3, dyq = (rnd(T, n, seed=31 + n).to(torch.bfloat16).to(DEV) for n in (H, I, 3 * H))
        xs, x3 = rnd(T, H, seed=41).to(torch.bfloat16).to(DEV), rnd(T, I, seed=42).to(torch.bfloat16).to(DEV)
        items = [(dy, x3, torch.empty(H, I, device=DEV)), (dy3, xs, torch.empty(I, H, device=DEV)),
                 (dy, xs, torch.empty(H, H, device=DEV)), (dyq, xs, torch.empty(3 * H, H, device=DEV))]
        hip.gemm_bf16x_dw_group(items, T)
        first = []
        for a_, b_, o in items:
            close(o, a_.double().cpu().t() @ b_.double().cpu(), rtol=3e-5, name=f"dW {tuple(o.shape)} over {T} rows")
            first.append(o.clone())
        side = torch.cuda.Stream()
        big = torch.randn(4096, 4096, device=DEV)
        for rep in range(4):
            for _, _, o in items:
                o.fill_(float("nan"))
            with torch.cuda.stream(side):  # a competing kernel: the group's blocks arrive unevenly
                for _ in range(3):
                    big @ big
            _ = hip._sk_scratch[(0, hip._st())][4096:4096 + (64 << 20)].view(torch.float32).sum()  # pre-read the slabs
            hip.gemm_bf16x_dw_group(items, T)
            torch.cuda.synchronize()
            for (_, _, o), f in zip(items, first):
                assert torch.equal(o, f), f"repeat {rep}: stream-K result changed"
        assert hip.streamk_error(DEV) == 0
    with pytest.raises(RuntimeError):
        hip.gemm_bf16x_dw_group([(dy[:, :128], xs, torch.empty(128, H, device=DEV))], T)  # M % 256 != 0


@pytest.mark.parametrize("T", [256, 1024, 4096])
def test_gemm_f32_dw_group(hip, T):
    """The four fp32 weight-gradient products of an encoder layer in ONE launch (mtvaf_gemm_f32_dw_group): BERT-base shapes at
    256 (BASELINE configs[0]), 1024 and 4096 token rows against the fp64 products, planned and forced split counts, with a
    k-tile list (rows outside the list exactly zero), fewer than four products, and the documented shape errors."""
    H, I = 768, 3072
    valid = torch.ones(T, dtype=torch.bool)
    valid[T // 2 + 5: T // 2 + 5 + T // 4] = False  # (a masked run that empties whole 32-row k-tiles)
    dy, dy3, dyq = (rnd(T, n, seed=31 + n) for n in (H, I, 3 * H))
    xs, x3 = rnd(T, H, seed=41), rnd(T, I, seed=42)
    for use_list in (False, True):
        if use_list:
            dy, dy3, dyq = (t * valid[:, None] for t in (dy, dy3, dyq))
            tiles = sorted(set(int(r) // 32 for r in torch.nonzero(valid).flatten()))
            assert len(tiles) < T // 32
            kt = (torch.tensor(tiles, dtype=torch.int32, device=DEV), torch.tensor([len(tiles)], dtype=torch.int32, device=DEV))
        else:
            kt = None
        d = [t.to(DEV) for t in (dy, dy3, dyq, xs, x3)]
        items = [(d[0], d[4], torch.empty(H, I, device=DEV)), (d[1], d[3], torch.empty(I, H, device=DEV)),
                 (d[0], d[3], torch.empty(H, H, device=DEV)), (d[2], d[3], torch.empty(3 * H, H, device=DEV))]
        refs = [a_.double().cpu().t() @ b_.double().cpu() for a_, b_, _ in items]
        for sp in (-1, 1, 2, 3):
            for _, _, o in items:
                o.fill_(float("nan"))
            hip.gemm_f32_dw_group(items, T, ktiles=kt, splits=sp)
            for (_, _, o), r in zip(items, refs):
                close(o, r, rtol=2e-5, name=f"dW {tuple(o.shape)} over {T} rows, splits {sp}, list {use_list}")
        first = [o.clone() for _, _, o in items]
        hip.gemm_f32_dw_group(items, T, ktiles=kt, splits=3)
        for (_, _, o), f in zip(items, first):
            assert torch.equal(o, f)  # ordered slab reduction: the same bits every time
        # the bias gradients that go with the weight gradients (column sums of the dY operands): out of the same call
        dbs = [None, torch.full((I,), float("nan"), device=DEV), None, torch.full((3 * H,), float("nan"), device=DEV)]
        for sp in (-1, 2):
            for _, _, o in items:
                o.fill_(float("nan"))
            for t in (dbs[1], dbs[3]):
                t.fill_(float("nan"))
            hip.gemm_f32_dw_group(items, T, ktiles=kt, splits=sp, dbias=dbs)
            for (_, _, o), r in zip(items, refs):
                close(o, r, rtol=2e-5, name=f"dW {tuple(o.shape)} with bias sums, splits {sp}, list {use_list}")
            close(dbs[1], d[1].double().cpu().sum(0), rtol=2e-5, name=f"bias gradient of product 1, splits {sp}, list {use_list}")
            close(dbs[3], d[2].double().cpu().sum(0), rtol=2e-5, name=f"bias gradient of product 3, splits {sp}, list {use_list}")
        again = [t.clone() for t in (dbs[1], dbs[3])]
        hip.gemm_f32_dw_group(items, T, ktiles=kt, splits=2, dbias=dbs)
        assert torch.equal(dbs[1], again[0]) and torch.equal(dbs[3], again[1])  # fixed summation order
        two = [(items[2][0], items[2][1], torch.empty(H, H, device=DEV)), (items[1][0], items[1][1], torch.empty(I, H, device=DEV))]
        hip.gemm_f32_dw_group(two, T, ktiles=kt)
        close(two[0][2], refs[2], rtol=2e-5, name="two products: first")
        close(two[1][2], refs[1], rtol=2e-5, name="two products: second")
    # which kernel: long reductions under the split arithmetic go to the split kernel's GROUP form, unsplit by plan at these
    # 432 tiles; few-token layers and the fp32 pipe keep the LDS-DMA ring -- and both produce the products above
    was = hip.f32_split()
    try:
        for mode in (True, False):
            hip.f32_split(mode)
            hip.prof_start(8)
            hip.gemm_f32_dw_group(items, T, ktiles=kt)
            recs = hip.prof_stop(8)
            assert len(recs) == 1
            on_split = mode and T > 1024
            assert recs[0][0]["cfg"] == (1225 if on_split else 1012), recs
            if on_split:
                assert recs[0][0]["splits"] == 1
            for (_, _, o), r in zip(items, refs):
                close(o, r, rtol=2e-5, name=f"dW {tuple(o.shape)} over {T} rows, split arithmetic {mode}")
            assert hip.dw_group_wanted(T, H, I) == (T <= 1024 or mode)
    finally:
        hip.f32_split(was)
    with pytest.raises(RuntimeError):
        hip.gemm_f32_dw_group([(d[0][:, :64], d[3], torch.empty(64, H, device=DEV))], T)  # M % 128 != 0
    with pytest.raises(RuntimeError):
        hip.gemm_f32_dw_group([(d[0], d[3][:, :100], torch.empty(H, 100, device=DEV))], T)  # N % 96 != 0


def _x3_operands(M, N, K, la, lb, seed, spread=0):
    """fp32 operands in the layouts (la, lb); spread > 0 scales rows of A / columns of B by 2^[-spread, spread] (every bf16 plane
    of the split is exercised across the exponent range); -> (a, b, fp64 product, fp64 |A|.|B|)"""
    g = torch.Generator().manual_seed(seed)
    A, B = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g)
    if spread:
        A = A * torch.exp2(torch.randint(-spread, spread + 1, (M, 1), generator=g).float())
        B = B * torch.exp2(torch.randint(-spread, spread + 1, (1, N), generator=g).float())
    a = (A if la == 0 else A.t().contiguous()).to(DEV)
    b = (B.t().contiguous() if lb == 0 else B).to(DEV)
    return a, b, A.double() @ B.double(), A.double().abs() @ B.double().abs()


@pytest.mark.parametrize("la,lb", [(0, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("M,N,K,spread,tile", [(512, 768, 768, 0, 5), (256, 384, 3072, 12, 5), (128, 128, 32, 30, 5), (192, 320, 96, 6, 3),
                                               (1024, 1536, 256, 3, 5), (512, 768, 768, 0, 6), (256, 384, 3072, 12, 6),
                                               (128, 96, 32, 30, 6), (1024, 1536, 256, 3, 6),
                                               (1152, 800, 384, 3, 5), (132, 388, 96, 12, 5),  # (tiles that hang over the result)
                                               (512, 768, 768, 0, 4), (256, 320, 3072, 12, 4), (128, 64, 32, 30, 4),   # 128x64 (round 5)
                                               (2432, 768, 256, 3, 4)])
def test_gemm_f32_split_accuracy(hip, la, lb, M, N, K, spread, tile):
    """fp32 GEMM by three-way bf16 operand splitting (mtvaf_gemm_f32x3, csrc/gemm_f32x3.hip) is an fp32 GEMM: against the fp64
    product its error is bounded element-wise by a few fp32 roundings of |A|.|B| (the six partial products are exact, the
    dropped terms are below 2^-26 |a||b|) and is not larger than the fp32 MFMA pipe's on the same operands -- all three
    operand layouts, every tile (cfg 5 = 128x128, 6 = 128x96 and 4 = 128x64 wave-specialised, 3 = 64x64), operands spread over
    2^+-30."""
    a, b, ref, mag = _x3_operands(M, N, K, la, lb, seed=M + N + K + 7 * la + lb, spread=spread)
    o_nat, o_spl = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    hip.gemm(a, la, b, lb, o_nat, M, N, K, compute="fp32")
    # (the tile is forced: left to itself the library keeps products of fewer than 96 tiles of 128 x 128 on the fp32 pipe)
    hip.prof_start(4)
    hip.gemm(a, la, b, lb, o_spl, M, N, K, compute="fp32x3", cfg=tile)
    recs = hip.prof_stop(4)
    assert recs[0][0]["cfg"] == {5: 225, 6: 226, 4: 224, 3: 203}[tile], recs  # (the kernel that was asked for really ran)
    e_nat = (o_nat.double().cpu() - ref).abs() / mag
    e_spl = (o_spl.double().cpu() - ref).abs() / mag
    # element-wise bound: accumulation of K terms in fp32 (worst case K 2^-24, in practice ~sqrt(K)) + the 2^-26 split remainder
    assert float(e_spl.max()) <= 2.0 ** -24 * (4 + K ** 0.5), (float(e_spl.max()), float(e_nat.max()))
    assert float(e_spl.max()) <= 1.25 * float(e_nat.max()) + 2.0 ** -25, (float(e_spl.max()), float(e_nat.max()))
    assert float(e_spl.pow(2).mean().sqrt()) <= 1.1 * float(e_nat.pow(2).mean().sqrt()) + 2.0 ** -27


@pytest.mark.parametrize("la,lb,M,N,K", [(0, 0, 1152, 800, 3840), (0, 1, 1152, 800, 6144), (1, 1, 6144, 800, 1152), (1, 1, 800, 3840, 1152),
                                          (0, 0, 260, 132, 64)])
def test_gemm_f32_split_ragged_tiles(hip, la, lb, M, N, K):
    """128 x 128 tiles of the wave-specialised split kernel that hang over the result (M, N multiples of 4: the prompt
    generator's 800-wide hidden layer, `models/bert_model.py:63-111`): clamped operand loads, guarded stores.  The library picks
    the kernel by itself for the four prompt-generator products; every epilogue and a split reduction give the values of the
    fp64 product; nothing is written outside the result (guard bands around a strided output stay NaN)."""
    a, b, ref, _ = _x3_operands(M, N, K, la, lb, seed=3 * M + N + K)
    bias = rnd(N, seed=5).to(DEV)
    was = hip.f32_split()
    try:
        hip.f32_split(True)
        big = torch.full((M + 8, N + 8), float("nan"), device=DEV)  # the result inside a wider, taller buffer (ldc = N + 8)
        rows = big[4:4 + M]
        hip.prof_start(4)
        hip.gemm(a, la, b, lb, rows, M, N, K, bias=bias, allow_split=True, ldc=N + 8)
        recs = hip.prof_stop(4)
        if M * N >= 96 * 128 * 128:
            assert recs[0][0]["cfg"] == 225, recs  # (chosen by the library, not forced)
        close(rows[:, :N], ref + bias.double().cpu(), rtol=3e-6, name="ragged, bias")
        guard = big.clone()
        guard[4:4 + M, :N] = float("nan")
        assert bool(torch.isnan(guard).all())  # nothing outside the result was written
        for sp in (1, 3):
            aux = torch.full((M, N), float("nan"), device=DEV)
            out2 = torch.full((M, N), float("nan"), device=DEV)
            hip.gemm(a, la, b, lb, out2, M, N, K, bias=bias, epi=hip.EPI_GELU, aux=aux, allow_split=True, splits=sp, compute="fp32x3", cfg=5)
            close(aux, ref + bias.double().cpu(), rtol=3e-6, name=f"ragged, saved pre-activation, splits {sp}")
            close(out2, torch.nn.functional.gelu(ref + bias.double().cpu()), rtol=3e-6, name=f"ragged, GELU, splits {sp}")
        acc0 = rnd(M, N, seed=6).to(DEV)
        out3 = acc0.clone()
        hip.gemm(a, la, b, lb, out3, M, N, K, accumulate=True, compute="fp32x3", cfg=5)
        close(out3, ref + acc0.double().cpu(), rtol=3e-6, name="ragged, accumulate")
    finally:
        hip.f32_split(was)


@pytest.mark.parametrize("la,lb", [(0, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("case", X3_CASES)
def test_gemm_f32_split_adversarial(hip, la, lb, case):
    """The split GEMM on operands chosen against it (tests/x3_cases.py; measured table: tools/f32x3_adversarial.py): exponents
    that vary per ELEMENT inside a row (not a per-row scale), rows whose sum cancels to 2^-12 of its terms, operands near
    2^+-120 / 2^+-58 with in-range products, |a| up to 2^127 (1 - 2^-9), values with all 24 significant bits set.  Bound as
    test_gemm_f32_split_accuracy: element-wise against the fp64 product, normalised by |A|.|B|, within a few fp32 roundings
    and not worse than the fp32 MFMA pipe on the same operands.  ONE documented limit: operands below 2^-109 put their second
    / third planes into bf16's subnormal range (8 exponent bits, as fp32, with 16 fewer mantissa bits per plane to spend
    below 2^-126), so the result keeps an ABSOLUTE error of at most K 2^-133 max|other operand| -- for `tiny_times_one` /
    `huge_times_tiny` that floor is what is asserted (values of 1e-36 against O(1) weights: far below anything the step reads;
    the fp32 pipe keeps full relative accuracy there)."""
    M, N, K = 256, 384, 512
    A, B = x3_operands(case, M, N, K, seed=5 + la + 2 * lb)
    ref, mag = A.double() @ B.double(), A.double().abs() @ B.double().abs()
    a = (A if la == 0 else A.t().contiguous()).to(DEV)
    b = (B.t().contiguous() if lb == 0 else B).to(DEV)
    o_nat, o_spl = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    was = hip.f32_split()
    try:
        hip.f32_split(False)
        hip.gemm(a, la, b, lb, o_nat, M, N, K)
    finally:
        hip.f32_split(was)
    hip.gemm(a, la, b, lb, o_spl, M, N, K, compute="fp32x3", cfg=5)
    assert bool(torch.isfinite(o_spl).all()), "the first plane must not round to infinity / NaN"
    e_nat = (o_nat.double().cpu() - ref).abs() / mag
    e_spl = (o_spl.double().cpu() - ref).abs()
    bound = 2.0 ** -24 * (4 + K ** 0.5) * mag
    if case in ("tiny_times_one", "huge_times_tiny"):
        tiny, other = (A, B) if case == "tiny_times_one" else (B, A)
        assert float(tiny.abs().max()) < 2.0 ** -109
        bound = bound + K * 2.0 ** -133 * float(other.abs().max())
        assert bool((e_spl <= bound).all()), float((e_spl / mag).max())
        return
    assert bool((e_spl <= bound).all()), (float((e_spl / mag).max()), float(e_nat.max()))
    e_spl = e_spl / mag
    assert float(e_spl.max()) <= 1.5 * float(e_nat.max()) + 2.0 ** -25, (float(e_spl.max()), float(e_nat.max()))
    assert float(e_spl.pow(2).mean().sqrt()) <= 1.25 * float(e_nat.pow(2).mean().sqrt()) + 2.0 ** -26


def test_gemm_f32_split_tile64_epilogues_splitk_ktiles_and_planner(hip):
    """The 128 x 64 layout of the wave-specialised split kernel (round 5; the N = 768 products of a packed batch): every epilogue,
    split-K, a k-tile list, bit-identical results to the 128 x 128 layout (same planes, same k order, same product sequence per
    element), and the planner takes it by itself exactly where it saves a round of the 256 CUs (2432 packed rows x 768), not at
    4096 rows."""
    M, N, K = 256, 320, 512
    x, w, bias = rnd(M, K, seed=1).to(DEV), rnd(N, K, seed=2).to(DEV), rnd(N, seed=3).to(DEV)
    ref = x.double().cpu() @ w.double().cpu().t() + bias.double().cpu()
    out, aux = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    hip.gemm(x, 0, w, 0, out, M, N, K, bias=bias, epi=hip.EPI_GELU, aux=aux, compute="fp32x3", cfg=4)
    close(aux, ref, rtol=2e-6, name="saved pre-activation")
    close(out, torch.nn.functional.gelu(ref), rtol=2e-6, name="GELU")
    dy, pre = rnd(M, N, seed=4).to(DEV), rnd(M, K, seed=5).to(DEV)
    dx = torch.empty(M, K, device=DEV)
    hip.gemm(dy, 0, w, 1, dx, M, K, N, epi=hip.EPI_DGELU, aux=pre, compute="fp32x3", cfg=4)
    p64 = pre.double().cpu()
    gp = 0.5 * (1 + torch.erf(p64 / 2 ** 0.5)) + p64 * torch.exp(-p64 * p64 / 2) / (2 * torch.pi) ** 0.5
    close(dx, (dy.double().cpu() @ w.double().cpu()) * gp, rtol=3e-6, name="GELU'")
    acc0 = rnd(M, N, seed=6)
    out.copy_(acc0)
    hip.gemm(x, 0, w, 0, out, M, N, K, accumulate=True, compute="fp32x3", cfg=4)
    close(out, ref - bias.double().cpu() + acc0.double(), rtol=2e-6, name="accumulate")
    for sp in (2, 4):
        out.fill_(float("nan"))
        hip.gemm(x, 0, w, 0, out, M, N, K, bias=bias, allow_split=True, splits=sp, compute="fp32x3", cfg=4)
        close(out, ref, rtol=2e-6, name=f"split-K {sp}")
    # the same bits as the 128 x 128 layout (N a multiple of both)
    M2, N2, K2 = 256, 384, 768
    for la, lb in ((0, 0), (0, 1), (1, 1)):
        a, b, _, _ = _x3_operands(M2, N2, K2, la, lb, seed=11 + la + lb)
        o4, o5 = torch.empty(M2, N2, device=DEV), torch.empty(M2, N2, device=DEV)
        hip.gemm(a, la, b, lb, o4, M2, N2, K2, compute="fp32x3", cfg=4)
        hip.gemm(a, la, b, lb, o5, M2, N2, K2, compute="fp32x3", cfg=5)
        assert torch.equal(o4, o5), (la, lb)
    # weight gradient over a k-tile list
    T, NO, KI = 1024, 256, 192
    valid = torch.ones(T, dtype=torch.bool)
    valid[300:700] = False
    dyw = (rnd(T, NO, seed=7) * valid[:, None]).to(DEV)
    xw = rnd(T, KI, seed=8).to(DEV)
    tiles = sorted(set(int(r) // 32 for r in torch.nonzero(valid).flatten()))
    kl, kc = torch.tensor(tiles, dtype=torch.int32, device=DEV), torch.tensor([len(tiles)], dtype=torch.int32, device=DEV)
    dw = torch.empty(NO, KI, device=DEV)
    refw = dyw.double().cpu().t() @ xw.double().cpu()
    was = hip.f32_split()
    try:
        assert hip.f32_split(True) is True
        for sp in (1, 3):
            hip.gemm_ktiles(dyw, xw, dw.fill_(float("nan")), NO, KI, T, kl, kc, cfg=4, splits=sp)
            close(dw, refw, rtol=3e-6, name=f"k-tile list, 128x64, splits {sp}")
        # the planner: a packed batch of 2432 rows takes 128x64 for N = 768 (228 tiles: one round), 4096 rows do not
        xs, ws = torch.randn(2432, 768, device=DEV), torch.randn(768, 768, device=DEV)
        o = torch.empty(2432, 768, device=DEV)
        hip.prof_start(8)
        hip.gemm(xs, 0, ws, 0, o, 2432, 768, 768)
        hip.gemm(xs, 0, ws, 1, o, 2432, 768, 768)
        xl, ol = torch.randn(4096, 768, device=DEV), torch.empty(4096, 768, device=DEV)
        hip.gemm(xl, 0, ws, 0, ol, 4096, 768, 768)
        recs = hip.prof_stop(8)
        assert [r[0]["cfg"] for r in recs] == [224, 224, 226], [r[0]["cfg"] for r in recs]
        close(o, xs.double().cpu() @ ws.double().cpu(), rtol=3e-6, name="planned 128x64 product")
    finally:
        hip.f32_split(was)


@pytest.mark.parametrize("M,N,K,spread", [(256, 384, 768, 0), (128, 128, 32, 12), (128, 256, 64, 20), (384, 128, 96, 3), (512, 768, 3072, 6),
                                          (640, 256, 64, 3), (896, 384, 32, 0)])  # (5 and 7 tile rows: a ragged last band of the tile walk)
def test_gemm_f32_presplit_planes_accuracy_and_epilogues(hip, M, N, K, spread):
    """mtvaf_f32_split_planes + mtvaf_gemm_f32p (csrc/gemm_f32p.hip, round 5: both operands as pre-split bf16 plane images,
    v_mfma_f32_16x16x32_bf16, nothing split inside the k-loop): the planes are those of the in-kernel split (x = p1 + p2 + p3 to
    2^-24 |x|, every plane a bf16), the product is an fp32 GEMM under the bound of `test_gemm_f32_split_accuracy` -- against the
    fp64 product and not worse than the fp32 MFMA pipe -- in both plane layouts (tile-blocked, natural), with 1, 2, 3 k-tiles
    (prologue / drain of the three-stage ring) and many, bias + GELU with the saved pre-activation, accumulate, and a
    deterministic 2-way split-K."""
    a, b, ref, mag = _x3_operands(M, N, K, 0, 0, seed=M + N + K, spread=spread)
    o_nat = torch.empty(M, N, device=DEV)
    hip.gemm(a, 0, b, 0, o_nat, M, N, K, compute="fp32")
    e_nat = (o_nat.double().cpu() - ref).abs() / mag
    for blocked in (True, False):
        pa, pb = hip.Planes(a, blocked), hip.Planes(b, blocked)
        # the image holds the three planes of every element: their sum is the element to 2^-24
        if blocked:
            img = pa.img.view(K // 32, 3, M, 32).float()
            back = img.sum(1).permute(1, 0, 2).reshape(M, K)
        else:
            back = pa.img.view(3, M, K).float().sum(0)
        assert float(((back.double() - a.double()).abs() / a.double().abs().clamp_min(1e-30)).max()) <= 2.0 ** -23
        out = torch.full((M, N), float("nan"), device=DEV)
        hip.gemm_planes(pa, pb, out)
        e = (out.double().cpu() - ref).abs() / mag
        assert float(e.max()) <= 2.0 ** -24 * (4 + K ** 0.5), (blocked, float(e.max()), float(e_nat.max()))
        assert float(e.max()) <= 1.25 * float(e_nat.max()) + 2.0 ** -25, (blocked, float(e.max()), float(e_nat.max()))
        assert float(e.pow(2).mean().sqrt()) <= 1.1 * float(e_nat.pow(2).mean().sqrt()) + 2.0 ** -27
    bias = rnd(N, seed=5).to(DEV)
    aux = torch.full((M, N), float("nan"), device=DEV)
    out = torch.full((M, N), float("nan"), device=DEV)
    hip.gemm_planes(pa, pb, out, bias=bias, epi=hip.EPI_GELU, aux=aux)
    scale = float(mag.max())
    close(aux, ref + bias.double().cpu(), rtol=3e-6, atol=3e-6 * scale, name="saved pre-activation")
    close(out, torch.nn.functional.gelu(ref + bias.double().cpu()), rtol=3e-6, atol=3e-6 * scale, name="GELU")
    acc0 = rnd(M, N, seed=6).to(DEV)
    out.copy_(acc0)
    hip.gemm_planes(pa, pb, out, accumulate=True)
    close(out, ref + acc0.double().cpu(), rtol=3e-6, atol=3e-6 * scale, name="accumulate")
    # the dX form: B [K, N] with the reduction index as its ROW (k-major: transposing fragment reads), natural plane image
    a2, b2, ref2, mag2 = _x3_operands(M, N, K, 0, 1, seed=M + N + K + 1, spread=spread)
    if N % 128 == 0:
        pa2, pb2 = hip.Planes(a2, True), hip.Planes(b2, False)
        o_km = torch.full((M, N), float("nan"), device=DEV)
        hip.gemm_planes(pa2, pb2, o_km, layout_b=hip.KM)
        e = (o_km.double().cpu() - ref2).abs() / mag2
        assert float(e.max()) <= 2.0 ** -24 * (4 + K ** 0.5), ("k-major B", float(e.max()))
    # the dW form: out[M, N] = A[K, M]^T . B[K, N], the reduction index is the ROW of both operands (natural plane images)
    a3, b3, ref3, mag3 = _x3_operands(M, N, K, 1, 1, seed=M + N + K + 2, spread=spread)
    if N % 128 == 0:
        pa3, pb3 = hip.Planes(a3, False), hip.Planes(b3, False)
        o_dw = torch.full((M, N), float("nan"), device=DEV)
        hip.gemm_planes(pa3, pb3, o_dw, layout_a=hip.KM, layout_b=hip.KM)
        e = (o_dw.double().cpu() - ref3).abs() / mag3
        assert float(e.max()) <= 2.0 ** -24 * (4 + K ** 0.5), ("k-major A and B", float(e.max()))
        # the TILE-BLOCKED images serve the k-major reads too (same planes at other addresses: the same bits), also for the dX form
        o_b = torch.full((M, N), float("nan"), device=DEV)
        hip.gemm_planes(hip.Planes(a3, True), hip.Planes(b3, True), o_b, layout_a=hip.KM, layout_b=hip.KM)
        assert torch.equal(o_b, o_dw), "k-major reads of tile-blocked images"
        o_b.fill_(float("nan"))
        hip.gemm_planes(pa2, hip.Planes(b2, True), o_b, layout_b=hip.KM)
        assert torch.equal(o_b, o_km), "k-major B from a tile-blocked image"
    if K >= 64:
        o1, o2 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
        hip.gemm_planes(pa, pb, o1, bias=bias, splits=2)
        hip.gemm_planes(pa, pb, o2, bias=bias, splits=2)
        close(o1, ref + bias.double().cpu(), rtol=3e-6, atol=3e-6 * scale, name="split-K 2")
        assert torch.equal(o1, o2), "split-K must be deterministic"


def test_gemm_f32_presplit_result_as_plane_image_and_column_partials(hip):
    """mtvaf_gemm_f32p_ep (round 5): the product leaves its epilogue as a tile-blocked plane image -- exactly the planes a split
    pass over the fp32 result would write (GELU forward with the saved pre-activation; GELU' backward with per-tile column sums: the
    FFN-1 bias gradient), with and without the fp32 copy."""
    M, N, K = 256, 384, 256
    x, w, bias = rnd(M, K, seed=51).to(DEV), rnd(N, K, seed=52).to(DEV), rnd(N, seed=53).to(DEV)
    pa, pb = hip.Planes(x, True), hip.Planes(w, True)
    out, pre = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    hip.gemm_planes(pa, pb, out, bias=bias, epi=hip.EPI_GELU, aux=pre)
    want = hip.Planes(out, True)
    for with_c in (True, False):
        o2, pre2 = torch.full((M, N), float("nan"), device=DEV), torch.full((M, N), float("nan"), device=DEV)
        got = hip.Planes(out, True, fill=False)
        got.img.fill_(float("nan"))
        hip.gemm_planes_ep(pa, pb, got, out=o2 if with_c else None, bias=bias, epi=hip.EPI_GELU, aux=pre2)
        assert torch.equal(got.img.view(torch.int16), want.img.view(torch.int16)), with_c
        assert torch.equal(pre2, pre)
        if with_c:
            assert torch.equal(o2, out)
    # the dX form with GELU' and the column partials
    dy, w2 = rnd(M, K, seed=54).to(DEV), rnd(K, N, seed=55).to(DEV)   # dX [M, N] = dy [M, K] . w2 [K, N]
    pd, pw = hip.Planes(dy, True), hip.Planes(w2, True)
    dx = torch.empty(M, N, device=DEV)
    hip.gemm_planes(pd, pw, dx, epi=hip.EPI_DGELU, aux=pre, layout_b=hip.KM)
    want = hip.Planes(dx, True)
    got = hip.Planes(dx, True, fill=False)
    part = torch.full((M // 128, N), float("nan"), device=DEV)
    hip.gemm_planes_ep(pd, pw, got, epi=hip.EPI_DGELU, aux=pre, colpart=part, layout_b=hip.KM)
    assert torch.equal(got.img.view(torch.int16), want.img.view(torch.int16))
    close(part.sum(0), dx.double().sum(0), rtol=1e-5, atol=1e-5 * float(dx.double().sum(0).abs().max()), name="column sums")
    close(part[1], dx[128:].double().sum(0), rtol=1e-5, atol=1e-5 * float(dx.double().sum(0).abs().max()), name="second tile row")


def test_packed_attention_writes_the_plane_images_of_its_results(hip):
    """Round 5: mtvaf_prefix_attn_varlen_fwd_planes / _bwd_planes also write the tile-blocked plane images of the context / of
    dQ | dK | dV (the bits of a split pass over the fp32 results, padding rows included) and leave the fp32 results unchanged."""
    L_ = hip.lib()
    B, S, Pn, NH, p = 5, 100, 36, 4, 0.1
    H = NH * 64
    lens = [S, 1, 53, 20, 77]
    Mv = sum(lens)
    Mp = (Mv + 127) // 128 * 128 + 128
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=DEV)
    qkv = torch.zeros(Mp, 3 * H, device=DEV)
    qkv[:Mv] = rnd(Mv, 3 * H, seed=71).to(DEV)
    pk, pv = rnd(B, Pn * H, seed=72).to(DEV), rnd(B, Pn * H, seed=73).to(DEV)
    dctx = torch.zeros(Mp, H, device=DEV)
    dctx[:Mv] = rnd(Mv, H, seed=74).to(DEV)
    c0, c1 = torch.full((Mp, H), float("nan"), device=DEV), torch.full((Mp, H), float("nan"), device=DEV)
    l0, l1 = torch.zeros(B, NH, S, device=DEV), torch.zeros(B, NH, S, device=DEV)
    hip.prefix_attn_varlen_fwd(qkv, pk, pv, cu, Mp - Mv, c0, l0, B, S, Pn, NH, p, 11, 5)
    img = hip.Planes(c0, True, fill=False)
    img.img.fill_(float("nan"))
    hip._ck(L_.mtvaf_prefix_attn_varlen_fwd_planes(hip._p(qkv), hip._p(pk), hip._p(pv), hip._p(cu), Mp - Mv, hip._p(c1), hip._p(l1), B, S, Pn, NH,
                                                   64, p, 11, 5, hip._p(img.img), Mp, hip._st()), "fwd planes")
    assert torch.equal(c0, c1) and torch.equal(l0, l1)
    assert torch.equal(img.img.view(torch.int16), hip.Planes(c0, True).img.view(torch.int16))
    d0, d1 = torch.full((Mp, 3 * H), float("nan"), device=DEV), torch.full((Mp, 3 * H), float("nan"), device=DEV)
    k0, v0, k1, v1 = (torch.zeros(B, Pn * H, device=DEV) for _ in range(4))
    de = torch.zeros(B, NH, S, device=DEV)
    hip.prefix_attn_varlen_bwd(dctx, qkv, pk, pv, cu, Mp - Mv, c0, l0, de, d0, k0, v0, B, S, Pn, NH, p, 11, 5)
    dimg = hip.Planes(d0, True, fill=False)
    dimg.img.fill_(float("nan"))
    hip._ck(L_.mtvaf_prefix_attn_varlen_bwd_planes(hip._p(dctx), hip._p(qkv), hip._p(pk), hip._p(pv), hip._p(cu), Mp - Mv, hip._p(c0), hip._p(l0),
                                                   hip._p(de), hip._p(d1), hip._p(k1), hip._p(v1), B, S, Pn, NH, 64, p, 11, 5, hip._p(dimg.img), Mp,
                                                   hip._st()), "bwd planes")
    assert torch.equal(d0, d1) and torch.equal(k0, k1) and torch.equal(v0, v1)
    assert torch.equal(dimg.img.view(torch.int16), hip.Planes(d0, True).img.view(torch.int16))


def test_layernorm_kernels_write_the_plane_images_of_their_outputs(hip):
    """Round 5: mtvaf_dropout_res_ln_fwd_planes / _bwd_rows_planes also write the tile-blocked plane image of their output -- the
    bits a split pass over the fp32 output would write -- and leave every fp32 result of the plain kernels unchanged (with and
    without the split-K slabs of the product in front; backward: no fp32 dx at all)."""
    L_ = hip.lib()
    M, H, ns = 384, 256, 2
    x, res = rnd(M, H, seed=61).to(DEV), rnd(M, H, seed=62).to(DEV)
    gamma, beta, bias = rnd(H, seed=63).to(DEV), rnd(H, seed=64).to(DEV), rnd(H, seed=65).to(DEV)
    slabs = rnd(ns, M, H, seed=66).to(DEV)
    for nslab in (0, ns):
        o0, mu0, rs0, xo0 = (torch.empty(M, H, device=DEV), torch.empty(M, device=DEV), torch.empty(M, device=DEV), torch.empty(M, H, device=DEV))
        o1, mu1, rs1, xo1 = (torch.empty(M, H, device=DEV), torch.empty(M, device=DEV), torch.empty(M, device=DEV), torch.empty(M, H, device=DEV))
        img = hip.Planes(o0, True, fill=False)
        img.img.fill_(float("nan"))
        if nslab:
            hip._ck(L_.mtvaf_dropout_res_ln_fwd_slabs(hip._p(slabs), nslab, hip._p(bias), hip._p(xo0), hip._p(res), hip._p(gamma), hip._p(beta),
                                                      hip._p(o0), hip._p(mu0), hip._p(rs0), M, H, 1e-12, 0.1, 77, 5, None, hip._st()), "ln")
        else:
            hip._ck(L_.mtvaf_dropout_res_ln_fwd(hip._p(x), hip._p(res), hip._p(gamma), hip._p(beta), hip._p(o0), hip._p(mu0), hip._p(rs0), M, H,
                                                1e-12, 0.1, 77, 5, None, hip._st()), "ln")
        hip._ck(L_.mtvaf_dropout_res_ln_fwd_planes(hip._p(slabs if nslab else x), nslab, hip._p(bias), hip._p(xo1), hip._p(res), hip._p(gamma),
                                                   hip._p(beta), hip._p(o1), hip._p(mu1), hip._p(rs1), M, H, 1e-12, 0.1, 77, 5, hip._p(img.img),
                                                   hip._st()), "ln planes")
        assert torch.equal(o0, o1) and torch.equal(mu0, mu1) and torch.equal(rs0, rs1)
        if nslab:
            assert torch.equal(xo0, xo1)
        assert torch.equal(img.img.view(torch.int16), hip.Planes(o0, True).img.view(torch.int16)), nslab
        # backward rows
        dout = rnd(M, H, seed=67).to(DEV)
        xin = xo0 if nslab else x
        nb = int(L_.mtvaf_ln_bwd_workspace_bytes(M, H))
        p0, p1 = torch.zeros(nb // 4, device=DEV), torch.zeros(nb // 4, device=DEV)
        dx0, dr0, dr1 = torch.empty(M, H, device=DEV), torch.empty(M, H, device=DEV), torch.empty(M, H, device=DEV)
        if nslab:
            hip._ck(L_.mtvaf_dropout_res_ln_bwd_rows_slabs(hip._p(dout), hip._p(slabs), nslab, hip._p(xin), hip._p(res), hip._p(gamma), hip._p(mu0),
                                                           hip._p(rs0), hip._p(dx0), hip._p(dr0), 0, M, H, 0.1, 77, 5, hip._p(p0), None, hip._st()), "b")
        else:
            hip._ck(L_.mtvaf_dropout_res_ln_bwd_rows(hip._p(dout), hip._p(xin), hip._p(res), hip._p(gamma), hip._p(mu0), hip._p(rs0), hip._p(dx0),
                                                     hip._p(dr0), 0, M, H, 0.1, 77, 5, hip._p(p0), None, hip._st()), "b")
        dimg = hip.Planes(dx0, True, fill=False)
        dimg.img.fill_(float("nan"))
        hip._ck(L_.mtvaf_dropout_res_ln_bwd_rows_planes(hip._p(dout), hip._p(slabs) if nslab else None, nslab, hip._p(xin), hip._p(res),
                                                        hip._p(gamma), hip._p(mu0), hip._p(rs0), None, hip._p(dr1), 0, M, H, 0.1, 77, 5, hip._p(p1),
                                                        hip._p(dimg.img), hip._st()), "b planes")
        assert torch.equal(dr0, dr1) and torch.equal(p0, p1)
        assert torch.equal(dimg.img.view(torch.int16), hip.Planes(dx0, True).img.view(torch.int16)), nslab


def test_gemm_f32_presplit_planes_grouped_weight_gradients(hip):
    """mtvaf_gemm_f32p_dw_group (research entry, round 5): the four weight-gradient products of a layer from plane images in ONE
    unsplit launch -- the same bits as one launch per product, from natural and from tile-blocked images, with fewer than four
    products too."""
    K, Hh, Ii = 384, 128, 256
    dys = [rnd(K, w, seed=30 + i).to(DEV) for i, w in enumerate((Hh, Ii, Hh, 3 * Hh))]
    xs = [rnd(K, w, seed=40 + i).to(DEV) for i, w in enumerate((Ii, Hh, Hh, Hh))]
    for blocked in (False, True):
        pas, pbs = [hip.Planes(t, blocked) for t in dys], [hip.Planes(t, blocked) for t in xs]
        singles = []
        for pa, pb in zip(pas, pbs):
            o = torch.empty(pa.cols, pb.cols, device=DEV)
            hip.gemm_planes(pa, pb, o, layout_a=hip.KM, layout_b=hip.KM)
            singles.append(o)
        close(singles[0], dys[0].double().cpu().t() @ xs[0].double().cpu(), rtol=3e-6, name="one product")
        for n in (4, 2, 1):
            outs = [torch.full_like(o, float("nan")) for o in singles[:n]]
            hip.gemm_planes_dw_group(list(zip(pas[:n], pbs[:n], outs)))
            for i in range(n):
                assert torch.equal(outs[i], singles[i]), (blocked, n, i)
        # ... with the column sums of an fp32 matrix (the QKV bias gradient) as extra blocks of the launch
        outs = [torch.full_like(o, float("nan")) for o in singles]
        cs = torch.full((3 * Hh,), float("nan"), device=DEV)
        part = rnd(37, 3 * 200, seed=50).to(DEV)  # (a second job of another shape: a column block of stacked partial rows)
        cs2 = torch.full((200,), float("nan"), device=DEV)
        hip.gemm_planes_dw_group(list(zip(pas, pbs, outs)), colsum=[(dys[3], cs), (part[:, 200:400], cs2)])
        for i in range(4):
            assert torch.equal(outs[i], singles[i]), (blocked, i)
        close(cs, dys[3].double().sum(0), rtol=1e-5, atol=1e-5 * float(dys[3].double().sum(0).abs().max()), name="column sums")
        close(cs2, part[:, 200:400].double().sum(0), rtol=1e-5, atol=1e-5, name="column sums of a column block")


@pytest.mark.parametrize("M,N,K", [(128, 256, 32), (256, 512, 64), (640, 768, 96), (384, 1024, 768), (896, 256, 2304), (2432, 2304, 160)])
def test_gemm_f32_presplit_wide_tile_same_bits(hip, M, N, K):
    """The 128 x 256 tile of the pre-split kernel (csrc/gemm_f32pw.hip, round 6; mtvaf_f32p_wide) issues the same MFMA products in the
    same order for every output element as the 128 x 128 tile: the SAME BITS in all three operand layouts (tile-blocked and natural
    images), with 1, 2, 3 k-tiles (prologue / drain of the two-stage ring) and many, a ragged last band of the tile walk (5, 7, 19
    tile rows), bias + GELU (saved pre-activation), accumulate, deterministic split-K, the plane-image epilogues (GELU forward; GELU'
    backward with per-tile column sums) and the grouped weight-gradient launch with column-sum jobs riding along."""
    was = hip.f32p_wide()

    def both(fn, mk, mask=15):
        res = []
        for m in (0, mask):
            hip.f32p_wide(m)
            o = mk()
            fn(o)
            res.append(o)
        return res
    eq = lambda x, y: torch.equal(x.view(torch.int32), y.view(torch.int32))  # (NaN-proof: a tile that was not written fails)
    nan = lambda *s_: torch.full(s_, float("nan"), device=DEV)
    try:
        bias = rnd(N, seed=5).to(DEV)
        for blocked in (True, False):
            a, b, ref, mag = _x3_operands(M, N, K, 0, 0, seed=M + N + K, spread=6)
            pa, pb = hip.Planes(a, blocked), hip.Planes(b, blocked)
            o0, o1 = both(lambda o: hip.gemm_planes(pa, pb, o), lambda: nan(M, N))
            assert eq(o0, o1), ("forward", blocked)
            assert float(((o1.double().cpu() - ref).abs() / mag).max()) <= 2.0 ** -24 * (4 + K ** 0.5)
            (o0, x0), (o1, x1) = both(lambda o: hip.gemm_planes(pa, pb, o[0], bias=bias, epi=hip.EPI_GELU, aux=o[1]), lambda: (nan(M, N), nan(M, N)))
            assert eq(o0, o1) and eq(x0, x1), ("GELU", blocked)
            acc0 = rnd(M, N, seed=6).to(DEV)
            o0, o1 = both(lambda o: hip.gemm_planes(pa, pb, o, accumulate=True), lambda: acc0.clone())
            assert eq(o0, o1), ("accumulate", blocked)
            if K >= 64:
                o0, o1 = both(lambda o: hip.gemm_planes(pa, pb, o, bias=bias, splits=2), lambda: nan(M, N))
                assert eq(o0, o1), ("split-K", blocked)
            a2, b2, ref2, mag2 = _x3_operands(M, N, K, 0, 1, seed=M + N + K + 1, spread=6)
            pa2, pb2 = hip.Planes(a2, True), hip.Planes(b2, blocked)
            o0, o1 = both(lambda o: hip.gemm_planes(pa2, pb2, o, layout_b=hip.KM), lambda: nan(M, N))
            assert eq(o0, o1), ("dX", blocked)
            assert float(((o1.double().cpu() - ref2).abs() / mag2).max()) <= 2.0 ** -24 * (4 + K ** 0.5)
            a3, b3, ref3, mag3 = _x3_operands(M, N, K, 1, 1, seed=M + N + K + 2, spread=6)
            pa3, pb3 = hip.Planes(a3, blocked), hip.Planes(b3, blocked)
            o0, o1 = both(lambda o: hip.gemm_planes(pa3, pb3, o, layout_a=hip.KM, layout_b=hip.KM), lambda: nan(M, N))
            assert eq(o0, o1), ("dW", blocked)
            assert float(((o1.double().cpu() - ref3).abs() / mag3).max()) <= 2.0 ** -24 * (4 + K ** 0.5)
        # the plane-image epilogues
        pa, pb = hip.Planes(a, True), hip.Planes(b, True)

        def mk_ep():
            im = hip.Planes(torch.empty(M, N, device=DEV), True, fill=False)
            im.img.fill_(float("nan"))
            return im, nan(M, N), nan(M, N), nan(M // 128, N)
        r0, r1 = both(lambda o: hip.gemm_planes_ep(pa, pb, o[0], out=o[1], bias=bias, epi=hip.EPI_GELU, aux=o[2]), mk_ep)
        assert torch.equal(r0[0].img.view(torch.int16), r1[0].img.view(torch.int16)) and eq(r0[1], r1[1]) and eq(r0[2], r1[2])
        pre = r0[2]
        r0, r1 = both(lambda o: hip.gemm_planes_ep(pa2, hip.Planes(b2, True), o[0], epi=hip.EPI_DGELU, aux=pre, colpart=o[3], layout_b=hip.KM), mk_ep)
        assert torch.equal(r0[0].img.view(torch.int16), r1[0].img.view(torch.int16)) and eq(r0[3], r1[3]), "GELU' + column partials"
        # the default mask leaves small launches (< 128 wide tiles) and N < 1024 on the 128 x 128 tile: same bits trivially, no fault
        o0, o1 = both(lambda o: hip.gemm_planes(pa, pb, o), lambda: nan(M, N), mask=7)
        assert eq(o0, o1)
    finally:
        hip.f32p_wide(was)


def test_gemm_f32_presplit_wide_tile_grouped_weight_gradients_same_bits(hip):
    """... and the grouped weight-gradient launch (four products, 128 x 256 tiles back to back, column-sum jobs as extra blocks)."""
    was = hip.f32p_wide()
    K, Hh, Ii = 416, 256, 512
    dys = [rnd(K, w, seed=30 + i).to(DEV) for i, w in enumerate((Hh, Ii, Hh, 3 * Hh))]
    xs = [rnd(K, w, seed=40 + i).to(DEV) for i, w in enumerate((Ii, Hh, Hh, Hh))]
    part = rnd(37, 3 * 200, seed=50).to(DEV)
    try:
        for blocked in (False, True):
            pas, pbs = [hip.Planes(t, blocked) for t in dys], [hip.Planes(t, blocked) for t in xs]
            res = []
            for m in (0, 4):
                hip.f32p_wide(m)
                outs = [torch.full((pa.cols, pb.cols), float("nan"), device=DEV) for pa, pb in zip(pas, pbs)]
                cs, cs2 = torch.full((3 * Hh,), float("nan"), device=DEV), torch.full((200,), float("nan"), device=DEV)
                hip.gemm_planes_dw_group(list(zip(pas, pbs, outs)), colsum=[(dys[3], cs), (part[:, 200:400], cs2)])
                outs2 = [torch.full_like(o, float("nan")) for o in outs[:2]]
                hip.gemm_planes_dw_group(list(zip(pas[:2], pbs[:2], outs2)))
                res.append(outs + [cs, cs2] + outs2)
            for x, y in zip(*res):
                assert torch.equal(x.view(torch.int32), y.view(torch.int32)), blocked
            close(res[1][0], dys[0].double().cpu().t() @ xs[0].double().cpu(), rtol=3e-6, name="one product of the group")
    finally:
        hip.f32p_wide(was)


def test_layernorm_adds_the_split_k_slabs_itself_bit_for_bit(hip):
    """Round 5: mtvaf_gemm_f32_slabs leaves a split-K plan's slabs unreduced and the LayerNorm behind the product adds them in the
    reduction launch's order (slab 0 + slab 1 + ... + bias; backward: (slab 0 + ...) + dout) -- forward Wo / FFN-2, backward the
    accumulating FFN-1 dX of a packed batch (modeling_bert.py:353-355, 433-435).  Same bits as product + reduction + LayerNorm:
    outputs, statistics, the stored dense output the backward pass reads, dx / dres and the column-sum partials."""
    import ctypes
    L_ = hip.lib()
    if not hip.f32_split():
        pytest.skip("the planner of the fp32 MFMA pipe does not split these products: nothing to hand over")
    M, H, K = 2432, 768, 3072   # 19 x 6 tiles: the planner splits
    x = rnd(M, K, seed=1).to(DEV)
    w = (rnd(H, K, seed=2) * 0.05).to(DEV)
    bias, res = rnd(H, seed=3).to(DEV), rnd(M, H, seed=4).to(DEV)
    gamma, beta = (rnd(H, seed=5) * 0.1 + 1).to(DEV), rnd(H, seed=6).to(DEV)
    wsb = L_.mtvaf_gemm_f32_workspace_bytes(M, H, K, 1)
    ws = hip.workspace(wsb, x.device)
    E = lambda *s_: torch.empty(*s_, device=DEV)
    # reference: product (+ its own reduction) then LayerNorm
    a0, o0, mu0, rs0 = E(M, H), E(M, H), E(M), E(M)
    hip.gemm(x, 0, w, 0, a0, M, H, K, bias=bias, allow_split=True)
    hip._ck(L_.mtvaf_dropout_res_ln_fwd(hip._p(a0), hip._p(res), hip._p(gamma), hip._p(beta), hip._p(o0), hip._p(mu0), hip._p(rs0), M, H,
                                        1e-12, 0.1, 77, 5, None, hip._st()), "ln")
    # fused: slabs kept, LayerNorm adds them
    a1, o1, mu1, rs1 = torch.full((M, H), float("nan"), device=DEV), E(M, H), E(M), E(M)
    ns = ctypes.c_int(0)
    hip._ck(L_.mtvaf_gemm_f32_slabs(0, 0, hip._p(x), K, hip._p(w), K, hip._p(a1), H, M, H, K, hip._p(bias), 0, hip._p(ws), wsb, ctypes.byref(ns),
                                    hip._st()), "slabs")
    assert ns.value > 1, "the planner was expected to split this product"
    assert bool(torch.isnan(a1).all())  # (C untouched)
    hip._ck(L_.mtvaf_dropout_res_ln_fwd_slabs(hip._p(ws), ns.value, hip._p(bias), hip._p(a1), hip._p(res), hip._p(gamma), hip._p(beta),
                                              hip._p(o1), hip._p(mu1), hip._p(rs1), M, H, 1e-12, 0.1, 77, 5, None, hip._st()), "ln slabs")
    assert torch.equal(a1, a0) and torch.equal(o1, o0) and torch.equal(mu1, mu0) and torch.equal(rs1, rs0)
    # backward: dX = dpre . W (accumulating into dh1) then LayerNorm backward, against the slab-adding form
    dpre, w1 = rnd(M, K, seed=7).to(DEV), (rnd(K, H, seed=8) * 0.05).to(DEV)
    base = rnd(M, H, seed=9).to(DEV)
    nb = L_.mtvaf_ln_bwd_workspace_bytes(M, H)
    part0, part1 = torch.zeros(nb // 4, device=DEV), torch.zeros(nb // 4, device=DEV)  # (the kernel writes a prefix of the workspace)
    d0 = base.clone()
    hip.gemm(dpre, 0, w1, 1, d0, M, H, K, accumulate=True, allow_split=True)
    dx0, dr0 = E(M, H), E(M, H)
    hip._ck(L_.mtvaf_dropout_res_ln_bwd_rows(hip._p(d0), hip._p(a0), hip._p(res), hip._p(gamma), hip._p(mu0), hip._p(rs0), hip._p(dx0),
                                             hip._p(dr0), 0, M, H, 0.1, 77, 5, hip._p(part0), None, hip._st()), "bwd rows")
    d1 = base.clone()
    hip._ck(L_.mtvaf_gemm_f32_slabs(0, 1, hip._p(dpre), K, hip._p(w1), H, hip._p(d1), H, M, H, K, None, 1, hip._p(ws), wsb, ctypes.byref(ns),
                                    hip._st()), "slabs bwd")
    assert ns.value > 1 and torch.equal(d1, base)
    dx1, dr1 = E(M, H), E(M, H)
    hip._ck(L_.mtvaf_dropout_res_ln_bwd_rows_slabs(hip._p(d1), hip._p(ws), ns.value, hip._p(a0), hip._p(res), hip._p(gamma), hip._p(mu0),
                                                   hip._p(rs0), hip._p(dx1), hip._p(dr1), 0, M, H, 0.1, 77, 5, hip._p(part1), None,
                                                   hip._st()), "bwd rows slabs")
    assert torch.equal(dx1, dx0) and torch.equal(dr1, dr0) and torch.equal(part1, part0)
    # an unsplit plan: the entry point behaves as mtvaf_gemm_f32
    xs = rnd(4096, 768, seed=10).to(DEV)
    ws_ = (rnd(3072, 768, seed=11) * 0.05).to(DEV)
    c0, c1 = E(4096, 3072), E(4096, 3072)
    hip.gemm(xs, 0, ws_, 0, c0, 4096, 3072, 768, allow_split=True)
    hip._ck(L_.mtvaf_gemm_f32_slabs(0, 0, hip._p(xs), 768, hip._p(ws_), 768, hip._p(c1), 3072, 4096, 3072, 768, None, 0, hip._p(ws), wsb,
                                    ctypes.byref(ns), hip._st()), "unsplit")
    assert ns.value == 1 and torch.equal(c1, c0)


def test_gemm_f32_split_epilogues_splitk_ktiles_and_fallback(hip):
    """The split kernel behind the fp32 entry points: every epilogue of the path (bias, bias + GELU with the saved
    pre-activation, GELU', tanh, accumulate), forced split-K, a k-tile list (weight gradient: dY exactly zero outside the listed
    32-row tiles), shapes it does not cover (run the fp32 pipe: same result), and the process-wide switch."""
    M, N, K = 256, 384, 512
    x, w, bias = rnd(M, K, seed=1).to(DEV), rnd(N, K, seed=2).to(DEV), rnd(N, seed=3).to(DEV)
    ref = x.double().cpu() @ w.double().cpu().t() + bias.double().cpu()
    out, aux = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    hip.gemm(x, 0, w, 0, out, M, N, K, bias=bias, compute="fp32x3", cfg=5)
    close(out, ref, rtol=2e-6, name="bias")
    hip.gemm(x, 0, w, 0, out, M, N, K, bias=bias, epi=hip.EPI_GELU, aux=aux, compute="fp32x3", cfg=5)
    close(aux, ref, rtol=2e-6, name="saved pre-activation")
    close(out, torch.nn.functional.gelu(ref), rtol=2e-6, name="GELU")
    dy = rnd(M, N, seed=4).to(DEV)
    wt = w.t().contiguous()  # [K, N] -> dX = dY . W with W as a KM operand: here dy [M, N] . w [N, K]
    dx = torch.empty(M, K, device=DEV)
    pre = rnd(M, K, seed=5).to(DEV)
    hip.gemm(dy, 0, w, 1, dx, M, K, N, epi=hip.EPI_DGELU, aux=pre, compute="fp32x3", cfg=5)
    p64 = pre.double().cpu()
    gp = 0.5 * (1 + torch.erf(p64 / 2 ** 0.5)) + p64 * torch.exp(-p64 * p64 / 2) / (2 * torch.pi) ** 0.5
    close(dx, (dy.double().cpu() @ w.double().cpu()) * gp, rtol=3e-6, name="GELU'")
    hip.gemm(x, 0, w, 0, out, M, N, K, bias=bias, epi=hip.EPI_TANH, compute="fp32x3", cfg=5)
    close(out, torch.tanh(ref), rtol=5e-5, name="tanh")  # (|pre-activation| up to ~90: its fp32 rounding alone is 1e-5 of tanh's range)
    acc0 = rnd(M, N, seed=6)
    out.copy_(acc0)
    hip.gemm(x, 0, w, 0, out, M, N, K, accumulate=True, compute="fp32x3", cfg=5)
    close(out, ref - bias.double().cpu() + acc0.double(), rtol=2e-6, name="accumulate")
    for sp in (2, 4):
        out.fill_(float("nan"))
        hip.gemm(x, 0, w, 0, out, M, N, K, bias=bias, allow_split=True, splits=sp, compute="fp32x3", cfg=5)
        close(out, ref, rtol=2e-6, name=f"split-K {sp}")
    del wt
    # weight gradient over a k-tile list
    T, NO, KI = 1024, 256, 384
    valid = torch.ones(T, dtype=torch.bool)
    valid[300:700] = False
    dyw = (rnd(T, NO, seed=7) * valid[:, None]).to(DEV)
    xw = rnd(T, KI, seed=8).to(DEV)
    tiles = sorted(set(int(r) // 32 for r in torch.nonzero(valid).flatten()))
    kl, kc = torch.tensor(tiles, dtype=torch.int32, device=DEV), torch.tensor([len(tiles)], dtype=torch.int32, device=DEV)
    dw = torch.empty(NO, KI, device=DEV)
    refw = dyw.double().cpu().t() @ xw.double().cpu()
    was = hip.f32_split()
    try:
        assert hip.f32_split(True) is True
        for sp in (-1, 1, 3):
            dw.fill_(float("nan"))
            hip.gemm_ktiles(dyw, xw, dw, NO, KI, T, kl, kc, splits=sp, cfg=5)
            close(dw, refw, rtol=2e-6, name=f"k-tile list through the switch, splits {sp}")
        # shapes outside the split kernels' cover run the fp32 pipe under the switch
        xs, ws_ = rnd(100, 72, seed=9).to(DEV), rnd(50, 72, seed=10).to(DEV)
        os_ = torch.empty(100, 50, device=DEV)
        hip.gemm(xs, 0, ws_, 0, os_, 100, 50, 72)
        close(os_, xs.double().cpu() @ ws_.double().cpu().t(), rtol=2e-6, name="uncovered shape")
        hip.gemm(x, 0, w, 0, out, M, N, K, bias=bias)  # (24 tiles: stays on the fp32 pipe)
        close(out, ref, rtol=2e-6, name="fp32 entry under the switch, small product")
        xb, wb = rnd(1024, 256, seed=11).to(DEV), rnd(1536, 256, seed=12).to(DEV)
        ob = torch.empty(1024, 1536, device=DEV)
        hip.prof_start(4)
        hip.gemm(xb, 0, wb, 0, ob, 1024, 1536, 256)  # (96 tiles: the split kernel, chosen by the library)
        recs = hip.prof_stop(4)
        assert recs and recs[0][0]["cfg"] >= 200, recs
        close(ob, xb.double().cpu() @ wb.double().cpu().t(), rtol=2e-6, name="fp32 entry under the switch, 96 tiles")
    finally:
        hip.f32_split(was)
    assert hip.f32_split() == was


@pytest.mark.parametrize("M,H,nslab", [(2432, 768, 2), (2432, 768, 0), (333, 768, 1), (100, 1024, 3), (64, 256, 2)])
def test_layernorm_backward_lean_kernel_against_the_wave_per_row_kernel(hip, M, H, nslab):
    """Round 6: ln_bwd_lean_kernel (a row over the block, <= 48 registers, 64 B of LDS: fits into the CUs a one-round GEMM launch
    occupies) against ln_bwd_kernel<0> (MTVAF_LN_LEAN=0) in ONE process: the dropout mask, dres accumulation and the slab sum are
    bit-identical by construction, the row statistics and column partials are summed in another order -> dx / dres / plane image
    within a few ulp of the row's scale, the finished column sums within 1e-5; and both against fp64 (modeling_bert.py:354-355)."""
    import os
    L_ = hip.lib()
    E = lambda *s_: torch.empty(*s_, device=DEV)
    x, res, dout = rnd(M, H, seed=1).to(DEV), rnd(M, H, seed=2).to(DEV), rnd(M, H, seed=3).to(DEV)
    gamma, beta = (1 + 0.1 * rnd(H, seed=4)).to(DEV), (0.1 * rnd(H, seed=5)).to(DEV)
    slabs = rnd(max(nslab, 1), M, H, seed=6).to(DEV)
    out, mu, rs = E(M, H), E(M), E(M)
    p, seed, off = 0.1, 77, 5
    hip._ck(L_.mtvaf_dropout_res_ln_fwd(hip._p(x), hip._p(res), hip._p(gamma), hip._p(beta), hip._p(out), hip._p(mu), hip._p(rs), M, H,
                                        1e-12, p, seed, off, None, hip._st()), "ln")
    nb = L_.mtvaf_ln_bwd_workspace_bytes(M, H)
    got = {}
    for lean in ("0", "1"):
        os.environ["MTVAF_LN_LEAN"] = lean
        try:
            part = torch.zeros(nb // 4, device=DEV)
            dx, dres = E(M, H), torch.ones(M, H, device=DEV)
            pl = hip.Planes(torch.zeros(M, H, device=DEV), True)
            hip._ck(L_.mtvaf_dropout_res_ln_bwd_rows_planes(hip._p(dout), hip._p(slabs) if nslab else None, nslab, hip._p(x), hip._p(res),
                                                            hip._p(gamma), hip._p(mu), hip._p(rs), hip._p(dx), hip._p(dres), 1, M, H, p, seed,
                                                            off, hip._p(part), hip._p(pl.img), hip._st()), "bwd rows planes")
            dg, db, dbx = E(H), E(H), E(H)
            hip._ck(L_.mtvaf_dropout_res_ln_bwd_finish(hip._p(part), M, H, hip._p(dg), hip._p(db), hip._p(dbx), 0, hip._st()), "finish")
            # the fp32 / bf16-copy entry points take the same kernel with PL = false
            dx2, dres2, part2 = E(M, H), E(M, H), torch.zeros(nb // 4, device=DEV)
            dsum = (slabs[:nslab].sum(0) + dout) if nslab else dout
            hip._ck(L_.mtvaf_dropout_res_ln_bwd_rows(hip._p(dsum), hip._p(x), hip._p(res), hip._p(gamma), hip._p(mu), hip._p(rs), hip._p(dx2),
                                                     hip._p(dres2), 0, M, H, p, seed, off, hip._p(part2), None, hip._st()), "bwd rows")
            got[lean] = (dx.clone(), dres.clone(), pl.img.clone(), dg, db, dbx, dx2, dres2)
        finally:
            os.environ.pop("MTVAF_LN_LEAN", None)
    a, b = got["0"], got["1"]
    scale = float(a[0].abs().max())
    for i, name in enumerate(("dx", "dres")):
        assert float((a[i] - b[i]).abs().max()) <= 4e-6 * max(scale, 1.0), name
    assert torch.equal(a[0] == 0, b[0] == 0)  # the same dropout mask
    for lean in ("0", "1"):  # the plane image is the three-way split of the dx beside it
        img = got[lean][2].view(H // 32, 3, M, 32).float().sum(1).permute(1, 0, 2).reshape(M, H)
        assert float((img - got[lean][0]).abs().max()) <= 1e-6 * max(scale, 1.0), lean
    for i, name in ((3, "dgamma"), (4, "dbeta"), (5, "dbias_x")):
        close(b[i], a[i], rtol=2e-5, name=name)
    # fp64 reference (dropout mask recovered from dx == 0 is not needed: compare the mask-free quantities)
    keep = torch.empty(M, H, device=DEV)
    hip.dropout(torch.ones(M, H, device=DEV), keep, p, seed, off)
    xd, rd = x.double().requires_grad_(True), res.double().requires_grad_(True)
    gd = gamma.double().requires_grad_(True)
    y = F.layer_norm(xd * keep.double() + rd, (H,), gd, beta.double(), 1e-12)
    dsum = ((slabs[:nslab].double().sum(0) if nslab else 0) + dout.double())
    (y * dsum).sum().backward()
    for lean in ("0", "1"):
        dx, dres, _, dg, db, dbx, dx2, dres2 = got[lean]
        close(dx, xd.grad, name=f"dx lean={lean}")
        close(dres - 1, rd.grad, name=f"dres lean={lean}")
        close(dg, gd.grad, rtol=5e-4, name=f"dgamma lean={lean}")
        close(dx2, xd.grad, name=f"dx (fp32 entry) lean={lean}")
        close(dres2, rd.grad, name=f"dres (fp32 entry) lean={lean}")
