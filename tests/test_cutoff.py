"""Cutoff augmentation (reference modules/augument.py): the batched cut rules against the oracle's per-sample
restatement (CPU), the fused apply kernel and the augmented span-model pass (GPU)."""
import types

import numpy as np
import pytest
import torch

import params as P
from oracle import mtvaf_oracle as O
from mtvaf_amd.modules.augument import Cutoff


def _batch(B=6, S=24, H=16, seed=0):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(3, S + 1, (B,), generator=g)
    lens[0] = S
    masks = (torch.arange(S)[None, :] < lens[:, None]).long()
    embeds = torch.randn(B, S, H, generator=g)
    return embeds, masks, lens, g


@pytest.mark.parametrize("ratio", [0.0, 0.1, 0.3, 0.5])
def test_span_cut_rule_matches_reference_loop(ratio):
    embeds, masks, lens, g = _batch()
    u = torch.rand(embeds.shape[0], generator=g)
    ref_e, ref_m = O.cutoff_span(embeds, masks, lens, ratio, u)
    keep = Cutoff.span_keep(lens, ratio, embeds.shape[1], u=u)
    assert torch.equal(embeds * keep[:, :, None], ref_e)
    assert torch.equal(masks * keep.long(), ref_m)


@pytest.mark.parametrize("ratio", [0.0, 0.1, 0.4])
def test_token_and_dim_cut_rules_match_reference_loop(ratio):
    embeds, masks, lens, g = _batch()
    B, S, H = embeds.shape
    cl = (lens.float() * ratio).long()
    draws = torch.rand(B, S, generator=g)
    zero_index = [(draws[i, :cl[i]] * lens[i].float()).long() for i in range(B)]
    ref_e, ref_m = O.cutoff_token(embeds, masks, zero_index)
    keep = Cutoff.index_keep(lens, cl, S, draws=draws)
    assert torch.equal(embeds * keep[:, :, None], ref_e)
    assert torch.equal((keep * masks).long(), ref_m)
    n = int(H * ratio)
    ddraws = torch.rand(B, H, generator=g)
    zi = [(ddraws[i, :n] * H).long() for i in range(B)]
    ref_e, ref_m = O.cutoff_dim(embeds, masks, zi)
    keepd = Cutoff.index_keep(torch.full((B,), H), torch.full((B,), n), H, draws=ddraws)
    assert torch.equal(embeds * keepd[:, None, :], ref_e) and torch.equal(ref_m, masks)


@pytest.mark.gpu
def test_mask_mul_kernel_and_backward():
    from mtvaf_amd import engine
    g = torch.Generator().manual_seed(1)
    x = torch.randn(5, 33, 768, generator=g)
    rk = (torch.rand(5 * 33, generator=g) > 0.3).float()
    ck = (torch.rand(5, 768, generator=g) > 0.1).float()
    for r, c in ((rk, None), (None, ck), (rk, ck)):
        xg = x.clone().cuda().requires_grad_(True)
        y = engine.MaskMulFunction.apply(xg, None if r is None else r.cuda(), None if c is None else c.cuda())
        ref = x * (1 if r is None else r.view(5, 33, 1)) * (1 if c is None else c.view(5, 1, 768))
        assert torch.equal(y.cpu(), ref)
        (y * 2.0).sum().backward()
        assert torch.equal(xg.grad.cpu(), 2.0 * torch.ones_like(x) * (1 if r is None else r.view(5, 33, 1)) *
                           (1 if c is None else c.view(5, 1, 768)))


@pytest.mark.gpu
@pytest.mark.parametrize("aug_type", ["span_cutoff", "token_cutoff", "dim_cutoff"])
def test_augmented_extraction_pass(aug_type):
    from test_model_gpu import DEV, LABELS, hf_config, make_args
    from mtvaf_amd.models.bert_model import TVNetSAModel
    cfg = P.EncCfg(vocab_size=500, hidden=128, heads=2, inter=256, layers=2, max_pos=64)
    args = make_args(use_prefix=False, gcn_layer_number=0, num_layers=0, aug_type=aug_type, aug_cutoff_ratio=0.0)
    args.bert_config = hf_config(cfg)
    torch.manual_seed(0)
    m = TVNetSAModel(LABELS, None, args).to(DEV).eval()
    B, S = 6, 32
    ids, mask, tt, _ = (t.to(DEV) for t in P.text_batch(cfg, 3, B, S, lo_id=5))
    plain = m.extraction(mask, ids, None, tt)
    same = m.extraction(mask, ids, None, tt, True)       # ratio 0: nothing is cut -> identical to the plain pass
    for a, b in zip(plain, same):
        assert torch.equal(a, b)
    args.aug_cutoff_ratio = 0.3
    st, en, seq = m.extraction(mask, ids, None, tt, True)
    assert not torch.equal(seq, plain[2]) and torch.isfinite(seq).all()
    m.train()
    st, en, seq = m.extraction(mask, ids, None, tt, True)
    (st.sum() + en.sum()).backward()
    gw = m.bert.embeddings.word_embeddings.weight.grad
    assert gw is not None and torch.isfinite(gw).all() and float(gw.abs().sum()) > 0
