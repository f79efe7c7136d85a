"""Size-independent properties of the path at the BASELINE configs[1] shape (BERT-base, B = 32, S = 128, 36 visual
prefix slots) -- checks that need no oracle run of that size: sentences are independent (batch permutation),
padding never leaks into valid tokens, the prefix is a SET of keys (slot permutation), and the training step is
deterministic for a fixed dropout seed."""
import numpy as np
import pytest
import torch

import params as P
from test_model_gpu import DEV, build_tvnet2, make_args

pytestmark = pytest.mark.gpu
B, S, PN = 32, 128, 36


@pytest.fixture(scope="module")
def setup():
    cfg = P.BASE_BERT
    m = build_tvnet2(cfg, make_args(use_prefix=False), sde=P.encoder_params(cfg, 31, std=0.03), sdh=P.head_params(cfg, 32))
    m.eval()
    rng = np.random.default_rng(33)
    lengths = [int(x) for x in rng.integers(16, S + 1, size=B)]
    lengths[0] = S
    ids, mask, tt, labels = P.text_batch(cfg, 34, B, S, lengths, lo_id=1000)
    labels[:, 0] = 9
    pkv = P.prefix_kv(35, cfg.layers, B, cfg.heads, PN, std=0.5)
    return m, ids.to(DEV), mask.to(DEV), tt.to(DEV), labels.to(DEV), [(k.to(DEV), v.to(DEV)) for k, v in pkv], lengths


def run(m, ids, mask, tt, labels, pkv):
    full = torch.cat([torch.ones(ids.shape[0], pkv[0][0].shape[2], dtype=mask.dtype, device=DEV), mask], 1)
    hs = m.bert(input_ids=ids, attention_mask=full, token_type_ids=tt, past_key_values=pkv)["last_hidden_state"]
    em = torch.nn.functional.linear(hs, m.fc.weight, m.fc.bias)
    mask_u8 = mask.to(torch.uint8)
    loss = -m.crf(em, labels, mask=mask_u8, reduction="mean")
    return hs, em, m.crf.decode(em, mask_u8), loss


def test_batch_permutation_equivariance(setup):
    m, ids, mask, tt, labels, pkv, _ = setup
    hs, em, tags, loss = run(m, ids, mask, tt, labels, pkv)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).to(DEV)
    hs2, em2, tags2, loss2 = run(m, ids[perm], mask[perm], tt[perm], labels[perm], [(k[perm], v[perm]) for k, v in pkv])
    # every reduction runs over the same operands in the same order for a given sentence: bit-identical rows
    assert torch.equal(hs2, hs[perm]) and torch.equal(em2, em[perm])
    assert tags2 == [tags[i] for i in perm.tolist()]
    assert abs(float(loss2) - float(loss)) <= 1e-5 * abs(float(loss))  # batch mean: summation order differs


def test_padding_does_not_leak_into_valid_tokens(setup):
    m, ids, mask, tt, labels, pkv, lengths = setup
    hs, em, tags, _ = run(m, ids, mask, tt, labels, pkv)
    ids2 = ids.clone()
    g = torch.Generator().manual_seed(2)
    junk = torch.randint(1000, 30000, ids.shape, generator=g).to(DEV)
    ids2 = torch.where(mask.bool(), ids, junk)  # arbitrary tokens in the padded positions
    hs2, em2, tags2, _ = run(m, ids2, mask, tt, labels, pkv)
    valid = mask.bool()
    assert torch.equal(hs2[valid], hs[valid]), "masked keys carry exactly zero probability: valid rows must not move"
    assert tags2 == tags
    from mtvaf_amd import engine
    if engine.LAST_PACK is not None:  # padding-free (the default): masked rows are not computed at all -- zeros either way
        assert float(hs[~valid].abs().max()) == 0.0 and float(hs2[~valid].abs().max()) == 0.0
    else:  # padded: the padded rows themselves do change (they are computed, not skipped)
        assert not torch.equal(hs2[~valid], hs[~valid])
    with engine.padding_free(False):  # the padded layout computes them: the junk must show there and nowhere else
        hp, _, tp, _ = run(m, ids, mask, tt, labels, pkv)
        hp2, _, tp2, _ = run(m, ids2, mask, tt, labels, pkv)
    assert engine.LAST_PACK is None
    assert torch.equal(hp2[valid], hp[valid]) and tp2 == tp == tags
    assert not torch.equal(hp2[~valid], hp[~valid])


def test_prefix_slots_are_a_set(setup):
    m, ids, mask, tt, labels, pkv, _ = setup
    hs, em, tags, loss = run(m, ids, mask, tt, labels, pkv)
    perm = torch.randperm(PN, generator=torch.Generator().manual_seed(3)).to(DEV)
    pkv2 = [(k[:, :, perm].contiguous(), v[:, :, perm].contiguous()) for k, v in pkv]
    hs2, em2, tags2, loss2 = run(m, ids, mask, tt, labels, pkv2)
    scale = float(hs.abs().max())
    assert float((hs2 - hs).abs().max()) <= 2e-5 * scale  # only the summation order over keys changes
    assert tags2 == tags
    assert abs(float(loss2) - float(loss)) <= 1e-5 * abs(float(loss))


def test_train_step_is_deterministic_for_a_fixed_seed(setup):
    from mtvaf_amd import engine
    m, ids, mask, tt, labels, pkv, _ = setup
    m.train()
    try:
        outs = []
        for _ in range(2):
            engine.RNG.__init__()  # same seed / offset sequence
            torch.manual_seed(0)
            m.zero_grad(set_to_none=True)
            full = torch.cat([torch.ones(B, PN, dtype=mask.dtype, device=DEV), mask], 1)
            hs = m.bert(input_ids=ids, attention_mask=full, token_type_ids=tt, past_key_values=pkv)["last_hidden_state"]
            (hs * hs).mean().backward()
            outs.append((hs.detach().clone(), m.bert.encoder.layer[5].intermediate.dense.weight.grad.clone(),
                         m.bert.encoder.layer[0].attention.self.key.bias.grad.clone()))
        # dropout masks are pure functions of (seed, site, element); split-K slabs are reduced in order; the dW stream
        # only changes WHEN kernels run: two runs agree bit for bit
        for a, b in zip(outs[0], outs[1]):
            assert torch.equal(a, b)
    finally:
        m.eval()
