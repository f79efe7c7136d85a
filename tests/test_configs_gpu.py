"""Parity of the ASSEMBLED `TVNetSAModel2.forward` on the code path `bench.py` times, and coverage of every BASELINE
configuration (C1 bs4/S64/P16 fp32, C2 bs32/S128/P36 fp32, C3 RoBERTa-base bf16, C4 bs64 bf16 DP,
C5 is in test_model_gpu.py::test_long_sequence_config5_shape_vs_oracle).

The timed path differs from a piecewise call of `m.bert` / `m.crf`: with `use_prefix=True` and B >= 8 the prompt
generator runs on the engine's second stream (the encoder waits for `PrefixKV.ready_event` right before its first
attention kernel), with B*S >= 1024 the Viterbi decode runs on the second stream next to the CRF forward algorithm, the
tags come back as `DeferredTags`, and the weight-gradient products run on the second stream in backward.  A missing
event wait on any of these shows up here as a mismatch against the CPU oracle
(`O.visual_prompt` -> `O.tvnet2_forward`, reference models/bert_model.py:480-588)."""
import os
import types

import numpy as np
import pytest
import torch

import params as P
from oracle import mtvaf_oracle as O
from test_model_gpu import DEV, LABELS, _prompt_inputs, build_tvnet2, close, hf_config, make_args

pytestmark = pytest.mark.gpu


def _assembled_case(cfg, B, S, n_aux, seed, std=0.03):
    sde, sdh, sdp = P.encoder_params(cfg, seed, std=std), P.head_params(cfg, seed + 1), P.prompt_params(seed + 2)
    ids, mask, tt, labels = P.text_batch(cfg, seed + 3, B, S, lo_id=5 if cfg.roberta else 1000)
    labels[:, 0] = 9
    feats, aux, lab = _prompt_inputs(seed + 4, B, n_aux)
    return sde, sdh, sdp, (ids, mask, tt, labels), (feats, aux, lab)


def _oracle(cfg, sde, sdh, sdp, text, vis, grad_names):
    ids, mask, tt, labels = text
    feats, aux, _ = vis
    B = ids.shape[0]
    sd = {**{"bert." + k: v.clone() for k, v in sde.items()}, **{k: v.clone() for k, v in sdh.items()},
          **{k: v.clone() for k, v in sdp.items()}}
    for n in grad_names:
        sd[n].requires_grad_(True)
    res, _, _ = O.visual_prompt(sd, feats.reshape(B, 4, -1), [aux[:, i].reshape(B, 4, -1) for i in range(aux.shape[1])],
                                num_layers=cfg.layers, num_heads=cfg.heads)
    loss, em, tags, hs = O.tvnet2_forward(sd, ids, mask, tt, labels, res, cfg.layers, cfg.heads, cfg.eps,
                                          roberta=cfg.roberta, pad_idx=cfg.pad_idx)
    loss.backward()
    return float(loss), em.detach(), tags, {n: sd[n].grad for n in grad_names}


def _run_model(m, text, vis):
    """-> (output, emissions).  The emissions are the tensor the HIP `fc` product (engine.LinearFunction: skinny GEMM behind
    the head dropout) handed to the CRF kernels, captured where the model passes it to the Viterbi decode -- not a
    recomputation from the hidden state."""
    ids, mask, tt, labels = (t.to(DEV) for t in text)
    feats, aux, lab = (t.to(DEV) for t in vis)
    cap = {}
    decode = m.crf.decode_deferred

    def spy(em, mask_u8):
        cap["em"] = em.detach().clone()
        return decode(em, mask_u8)
    m.crf.decode_deferred = spy
    m.zero_grad(set_to_none=True)
    try:
        out = m(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, imagelabel=lab, images=feats, aux_imgs=aux)
    finally:
        del m.crf.decode_deferred  # (the instance attribute shadowing the method)
    out.loss.backward()
    torch.cuda.synchronize()  # (the capture may have run on the second stream)
    return out, cap["em"]


def _bf16_report(tag, em, oem, loss, oloss, tags, otags, grads=None):
    """Measured deviation of a mixed-precision run from the fp32 oracle, printed (pytest -s / the captured log) so that the
    bounds asserted next to it are the observed figures plus a margin, not a guess."""
    rel = float((em - oem).norm() / oem.norm())
    lrel = abs(loss - oloss) / abs(oloss)
    agree = sum(a == b for ta, tb in zip(tags, otags) for a, b in zip(ta, tb)) / max(1, sum(len(t) for t in otags))
    g = {n: float((a - b).norm() / b.norm()) for n, (a, b) in (grads or {}).items()}
    print(f"[bf16 deviation] {tag}: emissions {rel:.3e}  loss {lrel:.3e}  tag agreement {agree:.4f}  "
          + "  ".join(f"{n.split('.')[-2] if '.' in n else n} {v:.3e}" for n, v in g.items()), flush=True)
    return rel, lrel, agree, g


# element-wise relative tolerance of the emissions (|ref| >= 1e-2 max |ref|): north_star's 1e-3, per element
ELEM_RTOL = 1e-3
GRADS = ["encoder_conv.2.weight", "projectors.0.weight", "bert.encoder.layer.5.intermediate.dense.weight",
         "bert.encoder.layer.0.attention.self.value.weight", "fc.weight", "crf.transitions"]


@pytest.mark.parametrize("B", [8, 32])
def test_assembled_forward_on_the_timed_path_vs_oracle(B, f32_arith, pad_mode):
    """BASELINE configs[1] (the bench workload): BERT-base 12 layers, S = 128, 8 aux crops -> P = 36, use_prefix=True.
    B = 8 (1024 tokens) and B = 32 (the bench batch): prompt generator and Viterbi on the second stream, DeferredTags.
    In both fp32 arithmetics (`f32_arith`): the library default that bench.py times, and the fp32 MFMA pipe; in both layouts
    (`pad_mode`): padding-free, the default that bench.py times, and padded."""
    from mtvaf_amd import engine
    from mtvaf_amd.modules.crf import DeferredTags
    if not engine.DW_SIDE_STREAM:
        pytest.skip("covers the second stream: run without MTVAF_DW_STREAM=0")
    cfg = P.BASE_BERT
    S, n_aux = 128, 8
    sde, sdh, sdp, text, vis = _assembled_case(cfg, B, S, n_aux, seed=41)
    oloss, oem, otags, ograds = _oracle(cfg, sde, sdh, sdp, text, vis, GRADS)
    m = build_tvnet2(cfg, make_args(alpha=0.0), sde=sde, sdh=sdh, sdp=sdp)
    m.eval()
    out, em = _run_model(m, text, vis)
    assert isinstance(out.logits, DeferredTags)
    assert (engine.LAST_PACK is not None) == (pad_mode == "unpad"), "the run did not use the layout the test asked for"
    if engine.UNPAD:  # (padding-free: masked positions are not computed)
        valid = text[1].bool()
        close(em[valid.to(DEV)], oem[valid], name="emissions of the unmasked tokens")
    else:
        close(em, oem, name="emissions")
    assert abs(float(out.loss) - oloss) <= 1e-3 * abs(oloss), (float(out.loss), oloss)
    assert list(out.logits) == otags, "decoded tags differ from the oracle"
    named = dict(m.named_parameters())
    for n in GRADS:
        close(named[n].grad, ograds[n], rtol=3e-3, name=n)
    # the measured margins behind the bounds above (printed: pytest -s / the captured log; the same report that
    # __graft_entry__.smoke() and bench.py put into their output), and an ELEMENT-WISE relative bound on the emissions next to
    # close()'s max-norm one: every element that is not tiny against the largest (|ref| >= 1e-2 max |ref|) within ELEM_RTOL of the
    # oracle's, relative to ITSELF
    import parity_report
    rep = parity_report.deviations(em, oem, float(out.loss), oloss, list(out.logits), otags, {n: named[n].grad for n in GRADS}, ograds,
                                   valid=text[1].bool() if engine.UNPAD else None)
    print(f"[{f32_arith}/{pad_mode} B={B}] " + parity_report.fmt(rep), flush=True)
    assert rep["elem_share"] >= 0.9, rep  # (the floor leaves almost every emission in)
    assert rep["emissions_elem_rel_max"] <= ELEM_RTOL and rep["emissions_max_rel"] <= 1e-3 and rep["worst_grad_rel"] <= 3e-3, rep
    # the same step with every stream serialised must agree bit for bit (eval mode: no dropout)
    ref = {n: named[n].grad.clone() for n in GRADS}
    ref_loss, ref_tags = float(out.loss), list(out.logits)
    engine.DW_SIDE_STREAM = False
    try:
        out2, em2 = _run_model(m, text, vis)
    finally:
        engine.DW_SIDE_STREAM = True
    assert float(out2.loss) == ref_loss and list(out2.logits) == ref_tags
    assert torch.equal(em2, em)
    for n in GRADS:
        assert torch.equal(named[n].grad, ref[n]), f"{n}: streams on/off differ"


def test_config1_bs4_seq64_three_aux_fp32():
    """BASELINE configs[0] shape (the reference's CPU-runnable case): bs 4, S 64, 3 aux crops -> P = 16, 12 layers, fp32.
    Small batch: single-stream host-bound path."""
    cfg = P.BASE_BERT
    sde, sdh, sdp, text, vis = _assembled_case(cfg, 4, 64, 3, seed=51)
    oloss, oem, otags, ograds = _oracle(cfg, sde, sdh, sdp, text, vis, GRADS)
    m = build_tvnet2(cfg, make_args(alpha=0.0), sde=sde, sdh=sdh, sdp=sdp)
    m.eval()
    out, em = _run_model(m, text, vis)
    close(em, oem, name="emissions")
    assert abs(float(out.loss) - oloss) <= 1e-3 * abs(oloss)
    assert list(out.logits) == otags
    named = dict(m.named_parameters())
    for n in GRADS:
        close(named[n].grad, ograds[n], rtol=3e-3, name=n)


@pytest.mark.parametrize("B,S,n_aux,lengths", [(3, 50, 2, [50, 1, 17]), (5, 33, 1, [33, 2, 33, 5, 16]), (2, 130, 8, [130, 64]),
                                              (1, 7, 3, [7])])
@pytest.mark.parametrize("unpad", [False, True])
def test_ragged_odd_shapes_on_the_assembled_path_vs_oracle(B, S, n_aux, lengths, unpad):
    """Shapes no tile divides (S = 50 / 33 / 130 / 7, B = 1 / 3 / 5, P = 8 / 12 / 16 / 36), one-token sentences beside full ones:
    the assembled TVNetSAModel2 step against the oracle -- emissions, loss, tags, parameter gradients; padded and
    padding-free (packed token rows: 68 / 89 / 194 / 7 of them)."""
    from mtvaf_amd import engine
    cfg = P.BASE_BERT
    sde, sdh, sdp = P.encoder_params(cfg, 71, std=0.03), P.head_params(cfg, 72), P.prompt_params(73)
    text = P.text_batch(cfg, 74, B, S, lengths=lengths, lo_id=1000)
    text[3][:, 0] = 9
    vis = _prompt_inputs(75, B, n_aux)
    oloss, oem, otags, ograds = _oracle(cfg, sde, sdh, sdp, text, vis, GRADS)
    m = build_tvnet2(cfg, make_args(alpha=0.0), sde=sde, sdh=sdh, sdp=sdp)
    m.eval()
    with engine.padding_free(unpad):
        out, em = _run_model(m, text, vis)
    valid = text[1].bool()
    close(em[valid.to(DEV)], oem[valid], name="emissions of the unmasked tokens")
    assert abs(float(out.loss) - oloss) <= 1e-3 * abs(oloss)
    assert list(out.logits) == otags
    named = dict(m.named_parameters())
    for n in GRADS:
        close(named[n].grad, ograds[n], rtol=3e-3, name=n)


ROBERTA_BASE = P.EncCfg(vocab_size=50265, hidden=768, heads=12, inter=3072, layers=12, max_pos=514, type_vocab=1, eps=1e-5,
                        roberta=True, pad_idx=1)


def _bf16(fn):
    from mtvaf_amd import hip
    hip.set_compute_dtype("bf16")
    try:
        r = fn()
        assert hip.streamk_errors() == 0, "a stream-K launch (grouped weight gradients) reported a timed-out wait"
        return r
    finally:
        hip.set_compute_dtype("fp32")


# Mixed-precision bounds against the fp32 oracle = the deviations measured on MI355X (printed by _bf16_report; round 3:
# emissions 2.7e-3 - 1.02e-2 (the largest at the C4 shape, BERT-base bs 64), loss 0.3-6.1e-4, tags 99.1-99.4 % at the C3 shapes (97.8-100 % on the 68 .. 194-token odd shapes),
# gradient norms 0.3-3.3e-2 -- the largest on the prompt generator's weights, whose gradient sums over every layer's prefix
# slots) plus a margin.  bf16 operands carry 2^-9 relative rounding, so north_star's 1e-3 / bit-exact tags cannot hold by
# construction in this mode.
# Round 6: the bounds are 1.25 x the WORST deviation observed over rounds 3 - 6 (the kernels are
# deterministic: the figures do not move from run to run) -- emissions 1.014e-2 (C4), loss 6.1e-4, tag disagreement 0.72 % of 4.8 k
# tokens (C4; 0.9 % bound), gradient norms 2.28e-2 at the BASELINE shapes and 3.31e-2 on the odd shapes (`encoder_conv.2.weight`).
BF16_EM, BF16_LOSS, BF16_TAGS, BF16_GRAD, BF16_GRAD_ODD = 1.27e-2, 7.6e-4, 0.991, 2.9e-2, 4.2e-2


@pytest.mark.parametrize("B", [8, 32])
def test_config3_roberta_base_bf16_vs_fp32_oracle(B):
    """BASELINE configs[2]: RoBERTa-base (12 layers, vocab 50265, 514 positions, eps 1e-5, pad id 1), S = 128, P = 36,
    bf16 compute, at B = 8 and at the configuration's own B = 32, against the fp32 oracle."""
    cfg = ROBERTA_BASE
    S, n_aux = 128, 8
    sde, sdh, sdp, text, vis = _assembled_case(cfg, B, S, n_aux, seed=61)
    oloss, oem, otags, ograds = _oracle(cfg, sde, sdh, sdp, text, vis, GRADS[:3])
    m = build_tvnet2(cfg, make_args(alpha=0.0, bert_name="roberta-base"), sde=sde, sdh=sdh, sdp=sdp)
    m.eval()
    out, em = _bf16(lambda: _run_model(m, text, vis))
    named = dict(m.named_parameters())
    valid = text[1].bool()
    rel, lrel, agree, g = _bf16_report(f"C3 RoBERTa-base B={B}", em.cpu()[valid], oem[valid], float(out.loss), oloss,
                                       list(out.logits), otags, {n: (named[n].grad.cpu(), ograds[n]) for n in GRADS[:3]})
    assert rel < BF16_EM and lrel < BF16_LOSS and agree >= BF16_TAGS, (rel, lrel, agree)
    assert all(v < BF16_GRAD for v in g.values()), g


def test_config4_bert_base_bs64_bf16_vs_fp32_oracle():
    """BASELINE configs[3], per-GPU shape: BERT-base, bs 64, S = 128, P = 36, bf16 compute -- 8192 token rows, where the four
    weight gradients of every layer run as ONE grouped stream-K launch of the 256x256 kernel -- against the fp32 oracle."""
    cfg = P.BASE_BERT
    B, S, n_aux = 64, 128, 8
    sde, sdh, sdp, text, vis = _assembled_case(cfg, B, S, n_aux, seed=67)
    oloss, oem, otags, ograds = _oracle(cfg, sde, sdh, sdp, text, vis, GRADS)
    m = build_tvnet2(cfg, make_args(alpha=0.0), sde=sde, sdh=sdh, sdp=sdp)
    m.eval()
    out, em = _bf16(lambda: _run_model(m, text, vis))
    named = dict(m.named_parameters())
    valid = text[1].bool()
    rel, lrel, agree, g = _bf16_report(f"C4 BERT-base B={B}", em.cpu()[valid], oem[valid], float(out.loss), oloss,
                                       list(out.logits), otags, {n: (named[n].grad.cpu(), ograds[n]) for n in GRADS})
    assert rel < BF16_EM and lrel < BF16_LOSS and agree >= BF16_TAGS, (rel, lrel, agree)
    assert all(v < BF16_GRAD for v in g.values()), g


@pytest.mark.parametrize("B,S,n_aux,lengths", [(3, 50, 2, [50, 1, 17]), (5, 33, 1, [33, 2, 33, 5, 16]), (2, 130, 8, [130, 64])])
@pytest.mark.parametrize("unpad", [False, True])
def test_ragged_odd_shapes_in_bf16_mode_track_the_fp32_oracle(B, S, n_aux, lengths, unpad):
    """The same odd shapes in the mixed-precision mode (bf16 kernels need 8-element operand rows and fall back where a shape
    breaks that), within the measured mixed-precision bounds of the fp32 oracle (a handful of tokens: tag agreement is
    counted over 7 .. 194 of them, so one flipped tag is 0.5 .. 1.5 %: bound 95 %)."""
    from mtvaf_amd import engine
    cfg = P.BASE_BERT
    sde, sdh, sdp = P.encoder_params(cfg, 71, std=0.03), P.head_params(cfg, 72), P.prompt_params(73)
    text = P.text_batch(cfg, 74, B, S, lengths=lengths, lo_id=1000)
    text[3][:, 0] = 9
    vis = _prompt_inputs(75, B, n_aux)
    oloss, oem, otags, ograds = _oracle(cfg, sde, sdh, sdp, text, vis, GRADS)
    m = build_tvnet2(cfg, make_args(alpha=0.0), sde=sde, sdh=sdh, sdp=sdp)
    m.eval()
    with engine.padding_free(unpad):
        out, em = _bf16(lambda: _run_model(m, text, vis))
    valid = text[1].bool()
    named = dict(m.named_parameters())
    rel, lrel, agree, g = _bf16_report(f"odd shape B={B} S={S} unpad={unpad}", em.cpu()[valid], oem[valid], float(out.loss), oloss,
                                       list(out.logits), otags, {n: (named[n].grad.cpu(), ograds[n]) for n in GRADS[:3]})
    assert rel < BF16_EM and lrel < BF16_LOSS and agree >= 0.95, (rel, lrel, agree)
    assert all(v < BF16_GRAD_ODD for v in g.values()), g


def _props_model(cfg, bert_name, dropout=0.0):
    from mtvaf_amd.models.bert_model import TVNetSAModel2
    args = make_args(alpha=0.0, bert_name=bert_name)
    args.bert_config = hf_config(cfg, dropout=dropout)
    torch.manual_seed(7)
    return TVNetSAModel2(LABELS, None, args).to(DEV)


def _emissions(m, ids, mask, tt, feats, aux):
    cap = {}
    hb = m.bert.register_forward_hook(lambda mod, inp, out: cap.update(out=out))
    with torch.no_grad():
        out = m(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=None, images=feats, aux_imgs=aux)
    hb.remove()
    return cap["out"]["last_hidden_state"], list(out.logits)


@pytest.mark.parametrize("case", ["C3 roberta bs32", "C4 bert bs64"])
def test_bf16_full_size_properties(case):
    """Size-independent properties of the path at the full C3 / C4 shapes IN bf16 MODE (assembled forward, second
    stream live): batch-permutation equivariance and padding that never leaks into valid tokens, both bit-identical
    (every kernel treats sentences independently and rounds per element), and bit-deterministic repeated calls."""
    roberta = case.startswith("C3")
    cfg = ROBERTA_BASE if roberta else P.BASE_BERT
    B, S, n_aux = (32, 128, 8) if roberta else (64, 128, 8)
    m = _props_model(cfg, "roberta-base" if roberta else "bert-base-uncased").eval()
    ids, mask, tt, _ = (t.to(DEV) for t in P.text_batch(cfg, 71, B, S, lo_id=5))
    feats, aux, _ = (t.to(DEV) for t in _prompt_inputs(72, B, n_aux))

    def run():
        h, tags = _emissions(m, ids, mask, tt, feats, aux)
        h2, tags2 = _emissions(m, ids, mask, tt, feats, aux)
        assert torch.equal(h, h2) and tags == tags2, "repeated forward differs"
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).to(DEV)
        hp, tagsp = _emissions(m, ids[perm], mask[perm], tt[perm], feats[perm], aux[perm])
        assert torch.equal(hp, h[perm]), "batch permutation changes rows"
        assert tagsp == [tags[int(i)] for i in perm]
        junk = ids.clone()
        junk[mask == 0] = 7 if not roberta else 9  # junk tokens under the padding (not the pad id)
        hj, tagsj = _emissions(m, junk, mask, tt, feats, aux)
        valid = mask.bool()
        assert torch.equal(hj[valid], h[valid]), "padding leaks into valid tokens"
        assert tagsj == tags
    _bf16(run)


def test_config4_bf16_training_step_is_deterministic_and_finite():
    """C4 per-GPU shape (bs 64, S 128, P 36, bf16 compute), train mode: two identical steps from the same dropout
    seed give bit-identical loss and gradients (weight-gradient stream on), and nothing overflows."""
    from mtvaf_amd import engine
    cfg = P.BASE_BERT
    B, S, n_aux = 64, 128, 8
    m = _props_model(cfg, "bert-base-uncased", dropout=0.1).train()
    ids, mask, tt, labels = (t.to(DEV) for t in P.text_batch(cfg, 81, B, S, lo_id=1000))
    feats, aux, _ = (t.to(DEV) for t in _prompt_inputs(82, B, n_aux))

    def step():
        engine.RNG.offset = 0
        torch.manual_seed(5)
        m.zero_grad(set_to_none=True)
        out = m(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, images=feats, aux_imgs=aux)
        out.loss.backward()
        torch.cuda.synchronize()
        return float(out.loss), {n: p.grad.clone() for n, p in m.named_parameters()
                                 if p.grad is not None and "word_embeddings" not in n}

    def run():
        l1, g1 = step()
        l2, g2 = step()
        assert l1 == l2 and np.isfinite(l1)
        for n in g1:
            assert torch.isfinite(g1[n]).all(), n
            assert torch.equal(g1[n], g2[n]), n
    _bf16(run)


def test_two_forwards_one_backward_accumulates_encoder_grads():
    """The reference's cutoff flow (modules/train.py:414-455): model(...) and model(..., second input) under ONE
    loss.backward().  Both encoder nodes see param.grad is None; the flat per-layer gradient buffer must not be handed
    to both (the second would overwrite the first's views: 2*g_last instead of g1 + g2).  Checked against the oracle's
    gradient of the summed loss."""
    cfg = P.EncCfg(vocab_size=300, hidden=128, heads=2, inter=256, layers=2, max_pos=64)
    sde, sdh = P.encoder_params(cfg, 1), P.head_params(cfg, 2)
    m = build_tvnet2(cfg, make_args(use_prefix=False), sde=sde, sdh=sdh)
    m.eval()
    b1 = P.text_batch(cfg, 3, 6, 40, lo_id=5)
    b2 = P.text_batch(cfg, 4, 6, 40, lo_id=5)
    names = ["bert.encoder.layer.0.intermediate.dense.weight", "bert.encoder.layer.1.attention.self.query.weight",
             "bert.encoder.layer.1.output.LayerNorm.bias", "fc.weight"]
    sd = {**{"bert." + k: v.clone() for k, v in sde.items()}, **{k: v.clone() for k, v in sdh.items()}}
    for n in names:
        sd[n].requires_grad_(True)
    tot = 0
    for ids, mask, tt, labels in (b1, b2):
        tot = tot + O.tvnet2_forward(sd, ids, mask, tt, labels, None, cfg.layers, cfg.heads, cfg.eps)[0]
    tot.backward()
    m.zero_grad(set_to_none=True)
    loss = 0
    for ids, mask, tt, labels in (b1, b2):
        loss = loss + m(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), token_type_ids=tt.to(DEV),
                        labels=labels.to(DEV)).loss
    loss.backward()
    assert abs(float(loss) - float(tot)) <= 1e-3 * abs(float(tot))
    named = dict(m.named_parameters())
    for n in names:
        close(named[n].grad, sd[n].grad, rtol=3e-3, name=n)
    # and the next ordinary step takes the zero-copy path again
    m.zero_grad(set_to_none=True)
    ids, mask, tt, labels = (t.to(DEV) for t in b1)
    m(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels).loss.backward()
    st = m.bert.encoder._stores[1]
    g = m.bert.encoder.layer[1].intermediate.dense.weight.grad
    from mtvaf_amd import engine
    if engine.NATIVE_EXEC and engine.DIRECT_GRADS:  # (the Python orchestration hands views to autograd, which may copy them)
        assert st.grad.data_ptr() <= g.data_ptr() < st.grad.data_ptr() + st.grad.numel() * 4


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_native_executor_equals_python_orchestration(dtype):
    """csrc/executor.hip composes the same kernels in the same order as the Python engine: loss, tags and every gradient
    of a training step (dropout live, fixed seed, second stream on) must agree bit for bit between MTVAF_NATIVE_EXEC=1
    (one C call per layer and direction, arenas, .grad adopted directly) and the per-kernel Python orchestration."""
    from mtvaf_amd import engine, hip
    cfg = P.EncCfg(vocab_size=30522, hidden=768, heads=12, inter=3072, layers=3, max_pos=512)
    hip.set_compute_dtype(dtype)
    try:
        m = _props_model(cfg, "bert-base-uncased", dropout=0.1).train()
        from test_model_gpu import _prompt_inputs
        ids, mask, tt, labels = (t.to(DEV) for t in P.text_batch(cfg, 91, 16, 128, lo_id=1000))
        feats, aux, _ = (t.to(DEV) for t in _prompt_inputs(92, 16, 8))

        def step(native):
            engine.NATIVE_EXEC = native
            engine.RNG.offset = 0
            torch.manual_seed(5)
            m.zero_grad(set_to_none=True)
            out = m(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, images=feats, aux_imgs=aux)
            out.loss.backward()
            torch.cuda.synchronize()
            return float(out.loss), list(out.logits), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        try:
            # (the Python orchestration knows the padded layout only: padding-free execution is the native executor's)
            with engine.padding_free(False):
                l1, t1, g1 = step(True)
                l0, t0, g0 = step(False)
        finally:
            engine.NATIVE_EXEC = True
        assert l1 == l0 and t1 == t0
        assert set(g1) == set(g0)
        for n in g1:
            if "word_embeddings" in n:  # float-atomic scatter-add: order-dependent in the last bits
                close(g1[n], g0[n], rtol=1e-4, name=n)
            else:
                assert torch.equal(g1[n], g0[n]), n
    finally:
        hip.set_compute_dtype("fp32")


# ---- BASELINE configs[4] (C5): S = 512, 36 visual regions ------------------------------------------------------------------
def test_config5_assembled_seq512_vs_oracle(f32_arith, pad_mode):
    """C5 shape on the ASSEMBLED model: TVNetSAModel2, BERT-base 12 layers, S = 512, 8 aux crops (P = 36), B = 4, fp32, against
    the oracle: prompt generator on the second stream, CRF / Viterbi at S = 512, weight-gradient k-tile lists (half the rows
    of a ragged S = 512 batch are padding), second stream in backward."""
    cfg = P.BASE_BERT
    B, S, n_aux = 4, 512, 8
    sde, sdh, sdp, text, vis = _assembled_case(cfg, B, S, n_aux, seed=91)
    oloss, oem, otags, ograds = _oracle(cfg, sde, sdh, sdp, text, vis, GRADS)
    m = build_tvnet2(cfg, make_args(alpha=0.0), sde=sde, sdh=sdh, sdp=sdp)
    m.eval()
    out, em = _run_model(m, text, vis)
    from mtvaf_amd import engine
    assert (engine.LAST_PACK is not None) == (pad_mode == "unpad"), "the run did not use the layout the test asked for"
    valid = text[1].bool()
    close(em[valid.to(DEV)], oem[valid], name="emissions of the unmasked tokens")
    assert abs(float(out.loss) - oloss) <= 1e-3 * abs(oloss), (float(out.loss), oloss)
    assert list(out.logits) == otags, "decoded tags differ from the oracle"
    named = dict(m.named_parameters())
    for n in GRADS:
        close(named[n].grad, ograds[n], rtol=3e-3, name=n)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_config5_full_size_properties(dtype):
    """C5 at its full size (B = 128, S = 512, P = 36: 65 536 token rows), assembled forward, fp32 and mixed precision:
    repeated calls bit-identical, batch permutation permutes rows bit for bit, junk under the padding never reaches a valid
    token."""
    from mtvaf_amd import hip
    cfg = P.BASE_BERT
    B, S, n_aux = 128, 512, 8
    m = _props_model(cfg, "bert-base-uncased").eval()
    ids, mask, tt, _ = (t.to(DEV) for t in P.text_batch(cfg, 73, B, S, lo_id=5))
    feats, aux, _ = (t.to(DEV) for t in _prompt_inputs(74, B, n_aux))
    hip.set_compute_dtype(dtype)
    try:
        h, tags = _emissions(m, ids, mask, tt, feats, aux)
        h2, tags2 = _emissions(m, ids, mask, tt, feats, aux)
        assert torch.equal(h, h2) and tags == tags2, "repeated forward differs"
        del h2
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).to(DEV)
        hp, tagsp = _emissions(m, ids[perm], mask[perm], tt[perm], feats[perm], aux[perm])
        assert torch.equal(hp, h[perm]), "batch permutation changes rows"
        assert tagsp == [tags[int(i)] for i in perm]
        del hp
        junk = ids.clone()
        junk[mask == 0] = 7
        hj, tagsj = _emissions(m, junk, mask, tt, feats, aux)
        valid = mask.bool()
        assert torch.equal(hj[valid], h[valid]), "padding leaks into valid tokens"
        assert tagsj == tags
    finally:
        hip.set_compute_dtype("fp32")


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_config5_full_size_training_step(dtype):
    """C5 at its full size as a TRAINING step: B = 128, S = 512, P = 36 -- up to 65 536 token rows, the regime
    of the 256 x 256 kernel's multi-round stream-K combine, of the grouped weight-gradient launches over tens of thousands of
    reduction rows and of the largest row x stride products -- forward + backward in both compute modes: every gradient finite,
    the step bit-deterministic, and (sentences are independent, the loss is the batch mean) every gradient of the full batch
    equal to the mean of the gradients of its four B = 32 chunks, which run other tile counts, split plans and packed row counts
    through the same kernels.  The oracle pins the shape at B = 4 (test_config5_assembled_seq512_vs_oracle)."""
    from mtvaf_amd import engine, hip
    cfg = P.BASE_BERT
    B, S, n_aux = 128, 512, 8
    hip.set_compute_dtype(dtype)
    try:
        m = _props_model(cfg, "bert-base-uncased").eval()  # (dropout off: chunked and full runs must draw the same arithmetic)
        ids, mask, tt, labels = (t.to(DEV) for t in P.text_batch(cfg, 73, B, S, lo_id=1000))
        labels[:, 0] = 9
        feats, aux, _ = (t.to(DEV) for t in _prompt_inputs(74, B, n_aux))

        def grads(sl):
            m.zero_grad(set_to_none=True)
            out = m(input_ids=ids[sl], attention_mask=mask[sl], token_type_ids=tt[sl], labels=labels[sl], images=feats[sl], aux_imgs=aux[sl])
            out.loss.backward()
            torch.cuda.synchronize()
            assert not hip.streamk_errors()
            return float(out.loss), list(out.logits), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
        full = slice(0, B)
        l1, t1, g1 = grads(full)
        rows = engine.LAST_PACK.Mp if engine.LAST_PACK is not None else B * S
        assert rows >= 24000, rows  # (a ragged S = 512 batch: tens of thousands of packed rows)
        l2, t2, g2 = grads(full)
        assert l1 == l2 and t1 == t2 and set(g1) == set(g2)
        for n in g1:
            assert bool(torch.isfinite(g1[n]).all()), n
            assert torch.equal(g1[n], g2[n]), f"{n}: two runs of the same step differ"
        del g2
        acc, lsum, tags = None, 0.0, []
        for c in range(4):
            lc, tc, gc = grads(slice(32 * c, 32 * (c + 1)))
            lsum += lc
            tags += tc
            acc = gc if acc is None else {n: acc[n] + gc[n] for n in acc}
        if dtype == "fp32":
            assert tags == t1, "decoded tags of the chunks differ from the full batch's"
        else:
            # (mixed precision: another packed row count is another tile / split plan, i.e. another fp32 summation order in front
            # of the bf16 roundings: a near-tie of the random-init emissions may flip)
            agree = sum(a == b for ta, tb in zip(tags, t1) for a, b in zip(ta, tb)) / sum(len(t) for t in t1)
            assert agree >= 0.995, agree
        assert abs(lsum / 4 - l1) <= (2e-3 if dtype == "bf16" else 1e-5) * abs(l1), (lsum / 4, l1)
        worst = 0.0
        for n in g1:
            ref = acc[n] / 4
            err = float((g1[n] - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
            worst = max(worst, err)
            # (the same products in another summation order: rounding only -- in bf16 mode the operands are the same bf16 values
            # either way, the results' fp32 sums differ as in fp32 mode, but bf16-rounded activations of the chain amplify them)
            # (measured, round 6: 6.9e-6 in fp32 mode, 3.7e-3 in bf16 mode; the kernels are deterministic)
            assert err <= (5e-3 if dtype == "bf16" else 2e-5), (n, err)
        print(f"[C5 full-size training step, {dtype}] {rows} packed rows, loss {l1:.4f}, worst gradient deviation from the "
              f"chunk mean {worst:.2e}", flush=True)
    finally:
        hip.set_compute_dtype("fp32")


def test_word_table_gradient_is_bit_reproducible():
    """Every gradient of a training step is bit-identical from run to run, the word table's included: its scatter-add (4096
    token rows into 30522 table rows; [CLS], [SEP] and repeated words share rows) is done by ONE owner wave per table row that
    sums its tokens in row order (mtvaf_embed_scatter_mode 0, the default).  The float-atomic form (mode 1, MTVAF_EMBED_ATOMIC=1)
    is kept for comparison: equal to rounding, rows touched by a single token equal bit for bit."""
    from mtvaf_amd import hip
    cfg = P.BASE_BERT
    B, S, n_aux = 32, 128, 8
    m = _props_model(cfg, "bert-base-uncased").eval()
    ids, mask, tt, labels = (t.to(DEV) for t in P.text_batch(cfg, 83, B, S, lo_id=1000))
    feats, aux, _ = (t.to(DEV) for t in _prompt_inputs(84, B, n_aux))
    w = m.bert.embeddings.word_embeddings.weight

    def grad():
        m.zero_grad(set_to_none=True)
        m(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, images=feats, aux_imgs=aux).loss.backward()
        torch.cuda.synchronize()
        return w.grad.detach().clone()
    assert hip.lib().mtvaf_embed_scatter_mode(-1) == 0
    runs = [grad() for _ in range(3)]
    for r in runs[1:]:
        assert torch.equal(r, runs[0]), "word-table gradient differs between two runs of the same step"
    valid_ids = ids[mask.bool()]
    uniq, counts = torch.unique(valid_ids, return_counts=True)
    single, shared = uniq[counts == 1], uniq[counts > 1]
    assert len(single) > 1000 and len(shared) >= 2
    untouched = torch.ones(w.shape[0], dtype=torch.bool, device=DEV)
    untouched[uniq] = False
    assert not runs[0][untouched].any(), "rows no token touches have exact-zero gradient"
    assert not runs[0][0].any(), "padding_idx row gets no gradient (nn.Embedding(padding_idx=0), modeling_bert.py:171)"
    # reference for the shared rows: the token rows' gradients summed in fp64 (dz is not exposed: compare with the atomic form)
    hip.lib().mtvaf_embed_scatter_mode(1)
    try:
        atomic = grad()
    finally:
        hip.lib().mtvaf_embed_scatter_mode(0)
    assert torch.equal(atomic[single], runs[0][single])
    scale = runs[0][shared].abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    assert float(((atomic[shared] - runs[0][shared]).abs() / scale).max()) < 1e-5
