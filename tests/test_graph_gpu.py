"""Whole-step HIP graph (mtvaf_amd.graph.GraphedTrainStep): a replay must be the eager step --
loss, decoded tags and every gradient bit for bit in eval mode (no dropout), against the CPU oracle at the BASELINE
configs[0] shape (bs 4, S 64, 3 aux crops -> P = 16), on NEW inputs copied into the captured buffers; in train mode
every replay draws fresh dropout masks from the device-side epoch, reproducibly, and consistently between the forward
and backward kernels of a replay."""
import pytest
import torch

import params as P
from oracle import mtvaf_oracle as O
from test_configs_gpu import GRADS, _assembled_case, _oracle
from test_model_gpu import DEV, build_tvnet2, close, make_args

pytestmark = pytest.mark.gpu


def _kw(text, vis):
    ids, mask, tt, labels = (t.to(DEV) for t in text)
    feats, aux, lab = (t.to(DEV) for t in vis)
    return dict(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, imagelabel=lab, images=feats, aux_imgs=aux)


def _eager(m, kw):
    m.zero_grad(set_to_none=True)
    out = m(**kw)
    out.loss.backward()
    torch.cuda.synchronize()
    return float(out.loss), list(out.logits), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}


def test_graph_replay_equals_eager_and_oracle_config1():
    from mtvaf_amd.graph import GraphedTrainStep
    cfg = P.BASE_BERT
    sde, sdh, sdp, text, vis = _assembled_case(cfg, 4, 64, 3, seed=51)
    _, _, _, text2, vis2 = _assembled_case(cfg, 4, 64, 3, seed=77)
    m = build_tvnet2(cfg, make_args(alpha=0.0), sde=sde, sdh=sdh, sdp=sdp)
    m.eval()
    kw1, kw2 = _kw(text, vis), _kw(text2, vis2)
    e1, e2 = _eager(m, kw1), _eager(m, kw2)
    g = GraphedTrainStep(m, kw1)
    try:
        for kw, (eloss, etags, egrads) in ((kw1, e1), (kw2, e2), (kw1, e1)):
            m.zero_grad(set_to_none=True)  # legal: the step re-attaches the graph's gradient tensors
            out = g(**{k: v for k, v in kw.items() if v is not None})
            torch.cuda.synchronize()
            assert float(out.loss) == eloss
            assert list(out.logits) == etags
            named = dict(m.named_parameters())
            assert set(egrads) == {n for n, p in named.items() if p.grad is not None}
            for n, ge in egrads.items():
                if "word_embeddings" in n:  # float-atomic scatter-add: order-dependent in the last bits
                    close(named[n].grad, ge, rtol=1e-4, name=n)
                else:
                    assert torch.equal(named[n].grad, ge), n
        oloss, oem, otags, ograds = _oracle(cfg, sde, sdh, sdp, text, vis, GRADS)
        assert abs(float(out.loss) - oloss) <= 1e-3 * abs(oloss)
        assert list(out.logits) == otags
        for n in GRADS:
            close(dict(m.named_parameters())[n].grad, ograds[n], rtol=3e-3, name=n)
    finally:
        g.close()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_graph_replay_draws_fresh_reproducible_dropout_masks(dtype):
    from mtvaf_amd import hip
    from mtvaf_amd.graph import GraphedTrainStep
    from test_configs_gpu import _props_model
    cfg = P.EncCfg(vocab_size=30522, hidden=768, heads=12, inter=3072, layers=2, max_pos=512)
    hip.set_compute_dtype(dtype)
    try:
        m = _props_model(cfg, "bert-base-uncased", dropout=0.1).train()
        ids, mask, tt, labels = (t.to(DEV) for t in P.text_batch(cfg, 81, 8, 128, lo_id=1000))
        from test_model_gpu import _prompt_inputs
        feats, aux, _ = (t.to(DEV) for t in _prompt_inputs(82, 8, 8))
        kw = dict(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, images=feats, aux_imgs=aux)
        g = GraphedTrainStep(m, kw)
        try:
            def run(epoch):
                g.set_epoch(epoch)
                out = g(**kw)
                torch.cuda.synchronize()
                gr = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None and "word_embeddings" not in n}
                return float(out.loss), gr
            l5, g5 = run(5)
            l6, g6 = run(6)
            l5b, g5b = run(5)
            assert l5 != l6, "two replays drew the same dropout masks"
            assert l5 == l5b
            for n in g5:
                assert torch.equal(g5[n], g5b[n]), n
            # forward and backward of a replay agree on the masks: central differences of the replayed loss along a weight
            # direction (same epoch => same masks) against the replayed gradient.  fp32 only (bf16 rounding swamps it).
            if dtype == "fp32":
                w = m.fc.weight
                d = torch.randn_like(w)
                d /= d.norm()
                ana = float((g5["fc.weight"].double() * d.double()).sum())
                eps = 1e-2
                with torch.no_grad():
                    w.add_(eps * d)
                lp, _ = run(5)
                with torch.no_grad():
                    w.add_(-2 * eps * d)
                lm, _ = run(5)
                with torch.no_grad():
                    w.add_(eps * d)
                num = (lp - lm) / (2 * eps)
                assert abs(num - ana) <= 2e-2 * max(1.0, abs(ana)), (num, ana)
        finally:
            g.close()
    finally:
        hip.set_compute_dtype("fp32")


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_graphed_training_trajectory_equals_eager(dtype):
    """Three optimizer steps with the graph (replay + eager AdamW between replays) leave the parameters where three eager
    steps leave them: the masters move between replays, so the bf16 weight images must be rebuilt inside the graph."""
    from mtvaf_amd import hip
    from mtvaf_amd.graph import GraphedTrainStep
    from mtvaf_amd.optim import AdamW
    from test_configs_gpu import _props_model
    from test_model_gpu import _prompt_inputs
    cfg = P.EncCfg(vocab_size=30522, hidden=768, heads=12, inter=3072, layers=2, max_pos=512)
    hip.set_compute_dtype(dtype)
    try:
        # eval mode: the model-level nn.Dropout modules would draw different host-seeded masks in the two models
        m1 = _props_model(cfg, "bert-base-uncased", dropout=0.0).eval()
        m2 = _props_model(cfg, "bert-base-uncased", dropout=0.0).eval()  # same seed: identical initial parameters
        ids, mask, tt, labels = (t.to(DEV) for t in P.text_batch(cfg, 81, 8, 128, lo_id=1000))
        feats, aux, _ = (t.to(DEV) for t in _prompt_inputs(82, 8, 8))
        kw = dict(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, images=feats, aux_imgs=aux)
        o1 = AdamW([p for p in m1.parameters() if p.requires_grad], lr=1e-3, model=m1, overlap=False)
        o2 = AdamW([p for p in m2.parameters() if p.requires_grad], lr=1e-3, model=m2, overlap=False)
        g = GraphedTrainStep(m2, kw)
        try:
            l1, l2 = [], []
            from mtvaf_amd import engine
            for _ in range(3):
                # (the captured step runs the padded layout -- the packed row count cannot reach the host inside a capture --
                # so the eager twin of this bit-level comparison runs padded too)
                with engine.padding_free(False):
                    out = m1(**kw)
                    out.loss.backward()
                o1.step()
                o1.zero_grad(set_to_none=True)
                l1.append(float(out.loss))
                out2 = g(**kw)
                o2.step()
                o2.zero_grad(set_to_none=True)
                l2.append(float(out2.loss))
            torch.cuda.synchronize()
            assert l1[0] == l2[0]
            assert l1[1] != l1[0], "the parameters did not move"
            for a, b in zip(l1, l2):  # (the word-table scatter-add is order-dependent in the last bits: later steps to 1e-5)
                assert abs(a - b) <= 1e-5 * abs(a), (l1, l2)
            n2 = dict(m2.named_parameters())
            for n, p in m1.named_parameters():
                if "key.bias" in n:
                    continue  # pure-noise gradient whose sign Adam amplifies (softmax is invariant to the key bias)
                torch.testing.assert_close(p, n2[n], rtol=0, atol=5e-5, msg=n)
        finally:
            g.close()
    finally:
        hip.set_compute_dtype("fp32")


def test_padding_free_execution_steps_aside_for_graph_capture():
    """engine.UNPAD reads the packed row count on the host once per step, which a capture cannot contain: with the switch on,
    GraphedTrainStep captures the PADDED step (same loss, tags and parameter gradients as the padding-free eager step) and
    leaves the switch on for eager steps afterwards."""
    from mtvaf_amd import engine
    from mtvaf_amd.graph import GraphedTrainStep
    cfg = P.BASE_BERT
    sde, sdh, sdp, text, vis = _assembled_case(cfg, 4, 64, 3, seed=53)
    m = build_tvnet2(cfg, make_args(alpha=0.0), sde=sde, sdh=sdh, sdp=sdp)
    m.eval()
    kw = _kw(text, vis)
    engine_unpad = engine.UNPAD
    try:
        engine.UNPAD = False
        ploss, ptags, pgrads = _eager(m, kw)
        engine.UNPAD = True
        uloss, utags, ugrads = _eager(m, kw)  # the eager step under the switch (packs when that saves a 128-row tile)
        g = GraphedTrainStep(m, kw)
        try:
            assert engine.UNPAD is True
            m.zero_grad(set_to_none=True)
            out = g(**{k: v for k, v in kw.items() if v is not None})
            torch.cuda.synchronize()
            assert float(out.loss) == ploss and list(out.logits) == ptags == utags  # the captured step is the padded one
            assert abs(float(out.loss) - uloss) <= 1e-5 * abs(uloss)
            named = dict(m.named_parameters())
            for n, ge in pgrads.items():
                if "word_embeddings" not in n:
                    assert torch.equal(named[n].grad, ge), n
        finally:
            g.close()
    finally:
        engine.UNPAD = engine_unpad


def test_graph_capture_refuses_an_attached_backward_hook():
    """AdamW(overlap=True) (build_optimizer's default) would apply real updates during the capture's warm-up passes."""
    import pytest as _pt
    from mtvaf_amd.graph import GraphedTrainStep
    from mtvaf_amd.optim import AdamW
    from test_optim_gpu import _batch, _model
    m, cfg = _model(layers=2)
    m.train()
    AdamW(m.parameters(), lr=1e-3, model=m, overlap=True)
    before = m.bert.encoder.layer[0].output.dense.weight.detach().clone()
    with _pt.raises(RuntimeError, match="backward hook"):
        GraphedTrainStep(m, _batch(cfg))
    assert torch.equal(before, m.bert.encoder.layer[0].output.dense.weight)
