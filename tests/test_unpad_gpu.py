"""Padding-free execution (engine.UNPAD): the encoder layers on the packed unmasked token rows.
Against the padded run of the SAME model on the same batch: loss, decoded tags, the last hidden state at unmasked
positions (zeros at masked ones) and every parameter gradient -- the two runs differ only in summation order (tile
plans depend on the row count) -- for ragged trailing padding and for masks with holes; and against the CPU oracle."""
import pytest
import torch

import params as P
from mtvaf_amd import engine
from test_configs_gpu import _props_model
from test_model_gpu import DEV, _prompt_inputs, close

pytestmark = pytest.mark.gpu


def _run(m, kw, unpad):
    was, engine.UNPAD = engine.UNPAD, unpad
    try:
        m.zero_grad(set_to_none=True)
        cap = {}
        hb = m.bert.register_forward_hook(lambda mod, inp, out: cap.update(out=out))
        out = m(**kw)
        hb.remove()
        out.loss.backward()
        torch.cuda.synchronize()
        packed = engine.LAST_PACK is not None
        hs = cap["out"]["hidden_states"]
        return (float(out.loss.detach()), list(out.logits), cap["out"]["last_hidden_state"].detach().clone(),
                {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}, packed,
                hs[1].detach().clone())
    finally:
        engine.UNPAD = was


@pytest.mark.parametrize("holes", [False, True])
def test_unpadded_run_equals_padded_run(holes):
    cfg = P.EncCfg(vocab_size=30522, hidden=768, heads=12, inter=3072, layers=4, max_pos=512)
    m = _props_model(cfg, "bert-base-uncased", dropout=0.0).eval()
    B, S = 16, 128
    ids, mask, tt, labels = (t.to(DEV) for t in P.text_batch(cfg, 91, B, S, lo_id=1000))
    if holes:
        g = torch.Generator().manual_seed(3)
        drop = (torch.rand(B, S, generator=g) < 0.15).to(DEV)
        drop[:, 0] = False
        mask = mask * (~drop).to(mask.dtype)
    feats, aux, _ = (t.to(DEV) for t in _prompt_inputs(92, B, 8))
    kw = dict(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, images=feats, aux_imgs=aux)
    l0, t0, h0, g0, p0, mid0 = _run(m, kw, False)
    l1, t1, h1, g1, p1, mid1 = _run(m, kw, True)
    assert not p0 and p1, "the second run did not pack"
    assert abs(l0 - l1) <= 2e-6 * abs(l0), (l0, l1)
    assert t0 == t1
    valid = mask.bool()
    close(h1[valid], h0[valid], rtol=2e-5, name="last hidden state at unmasked positions")
    assert float(h1[~valid].abs().max()) == 0.0
    close(mid1[valid], mid0[valid], rtol=2e-5, name="lazy intermediate hidden state")
    assert set(g0) == set(g1)
    for n in g0:
        if "word_embeddings" in n:  # float-atomic scatter-add
            close(g1[n], g0[n], rtol=1e-4, atol=2e-6 * float(g0[n].abs().max()), name=n)
        else:
            # (4e-6 of the largest entry: with MTVAF_F32_SPLIT=1 a product can run the split kernel at 4096 rows and the fp32
            # pipe at the packed row count -- two correct fp32 roundings of the same sums, 1.6e-6 apart at most as measured)
            close(g1[n], g0[n], rtol=2e-5, atol=4e-6 * float(g0[n].abs().max()) + 1e-9, name=n)


def test_unpadded_training_step_with_dropout_is_finite_and_deterministic():
    """Train mode (dropout live): two unpadded steps from the same state and RNG offsets give identical results, every
    gradient is finite (the rows that pad the packed image never leak into a weight-gradient sum)."""
    cfg = P.EncCfg(vocab_size=30522, hidden=768, heads=12, inter=3072, layers=2, max_pos=512)
    m = _props_model(cfg, "bert-base-uncased", dropout=0.1).train()
    ids, mask, tt, labels = (t.to(DEV) for t in P.text_batch(cfg, 93, 8, 128, lo_id=1000))
    feats, aux, _ = (t.to(DEV) for t in _prompt_inputs(94, 8, 8))
    kw = dict(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, images=feats, aux_imgs=aux)
    res = []
    for _ in range(2):
        engine.RNG.reset(1234) if hasattr(engine.RNG, "reset") else None
        torch.manual_seed(5)
        res.append(_run(m, kw, True))
    for r in res:
        assert r[4]
        assert all(torch.isfinite(g).all() for g in r[3].values())


def test_unpadded_run_in_bf16_mode_tracks_the_padded_bf16_run():
    """Mixed-precision mode: the packed run uses the same bf16 kernels on fewer rows; tile plans (and with them the fp32
    summation order in front of a bf16 rounding) depend on the row count, so the comparison carries the bf16 tolerance."""
    from mtvaf_amd import hip
    cfg = P.EncCfg(vocab_size=30522, hidden=768, heads=12, inter=3072, layers=4, max_pos=512)
    hip.set_compute_dtype("bf16")
    try:
        m = _props_model(cfg, "bert-base-uncased", dropout=0.0).eval()
        B, S = 16, 128
        ids, mask, tt, labels = (t.to(DEV) for t in P.text_batch(cfg, 95, B, S, lo_id=1000))
        feats, aux, _ = (t.to(DEV) for t in _prompt_inputs(96, B, 8))
        kw = dict(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, images=feats, aux_imgs=aux)
        l0, t0, h0, g0, p0, _ = _run(m, kw, False)
        hip.prof_start(256)
        l1, t1, h1, g1, p1, _ = _run(m, kw, True)
        recs = hip.prof_stop(256)
        assert not p0 and p1
        # (round 5) packed rows are padded to whole 256-row tiles in this mode, so that a layer's four weight gradients still go
        # out as ONE grouped stream-K launch of the 256 x 256 kernel (at 128-row granules a 2432-row batch fell back to a split
        # launch per weight gradient: C4 6368 -> 7769 sentences/s with the granule)
        enc = [k for k, _ in recs if k["la"] == 1 and k["lb"] == 1 and k["cfg"] >= 300]  # (the encoder's bf16-operand kernels)
        rows = {k["K"] for k in enc}
        assert rows and all(r % 256 == 0 for r in rows), rows
        syms = [hip.kernel_symbol(k["cfg"], k["la"], k["lb"], k["fast"]) for k in enc]
        assert syms and all(s_.startswith("gemm_bf16_p256_kernel<true, true, true>") for s_ in syms), sorted(set(syms))
        assert abs(l0 - l1) <= 5e-3 * abs(l0), (l0, l1)
        agree = sum(a == b for x, y in zip(t0, t1) for a, b in zip(x, y)) / sum(len(x) for x in t0)
        assert agree > 0.97, agree
        valid = mask.bool()
        close(h1[valid], h0[valid], rtol=3e-2, atol=3e-2 * float(h0[valid].abs().max()), name="last hidden state (bf16)")
        assert float(h1[~valid].abs().max()) == 0.0
        for n in ("bert.encoder.layer.0.attention.self.query.weight", "bert.encoder.layer.3.output.dense.weight",
                  "bert.encoder.layer.1.intermediate.dense.bias", "bert.encoder.layer.2.attention.self.key.bias", "fc.weight"):
            assert torch.isfinite(g1[n]).all()
            rel = float((g1[n] - g0[n]).norm() / g0[n].norm())
            assert rel < 5e-2, (n, rel)
    finally:
        hip.set_compute_dtype("fp32")


def test_unpadded_encoder_entry_without_the_early_packing_hook():
    """`BertModel.get_bert_output` (the Cutoff augmentation entry, reference models/modeling_bert.py:1127-1157) reaches the
    encoder without `Packing.begin`: the maps are then built on the spot (the fallback with a plain host sync)."""
    cfg = P.EncCfg(vocab_size=30522, hidden=768, heads=12, inter=3072, layers=2, max_pos=512)
    m = _props_model(cfg, "bert-base-uncased", dropout=0.0).eval()
    ids, mask, tt, _ = (t.to(DEV) for t in P.text_batch(cfg, 97, 8, 128, lo_id=1000))
    with torch.no_grad():
        emb = m.bert.get_embedding_output(ids, tt)
        with engine.padding_free(False):
            seq0, _ = m.bert.get_bert_output(emb, attention_mask=mask)
            assert engine.LAST_PACK is None
        with engine.padding_free(True):
            seq1, _ = m.bert.get_bert_output(emb, attention_mask=mask)
            assert engine.LAST_PACK is not None
    valid = mask.bool()
    close(seq1[valid], seq0[valid], rtol=2e-5, name="get_bert_output, unpadded")
    assert float(seq1[~valid].abs().max()) == 0.0


def test_unpadded_steps_with_changing_batches():
    """Consecutive steps with different batch shapes and masks (the packed row count changes every step: arenas, layouts
    and tile plans follow it) -- each step against its padded twin."""
    cfg = P.EncCfg(vocab_size=30522, hidden=768, heads=12, inter=3072, layers=2, max_pos=512)
    m = _props_model(cfg, "bert-base-uncased", dropout=0.0).eval()
    for i, (B, S) in enumerate([(8, 128), (24, 64), (5, 200), (8, 128)]):
        ids, mask, tt, labels = (t.to(DEV) for t in P.text_batch(cfg, 200 + i, B, S, lo_id=1000))
        feats, aux, _ = (t.to(DEV) for t in _prompt_inputs(300 + i, B, 3))
        kw = dict(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, images=feats, aux_imgs=aux)
        l0, t0, h0, g0, _, _ = _run(m, kw, False)
        l1, t1, h1, g1, packed, _ = _run(m, kw, True)
        assert abs(l0 - l1) <= 2e-6 * abs(l0), (i, l0, l1)
        assert t0 == t1
        for n in ("bert.encoder.layer.0.attention.self.query.weight", "bert.encoder.layer.1.output.dense.bias",
                  "bert.embeddings.position_embeddings.weight", "encoder_conv.2.weight"):
            close(g1[n], g0[n], rtol=2e-5, atol=1e-6 * float(g0[n].abs().max()) + 1e-9, name=f"step {i} {n}")


def test_unpadded_bench_workload_against_the_cpu_oracle():
    """BASELINE configs[1] (the bench workload: BERT-base 12 layers, bs 32, S = 128, 8 aux crops -> P = 36) with the encoder
    on packed rows, directly against the CPU oracle: loss 1e-3, decoded tags identical, emissions at unmasked positions,
    the gradients the padded parity test checks."""
    from test_configs_gpu import GRADS, _assembled_case, _oracle, _run_model
    from test_model_gpu import build_tvnet2, make_args
    cfg = P.BASE_BERT
    B, S, n_aux = 32, 128, 8
    sde, sdh, sdp, text, vis = _assembled_case(cfg, B, S, n_aux, seed=41)
    oloss, oem, otags, ograds = _oracle(cfg, sde, sdh, sdp, text, vis, GRADS)
    m = build_tvnet2(cfg, make_args(alpha=0.0), sde=sde, sdh=sdh, sdp=sdp)
    m.eval()
    with engine.padding_free(True):
        out, em = _run_model(m, text, vis)
        assert engine.LAST_PACK is not None
    valid = text[1].bool()
    close(em.cpu()[valid], torch.as_tensor(oem)[valid], name="emissions at unmasked positions")
    assert abs(float(out.loss) - oloss) <= 1e-3 * abs(oloss), (float(out.loss), oloss)
    assert list(out.logits) == otags, "decoded tags differ from the oracle"
    named = dict(m.named_parameters())
    for n in GRADS:
        close(named[n].grad, ograds[n], rtol=3e-3, name=n)


def test_weight_gradients_skip_padded_k_tiles_without_changing_anything():
    """Padded run (the default): the dW products skip the 32-row k-tiles of the token axis that hold only masked tokens
    (engine.SKIP_PAD_DW) -- their dY rows are exact zeros, so every parameter gradient must equal the full reduction up
    to the order in which split-K slabs are cut; loss, tags and hidden states are untouched (forward is identical)."""
    cfg = P.EncCfg(vocab_size=30522, hidden=768, heads=12, inter=3072, layers=3, max_pos=512)
    m = _props_model(cfg, "bert-base-uncased", dropout=0.0).eval()
    B, S = 32, 128
    ids, mask, tt, labels = (t.to(DEV) for t in P.text_batch(cfg, 98, B, S, lo_id=1000))
    g = torch.Generator().manual_seed(4)
    drop = (torch.rand(B, S, generator=g) < 0.1).to(DEV)
    drop[:, 0] = False
    mask = mask * (~drop).to(mask.dtype)  # (holes as well as trailing padding)
    feats, aux, _ = (t.to(DEV) for t in _prompt_inputs(99, B, 8))
    kw = dict(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, images=feats, aux_imgs=aux)
    res = []
    for skip in (False, True):
        engine.SKIP_PAD_DW = skip
        try:
            res.append(_run(m, kw, False))
        finally:
            engine.SKIP_PAD_DW = True
    (l0, t0, h0, g0, _, _), (l1, t1, h1, g1, _, _) = res
    assert l0 == l1 and t0 == t1 and torch.equal(h0, h1)
    for n in g0:
        if "word_embeddings" in n:
            close(g1[n], g0[n], rtol=1e-4, atol=2e-6 * float(g0[n].abs().max()), name=n)
        else:
            close(g1[n], g0[n], rtol=1e-5, atol=1e-6 * float(g0[n].abs().max()) + 1e-9, name=n)


def test_bf16_weight_gradients_skip_padded_k_tiles():
    """The same in the mixed-precision mode (64-row k-tiles; dY is stored as bf16: zeros stay zeros): gradients of the
    padded run with and without the k-tile list."""
    from mtvaf_amd import hip
    cfg = P.EncCfg(vocab_size=30522, hidden=768, heads=12, inter=3072, layers=3, max_pos=512)
    hip.set_compute_dtype("bf16")
    try:
        m = _props_model(cfg, "bert-base-uncased", dropout=0.0).eval()
        B, S = 32, 128
        ids, mask, tt, labels = (t.to(DEV) for t in P.text_batch(cfg, 101, B, S, lo_id=1000))
        feats, aux, _ = (t.to(DEV) for t in _prompt_inputs(102, B, 8))
        kw = dict(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, images=feats, aux_imgs=aux)
        res = []
        for skip in (False, True):
            engine.SKIP_PAD_DW, engine.SKIP_PAD_DW_BF16 = skip, skip  # (off by default in this mode: measured slower)
            try:
                res.append(_run(m, kw, False))
            finally:
                engine.SKIP_PAD_DW, engine.SKIP_PAD_DW_BF16 = True, False
        (l0, t0, h0, g0, _, _), (l1, t1, h1, g1, _, _) = res
        assert l0 == l1 and t0 == t1 and torch.equal(h0, h1)
        for n in g0:
            if "word_embeddings" in n:
                close(g1[n], g0[n], rtol=1e-4, atol=2e-6 * float(g0[n].abs().max()), name=n)
            else:  # (fp32 accumulation of the same bf16 products: only the cut of the split-K slabs moves)
                close(g1[n], g0[n], rtol=1e-5, atol=2e-6 * float(g0[n].abs().max()) + 1e-9, name=n)
    finally:
        hip.set_compute_dtype("fp32")


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_gradients_of_masked_token_rows_are_exact_zeros(dtype):
    """The premise of DESIGN 4.5b, checked on its own with the lever switched OFF: in the padded run the gradient that
    reaches a masked token row is exactly 0.0 at the encoder output AND after twelve... (here four) layers at the encoder
    input -- trailing padding and holes, dropout live -- so the terms the weight-gradient products skip are zeros."""
    from mtvaf_amd import hip
    cfg = P.EncCfg(vocab_size=30522, hidden=768, heads=12, inter=3072, layers=4, max_pos=512)
    hip.set_compute_dtype(dtype)
    engine.SKIP_PAD_DW = False
    try:
        m = _props_model(cfg, "bert-base-uncased", dropout=0.1).train()
        B, S = 16, 128
        ids, mask, tt, labels = (t.to(DEV) for t in P.text_batch(cfg, 111, B, S, lo_id=1000))
        g = torch.Generator().manual_seed(9)
        drop = (torch.rand(B, S, generator=g) < 0.1).to(DEV)
        drop[:, 0] = False
        mask = mask * (~drop).to(mask.dtype)
        feats, aux, _ = (t.to(DEV) for t in _prompt_inputs(112, B, 8))
        grads = {}
        def grab(name):
            def hook(gr):
                grads[name] = gr.detach().clone()
            return hook

        def on_embeddings(mod, inp, out):  # (a forward hook must return None, or it replaces the output)
            out.register_hook(grab("encoder input"))

        def on_encoder(mod, inp, out):
            out.last_hidden_state.register_hook(grab("encoder output"))

        h_in = m.bert.embeddings.register_forward_hook(on_embeddings)
        h_out = m.bert.encoder.register_forward_hook(on_encoder)
        out = m(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, images=feats, aux_imgs=aux)
        out.loss.backward()
        h_in.remove()
        h_out.remove()
        torch.cuda.synchronize()
        masked = ~mask.bool()
        assert set(grads) == {"encoder input", "encoder output"}
        for name, gr in grads.items():
            assert float(gr[mask.bool()].abs().max()) > 0.0
            assert float(gr[masked].abs().max()) == 0.0, f"{name}: a masked token row carries a nonzero gradient"
    finally:
        engine.SKIP_PAD_DW = True
        hip.set_compute_dtype("fp32")


@pytest.mark.parametrize("unpad", [False, True])
def test_contract_check_trips_on_a_head_that_reads_masked_positions(unpad):
    """MTVAF_CHECK_CONTRACT / engine.CHECK_CONTRACT (round 5): `BertModel.allow_unpad` is the caller's promise that nothing
    downstream reads hidden states at masked positions -- the k-tile lists, the attention backward's shortened query loops and
    padding-free execution all rest on it, and the engine cannot see the head.  With the switch on, the backward pass reads
    the incoming hidden-state gradients back and raises when a masked row is not exactly zero: a head that sums over ALL
    positions trips it (in the padded layout and in the padding-free one); the masked head (reference: the CRF of
    bert_model.py:511, 521 reads through the mask only) passes."""
    cfg = P.EncCfg(vocab_size=30522, hidden=768, heads=12, inter=3072, layers=2, max_pos=512)
    m = _props_model(cfg, "bert-base-uncased", dropout=0.0).eval()
    assert m.bert.allow_unpad
    B, S = 8, 128
    ids, mask, tt, _ = (t.to(DEV) for t in P.text_batch(cfg, 131, B, S, lo_id=1000))
    assert int((mask == 0).sum()) > 128
    was = engine.CHECK_CONTRACT
    engine.CHECK_CONTRACT = True
    try:
        with engine.padding_free(unpad):
            h = m.bert(input_ids=ids, attention_mask=mask, token_type_ids=tt)["last_hidden_state"]
            assert (engine.LAST_PACK is not None) == unpad
            (h * mask[..., None]).sum().backward()  # masked head: fine
            h = m.bert(input_ids=ids, attention_mask=mask, token_type_ids=tt)["last_hidden_state"]
            with pytest.raises(RuntimeError, match="masked-rows contract violated"):
                (h + 1.0).pow(2).sum().backward()  # reads every position
    finally:
        engine.CHECK_CONTRACT = was
        m.zero_grad(set_to_none=True)


def test_presplit_operand_path_against_the_in_kernel_split_path():
    """Round 5 (engine.F32_PLANES): on packed rows the encoder's GEMM operands are plane images read by the pre-split kernels
    (csrc/gemm_f32p.hip: forward, dX and the grouped weight gradients) instead of fp32 tensors split tile by tile.  The same
    products of the same planes in a different accumulation grouping: three optimizer steps (AdamW updates issued from inside
    backward, the weights' images rewritten behind them) track the in-kernel-split run to rounding -- loss, tags, every gradient,
    the weights after the steps; and after the masters change behind the optimizer's back (load_state_dict, an in-place update
    through a Parameter) the next forward has rebuilt the images."""
    import copy
    from mtvaf_amd import hip
    from mtvaf_amd.optim import AdamW
    if not hip.f32_split():
        pytest.skip("plane images belong to the split arithmetic")
    cfg = P.EncCfg(vocab_size=30522, hidden=768, heads=12, inter=3072, layers=2, max_pos=512)
    B, S = 16, 128
    ids, mask, tt, labels = (t.to(DEV) for t in P.text_batch(cfg, 95, B, S, lo_id=1000))
    feats, aux, _ = (t.to(DEV) for t in _prompt_inputs(96, B, 8))
    kw = dict(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, images=feats, aux_imgs=aux)
    state = copy.deepcopy(_props_model(cfg, "bert-base-uncased", dropout=0.1).state_dict())

    def run(planes):
        was_p, was_u = engine.F32_PLANES, engine.UNPAD
        engine.F32_PLANES, engine.UNPAD = planes, True
        try:
            m = _props_model(cfg, "bert-base-uncased", dropout=0.1).train()
            m.load_state_dict(state)
            opt = AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-2, model=m, overlap=True)
            out = []
            for step in range(3):
                engine.RNG.offset = 1000 * step  # (both runs draw the same dropout masks)
                torch.manual_seed(7 + step)
                o = m(**kw)
                o.loss.backward()
                grads = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
                opt.step()
                opt.zero_grad(set_to_none=True)
                out.append((float(o.loss.detach()), list(o.logits), grads))
            torch.cuda.synchronize()
            used = m.bert.encoder._stores[0].weights._pl is not None
            after = {n: p.detach().clone() for n, p in m.named_parameters() if "encoder.layer" in n}
            with torch.no_grad():
                m.bert.encoder.layer[1].intermediate.dense.weight.mul_(0.5)
            sd = {k: (v * 1.25 if k.endswith("layer.0.attention.output.dense.weight") else v) for k, v in m.state_dict().items()}
            m.load_state_dict(sd)
            m.eval()
            o = m(**kw)
            torch.cuda.synchronize()
            return out, float(o.loss.detach()), list(o.logits), used, after
        finally:
            engine.F32_PLANES, engine.UNPAD = was_p, was_u

    r1, l1, t1, used1, w1 = run(True)
    r0, l0, t0, used0, w0 = run(False)
    assert used1 and not used0, "the plane images were not built / were built with the switch off"
    for step, ((la, ta, ga), (lb, tb, gb)) in enumerate(zip(r1, r0)):
        assert abs(la - lb) <= (3e-6 if step == 0 else 2e-5) * abs(lb), (step, la, lb)
        assert ta == tb, step
        assert set(ga) == set(gb)
        for n in ga:
            # (step 0: the same weights, rounding of the products only; later steps: two trajectories one rounding apart)
            tol = (1e-4 if "word_embeddings" in n else 3e-5) * (1 if step == 0 else 10)
            close(ga[n], gb[n], rtol=tol, atol=(6e-6 if step == 0 else 6e-5) * float(gb[n].abs().max()) + 1e-9, name=f"step {step} {n}")
    assert abs(l1 - l0) <= 2e-5 * abs(l0) and t1 == t0
    for n in w0:
        # (AdamW normalises the gradient: where it is tiny, one rounding of it moves the update by a visible fraction of lr = 1e-4)
        # -- so: no element further apart than the three steps can carry a sign flip (3 lr), all but a handful within 0.3 lr
        err = (w1[n] - w0[n]).abs()
        assert float(err.max()) <= 3e-4 and float((err > 3e-5).float().mean()) <= 1e-5, (n, float(err.max()), int((err > 3e-5).sum()))



@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_weight_images_after_a_write_through_data_need_invalidation(mode):
    """the cached GEMM-operand images of the weights (plane images of the fp32 pre-split path, bf16 shadows of
    the mixed-precision mode) are checked for freshness through the optimizer epoch and the Parameters' version counters; a write
    through `p.data` moves neither.  The documented contract: such a writer calls `engine.invalidate_weight_images()`; the next
    forward then multiplies by the NEW weights (the same loss as a model that was loaded with them), and the debugging switch
    `engine.WEIGHT_IMAGES_REBUILD` (MTVAF_WEIGHT_IMAGES=rebuild) makes every forward rebuild without being told."""
    from mtvaf_amd import hip
    if mode == "fp32" and not hip.f32_split():
        pytest.skip("plane images belong to the split arithmetic")
    was_mode = hip.COMPUTE
    hip.set_compute_dtype(mode)
    try:
        cfg = P.EncCfg(vocab_size=30522, hidden=768, heads=12, inter=3072, layers=2, max_pos=512)
        B, S = 16, 128
        ids, mask, tt, labels = (t.to(DEV) for t in P.text_batch(cfg, 95, B, S, lo_id=1000))
        feats, aux, _ = (t.to(DEV) for t in _prompt_inputs(96, B, 8))
        kw = dict(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, images=feats, aux_imgs=aux)
        with engine.padding_free(True):
            m = _props_model(cfg, "bert-base-uncased", dropout=0.0).eval()
            with torch.no_grad():
                l0 = float(m(**kw).loss)
                w = m.bert.encoder.layer[1].intermediate.dense.weight
                w.data.mul_(0.5)                      # behind the version counter's back (an EMA / SWA swap looks like this)
                l_stale = float(m(**kw).loss)
                engine.invalidate_weight_images(m)
                l_new = float(m(**kw).loss)
                # the reference value: a fresh model loaded with the modified weights
                m2 = _props_model(cfg, "bert-base-uncased", dropout=0.0).eval()
                m2.load_state_dict(m.state_dict())
                l_ref = float(m2(**kw).loss)
                # the debugging switch: no call needed
                w.data.mul_(2.0)
                was = engine.WEIGHT_IMAGES_REBUILD, engine.BF16_WCACHE
                engine.WEIGHT_IMAGES_REBUILD, engine.BF16_WCACHE = True, False
                try:
                    l_back = float(m(**kw).loss)
                finally:
                    engine.WEIGHT_IMAGES_REBUILD, engine.BF16_WCACHE = was
        st = m.bert.encoder._stores[0].weights
        assert (st._pl is not None) if mode == "fp32" else (st._h is not None), "the run did not use the cached weight images"
        assert l_stale == l0, "the .data write was seen without invalidation: the caveat this test documents is gone -- update it"
        assert abs(l_new - l_ref) <= 2e-6 * abs(l_ref), (l_new, l_ref)
        assert abs(l_new - l0) > 1e-4 * abs(l0), "halving an FFN weight did not move the loss"
        assert abs(l_back - l0) <= 2e-6 * abs(l0), (l_back, l0)
    finally:
        hip.set_compute_dtype(was_mode)


@pytest.mark.gpu
def test_ordered_packing_is_placement_only():
    """Round 6: `mtvaf_build_packing_ordered` stores, behind the B + 1 row offsets, the order in which the varlen attention launches
    take the sentences (longest first, ties by index; cu[0] = -1 marks the list).  Placement only: the offsets are those of
    `mtvaf_build_packing`, the list is a permutation sorted by length, and forward + backward attention produce the same bits with
    and without it (fp32 and bf16 kernels)."""
    from mtvaf_amd import hip
    L_ = hip.lib()
    rnd = lambda *shape, seed=0: torch.randn(*shape, generator=torch.Generator().manual_seed(seed))
    B, S, Pn, NH, p = 7, 100, 36, 4, 0.1
    H = NH * 64
    lens = [S, 1, 53, 20, 77, 53, 99]
    mask = torch.zeros(B, S)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
    addmask = torch.cat([torch.zeros(B, Pn), (1 - mask) * -10000.0], 1).to(DEV).contiguous()

    def pack(fn, ncu):
        cu = torch.full((ncu,), 12345, dtype=torch.int32, device=DEV)
        inv = torch.empty(B * S, dtype=torch.int32, device=DEV)
        rowmap = torch.empty(B * S, dtype=torch.int32, device=DEV)
        mv = torch.empty(1, dtype=torch.int32, device=DEV)
        hip._ck(fn(hip._p(addmask), B, Pn + S, Pn, S, hip._p(cu), hip._p(inv), hip._p(rowmap), hip._p(mv), hip._st()), "packing")
        return cu, inv, rowmap, int(mv)
    cu0, inv0, map0, mv0 = pack(L_.mtvaf_build_packing, B + 1)
    cu1, inv1, map1, mv1 = pack(L_.mtvaf_build_packing_ordered, 2 * B + 1)
    assert mv0 == mv1 == sum(lens) and torch.equal(inv0, inv1) and torch.equal(map0, map1)
    assert int(cu0[0]) == 0 and int(cu1[0]) == -1 and torch.equal(cu0[1:], cu1[1:B + 1])
    order = cu1[B + 1:].tolist()
    assert order == sorted(range(B), key=lambda b: (-lens[b], b)), order
    Mv = mv0
    Mp = (Mv + 127) // 128 * 128 + 128
    for dtype in ("fp32", "bf16"):
        tdt = torch.float32 if dtype == "fp32" else torch.bfloat16
        qkv = torch.zeros(Mp, 3 * H, device=DEV, dtype=tdt)
        qkv[:Mv] = rnd(Mv, 3 * H, seed=81).to(DEV).to(tdt)
        pk, pv = rnd(B, Pn * H, seed=82).to(DEV).to(tdt), rnd(B, Pn * H, seed=83).to(DEV).to(tdt)
        dctx = torch.zeros(Mp, H, device=DEV, dtype=tdt)
        dctx[:Mv] = rnd(Mv, H, seed=84).to(DEV).to(tdt)
        res = []
        for cu in (cu0, cu1):
            ctx = torch.full((Mp, H), float("nan"), device=DEV, dtype=tdt)
            lse = torch.zeros(B, NH, S, device=DEV)
            dqkv = torch.full((Mp, 3 * H), float("nan"), device=DEV, dtype=tdt)
            dk, dv = torch.zeros(B, Pn * H, device=DEV), torch.zeros(B, Pn * H, device=DEV)
            if dtype == "fp32":
                de = torch.zeros(B, NH, S, device=DEV)
                hip.prefix_attn_varlen_fwd(qkv, pk, pv, cu, Mp - Mv, ctx, lse, B, S, Pn, NH, p, 11, 5)
                hip.prefix_attn_varlen_bwd(dctx, qkv, pk, pv, cu, Mp - Mv, ctx, lse, de, dqkv, dk, dv, B, S, Pn, NH, p, 11, 5)
                res.append((ctx, lse, dqkv, dk, dv, de))
            else:
                nqt, nkt = (S + 63) // 64, (Pn + S + 63) // 64
                pq, pkv = torch.zeros(B * nqt, H, device=DEV), torch.zeros(B * nkt, 2 * H, device=DEV)
                hip._ck(L_.mtvaf_prefix_attn_bf16_varlen_fwd(hip._p(qkv), hip._p(pk), hip._p(pv), hip._p(cu), Mp - Mv, hip._p(ctx), hip._p(lse),
                                                             B, S, Pn, NH, 64, p, 11, 5, hip._st()), "bf16 varlen fwd")
                hip._ck(L_.mtvaf_prefix_attn_bf16_varlen_bwd(hip._p(dctx), hip._p(qkv), hip._p(pk), hip._p(pv), hip._p(cu), Mp - Mv, hip._p(ctx),
                                                             hip._p(lse), hip._p(dqkv), hip._p(dk), hip._p(dv), hip._p(pq), hip._p(pkv), B, S, Pn,
                                                             NH, 64, p, 11, 5, hip._st()), "bf16 varlen bwd")
                res.append((ctx, lse, dqkv, dk, dv, pq, pkv))
        torch.cuda.synchronize()
        for x, y in zip(*res):
            assert torch.equal(x.view(torch.int16 if x.dtype == torch.bfloat16 else torch.int32),
                               y.view(torch.int16 if y.dtype == torch.bfloat16 else torch.int32)), dtype
