"""mtvaf_amd.optim.AdamW (csrc/optim.hip) against torch.optim.AdamW on the reference trainer's three parameter groups
(modules/train.py:894-926), and the in-backward (overlap=True) schedule against the plain one."""
import copy
import types

import pytest
import torch

import params as P
from test_model_gpu import DEV, LABELS, _prompt_inputs, hf_config, make_args

pytestmark = pytest.mark.gpu


def _model(layers=3, dropout=0.0):
    from mtvaf_amd.models.bert_model import TVNetSAModel2
    cfg = P.EncCfg(vocab_size=600, hidden=128, heads=2, inter=256, layers=layers, max_pos=64)
    args = make_args(alpha=0.0)
    args.bert_config = hf_config(cfg, dropout=dropout)
    torch.manual_seed(11)
    return TVNetSAModel2(LABELS, None, args).to(DEV), cfg


def _batch(cfg, B=16, S=64, seed=5):
    ids, mask, tt, labels = (t.to(DEV) for t in P.text_batch(cfg, seed, B, S, lo_id=5))
    feats, aux, _ = (t.to(DEV) for t in _prompt_inputs(seed + 1, B, 3))
    return dict(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, images=feats, aux_imgs=aux)


def test_adamw_kernels_match_torch_adamw_on_the_reference_groups():
    from mtvaf_amd.optim import AdamW, linear_schedule_with_warmup, reference_param_groups
    m, cfg = _model()
    m.eval()
    m2 = copy.deepcopy(m)
    ours = AdamW(reference_param_groups(m, 3e-3), model=m)
    ref = torch.optim.AdamW(reference_param_groups(m2, 3e-3))
    s1 = linear_schedule_with_warmup(ours, 2, 10)
    s2 = linear_schedule_with_warmup(ref, 2, 10)
    batch = _batch(cfg)
    n1, n2 = dict(m.named_parameters()), dict(m2.named_parameters())
    for it in range(4):
        # both optimizers are fed the SAME gradient sequence (computed once, by m): a parameter whose true gradient is
        # zero (the key bias) has a rounding-noise gradient and Adam turns its sign into a full +-lr step, so two
        # independently evolving models would not stay comparable at 1e-6
        m(**batch).loss.backward()
        for n, p in n1.items():
            n2[n].grad = None if p.grad is None else p.grad.clone()
        for opt, sch in ((ours, s1), (ref, s2)):
            opt.step()
            sch.step()
        ours.zero_grad(set_to_none=True)
    for n, p in n1.items():
        torch.testing.assert_close(p, n2[n], rtol=1e-6, atol=1e-7, msg=n)
    # the layer-flat path was taken (one launch per layer) and its state is exposed per parameter
    w = m.bert.encoder.layer[1].intermediate.dense.weight
    st = ours.state[w]
    flat = ours._flat_state[("layer", 1)]
    assert st["exp_avg"].data_ptr() >= flat["m"].data_ptr() and st["step"] == 4 == flat["step"]
    torch.testing.assert_close(st["exp_avg"], ref.state[n2["bert.encoder.layer.1.intermediate.dense.weight"]]["exp_avg"],
                               rtol=1e-5, atol=1e-8)
    # reference quirk: projectors are in no group and never move
    assert "projectors.0.weight" in n1 and ours.state.get(n1["projectors.0.weight"], {}) == {}



@pytest.mark.parametrize("B", [16, 32])
def test_overlap_update_inside_backward_equals_the_plain_schedule(B):
    """overlap=True enqueues each layer's update from the backward hook (second stream, M >= 1024); parameters after
    3 steps must equal the plain step() schedule, in train mode with dropout (same seeds).  B = 32 (2048 token rows) is
    where the in-backward updates become background launches (128 blocks)."""
    from mtvaf_amd import engine
    from mtvaf_amd.optim import AdamW, reference_param_groups
    outs = []
    for overlap in (False, True):
        m, cfg = _model(dropout=0.1)
        m.train()
        opt = AdamW(reference_param_groups(m, 1e-3), model=m, overlap=overlap)
        batch = _batch(cfg, B=B)
        engine.RNG.offset = 0
        torch.manual_seed(3)
        losses = []
        for it in range(3):
            out = m(**batch)
            out.loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            losses.append(float(out.loss))
        torch.cuda.synchronize()
        outs.append((losses, {n: p.detach().clone() for n, p in m.named_parameters()}))
        if overlap:
            assert m.bert.encoder.grad_sink.settle_params
    # (the word-table gradient is a float-atomic scatter-add: order-dependent in the last bits, so the two runs are
    # compared to rounding noise, not bit for bit; the key bias has a pure-noise gradient whose SIGN Adam amplifies to a
    # full +-lr step -- it does not influence the function, softmax is invariant to it)
    torch.testing.assert_close(torch.tensor(outs[0][0]), torch.tensor(outs[1][0]), rtol=1e-5, atol=0)
    for n, p in outs[0][1].items():
        if "key.bias" in n:
            continue
        # Adam turns a relative gradient perturbation e into a step perturbation ~ lr * e: the crf / fc group runs at the
        # reference's lr 5e-2 (modules/train.py:911), 50x the encoder's 1e-3 here
        atol = 2.5e-4 if (n.startswith("crf") or n.startswith("fc")) else 5e-6
        torch.testing.assert_close(p, outs[1][1][n], rtol=0, atol=atol, msg=n)


def test_background_update_is_the_full_width_update():
    """max_blocks caps the grid of the per-layer launch (a background update beside MFMA-bound kernels): element-wise
    arithmetic, so parameters, moments and the bf16 shadow must be bit-identical for every grid, ragged tail included."""
    from mtvaf_amd import hip
    n = 7 * 1024 * 1024 + 13
    g0 = torch.Generator(device="cpu").manual_seed(2)
    base = [torch.randn(n, generator=g0).to(DEV) for _ in range(3)] + [torch.rand(n, generator=g0).to(DEV) * 1e-3]
    outs = []
    for blocks in (0, 128, 3):
        p, g, m, v = (t.clone() for t in base)
        sh = torch.empty(n, dtype=torch.bfloat16, device=DEV)
        for step in (1, 2):
            hip.adamw(p, g, m, v, 1e-3, 0.9, 0.999, 1e-8, 1e-2, step, p_bf16=sh, max_blocks=blocks)
        outs.append((p, m, v, sh))
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)
    # and the optimizer asks for it only where a layer's backward pass is long enough to hide it
    from mtvaf_amd.optim import AdamW
    assert AdamW.BACKGROUND_BLOCKS == 128 and AdamW.BACKGROUND_MIN_ROWS == 2048


def test_overlap_contract_violation_raises():
    from mtvaf_amd.optim import AdamW
    m, cfg = _model(layers=2)
    m.eval()
    opt = AdamW(m.parameters(), lr=1e-3, model=m, overlap=True)
    batch = _batch(cfg)
    m(**batch).loss.backward()
    m.zero_grad(set_to_none=True)
    with pytest.raises(RuntimeError, match="second backward"):
        m(**batch).loss.backward()
    opt.step()


def test_second_backward_without_zero_grad_raises_under_overlap():
    """Gradient accumulation (backward, backward, step -- no zero_grad in between, train.py:616-625) under overlap=True:
    the first pass updates the encoder from inside its backward, so the second must refuse instead of accumulating on top
    of moved weights."""
    from mtvaf_amd.optim import AdamW
    m, cfg = _model(layers=2)
    m.eval()
    opt = AdamW(m.parameters(), lr=1e-3, model=m, overlap=True)
    batch = _batch(cfg)
    m(**batch).loss.backward()
    with pytest.raises(RuntimeError, match="second backward"):
        m(**batch).loss.backward()
    opt.step()


def test_build_optimizer_with_gradient_accumulation_equals_torch_adamw():
    """build_optimizer(gradient_accumulation_steps=2) must not update from inside the backward pass: two accumulated
    micro-batches then one step, twice, equal torch.optim.AdamW on the same groups."""
    from mtvaf_amd.optim import build_optimizer, reference_param_groups
    m, cfg = _model(layers=2)
    m.eval()
    ref = copy.deepcopy(m)
    args = types.SimpleNamespace(lr=1e-3, warmup_ratio=0.0, use_prefix=True, gradient_accumulation_steps=2)
    opt, _ = build_optimizer(m, args, 100)
    assert opt.overlap is False and m.bert.encoder.grad_sink.on_layer_done is None
    topt = torch.optim.AdamW(reference_param_groups(ref, 1e-3), lr=1e-3)
    batches = [_batch(cfg, seed=5), _batch(cfg, seed=9)]
    for mod, o in ((m, opt), (ref, topt)):
        for _ in range(2):
            for b in batches:
                (mod(**b).loss / 2).backward()
            o.step()
            o.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        if "key.bias" in n or "word_embeddings" in n:
            continue  # pure-noise gradient whose sign Adam amplifies / float-atomic scatter-add
        atol = 2.5e-4 if (n.startswith("crf") or n.startswith("fc")) else 5e-6
        torch.testing.assert_close(p, q, rtol=0, atol=atol, msg=n)
    # and with accumulation 1 the in-backward updates are on
    m2, _ = _model(layers=2)
    args.gradient_accumulation_steps = 1
    opt2, _ = build_optimizer(m2, args, 100)
    assert opt2.overlap is True and m2.bert.encoder.grad_sink.on_layer_done is not None


def test_state_dict_round_trip_keeps_the_flat_moments_and_step():
    """load_state_dict into a fresh optimizer: the encoder's Adam moments and bias-correction step survive (the flat
    per-layer buffers are rebuilt from the loaded per-parameter entries); a resumed run equals the uninterrupted one."""
    from mtvaf_amd.optim import AdamW
    batch = None
    finals = []
    for resume in (False, True):
        m, cfg = _model(layers=2)
        m.eval()
        batch = batch or _batch(cfg)
        opt = AdamW(m.parameters(), lr=1e-3, model=m, overlap=True)
        for it in range(4):
            if resume and it == 2:
                sd_o, sd_m = copy.deepcopy(opt.state_dict()), copy.deepcopy(m.state_dict())
                m, _ = _model(layers=2)
                m.eval()
                m.load_state_dict(sd_m)
                opt = AdamW(m.parameters(), lr=1e-3, model=m, overlap=True)
                opt.load_state_dict(sd_o)
            m(**batch).loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        p0 = m.bert.encoder.layer[1].intermediate.dense.weight
        assert int(opt.state[p0]["step"]) == 4
        finals.append({n: p.detach().clone() for n, p in m.named_parameters()})
    for n, p in finals[0].items():
        if "key.bias" in n or "word_embeddings" in n:
            continue
        atol = 2.5e-4 if (n.startswith("crf") or n.startswith("fc")) else 5e-6
        torch.testing.assert_close(finals[1][n], p, rtol=0, atol=atol, msg=n)


def test_tensor_hooks_on_encoder_parameters_still_fire():
    """The zero-copy gradient path assigns .grad itself and bypasses AccumulateGrad; with a hook registered on an encoder
    parameter the backward must go through autograd so that the hook runs (Python-visible hooks only: torch DDP's C++ reducer
    hooks are not detectable -- INTEGRATION.md, `MTVAF_DIRECT_GRADS=0`)."""
    m, cfg = _model(layers=2)
    m.eval()
    batch = _batch(cfg)
    m(**batch).loss.backward()
    w = m.bert.encoder.layer[0].output.dense.weight
    want = w.grad.detach().clone()
    m.zero_grad(set_to_none=True)
    seen = []
    h = w.register_post_accumulate_grad_hook(lambda p: seen.append(p.grad.detach().clone()))
    m(**batch).loss.backward()
    torch.cuda.synchronize()
    h.remove()
    assert len(seen) == 1
    torch.testing.assert_close(seen[0], want, rtol=1e-5, atol=1e-7)


def test_grad_wire_format_kernels():
    """pack / reduce / unpack of the bf16 gradient exchange against torch arithmetic (bit-exact: RNE casts, fp32 sums
    in rank order)."""
    from mtvaf_amd import hip
    g = torch.Generator(device="cpu").manual_seed(0)
    n, W = 100003, 4
    src = (torch.randn(n, generator=g) * 3).to(DEV)
    chunk = ((n + W - 1) // W + 7) // 8 * 8
    send = torch.full((W * chunk,), 7.0, dtype=torch.bfloat16, device=DEV)
    hip.grad_pack_bf16(src, send, n, W * chunk)
    assert torch.equal(send[:n], src.to(torch.bfloat16)) and not send[n:].any()
    recv = (torch.randn(W * chunk, generator=g)).to(torch.bfloat16).to(DEV)
    shard = torch.empty(chunk, dtype=torch.bfloat16, device=DEV)
    hip.grad_reduce_bf16(recv, shard, W, chunk, 1.0 / W)
    acc = torch.zeros(chunk, device=DEV)
    for r in range(W):
        acc += recv[r * chunk:(r + 1) * chunk].float()
    assert torch.equal(shard, (acc * (1.0 / W)).to(torch.bfloat16))
    dst = torch.zeros(n, device=DEV)
    hip.grad_unpack_bf16(send, dst, n)
    assert torch.equal(dst, send[:n].float())


def test_adamw_rewrites_the_plane_images_of_the_weights_inside_a_flat_buffer():
    """Round 5: mtvaf_adamw_planes = mtvaf_adamw (same bits of p, m, v) + the tile-blocked plane images of the row-major matrices
    inside the flat buffer, equal to a split pass over the UPDATED weights (the pre-split operand path's weight images)."""
    from mtvaf_amd import hip
    torch.manual_seed(3)
    shapes = [(96, 64), (64, 64), (128, 64), (64, 128)]
    gaps = [96, 64 + 8, 128 + 4, 64]  # (biases / LayerNorm vectors between the matrices)
    n = sum(r * c for r, c in shapes) + sum(gaps)
    p0 = torch.randn(n, device="cuda")
    g = torch.randn(n, device="cuda") * 0.1
    m0, v0 = torch.randn(n, device="cuda") * 0.01, torch.rand(n, device="cuda") * 0.01
    segs, off = [], 0
    for (r, c), gap in zip(shapes, gaps):
        segs.append((off, r, c, torch.full((6 * r * c,), 0x7f, dtype=torch.uint8, device="cuda")))
        off += r * c + gap
    pa, ma, va = p0.clone(), m0.clone(), v0.clone()
    pb, mb, vb = p0.clone(), m0.clone(), v0.clone()
    hip.adamw(pa, g, ma, va, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 3)
    hip.adamw_planes(pb, g, mb, vb, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 3, segs)
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    for (o, r, c, img) in segs:
        want = torch.empty_like(img)
        hip.split_planes_blocked(pa[o:o + r * c].view(r, c), want)
        assert torch.equal(img, want), (o, r, c)
