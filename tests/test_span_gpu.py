"""Span-model (TVNetSAModel) heads on the MI355X: the fused span-pooling / distant-CE / CE kernels against the
CPU oracle (incl. the reference's clipping and all-masked edge cases) and the drop-in module against the golden
fixture captured from the reference class."""
import numpy as np
import pytest
import torch

import params as P
from oracle import mtvaf_oracle as O
from test_model_gpu import DEV, close, hf_config, load, make_args, LABELS, _prompt_inputs

pytestmark = pytest.mark.gpu


def _oracle_pool(seq, mask, starts, ends, wu, bu):
    emb, sm = O.span_representation(starts, ends, seq, mask)
    score = torch.nn.functional.linear(emb, wu, bu).squeeze(-1)
    return O.self_att_representation(emb, score, sm)


def _case(name, B, S, M, H, seed):
    rng = np.random.default_rng(seed)
    seq = torch.from_numpy(rng.standard_normal((B, S, H), dtype=np.float32))
    wu = torch.from_numpy(rng.standard_normal((1, H), dtype=np.float32) * 0.2)
    bu = torch.from_numpy(rng.standard_normal((1,), dtype=np.float32))
    lengths = rng.integers(3, S + 1, size=B)
    lengths[0] = S
    mask = (torch.arange(S)[None, :] < torch.from_numpy(lengths)[:, None]).long()
    starts = torch.zeros(B, M, dtype=torch.long)
    ends = torch.zeros(B, M, dtype=torch.long)
    for b in range(B):
        for m in range(M):
            a = int(rng.integers(0, lengths[b]))
            starts[b, m], ends[b, m] = a, min(int(lengths[b]) - 1, a + int(rng.integers(0, 6)))
    if name == "clip_past_text":     # spans of the LAST sentence running past the flattened text: clipped to the last token
        starts[-1, 0], ends[-1, 0] = int(lengths[-1]) - 1, int(lengths[-1]) + 3
        starts[-1, 1], ends[-1, 1] = int(lengths[-1]) - 2, int(lengths[-1]) + 1
    if name == "cross_sentence":     # a span leaving its sentence reads the NEXT sentence's tokens (flattened indexing)
        starts[0, 0], ends[0, 0] = S - 2, S + 2
    if name == "empty_span":         # end < start: every position masked -> plain softmax over JR clipped positions
        starts[1, 0], ends[1, 0] = 5, 2
        starts[0, 1], ends[0, 1] = 0, -1
    if name == "holes":              # non-prefix mask: compaction must follow nonzero(), not lengths
        mask[1, 1] = 0
        mask[2 % B, 0] = 0
    return seq, mask, starts, ends, wu, bu


@pytest.mark.parametrize("name,B,S,M,H", [("plain", 4, 16, 5, 128), ("plain", 32, 128, 20, 768), ("clip_past_text", 3, 12, 4, 128),
                                           ("cross_sentence", 3, 12, 4, 64), ("empty_span", 3, 12, 4, 128),
                                           ("holes", 3, 12, 4, 1024), ("plain", 1, 8, 1, 4), ("plain", 70, 9, 3, 36)])
def test_span_pool_fwd_bwd_vs_oracle(name, B, S, M, H):
    from mtvaf_amd import engine, hip
    seq, mask, starts, ends, wu, bu = _case(name, B, S, M, H, seed=B * 131 + S)
    so, wo, bo = (t.clone().requires_grad_(True) for t in (seq, wu, bu))
    ref = _oracle_pool(so, mask, starts, ends, wo, bo)
    gw = torch.from_numpy(np.random.default_rng(5).standard_normal(tuple(ref.shape), dtype=np.float32))
    (ref * gw).sum().backward()
    sg, wg, bg = (t.clone().to(DEV).requires_grad_(True) for t in (seq, wu, bu))
    index = hip.span_index(mask.to(DEV).to(torch.uint8), starts.to(DEV), ends.to(DEV))
    got = engine.SpanPoolFunction.apply(sg, wg, bg, index, M)
    close(got, ref, rtol=1e-4, name="pooled")
    (got * gw.to(DEV)).sum().backward()
    close(sg.grad, so.grad, rtol=1e-3, name="dseq")
    close(wg.grad, wo.grad, rtol=1e-3, name="dw_unary")
    close(bg.grad, bo.grad, rtol=1e-3, atol=1e-4 * float(gw.abs().sum()) * 1e-3 + 1e-5, name="db_unary")
    # deterministic: a second backward gives the same bits
    sg2 = seq.clone().to(DEV).requires_grad_(True)
    got2 = engine.SpanPoolFunction.apply(sg2, wg.detach(), bg.detach(), index, M)
    (got2 * gw.to(DEV)).sum().backward()
    assert torch.equal(sg2.grad, sg.grad)


def test_span_index_block():
    from mtvaf_amd import hip
    B, S, M = 5, 11, 3
    _, mask, starts, ends, _, _ = _case("holes", B, S, M, 8, 3)
    index = hip.span_index(mask.to(DEV).to(torch.uint8), starts.to(DEV), ends.to(DEV)).cpu()
    flat = mask.reshape(-1).nonzero().squeeze(-1)
    T = flat.numel()
    assert torch.equal(index[:T].long(), flat)
    assert int(index[-2]) == T and int(index[-1]) == min(S, int((ends - starts + 1).max()))
    woff = torch.cumsum(mask.sum(-1), 0) - mask.sum(-1)
    soff = index[2 * B * S + B: 2 * B * S + B + B * M].long()
    assert torch.equal(soff, (starts + woff[:, None]).reshape(-1))


@pytest.mark.parametrize("B,S", [(3, 16), (32, 128), (2, 512), (1, 5)])
def test_distant_ce_pair(B, S):
    from mtvaf_amd import engine
    rng = np.random.default_rng(B + S)
    z = torch.from_numpy(rng.standard_normal((B, S, 2), dtype=np.float32) * 3)
    sp = torch.from_numpy((rng.random((B, S)) > 0.8).astype(np.int64))
    ep = torch.from_numpy((rng.random((B, S)) > 0.8).astype(np.int64))
    sp[:, 1] = 1
    ep[:, 2] = 1
    zo = z.clone().requires_grad_(True)
    ref = (O.distant_cross_entropy(zo[..., 0], sp) + O.distant_cross_entropy(zo[..., 1], ep)) / 2
    (ref * 1.7).backward()
    zg = z.clone().to(DEV).requires_grad_(True)
    got = engine.DistantCEPairFunction.apply(zg, sp.to(DEV), ep.to(DEV))
    assert abs(float(got) - float(ref)) <= 1e-5 * abs(float(ref)) + 1e-6
    (got * 1.7).backward()
    close(zg.grad, zo.grad, rtol=1e-4, name="dlogits")


@pytest.mark.parametrize("N,C,ignore", [(15, 4, False), (640, 4, False), (33, 11, True), (1, 2, False)])
def test_cross_entropy(N, C, ignore):
    from mtvaf_amd import engine
    rng = np.random.default_rng(N + C)
    z = torch.from_numpy(rng.standard_normal((N, C), dtype=np.float32) * 2)
    lab = torch.from_numpy(rng.integers(0, C, size=N).astype(np.int64))
    if ignore:
        lab[::3] = -100
    zo = z.clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(zo, lab)
    (ref * 0.3).backward()
    zg = z.clone().to(DEV).requires_grad_(True)
    got = engine.CrossEntropyFunction.apply(zg, lab.to(DEV))
    assert abs(float(got) - float(ref)) <= 1e-5 * abs(float(ref)) + 1e-6
    (got * 0.3).backward()
    close(zg.grad, zo.grad, rtol=1e-4, name="dlogits")


def build_tvnet1(cfg, args, sd):
    from mtvaf_amd.models.bert_model import TVNetSAModel
    args.bert_config = hf_config(cfg)
    m = TVNetSAModel(LABELS, None, args)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    return m.to(DEV)


def test_tvnet1_matches_reference_golden():
    fx = load("tvnet1_tiny_B3S16")
    cfg = P.TINY_BERT_L8
    seed, B, S, M = int(fx["seed"]), int(fx["B"]), int(fx["S"]), int(fx["M"])
    lengths = [int(x) for x in fx["lengths"]]
    sd = {**{"bert." + k: v for k, v in P.encoder_params(cfg, seed).items()}, **P.span_head_params(cfg, seed + 3)}
    m = build_tvnet1(cfg, make_args(use_prefix=False, gcn_layer_number=0, num_layers=0), sd)
    m.eval()
    ids, mask, tt, _ = P.text_batch(cfg, seed + 1, B, S, lengths)
    starts, ends, spos, epos, pol, lm = P.span_batch(cfg, seed + 2, B, S, M, lengths)
    g = lambda t: t.to(DEV)
    out = m(input_ids=g(ids), attention_mask=g(mask), token_type_ids=g(tt), start_positions=g(spos), end_positions=g(epos),
            span_starts=g(starts), span_ends=g(ends), polarity_labels=g(pol), label_masks=g(lm))
    assert abs(float(out.loss) - float(fx["loss"])) <= 1e-3 * abs(float(fx["loss"]))
    close(out.logits, fx["logits"], name="span logits")
    assert tuple(out.logits.shape) == (B, M, 4)
    assert torch.equal(out.logits.argmax(-1).cpu(), torch.from_numpy(fx["logits"]).argmax(-1))  # predicted classes exact
    st, en, seq = m.extraction(g(mask), g(ids), None, g(tt))
    close(st, fx["start_logits"], name="start logits")
    close(en, fx["end_logits"], name="end logits")
    out.loss.backward()
    named = dict(m.named_parameters())
    for k, pn in {"g_dense_w": "dense.weight", "g_unary_w": "unary_affine.weight", "g_unary_b": "unary_affine.bias",
                  "g_binary_w": "binary_affine.weight", "g_cls_b": "classifier.bias",
                  "g_o1_w": f"bert.encoder.layer.{cfg.layers - 1}.output.dense.weight"}.items():
        close(named[pn].grad, fx[k], rtol=3e-3, name=k)


def test_tvnet1_full_size_with_prefix_vs_oracle():
    """bert-base dims, S = 128, 36 prefix slots (via get_visual_prompt), 20 candidate spans per sentence."""
    cfg = P.BASE_BERT
    B, S, M, n_aux = 4, 128, 20, 8
    seed = 77
    sde, sdh, sdp = P.encoder_params(cfg, seed, std=0.03), P.span_head_params(cfg, seed + 3), P.prompt_params(seed + 20)
    sd = {**{"bert." + k: v for k, v in sde.items()}, **sdh, **sdp}
    sd = {k: v for k, v in sd.items() if not k.startswith(("img_classifier", "aux_img_classifier"))}
    m = build_tvnet1(cfg, make_args(gcn_layer_number=0, num_layers=0), sd)
    m.eval()
    lengths = [S, 90, 40, 7]
    ids, mask, tt, _ = P.text_batch(cfg, seed + 1, B, S, lengths, lo_id=1000)
    starts, ends, spos, epos, pol, lm = P.span_batch(cfg, seed + 2, B, S, M, lengths)
    feats, aux, _ = _prompt_inputs(seed + 5, B, n_aux)
    g = lambda t: t.to(DEV)
    out = m(input_ids=g(ids), attention_mask=g(mask), token_type_ids=g(tt), start_positions=g(spos), end_positions=g(epos),
            span_starts=g(starts), span_ends=g(ends), polarity_labels=g(pol), label_masks=g(lm), images=g(feats),
            aux_imgs=g(aux))
    # oracle: prompt -> encoder -> heads
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    pkv, _, _ = O.visual_prompt(sdo, feats.reshape(B, 4, -1), [aux[:, i].reshape(B, 4, -1) for i in range(n_aux)], cfg.layers,
                          cfg.heads)
    full = torch.cat([torch.ones(B, 4 * (1 + n_aux), dtype=mask.dtype), mask], 1)
    hs = O.bert_model(sdo, ids, full, tt, pkv, cfg.layers, cfg.heads, cfg.eps, prefix="bert.")
    oloss, ologits, ost, oen = O.tvnet1_heads(sdo, hs[-1], mask, starts, ends, spos, epos, pol, lm)
    assert abs(float(out.loss) - float(oloss)) <= 1e-3 * abs(float(oloss))
    close(out.logits, ologits, name="span logits")
    assert torch.equal(out.logits.argmax(-1).cpu(), ologits.argmax(-1))
    out.loss.backward()
    oloss.backward()
    named = dict(m.named_parameters())
    for pn in ["dense.weight", "unary_affine.weight", "binary_affine.weight", "classifier.weight",
               "bert.encoder.layer.11.output.dense.weight", "bert.encoder.layer.0.attention.self.query.weight",
               "encoder_conv.2.bias", "projectors.3.weight"]:
        close(named[pn].grad, sdo[pn].grad, rtol=5e-3, name=pn)


def test_tvnet1_trains_and_rejects_unbuilt_branches():
    from mtvaf_amd.models.bert_model import TVNetSAModel
    cfg = P.EncCfg(vocab_size=500, hidden=128, heads=2, inter=256, layers=2, max_pos=64)
    args = make_args(use_prefix=False, gcn_layer_number=0, num_layers=0)
    args.bert_config = hf_config(cfg, dropout=0.1)
    torch.manual_seed(0)
    m = TVNetSAModel(LABELS, None, args).to(DEV).train()
    B, S, M = 8, 32, 6
    lengths = [S] * B
    ids, mask, tt, _ = (t.to(DEV) for t in P.text_batch(cfg, 3, B, S, lengths, lo_id=5))
    batch = [t.to(DEV) for t in P.span_batch(cfg, 4, B, S, M, lengths)]
    starts, ends, spos, epos, pol, lm = batch
    opt = torch.optim.AdamW(m.parameters(), lr=2e-3)
    losses = []
    for _ in range(40):
        out = m(input_ids=ids, attention_mask=mask, token_type_ids=tt, start_positions=spos, end_positions=epos,
                span_starts=starts, span_ends=ends, polarity_labels=pol, label_masks=lm)
        out.loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(float(out.loss))
    assert losses[-1] < 0.6 * losses[0], losses
    bad = make_args(use_prefix=False, gcn_layer_number=2, num_layers=0)
    bad.bert_config = hf_config(cfg)
    with pytest.raises(NotImplementedError):
        TVNetSAModel(LABELS, None, bad)
