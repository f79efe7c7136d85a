"""End-to-end parity of the drop-in modules (BertModel / RobertaModel / TVNetSAModel2) on the MI355X
against the golden vectors captured from the reference and against the CPU oracle."""
import os
import types

import numpy as np
import pytest
import torch
from transformers import BertConfig, RobertaConfig

import params as P
from oracle import mtvaf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return dict(np.load(os.path.join(G, name + ".npz")))


def hf_config(cfg: P.EncCfg, dropout=0.0):
    kw = dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers,
              num_attention_heads=cfg.heads, intermediate_size=cfg.inter, max_position_embeddings=cfg.max_pos,
              type_vocab_size=cfg.type_vocab, layer_norm_eps=cfg.eps, hidden_dropout_prob=dropout,
              attention_probs_dropout_prob=dropout, hidden_act="gelu")
    return RobertaConfig(pad_token_id=cfg.pad_idx, **kw) if cfg.roberta else BertConfig(pad_token_id=0, **kw)


def close(got, ref, rtol=1e-3, atol=None, name=""):
    got = torch.as_tensor(got).detach().float().cpu()
    ref = torch.as_tensor(ref).detach().float().cpu()
    if atol is None:
        atol = rtol * float(ref.abs().max()) + 1e-7
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert not bool(bad.any()), f"{name}: max err {float(err.max()):.3e} (ref max {float(ref.abs().max()):.3e}), " \
                                f"{int(bad.sum())}/{bad.numel()} bad"


def build_encoder(cfg):
    from mtvaf_amd.models.modeling_bert import BertModel
    from mtvaf_amd.models.modeling_roberta import RobertaModel
    m = (RobertaModel if cfg.roberta else BertModel)(hf_config(cfg))
    return m


@pytest.mark.parametrize("name,cfg", [("enc_tiny_bert_P0", P.TINY_BERT), ("enc_tiny_bert_P4", P.TINY_BERT),
                                      ("enc_tiny_bert_P16", P.TINY_BERT), ("enc_tiny_bert_P36", P.TINY_BERT),
                                      ("enc_tiny_roberta_P4", P.TINY_ROBERTA)])
def test_encoder_matches_reference_golden(name, cfg, f32_arith, pad_mode):
    fx = load(name)
    seed, B, S, Pfx = int(fx["seed"]), int(fx["B"]), int(fx["S"]), int(fx["P"])
    m = build_encoder(cfg)
    missing, unexpected = m.load_state_dict(P.encoder_params(cfg, seed), strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    m.to(DEV).train()
    ids, mask, tt = (torch.from_numpy(fx[k]).to(DEV) for k in ("ids", "mask", "tt"))
    pkv = P.prefix_kv(seed + 2, cfg.layers, B, cfg.heads, Pfx)
    if pkv is not None:
        pkv = [(k.to(DEV).requires_grad_(True), v.to(DEV).requires_grad_(True)) for k, v in pkv]
    full = torch.cat([torch.ones(B, Pfx, dtype=mask.dtype, device=DEV), mask], 1) if Pfx else mask
    out = m(input_ids=ids, attention_mask=full, token_type_ids=tt, past_key_values=pkv, output_attentions=True,
            output_hidden_states=True, return_dict=True)
    hs = out["hidden_states"]
    assert len(hs) == cfg.layers + 1
    for i, h in enumerate(hs):
        close(h, fx[f"h{i}"], name=f"h{i}")
    close(out["pooler_output"], fx["pooler"], name="pooler")
    (out["last_hidden_state"] * torch.from_numpy(fx["grad_seed_w"]).to(DEV)).sum().backward()
    named = dict(m.named_parameters())
    L = cfg.layers - 1
    pairs = {"g_word": "embeddings.word_embeddings.weight", "g_pos": "embeddings.position_embeddings.weight",
             "g_type": "embeddings.token_type_embeddings.weight", "g_emb_ln_w": "embeddings.LayerNorm.weight",
             "g_q0_w": "encoder.layer.0.attention.self.query.weight",
             "g_k0_b": "encoder.layer.0.attention.self.key.bias",
             "g_v1_w": f"encoder.layer.{L}.attention.self.value.weight",
             "g_ao0_w": "encoder.layer.0.attention.output.dense.weight",
             "g_ln0_w": "encoder.layer.0.attention.output.LayerNorm.weight",
             "g_ln0_b": "encoder.layer.0.attention.output.LayerNorm.bias",
             "g_i0_w": "encoder.layer.0.intermediate.dense.weight",
             "g_o0_w": "encoder.layer.0.output.dense.weight", "g_o0_b": "encoder.layer.0.output.dense.bias"}
    for k, pn in pairs.items():
        # the key bias shifts every score of a query equally: its true gradient is 0 and the reference
        # value is pure rounding noise (~1e-7), so compare it absolutely
        close(named[pn].grad, fx[k], rtol=2e-3, atol=2e-6 if k == "g_k0_b" else None, name=k)
    if pkv is not None:
        close(pkv[0][0].grad, fx["g_pk0"], rtol=2e-3, name="g_pk0")
        close(pkv[0][1].grad, fx["g_pv0"], rtol=2e-3, name="g_pv0")
        close(pkv[-1][0].grad, fx["g_pkL"], rtol=2e-3, name="g_pkL")


def _prompt_inputs(seed, B, n_aux):
    rng = np.random.default_rng(seed)
    feats = torch.from_numpy(np.abs(rng.standard_normal((B, 3840, 2, 2), dtype=np.float32)))
    aux = [torch.from_numpy(np.abs(rng.standard_normal((B, 3840, 2, 2), dtype=np.float32))) for _ in range(n_aux)]
    lab = torch.softmax(torch.from_numpy(rng.standard_normal((B, 2089), dtype=np.float32)), -1)
    return feats, torch.stack(aux, 1), lab


def make_args(**kw):
    base = dict(bert_name="bert-base-uncased", use_prefix=True, vao=False, noauxloss=True, use_probe=False, n_gpu=1,
                alpha=0.5, beta=0.0, prefix_len=4, prefix_dim=768, device=DEV, resnet_root=None, use_152=False)
    base.update(kw)
    return types.SimpleNamespace(**base)


LABELS = ["O", "B-NEU", "I-NEU", "B-POS", "I-POS", "B-NEG", "I-NEG", "X", "[CLS]", "[SEP]"]


def build_tvnet2(cfg, args, sde=None, sdh=None, sdp=None):
    from mtvaf_amd.models.bert_model import TVNetSAModel2
    args.bert_config = hf_config(cfg)
    m = TVNetSAModel2(LABELS, None, args)
    sd = {}
    if sde is not None:
        sd.update({"bert." + k: v for k, v in sde.items()})
    if sdh is not None:
        sd.update(sdh)
    if sdp is not None:
        sd.update(sdp)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    return m.to(DEV)


@pytest.mark.parametrize("name", ["prompt_novao", "prompt_vao"])
def test_visual_prompt_matches_reference_golden(name):
    fx = load(name)
    seed, B, n_aux, vao = int(fx["seed"]), int(fx["B"]), int(fx["n_aux"]), bool(fx["vao"])
    args = make_args(vao=vao)
    m = build_tvnet2(P.EncCfg(vocab_size=64, hidden=768, heads=12, inter=128, layers=12, max_pos=32), args,
                     sdp=P.prompt_params(seed))
    m.eval()  # VAO dropout(0.2) off, as in the fixture
    feats, aux, lab = _prompt_inputs(seed + 1, B, n_aux)
    res, loss, auxl = m.get_visual_prompt(feats.to(DEV), aux.to(DEV), lab.to(DEV))
    assert len(res) == 12 and tuple(res[0][0].shape) == (B, 12, 4 * (1 + n_aux), 64)
    close(res[0][0], fx["k0"], name="k0")
    close(res[0][1], fx["v0"], name="v0")
    close(res[7][0], fx["k7"], name="k7")
    close(res[11][1], fx["v11"], name="v11")
    gk = torch.from_numpy(fx["grad_seed_k"]).to(DEV)
    tot = sum(((k * gk).sum() + (v * gk).sum() * 0.5) * (1 + 0.1 * i) for i, (k, v) in enumerate(res))
    if vao:
        close(loss, fx["loss"], rtol=1e-4, name="vao loss")
        close(torch.stack(auxl), fx["aux_losses"], rtol=1e-4, name="vao aux")
        tot = tot + 3.0 * (loss + sum(auxl))
    tot.backward()
    named = dict(m.named_parameters())
    chk = {"g_enc0_b": named["encoder_conv.0.bias"].grad, "g_enc2_b": named["encoder_conv.2.bias"].grad,
           "g_enc0_w_rows": named["encoder_conv.0.weight"].grad[:4],
           "g_enc2_w_rows": named["encoder_conv.2.weight"].grad[:4],
           "g_proj0_w": named["projectors.0.weight"].grad, "g_proj11_b": named["projectors.11.bias"].grad}
    if vao:
        chk["g_cls_b"] = named["img_classifier.bias"].grad
        chk["g_aux2_b"] = named["aux_img_classifier.2.bias"].grad
    for k, g in chk.items():
        close(g, fx[k], rtol=3e-3, name=k)


def test_tvnet2_matches_reference_golden_base_dims(f32_arith, pad_mode):
    fx = load("tvnet2_base_B2S16")
    seed, B, S, n_aux = int(fx["seed"]), int(fx["B"]), int(fx["S"]), int(fx["n_aux"])
    cfg = P.BASE_BERT
    m = build_tvnet2(cfg, make_args(), sde=P.encoder_params(cfg, seed, std=0.03), sdh=P.head_params(cfg, seed + 10),
                     sdp=P.prompt_params(seed + 20))
    m.eval()
    ids, mask, tt, labels = P.text_batch(cfg, seed + 1, B, S, lo_id=1000)
    labels[:, 0] = 9
    feats, aux, lab = _prompt_inputs(seed + 2, B, n_aux)
    captured = {}
    h = m.bert.register_forward_hook(lambda mod, inp, out: captured.update(out=out))
    decode = m.crf.decode_deferred

    def spy(em_, mask_u8):  # the emissions the HIP `fc` product hands to the CRF kernels (not a recomputation)
        captured["em"] = em_.detach().clone()
        return decode(em_, mask_u8)
    m.crf.decode_deferred = spy
    out = m(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), token_type_ids=tt.to(DEV), labels=labels.to(DEV),
            imagelabel=lab.to(DEV), images=feats.to(DEV), aux_imgs=aux.to(DEV))
    h.remove()
    del m.crf.decode_deferred
    torch.cuda.synchronize()
    close(captured["out"]["last_hidden_state"], fx["h12"], name="h12")
    close(captured["out"]["hidden_states"][7], fx["h7"], name="h7")
    close(captured["em"], fx["emissions"], name="emissions")
    assert abs(float(out.loss) - float(fx["loss"])) <= 1e-3 * abs(float(fx["loss"]))
    exp = [[int(t) for t in row if t >= 0] for row in fx["tags"]]
    assert out.logits == exp  # predicted-class indices bit-exact
    out.loss.backward()
    named = dict(m.named_parameters())
    close(named["fc.weight"].grad, fx["g_fc_w"], rtol=3e-3, name="g_fc_w")
    close(named["crf.transitions"].grad, fx["g_trans"], rtol=3e-3, name="g_trans")
    close(named["bert.encoder.layer.11.attention.self.query.bias"].grad, fx["g_q11_b"], rtol=3e-3, name="g_q11_b")
    close(named["encoder_conv.2.bias"].grad, fx["g_enc2_b"], rtol=3e-3, name="g_enc2_b")


def test_full_size_step_vs_oracle_and_grad_sink(f32_arith, pad_mode):
    """BASELINE config-2 shape (S=128, P=36) at B=4 against the CPU oracle: emissions/loss 1e-3,
    tags bit-exact; gradients land in the flat per-layer buffers without a copy."""
    cfg = P.BASE_BERT
    B, S, Pn = 4, 128, 36
    sde, sdh = P.encoder_params(cfg, 7, std=0.03), P.head_params(cfg, 8)
    m = build_tvnet2(cfg, make_args(use_prefix=False), sde=sde, sdh=sdh)
    m.eval()
    ids, mask, tt, labels = P.text_batch(cfg, 9, B, S, lo_id=1000)
    labels[:, 0] = 9
    pkv = P.prefix_kv(10, cfg.layers, B, cfg.heads, Pn, std=0.5)
    sd = {**{"bert." + k: v for k, v in sde.items()}, **sdh}
    oloss, oem, otags, ohs = O.tvnet2_forward(sd, ids, mask, tt, labels, pkv, cfg.layers, cfg.heads, cfg.eps)
    full = torch.cat([torch.ones(B, Pn, dtype=mask.dtype), mask], 1).to(DEV)
    gp = [(k.to(DEV), v.to(DEV)) for k, v in pkv]
    bo = m.bert(input_ids=ids.to(DEV), attention_mask=full, token_type_ids=tt.to(DEV), past_key_values=gp,
                output_hidden_states=True)
    from mtvaf_amd import engine
    valid = mask.bool().to(DEV)
    if engine.LAST_PACK is not None:  # padding-free: zeros at masked positions instead of the reference's don't-care values
        assert pad_mode == "unpad"
        assert float(bo["last_hidden_state"][~valid].abs().max()) == 0.0
    else:
        close(bo["last_hidden_state"], ohs[-1], name="last hidden")
    close(bo["last_hidden_state"][valid], ohs[-1][valid.cpu()], name="last hidden at unmasked positions")
    em = torch.nn.functional.linear(bo["last_hidden_state"], m.fc.weight, m.fc.bias)
    close(em[valid], oem[valid.cpu()], name="emissions")
    mask_u8 = mask.to(DEV).to(torch.uint8)
    assert m.crf.decode(em, mask_u8) == otags
    loss = -m.crf(em, labels.to(DEV), mask=mask_u8, reduction="mean")
    assert abs(float(loss) - float(oloss)) <= 1e-3 * abs(float(oloss))
    loss.backward()
    st = m.bert.encoder._stores[3]
    g = m.bert.encoder.layer[3].intermediate.dense.weight.grad
    assert st.grad is not None and st.grad.data_ptr() <= g.data_ptr() < st.grad.data_ptr() + st.grad.numel() * 4, \
        "parameter gradient was copied instead of adopted from the flat layer buffer"
    # second backward without zero_grad must ACCUMULATE (fallback path)
    g1 = g.clone()
    bo = m.bert(input_ids=ids.to(DEV), attention_mask=full, token_type_ids=tt.to(DEV), past_key_values=gp)
    em = torch.nn.functional.linear(bo["last_hidden_state"], m.fc.weight, m.fc.bias)
    (-m.crf(em, labels.to(DEV), mask=mask_u8, reduction="mean")).backward()
    close(m.bert.encoder.layer[3].intermediate.dense.weight.grad, 2 * g1, rtol=1e-4, name="grad accumulation")


def test_training_reduces_loss_and_dropout_is_live():
    cfg = P.EncCfg(vocab_size=500, hidden=128, heads=2, inter=256, layers=2, max_pos=64)
    from mtvaf_amd.models.bert_model import TVNetSAModel2
    args = make_args(use_prefix=False)
    args.bert_config = hf_config(cfg, dropout=0.1)
    torch.manual_seed(0)
    m = TVNetSAModel2(LABELS, None, args).to(DEV)
    ids, mask, tt, labels = (t.to(DEV) for t in P.text_batch(cfg, 3, 16, 32, lo_id=5))
    opt = torch.optim.AdamW(m.parameters(), lr=3e-3)
    m.train()
    a = m(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels).loss
    b = m(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels).loss
    assert float(a) != float(b), "dropout masks must differ between forwards in train mode"
    losses = []
    for _ in range(30):
        out = m(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels)
        out.loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(float(out.loss))
    assert losses[-1] < 0.5 * losses[0], losses
    m.eval()
    e1 = m(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels).loss
    e2 = m(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels).loss
    assert float(e1) == float(e2)


def test_long_sequence_config5_shape_vs_oracle():
    """BASELINE config-5 geometry (S = 512, P = 36) at B = 2: several key/query tiles, prefix + ragged lengths."""
    cfg = P.EncCfg(vocab_size=2000, hidden=768, heads=12, inter=3072, layers=2, max_pos=512)
    B, S, Pn = 2, 512, 36
    sde = P.encoder_params(cfg, 21, std=0.03)
    m = build_encoder(cfg)
    m.load_state_dict(sde, strict=False)
    m.to(DEV).train()
    ids, mask, tt, _ = P.text_batch(cfg, 22, B, S, lengths=[512, 301], lo_id=5)
    pkv = P.prefix_kv(23, cfg.layers, B, cfg.heads, Pn, std=0.5)
    full = torch.cat([torch.ones(B, Pn, dtype=mask.dtype), mask], 1)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sde.items()}
    ohs = O.bert_model(sdg, ids, full, tt, pkv, cfg.layers, cfg.heads, cfg.eps)
    gw = torch.randn(B, S, cfg.hidden, generator=torch.Generator().manual_seed(5))
    (ohs[-1] * gw).sum().backward()
    gp = [(k.to(DEV).requires_grad_(True), v.to(DEV).requires_grad_(True)) for k, v in pkv]
    out = m(input_ids=ids.to(DEV), attention_mask=full.to(DEV), token_type_ids=tt.to(DEV), past_key_values=gp,
            output_hidden_states=True)
    close(out["last_hidden_state"], ohs[-1], name="S512 last hidden")
    (out["last_hidden_state"] * gw.to(DEV)).sum().backward()
    named = dict(m.named_parameters())
    for n in ("encoder.layer.0.attention.self.key.weight", "encoder.layer.1.output.dense.weight",
              "embeddings.position_embeddings.weight", "encoder.layer.0.attention.output.LayerNorm.weight"):
        close(named[n].grad, sdg[n].grad, rtol=3e-3, name=n)


def test_roberta_base_dims_and_single_sentence():
    """RoBERTa-base geometry (vocab 50265, 514 positions, eps 1e-5, pad id 1) with B = 1 and a short sequence."""
    cfg = P.EncCfg(vocab_size=50265, hidden=768, heads=12, inter=3072, layers=2, max_pos=514, type_vocab=1, eps=1e-5,
                   roberta=True, pad_idx=1)
    sde = P.encoder_params(cfg, 31, std=0.03)
    m = build_encoder(cfg)
    m.load_state_dict(sde, strict=False)
    m.to(DEV).eval()
    ids = torch.tensor([[0, 713, 16, 1, 10, 1296, 2, 0, 0, 0, 0]])  # a <pad>=1 inside, zeros (= <s>) as the dataset pads
    mask = torch.tensor([[1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0]])
    tt = torch.zeros_like(ids)
    ohs = O.bert_model(sde, ids, mask, tt, None, cfg.layers, cfg.heads, cfg.eps, roberta=True, pad_idx=1)
    out = m(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), token_type_ids=tt.to(DEV), output_hidden_states=True)
    close(out["hidden_states"][0], ohs[0], name="roberta embeddings")
    close(out["last_hidden_state"], ohs[-1], name="roberta last hidden")
    emb = m.get_embedding_output(ids.to(DEV), tt.to(DEV))
    seq, pooled = m.get_bert_output(emb, attention_mask=mask.to(DEV))
    close(seq, ohs[-1], name="get_bert_output")
    close(pooled, O.bert_pooler(sde, ohs[-1]), name="pooler")


def test_backward_is_deterministic_and_eval_has_no_grad_overhead():
    cfg = P.EncCfg(vocab_size=300, hidden=128, heads=2, inter=256, layers=2, max_pos=64)
    m = build_tvnet2(cfg, make_args(use_prefix=False), sde=P.encoder_params(cfg, 1), sdh=P.head_params(cfg, 2))
    m.eval()
    ids, mask, tt, labels = (t.to(DEV) for t in P.text_batch(cfg, 3, 6, 40, lo_id=5))
    grads = []
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        m(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels).loss.backward()
        grads.append({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    for n in grads[0]:
        if "word_embeddings" in n:
            continue  # scatter-add with float atomics: order-dependent in the last bits
        assert torch.equal(grads[0][n], grads[1][n]), n
    with torch.no_grad():
        out = m(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels)
    assert out.loss.grad_fn is None and len(out.logits) == 6
    assert [len(t) for t in out.logits] == mask.sum(1).tolist()


def test_bf16_compute_mode_tracks_the_fp32_oracle():
    """BASELINE configs 3-4 arithmetic (bf16 MFMA, fp32 accumulation / storage / statistics) against the fp32
    oracle at BERT-base width: bf16 operand rounding (2^-9 relative) bounds the deviation."""
    from mtvaf_amd import hip
    cfg = P.EncCfg(vocab_size=3000, hidden=768, heads=12, inter=3072, layers=4, max_pos=128)
    B, S, Pn = 4, 64, 16
    sde, sdh = P.encoder_params(cfg, 7, std=0.03), P.head_params(cfg, 8)
    m = build_tvnet2(cfg, make_args(use_prefix=False), sde=sde, sdh=sdh)
    m.eval()
    ids, mask, tt, labels = P.text_batch(cfg, 9, B, S, lo_id=100)
    labels[:, 0] = 9
    pkv = P.prefix_kv(10, cfg.layers, B, cfg.heads, Pn, std=0.5)
    sd = {**{"bert." + k: v.clone().requires_grad_(True) for k, v in sde.items()}, **sdh}
    oloss, oem, otags, ohs = O.tvnet2_forward(sd, ids, mask, tt, labels, pkv, cfg.layers, cfg.heads, cfg.eps)
    oloss.backward()
    full = torch.cat([torch.ones(B, Pn, dtype=mask.dtype), mask], 1).to(DEV)
    gp = [(k.to(DEV), v.to(DEV)) for k, v in pkv]
    hip.set_compute_dtype("bf16")
    try:
        bo = m.bert(input_ids=ids.to(DEV), attention_mask=full, token_type_ids=tt.to(DEV), past_key_values=gp)
        from mtvaf_amd import engine
        em = engine.LinearFunction.apply(bo["last_hidden_state"], m.fc.weight, m.fc.bias, False)
        mask_u8 = mask.to(DEV).to(torch.uint8)
        loss = -m.crf(em, labels.to(DEV), mask=mask_u8, reduction="mean")
        loss.backward()
        tags = m.crf.decode(em, mask_u8)
    finally:
        hip.set_compute_dtype("fp32")
    rel = float((bo["last_hidden_state"].cpu() - ohs[-1]).norm() / ohs[-1].norm())
    assert rel < 2e-2, rel
    assert abs(float(loss) - float(oloss)) <= 2e-2 * abs(float(oloss)), (float(loss), float(oloss))
    agree = sum(a == b for ta, tb in zip(tags, otags) for a, b in zip(ta, tb)) / sum(len(t) for t in otags)
    assert agree > 0.9, agree
    g = m.bert.encoder.layer[1].intermediate.dense.weight.grad.cpu()
    go = sd["bert.encoder.layer.1.intermediate.dense.weight"].grad
    assert float((g - go).norm() / go.norm()) < 5e-2


def test_bf16_operand_path_matches_the_fp32_operand_bf16_kernels():
    """Same arithmetic model (operands rounded to bf16, fp32 accumulation), two implementations: the bf16-operand
    DMA kernels fed by cast / transpose passes vs the kernels that round fp32 tiles while staging them.  Loss and
    every encoder gradient must agree to accumulation-order noise."""
    from mtvaf_amd import engine, hip
    cfg = P.EncCfg(vocab_size=3000, hidden=768, heads=12, inter=3072, layers=3, max_pos=128)
    B, S, Pn = 4, 128, 36
    m = build_tvnet2(cfg, make_args(use_prefix=False), sde=P.encoder_params(cfg, 17, std=0.03), sdh=P.head_params(cfg, 18))
    m.eval()
    ids, mask, tt, labels = (t.to(DEV) for t in P.text_batch(cfg, 19, B, S, lo_id=100))
    labels[:, 0] = 9
    gp = [(k.to(DEV), v.to(DEV)) for k, v in P.prefix_kv(20, cfg.layers, B, cfg.heads, Pn, std=0.5)]
    full = torch.cat([torch.ones(B, Pn, dtype=mask.dtype, device=DEV), mask], 1)
    res = {}
    hip.set_compute_dtype("bf16")
    was = engine.BF16_OPERANDS
    try:
        for flag in (True, False):
            engine.BF16_OPERANDS = flag
            m.zero_grad(set_to_none=True)
            hs = m.bert(input_ids=ids, attention_mask=full, token_type_ids=tt, past_key_values=gp)["last_hidden_state"]
            em = torch.nn.functional.linear(hs, m.fc.weight, m.fc.bias)
            loss = -m.crf(em, labels, mask=mask.to(torch.uint8), reduction="mean")
            loss.backward()
            res[flag] = (float(loss), hs.detach().clone(), {n: p.grad.clone() for n, p in m.bert.named_parameters()
                                                            if p.grad is not None})
    finally:
        engine.BF16_OPERANDS = was
        hip.set_compute_dtype("fp32")
    # a 1e-6 difference upstream flips individual bf16 roundings downstream (2^-9 each), so the two runs agree to a
    # few 1e-3 in norm, not element by element; a wrong operand or transpose would be an O(1) error
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
    assert abs(res[True][0] - res[False][0]) <= 1e-3 * abs(res[False][0])
    assert rel(res[True][1], res[False][1]) < 5e-3
    assert len(res[True][2]) == len(res[False][2]) > 40
    for n, g in res[False][2].items():
        if float(g.abs().max()) > 1e-6:   # (the key bias has a true gradient of 0: pure rounding noise)
            assert rel(res[True][2][n], g) < 2e-2, (n, rel(res[True][2][n], g))


def test_deferred_tags_behave_like_the_eager_list():
    cfg = P.EncCfg(vocab_size=300, hidden=128, heads=2, inter=256, layers=2, max_pos=64)
    m = build_tvnet2(cfg, make_args(use_prefix=False), sde=P.encoder_params(cfg, 1), sdh=P.head_params(cfg, 2))
    m.eval()
    ids, mask, tt, labels = (t.to(DEV) for t in P.text_batch(cfg, 3, 6, 40, lo_id=5))
    out = m(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels)
    from mtvaf_amd.modules.crf import DeferredTags
    assert isinstance(out.logits, list) and isinstance(out.logits, DeferredTags)
    em_hook = {}
    h = m.fc.register_forward_hook(lambda *a: None)
    h.remove()
    bo = m.bert(input_ids=ids, attention_mask=mask, token_type_ids=tt)
    em = torch.nn.functional.linear(bo["last_hidden_state"], m.fc.weight, m.fc.bias)
    eager = m.crf.decode(em, mask.to(torch.uint8))
    assert len(out.logits) == 6 and out.logits == eager and list(out.logits) == eager
    assert out.logits[2][1] == eager[2][1] and [len(t) for t in out.logits] == mask.sum(1).tolist()
    import copy, json
    assert json.loads(json.dumps(out.logits)) == eager and copy.deepcopy(out.logits) == eager


def test_crf_module_nll_mean_is_the_negated_mean_log_likelihood():
    """``crf.nll_mean`` (one autograd node, used by the model's forward) against the public torchcrf-style call the
    reference makes, ``-1 * crf(emissions, tags, mask=mask, reduction='mean')`` (models/bert_model.py:521): same value,
    same gradients, bit for bit."""
    from mtvaf_amd.modules.crf import CRF
    torch.manual_seed(3)
    crf = CRF(11, batch_first=True).to(DEV)
    em = torch.randn(6, 40, 11, device=DEV, requires_grad=True)
    tags = torch.randint(0, 11, (6, 40), device=DEV)
    mask = (torch.arange(40, device=DEV)[None] < torch.tensor([40, 33, 1, 17, 40, 8], device=DEV)[:, None]).to(torch.uint8)
    a = -1 * crf(em, tags, mask=mask, reduction="mean")
    ga = torch.autograd.grad(a, [em, crf.transitions, crf.start_transitions, crf.end_transitions])
    b = crf.nll_mean(em, tags, mask=mask)
    gb = torch.autograd.grad(b, [em, crf.transitions, crf.start_transitions, crf.end_transitions])
    assert torch.equal(a, b)
    for x, y in zip(ga, gb):
        assert torch.equal(x, y)
