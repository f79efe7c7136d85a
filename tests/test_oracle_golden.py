"""Oracle restatement vs. the committed golden vectors (captured from the real reference modules by
tests/golden/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

import params as P
from oracle import mtvaf_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return dict(np.load(os.path.join(G, name + ".npz")))


def run_encoder(cfg, fx):
    seed, B, S, Pfx = int(fx["seed"]), int(fx["B"]), int(fx["S"]), int(fx["P"])
    sd = {k: v.clone().requires_grad_(True) for k, v in P.encoder_params(cfg, seed).items()}
    ids, mask, tt = (torch.from_numpy(fx[k]) for k in ("ids", "mask", "tt"))
    pkv = P.prefix_kv(seed + 2, cfg.layers, B, cfg.heads, Pfx)
    if pkv is not None:
        pkv = [(k.requires_grad_(True), v.requires_grad_(True)) for k, v in pkv]
    full = torch.cat([torch.ones(B, Pfx, dtype=mask.dtype), mask], 1) if Pfx else mask
    hs = O.bert_model(sd, ids, full, tt, pkv, cfg.layers, cfg.heads, cfg.eps, roberta=cfg.roberta,
                      pad_idx=cfg.pad_idx)
    return sd, pkv, hs, ids, full


@pytest.mark.parametrize("name,cfg", [("enc_tiny_bert_P0", P.TINY_BERT), ("enc_tiny_bert_P4", P.TINY_BERT),
                                      ("enc_tiny_bert_P16", P.TINY_BERT), ("enc_tiny_bert_P36", P.TINY_BERT),
                                      ("enc_tiny_roberta_P4", P.TINY_ROBERTA)])
def test_encoder_forward_backward(name, cfg):
    fx = load(name)
    sd, pkv, hs, ids, full = run_encoder(cfg, fx)
    for i, h in enumerate(hs):
        np.testing.assert_allclose(h.detach().numpy(), fx[f"h{i}"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(O.bert_pooler(sd, hs[-1]).detach().numpy(), fx["pooler"], rtol=1e-4, atol=2e-5)
    (hs[-1] * torch.from_numpy(fx["grad_seed_w"])).sum().backward()
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
        ref = fx[k]
        np.testing.assert_allclose(sd[pn].grad.numpy(), ref, rtol=2e-3, atol=2e-5 * max(1.0, np.abs(ref).max()))
    if pkv is not None:
        np.testing.assert_allclose(pkv[0][0].grad.numpy(), fx["g_pk0"], rtol=2e-3, atol=1e-5)
        np.testing.assert_allclose(pkv[0][1].grad.numpy(), fx["g_pv0"], rtol=2e-3, atol=1e-5)
        np.testing.assert_allclose(pkv[-1][0].grad.numpy(), fx["g_pkL"], rtol=2e-3, atol=1e-5)


def test_attention_probs_layer0():
    fx = load("enc_tiny_bert_P16")
    cfg = P.TINY_BERT
    sd, pkv, hs, ids, full = run_encoder(cfg, fx)
    _, probs = O.prefix_self_attention(hs[0], O.extended_attention_mask(full), sd,
                                       "encoder.layer.0.attention.self.", cfg.heads, pkv[0], return_probs=True)
    np.testing.assert_allclose(probs.detach().numpy(), fx["attn_l0"], rtol=1e-4, atol=1e-6)
    assert probs.shape[-1] == 16 + 16  # T = P + S: prefix keys in FRONT (modeling_bert.py:285)


def test_roberta_position_ids_quirk():
    # dataset pads with 0 (= <s> for RoBERTa) so padded positions keep counting
    ids = torch.tensor([[5, 6, 1, 7, 0, 0]])
    assert O.roberta_position_ids(ids, 1).tolist() == [[2, 3, 1, 4, 5, 6]]


def _prompt_inputs(seed, B, n_aux):
    rng = np.random.default_rng(seed)
    feats = torch.from_numpy(np.abs(rng.standard_normal((B, 3840, 2, 2), dtype=np.float32)))
    aux = [torch.from_numpy(np.abs(rng.standard_normal((B, 3840, 2, 2), dtype=np.float32))) for _ in range(n_aux)]
    lab = torch.softmax(torch.from_numpy(rng.standard_normal((B, 2089), dtype=np.float32)), -1)
    return feats.reshape(B, 4, -1), [a.reshape(B, 4, -1) for a in aux], lab


@pytest.mark.parametrize("name", ["prompt_novao", "prompt_vao"])
def test_visual_prompt(name):
    fx = load(name)
    seed, B, n_aux, vao = int(fx["seed"]), int(fx["B"]), int(fx["n_aux"]), bool(fx["vao"])
    sd = {k: v.requires_grad_(True) for k, v in P.prompt_params(seed).items()}
    feats, aux, lab = _prompt_inputs(seed + 1, B, n_aux)
    res, loss, auxl = O.visual_prompt(sd, feats, aux, vao=vao, imagelabel=lab)
    assert res[0][0].shape == (B, 12, 4 * (1 + n_aux), 64)
    np.testing.assert_allclose(res[0][0].detach().numpy(), fx["k0"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(res[0][1].detach().numpy(), fx["v0"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(res[7][0].detach().numpy(), fx["k7"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(res[11][1].detach().numpy(), fx["v11"], rtol=1e-4, atol=1e-5)
    sums = np.array([[float(k.double().sum()), float(v.double().sum())] for k, v in res])
    np.testing.assert_allclose(sums, fx["kv_sums"], rtol=1e-4, atol=1e-3)
    gk = torch.from_numpy(fx["grad_seed_k"])
    tot = sum(((k * gk).sum() + (v * gk).sum() * 0.5) * (1 + 0.1 * i) for i, (k, v) in enumerate(res))
    if vao:
        np.testing.assert_allclose(float(loss), float(fx["loss"]), rtol=1e-5)
        np.testing.assert_allclose([float(a) for a in auxl], fx["aux_losses"], rtol=1e-5)
        tot = tot + 3.0 * (loss + sum(auxl))
    tot.backward()
    chk = {"g_enc0_b": sd["encoder_conv.0.bias"].grad, "g_enc2_b": sd["encoder_conv.2.bias"].grad,
           "g_enc0_w_rows": sd["encoder_conv.0.weight"].grad[:4], "g_enc2_w_rows": sd["encoder_conv.2.weight"].grad[:4],
           "g_proj0_w": sd["projectors.0.weight"].grad, "g_proj11_b": sd["projectors.11.bias"].grad}
    if vao:
        chk["g_cls_b"] = sd["img_classifier.bias"].grad
        chk["g_aux2_b"] = sd["aux_img_classifier.2.bias"].grad
    for k, g in chk.items():
        ref = fx[k]
        np.testing.assert_allclose(g.numpy(), ref, rtol=2e-3, atol=2e-5 * max(1.0, np.abs(ref).max()), err_msg=k)


def test_tvnet2_end_to_end_base_dims():
    fx = load("tvnet2_base_B2S16")
    seed, B, S, n_aux = int(fx["seed"]), int(fx["B"]), int(fx["S"]), int(fx["n_aux"])
    cfg = P.BASE_BERT
    sd = {**{"bert." + k: v for k, v in P.encoder_params(cfg, seed, std=0.03).items()},
          **P.head_params(cfg, seed + 10)}
    sdp = P.prompt_params(seed + 20)
    ids, mask, tt, labels = P.text_batch(cfg, seed + 1, B, S, lo_id=1000)
    labels[:, 0] = 9
    feats, aux, lab = _prompt_inputs(seed + 2, B, n_aux)
    pk, _, _ = O.visual_prompt(sdp, feats, aux)
    loss, em, tags, hs = O.tvnet2_forward(sd, ids, mask, tt, labels, pk, cfg.layers, cfg.heads, cfg.eps)
    np.testing.assert_allclose(em.numpy(), fx["emissions"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(hs[7].numpy(), fx["h7"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(float(loss), float(fx["loss"]), rtol=1e-5)
    exp = [[int(t) for t in row if t >= 0] for row in fx["tags"]]
    assert tags == exp


def test_span_model_heads_golden():
    """TVNetSAModel (span variant) heads + loss on the 8-layer tiny encoder vs the fixture captured from the
    reference class (models/bert_model.py:246-376)."""
    fx = load("tvnet1_tiny_B3S16")
    cfg = P.TINY_BERT_L8
    seed, B, S, M = int(fx["seed"]), int(fx["B"]), int(fx["S"]), int(fx["M"])
    lengths = [int(x) for x in fx["lengths"]]
    sde = {k: v.clone().requires_grad_(True) for k, v in P.encoder_params(cfg, seed).items()}
    sdh = {k: v.clone().requires_grad_(True) for k, v in P.span_head_params(cfg, seed + 3).items()}
    ids, mask, tt, _ = P.text_batch(cfg, seed + 1, B, S, lengths)
    starts, ends, spos, epos, pol, lm = P.span_batch(cfg, seed + 2, B, S, M, lengths)
    hs = O.bert_model(sde, ids, mask, tt, None, cfg.layers, cfg.heads, cfg.eps)
    loss, logits, st, en = O.tvnet1_heads(sdh, hs[-1], mask, starts, ends, spos, epos, pol, lm)
    assert abs(float(loss) - float(fx["loss"])) < 1e-5 * abs(float(fx["loss"]))
    np.testing.assert_allclose(logits.detach().numpy(), fx["logits"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(st.detach().numpy(), fx["start_logits"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(en.detach().numpy(), fx["end_logits"], rtol=1e-4, atol=2e-5)
    loss.backward()
    for k, pn in {"g_dense_w": "dense.weight", "g_unary_w": "unary_affine.weight", "g_unary_b": "unary_affine.bias",
                  "g_binary_w": "binary_affine.weight", "g_cls_b": "classifier.bias"}.items():
        ref = fx[k]
        np.testing.assert_allclose(sdh[pn].grad.numpy(), ref, rtol=2e-3, atol=2e-5 * max(1.0, np.abs(ref).max()))
    ref = fx["g_o1_w"]
    np.testing.assert_allclose(sde[f"encoder.layer.{cfg.layers - 1}.output.dense.weight"].grad.numpy(), ref, rtol=2e-3,
                               atol=2e-5 * max(1.0, np.abs(ref).max()))
