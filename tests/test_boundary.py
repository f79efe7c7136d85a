"""Drop-in boundary checks that need no GPU: parameter names / optimizer grouping of the reference
trainer (modules/train.py:894-926), state_dict round trip, loud failure without a GPU."""
import types

import pytest
import torch
from transformers import BertConfig

LABELS = ["O", "B-NEU", "I-NEU", "B-POS", "I-POS", "B-NEG", "I-NEG", "X", "[CLS]", "[SEP]"]


def tiny_model(use_prefix=True):
    from mtvaf_amd.models.bert_model import TVNetSAModel2
    cfg = BertConfig(vocab_size=64, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                     max_position_embeddings=64)
    args = types.SimpleNamespace(bert_name="bert-base-uncased", bert_config=cfg, use_prefix=use_prefix, vao=False,
                                 noauxloss=True, use_probe=False, n_gpu=1, alpha=0.0, prefix_len=4, prefix_dim=768,
                                 device="cpu", resnet_root=None, use_152=False)
    return TVNetSAModel2(LABELS, None, args)


def test_parameter_names_match_reference_checkpoint_layout():
    m = tiny_model()
    names = [n for n, _ in m.named_parameters()]
    must = ["bert.embeddings.word_embeddings.weight", "bert.embeddings.position_embeddings.weight",
            "bert.embeddings.token_type_embeddings.weight", "bert.embeddings.LayerNorm.weight",
            "bert.embeddings.LayerNorm.bias", "bert.encoder.layer.0.attention.self.query.weight",
            "bert.encoder.layer.0.attention.self.key.bias", "bert.encoder.layer.1.attention.self.value.weight",
            "bert.encoder.layer.0.attention.output.dense.weight", "bert.encoder.layer.0.attention.output.LayerNorm.bias",
            "bert.encoder.layer.1.intermediate.dense.weight", "bert.encoder.layer.1.output.dense.bias",
            "bert.encoder.layer.1.output.LayerNorm.weight", "bert.pooler.dense.weight", "fc.weight", "fc.bias",
            "crf.start_transitions", "crf.end_transitions", "crf.transitions", "encoder_conv.0.weight",
            "encoder_conv.2.bias", "projectors.0.weight", "projectors.1.bias", "img_classifier.weight",
            "aux_img_classifier.2.bias"]
    for n in must:
        assert n in names, n
    assert "bert.embeddings.position_ids" in m.state_dict()
    assert m.num_labels == 11 and m.fc.out_features == 11


def test_reference_trainer_param_groups_by_name():
    """Mirrors the name matching of modules/train.py:894-921 against our module."""
    m = tiny_model()
    bert = [n for n, _ in m.named_parameters() if "bert" in n]
    conv = [n for n, _ in m.named_parameters() if "encoder_conv" in n or "gates" in n]
    head = [n for n, _ in m.named_parameters() if "crf" in n or n.startswith("fc")]
    assert len(bert) == 5 + 16 * 2 + 2 and len(conv) == 4 and len(head) == 5
    assert not (set(bert) & set(conv)) and not (set(bert) & set(head))


def test_state_dict_round_trip_and_qkv_packing_survives():
    m, m2 = tiny_model(), tiny_model()
    m2.load_state_dict(m.state_dict())
    for (n1, p1), (n2, p2) in zip(m.named_parameters(), m2.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2)
    enc = m.bert.encoder
    stores, _ = enc._prepare()
    q = enc.layer[0].attention.self.query.weight
    k = enc.layer[0].attention.self.key.weight
    assert k.data_ptr() == q.data_ptr() + q.numel() * 4, "Q/K/V must be views of one packed [3H,H] operand"
    before = q.detach().clone()
    m.load_state_dict(m2.state_dict())  # in-place copy keeps the packing
    assert stores[0].valid(enc.layer[0]) and torch.equal(q, before)
    sd = m.state_dict()
    assert sd["bert.encoder.layer.0.attention.self.query.weight"].shape == (128, 128)


def test_no_cpu_fallback():
    m = tiny_model(use_prefix=False)
    ids = torch.randint(3, 60, (2, 8))
    with pytest.raises((RuntimeError, AssertionError)):
        m(input_ids=ids, attention_mask=torch.ones(2, 8, dtype=torch.long), token_type_ids=torch.zeros_like(ids),
          labels=torch.ones(2, 8, dtype=torch.long))


def test_span_model_parameter_names_and_surface():
    """TVNetSAModel (span variant) keeps the reference's module names (models/bert_model.py:205-232) and methods."""
    from mtvaf_amd.models.bert_model import TVNetSAModel, flatten, reconstruct
    cfg = BertConfig(vocab_size=64, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                     max_position_embeddings=64)
    args = types.SimpleNamespace(bert_name="bert-base-uncased", bert_config=cfg, use_prefix=True, use_probe=False, n_gpu=1,
                                 prefix_len=4, prefix_dim=768, device="cpu", resnet_root=None, use_152=False,
                                 gcn_layer_number=0, num_layers=0)
    m = TVNetSAModel(LABELS, None, args)
    names = {n for n, _ in m.named_parameters()}
    for n in ["dense.weight", "dense.bias", "unary_affine.weight", "unary_affine.bias", "binary_affine.weight",
              "binary_affine.bias", "classifier.weight", "classifier.bias", "fc.weight", "encoder_conv.0.weight",
              "encoder_conv.2.bias", "projectors.1.weight", "bert.encoder.layer.1.output.dense.weight"]:
        assert n in names, n
    assert m.classifier.out_features == 4 and m.unary_affine.out_features == 1 and m.binary_affine.out_features == 2
    assert not any(n.startswith(("crf", "img_classifier")) for n in names)
    for meth in ("forward", "extraction", "classification", "get_visual_prompt"):
        assert callable(getattr(m, meth))
    x = torch.arange(24).view(2, 3, 4)
    assert flatten(x).shape == (6, 4) and flatten(x[..., 0]).shape == (6,)
    assert reconstruct(flatten(x), x[..., 0]).shape == (2, 3, 4)
    with pytest.raises(RuntimeError):  # no GPU here: the product path must refuse, not fall back
        ids = torch.ones(1, 4, dtype=torch.long)
        m.extraction(torch.ones(1, 4, dtype=torch.long), ids, None, torch.zeros_like(ids))


def test_reference_optimizer_groups_and_schedule():
    """mtvaf_amd.optim mirrors modules/train.py:887-926: groups by name, lr 5e-2 for crf/fc, projectors in no
    group (reference quirk), linear warm-up schedule equal to transformers' implementation."""
    from transformers import get_linear_schedule_with_warmup
    from mtvaf_amd.optim import build_optimizer, reference_param_groups
    m = tiny_model()
    groups = reference_param_groups(m, 3e-5)
    assert [len(g["params"]) for g in groups] == [5 + 16 * 2 + 2, 4, 5]
    assert [g["lr"] for g in groups] == [3e-5, 3e-5, 5e-2] and all(g["weight_decay"] == 1e-2 for g in groups)
    in_groups = {id(p) for g in groups for p in g["params"]}
    named = dict(m.named_parameters())
    assert id(named["projectors.0.weight"]) not in in_groups and id(named["img_classifier.bias"]) not in in_groups
    args = types.SimpleNamespace(lr=3e-5, warmup_ratio=0.01, use_prefix=True)
    opt, sched = build_optimizer(m, args, train_num_steps=250)
    ref_opt = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=3e-5)
    ref = get_linear_schedule_with_warmup(ref_opt, num_warmup_steps=0.01 * 250, num_training_steps=250)
    for _ in range(20):
        assert abs(sched.get_last_lr()[0] - ref.get_last_lr()[0]) < 1e-12
        assert abs(sched.get_last_lr()[2] / 5e-2 - ref.get_last_lr()[0] / 3e-5) < 1e-9
        opt.step(); sched.step(); ref_opt.step(); ref.step()
    one = reference_param_groups(tiny_model(use_prefix=False), 1e-5, use_prefix=False)
    assert len(one) == 1 and len(one[0]["params"]) == len(list(tiny_model(use_prefix=False).parameters()))
