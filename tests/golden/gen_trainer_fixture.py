"""Trainer-contract fixture: drives the REAL reference trainer code (modules/train.py, imported from /root/reference in the
authoring container through the shim of gen_golden.py) against the drop-in `mtvaf_amd` model on CPU tensors -- no forward
pass, so no GPU is needed -- and records what the reference code does with it:

  * `SATrainer2.multiModal_before_train` (modules/train.py:894-926): optimizer-group membership by parameter NAME,
    per-group lr / weight decay, which parameters end up frozen, the LR-schedule factors of the first steps;
  * `SATrainer2.bert_before_train` (:887-892): the single text-only group;
  * `SATrainer2.load_pretrained2` (:958-975, positional + shape-driven) and `load_bert` (:977-987, by key): a checkpoint
    saved from the REFERENCE `TVNetSAModel2` class (its own state_dict order) is loaded into the drop-in by the
    reference's loader; every tensor of the checkpoint carries a unique fill value, so reading the drop-in's state_dict
    back gives the (drop-in key <- checkpoint key) map the loader actually applied;
  * the state_dict key order and shapes of both models.

    python tests/golden/gen_trainer_fixture.py        # writes tests/golden/trainer_contract.json

`tests/test_boundary.py` asserts the committed fixture against the current drop-in (same names, same order, same shapes
=> the reference's positional loaders and name-matched groups behave exactly as recorded) and against
`mtvaf_amd.optim.reference_param_groups`.  Nothing of the reference's source text is stored: the fixture holds names,
shapes and numbers only.
"""
from __future__ import annotations

import json
import logging
import os
import sys
import tempfile
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

import gen_golden as G  # noqa: E402
import params as P  # noqa: E402

LABELS = ["O", "B-NEU", "I-NEU", "B-POS", "I-POS", "B-NEG", "I-NEG", "X", "[CLS]", "[SEP]"]
CFG = P.EncCfg(vocab_size=120, hidden=768, heads=12, inter=96, layers=2, max_pos=40)  # hidden 768: the reference hard-codes it


def _args(tmp, **kw):
    a = dict(bert_name="bert-base-uncased", use_prefix=True, vao=False, noauxloss=True, use_probe=False, n_gpu=1, alpha=0.0,
             prefix_len=4, prefix_dim=768, device="cpu", resnet_root=tmp, use_152=False, use_101=False, use_34=False,
             use_18=False, lr=3e-5, warmup_ratio=0.01, gradient_accumulation_steps=1, num_epochs=2, local_rank=-1,
             load_path=os.path.join(tmp, "trained.pth"), beta=0.0)
    a.update(kw)
    return types.SimpleNamespace(**a)


def main():
    G.install_shim()
    import torchvision.models as tvm
    from mtvaf_amd.models import resnet as our_resnet
    for n in ("resnet18", "resnet34", "resnet50", "resnet101", "resnet152"):
        setattr(tvm, n, (lambda ctor: (lambda pretrained=False, **k: ctor()))(getattr(our_resnet, n)))
    import sklearn.metrics  # noqa: F401  (imported by the reference trainer)
    sys.modules["seqeval.metrics"].classification_report = lambda *a, **k: ""
    tmp = tempfile.mkdtemp(prefix="mtvaf_trainer_fixture_")
    torch.manual_seed(0)
    torch.save(our_resnet.resnet50().state_dict(), os.path.join(tmp, "resnet50.pth"))

    # the reference classes, with from_pretrained returning a random-init model of CFG (no weights on disk: SURVEY 8c (6))
    import models.bert_model as rbm
    import models.modeling_bert as rmb
    rmb.BertModel.from_pretrained = classmethod(lambda cls, name, *a, **k: cls(G.ref_config(CFG)))
    import modules.train as rtrain

    ref_model = rbm.TVNetSAModel2(LABELS, None, _args(tmp))
    ref_sd = ref_model.state_dict()
    ref_keys = list(ref_sd.keys())

    from mtvaf_amd.models.bert_model import TVNetSAModel2
    from transformers import BertConfig

    def dropin(**kw):
        a = _args(tmp, **kw)
        a.bert_config = BertConfig(vocab_size=CFG.vocab_size, hidden_size=CFG.hidden, num_hidden_layers=CFG.layers,
                                   num_attention_heads=CFG.heads, intermediate_size=CFG.inter,
                                   max_position_embeddings=CFG.max_pos, pad_token_id=0)
        return TVNetSAModel2(LABELS, None, a)

    log = logging.getLogger("fixture")
    out = {"_generated_by": "tests/golden/gen_trainer_fixture.py (reference modules/train.py imported in the authoring "
                            "container; names / shapes / numbers only)",
           "config": {"hidden": CFG.hidden, "layers": CFG.layers, "inter": CFG.inter, "vocab": CFG.vocab_size,
                      "max_pos": CFG.max_pos}}
    out["reference_state_dict"] = [[k, list(v.shape)] for k, v in ref_sd.items()]

    # ---- multiModal_before_train on the drop-in -------------------------------------------------------------------
    m = dropin()
    out["dropin_state_dict"] = [[k, list(v.shape)] for k, v in m.state_dict().items()]
    out["dropin_named_parameters"] = [n for n, _ in m.named_parameters()]
    tr = rtrain.SATrainer2(train_data=list(range(50)), model=m, args=_args(tmp), logger=log)
    tr.multiModal_before_train()
    names = {id(p): n for n, p in m.named_parameters()}
    out["multimodal_groups"] = [{"lr": g.get("initial_lr", g["lr"]), "weight_decay": g["weight_decay"],
                                 "params": [names[id(p)] for p in g["params"]]} for g in tr.optimizer.param_groups]
    out["multimodal_frozen"] = [n for n, p in m.named_parameters() if not p.requires_grad]
    grouped = {n for g in out["multimodal_groups"] for n in g["params"]}
    out["multimodal_ungrouped"] = [n for n, p in m.named_parameters() if n not in grouped]
    out["train_num_steps"] = tr.train_num_steps
    facs = []
    for _ in range(6):
        facs.append([g["lr"] for g in tr.optimizer.param_groups])
        tr.optimizer.step()
        tr.scheduler.step()
    out["multimodal_lr_first_steps"] = facs

    # ---- bert_before_train (text-only) ----------------------------------------------------------------------------
    m2 = dropin(use_prefix=False)
    tr2 = rtrain.SATrainer2(train_data=list(range(50)), model=m2, args=_args(tmp, use_prefix=False), logger=log)
    tr2.bert_before_train()
    n2 = {id(p): n for n, p in m2.named_parameters()}
    out["text_only_groups"] = [{"lr": g.get("initial_lr", g["lr"]), "weight_decay": g["weight_decay"], "params": [n2[id(p)] for p in g["params"]]}
                               for g in tr2.optimizer.param_groups]

    # ---- the reference's checkpoint loaders, reference-class checkpoint -> drop-in ---------------------------------
    marked = {}
    for i, (k, v) in enumerate(ref_sd.items()):
        marked[k] = torch.full_like(v, float(i + 1)) if v.is_floating_point() else v.clone()
    torch.save(marked, os.path.join(tmp, "trained.pth"))
    value_to_key = {float(i + 1): k for i, k in enumerate(ref_keys) if ref_sd[k].is_floating_point()}

    def applied_map(model):
        res = {}
        for k, v in model.state_dict().items():
            if v.is_floating_point() and v.numel() and float(v.flatten()[0]) in value_to_key and bool((v == v.flatten()[0]).all()):
                res[k] = value_to_key[float(v.flatten()[0])]
        return res

    for loader in ("load_pretrained2", "load_bert"):
        mm = dropin()
        with torch.no_grad():
            for p in mm.parameters():
                p.fill_(-7.0)
        t = rtrain.SATrainer2(model=mm, args=_args(tmp), logger=log)
        getattr(t, loader)()
        out[loader + "_applied"] = applied_map(mm)

    path = os.path.join(HERE, "trainer_contract.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path, {k: (len(v) if hasattr(v, "__len__") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
