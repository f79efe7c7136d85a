"""Deterministic parameter / input factories shared by the golden generator and the tests.

Weights are drawn from ``numpy.random.Generator(PCG64(seed))`` (bit-stable across numpy versions and
platforms), so golden fixtures only need to store *outputs*; every consumer regenerates identical
parameters from (seed, config).  Nothing here depends on the reference checkout.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List

import numpy as np
import torch


@dataclass
class EncCfg:
    vocab_size: int = 64
    hidden: int = 128
    heads: int = 2
    inter: int = 256
    layers: int = 2
    max_pos: int = 64
    type_vocab: int = 2
    eps: float = 1e-12
    roberta: bool = False
    pad_idx: int = 0
    num_labels: int = 11


TINY_BERT = EncCfg()
TINY_BERT_L8 = EncCfg(layers=8)  # the span model reads hidden_states[7] (bert_model.py:331): needs >= 7 layers
TINY_ROBERTA = EncCfg(type_vocab=1, eps=1e-5, roberta=True, pad_idx=1, max_pos=66)
BASE_BERT = EncCfg(vocab_size=30522, hidden=768, heads=12, inter=3072, layers=12, max_pos=512)
BASE_ROBERTA = EncCfg(vocab_size=50265, hidden=768, heads=12, inter=3072, layers=12, max_pos=514,
                      type_vocab=1, eps=1e-5, roberta=True, pad_idx=1)


def _normal(rng, shape, std):
    return torch.from_numpy((rng.standard_normal(shape, dtype=np.float32) * np.float32(std)))


def _uniform(rng, shape, lo, hi):
    return torch.from_numpy(rng.uniform(lo, hi, size=shape).astype(np.float32))


def encoder_params(cfg: EncCfg, seed: int, prefix: str = "", std: float = 0.05,
                   with_pooler: bool = True) -> Dict[str, torch.Tensor]:
    """HF/reference-named BertModel / RobertaModel state dict (models/modeling_bert.py key names).
    Non-trivial LayerNorm affine and biases so that every term is exercised."""
    rng = np.random.default_rng(seed)
    H, I = cfg.hidden, cfg.inter
    sd = {}
    e = prefix + "embeddings."
    sd[e + "word_embeddings.weight"] = _normal(rng, (cfg.vocab_size, H), std)
    sd[e + "position_embeddings.weight"] = _normal(rng, (cfg.max_pos, H), std)
    sd[e + "token_type_embeddings.weight"] = _normal(rng, (cfg.type_vocab, H), std)
    sd[e + "LayerNorm.weight"] = 1.0 + _normal(rng, (H,), 0.1)
    sd[e + "LayerNorm.bias"] = _normal(rng, (H,), 0.1)
    for i in range(cfg.layers):
        p = f"{prefix}encoder.layer.{i}."
        for name, (o, k) in {"attention.self.query": (H, H), "attention.self.key": (H, H),
                             "attention.self.value": (H, H), "attention.output.dense": (H, H),
                             "intermediate.dense": (I, H), "output.dense": (H, I)}.items():
            sd[p + name + ".weight"] = _normal(rng, (o, k), std)
            sd[p + name + ".bias"] = _normal(rng, (o,), 0.02)
        for ln in ("attention.output.LayerNorm", "output.LayerNorm"):
            sd[p + ln + ".weight"] = 1.0 + _normal(rng, (H,), 0.1)
            sd[p + ln + ".bias"] = _normal(rng, (H,), 0.1)
    if with_pooler:
        sd[prefix + "pooler.dense.weight"] = _normal(rng, (H, H), std)
        sd[prefix + "pooler.dense.bias"] = _normal(rng, (H,), 0.02)
    return sd


def head_params(cfg: EncCfg, seed: int) -> Dict[str, torch.Tensor]:
    """fc (models/bert_model.py:465) + CRF parameters (uniform(-0.1, 0.1) as pytorch-crf initialises)."""
    rng = np.random.default_rng(seed)
    C = cfg.num_labels
    return {
        "fc.weight": _normal(rng, (C, cfg.hidden), 0.05),
        "fc.bias": _normal(rng, (C,), 0.02),
        "crf.start_transitions": _uniform(rng, (C,), -0.1, 0.1),
        "crf.end_transitions": _uniform(rng, (C,), -0.1, 0.1),
        "crf.transitions": _uniform(rng, (C, C), -0.1, 0.1),
    }


def prompt_params(seed: int, feat_dim: int = 3840, mid: int = 800, hidden: int = 768, layers: int = 12,
                  n_anp: int = 2089, n_aux_cls: int = 3) -> Dict[str, torch.Tensor]:
    """encoder_conv / projectors / img_classifier / aux_img_classifier (models/bert_model.py:446-461)."""
    rng = np.random.default_rng(seed)
    out = 4 * 2 * hidden
    sd = {
        "encoder_conv.0.weight": _normal(rng, (mid, feat_dim), 0.02),
        "encoder_conv.0.bias": _normal(rng, (mid,), 0.02),
        "encoder_conv.2.weight": _normal(rng, (out, mid), 0.03),
        "encoder_conv.2.bias": _normal(rng, (out,), 0.02),
        "img_classifier.weight": _normal(rng, (n_anp, out), 0.02),
        "img_classifier.bias": _normal(rng, (n_anp,), 0.02),
    }
    for i in range(layers):
        sd[f"projectors.{i}.weight"] = _normal(rng, (4, out), 0.05)
        sd[f"projectors.{i}.bias"] = _normal(rng, (4,), 0.05)
    for k in range(n_aux_cls):
        sd[f"aux_img_classifier.{k}.weight"] = _normal(rng, (n_anp, out), 0.02)
        sd[f"aux_img_classifier.{k}.bias"] = _normal(rng, (n_anp,), 0.02)
    return sd


def text_batch(cfg: EncCfg, seed: int, B: int, S: int, lengths: List[int] = None, lo_id: int = 3):
    """Twitter-shaped synthetic text batch: ids uniform, first token CLS-like,
    padded with 0 (modules/dataset.py:414-415 pads with 0 even for RoBERTa), ragged lengths,
    labels 1..C-1 on valid tokens, 0 on pad."""
    rng = np.random.default_rng(seed)
    if lengths is None:
        lengths = [int(x) for x in rng.integers(max(2, S // 4), S + 1, size=B)]
        lengths[0] = S
    ids = rng.integers(lo_id, cfg.vocab_size, size=(B, S)).astype(np.int64)
    mask = np.zeros((B, S), dtype=np.int64)
    labels = rng.integers(1, cfg.num_labels, size=(B, S)).astype(np.int64)
    for b, L in enumerate(lengths):
        mask[b, :L] = 1
        ids[b, L:] = 0
        labels[b, L:] = 0
    tt = np.zeros((B, S), dtype=np.int64)
    return (torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(tt), torch.from_numpy(labels))


def prefix_kv(seed: int, layers: int, B: int, heads: int, P: int, D: int = 64, std: float = 0.5):
    """Synthetic per-layer visual prefix (K, V) [B, heads, P, D]."""
    if P == 0:
        return None
    rng = np.random.default_rng(seed)
    return [(_normal(rng, (B, heads, P, D), std), _normal(rng, (B, heads, P, D), std)) for _ in range(layers)]


def span_batch(cfg, seed, B, S, M, lengths):
    """Synthetic span supervision: M candidate spans per sentence inside the valid tokens, multi-hot start/end
    positions (distant supervision), polarity labels 0..3, a few masked (padding) span slots."""
    rng = np.random.default_rng(seed)
    starts = np.zeros((B, M), dtype=np.int64)
    ends = np.zeros((B, M), dtype=np.int64)
    spos = np.zeros((B, S), dtype=np.int64)
    epos = np.zeros((B, S), dtype=np.int64)
    for b, L in enumerate(lengths):
        for m in range(M):
            a = int(rng.integers(1, max(2, L - 1)))
            w = int(rng.integers(1, 5))
            starts[b, m], ends[b, m] = a, min(L - 1, a + w - 1)
        for _ in range(2):
            a = int(rng.integers(1, L))
            spos[b, a] = 1
            epos[b, min(L - 1, a + int(rng.integers(0, 3)))] = 1
    pol = rng.integers(0, 4, size=(B, M)).astype(np.int64)
    lm = (rng.random((B, M)) > 0.3).astype(np.int64)
    lm[:, 0] = 1
    return tuple(torch.from_numpy(x) for x in (starts, ends, spos, epos, pol, lm))


def span_head_params(cfg, seed):
    rng = np.random.default_rng(seed)
    H = cfg.hidden
    n = lambda shape, std: torch.from_numpy(rng.standard_normal(shape, dtype=np.float32) * np.float32(std))
    return {"dense.weight": n((H, H), 0.05), "dense.bias": n((H,), 0.02), "unary_affine.weight": n((1, H), 0.1),
            "unary_affine.bias": n((1,), 0.02), "binary_affine.weight": n((2, H), 0.1), "binary_affine.bias": n((2,), 0.02),
            "classifier.weight": n((4, H), 0.1), "classifier.bias": n((4,), 0.02)}


def resnet_params(model: "torch.nn.Module", seed: int) -> Dict[str, torch.Tensor]:
    """Seeded (numpy PCG64) state_dict for a torchvision-layout ResNet trunk (mtvaf_amd/models/resnet.py): He-scaled
    convolutions, BatchNorm scale in [0.5, 1.5], small BatchNorm shifts, non-trivial running statistics -- so that
    train-mode and eval-mode BatchNorm give visibly different pyramids."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for k, v in model.state_dict().items():
        shape = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros(shape, dtype=v.dtype)
        elif k.endswith("running_var"):
            out[k] = _uniform(rng, shape, 0.5, 2.0)
        elif k.endswith("running_mean"):
            out[k] = _normal(rng, shape, 0.2)
        elif v.dim() == 4:  # convolution: fan_out scaling (what torchvision's own initialisation uses)
            out[k] = _normal(rng, shape, float(np.sqrt(2.0 / (shape[0] * shape[2] * shape[3]))))
        elif v.dim() == 1 and k.endswith("weight"):
            out[k] = _uniform(rng, shape, 0.5, 1.5)
        elif v.dim() == 1:
            out[k] = _normal(rng, shape, 0.1)
        else:
            out[k] = _normal(rng, shape, 0.02)
    return out


def image_batch(seed: int, B: int, n_aux: int, hw: int = 64):
    rng = np.random.Generator(np.random.PCG64(seed))
    return _normal(rng, (B, 3, hw, hw), 1.0), _normal(rng, (B, n_aux, 3, hw, hw), 1.0)
