"""Host-side helpers for the unchanged reference trainer.

``modules/train.py:627-647`` (train), ``:722-735`` / ``:805-818`` (evaluate / test) rebuild ``y_true`` / ``y_pred`` for
seqeval with a B x S Python double loop over ``labels.to('cpu')``, ``attention_mask.to('cpu')`` and the decoded tag lists:
~4k interpreted iterations per step at bs 32 / S 128, more than the GPU needs for the whole forward pass in bf16 mode.
``label_sequences`` produces the same two lists of label-name lists with array operations: one packed device->host copy
(the ``DeferredTags`` of ``TVNetSAModel2.forward`` already holds the tags in pinned memory), boolean masks, one lookup.

Rule restated from the reference loop: per sentence, walk the columns from 1 (column 0 is ``[CLS]``) while
``attention_mask`` is 1 and stop at the first 0; keep a position unless its gold label is ``"X"`` or ``"[SEP]"``; emit
``label_map^-1[label]`` and ``label_map^-1[predicted tag]`` (id 0 reads ``"PAD"``)."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from .modules.crf import DeferredTags


def _tags_array(logits, B: int, S: int) -> np.ndarray:
    """Decoded tags as a [B, S] int array (padded with 0 beyond each sentence's length)."""
    if isinstance(logits, DeferredTags):
        packed = logits.packed()
        if packed is not None:
            return np.where(packed[:, :S] < 0, 0, packed[:, :S])
    out = np.zeros((B, S), dtype=np.int64)
    for r, row in enumerate(logits):
        n = min(len(row), S)
        out[r, :n] = row[:n]
    return out


def label_sequences(labels: torch.Tensor, attention_mask: torch.Tensor, logits: Sequence[Sequence[int]],
                    label_map: Dict[str, int]) -> Tuple[List[List[str]], List[List[str]]]:
    """-> (y_true, y_pred) exactly as the loops of modules/train.py:627-647 / :722-735 / :805-818 build them."""
    lab = labels.detach().to("cpu").numpy()
    msk = attention_mask.detach().to("cpu").numpy().astype(bool)
    B, S = lab.shape
    tags = _tags_array(logits, B, S)
    id2 = {idx: name for name, idx in label_map.items()}
    id2[0] = "PAD"
    top = int(max(max(id2), int(lab.max(initial=0)), int(tags.max(initial=0))))
    names = np.array([id2.get(i, "PAD") for i in range(top + 1)], dtype=object)
    run = np.logical_and.accumulate(msk[:, 1:], axis=1)  # the reference breaks at the first mask 0
    skip = [label_map[n] for n in ("X", "[SEP]") if n in label_map]
    keep = run & ~np.isin(lab[:, 1:], skip)
    y_true = [names[lab[r, 1:][keep[r]]].tolist() for r in range(B)]
    y_pred = [names[tags[r, 1:][keep[r]]].tolist() for r in range(B)]
    return y_true, y_pred
