"""mtvaf_amd.metrics.label_sequences against the reference trainer's loop (modules/train.py:627-647), restated literally,
on random batches including masks with holes (the reference stops at the first 0), X / [SEP] gold labels, predicted PAD."""
import numpy as np
import torch

LABELS = ["O", "B-NEU", "I-NEU", "B-POS", "I-POS", "B-NEG", "I-NEG", "X", "[CLS]", "[SEP]"]
LABEL_MAP = {label: idx for idx, label in enumerate(LABELS, 1)}


def reference_loop(label_ids, input_mask, logits, label_map_in):
    label_map = {idx: label for label, idx in label_map_in.items()}
    label_map[0] = "PAD"
    y_true, y_pred = [], []
    for row, mask_line in enumerate(input_mask):
        true_label, true_predict = [], []
        for column, mask in enumerate(mask_line):
            if column == 0:
                continue
            if mask:
                if label_map[label_ids[row][column]] != "X" and label_map[label_ids[row][column]] != "[SEP]":
                    true_label.append(label_map[label_ids[row][column]])
                    true_predict.append(label_map[logits[row][column]])
            else:
                break
        y_true.append(true_label)
        y_pred.append(true_predict)
    return y_true, y_pred


def test_label_sequences_equal_the_reference_loop():
    from mtvaf_amd.metrics import label_sequences
    rng = np.random.default_rng(0)
    for B, S in ((1, 2), (7, 16), (32, 128)):
        lens = rng.integers(1, S + 1, B)
        mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int64)
        if B > 2:
            mask[1, min(3, S - 1)] = 0  # a hole: everything behind it is ignored
        labels = rng.integers(1, 11, (B, S)) * mask
        labels[:, 0] = LABEL_MAP["[CLS]"]
        logits = [rng.integers(0, 11, int(mask[b].sum())).tolist() for b in range(B)]
        want = reference_loop(labels, mask, [row + [0] * S for row in logits], LABEL_MAP)
        got = label_sequences(torch.from_numpy(labels), torch.from_numpy(mask), logits, LABEL_MAP)
        assert got == want


def test_label_sequences_reads_deferred_tags_without_python_lists():
    from mtvaf_amd.metrics import label_sequences
    from mtvaf_amd.modules.crf import DeferredTags
    rng = np.random.default_rng(1)
    B, S = 5, 12
    lens = rng.integers(2, S + 1, B)
    mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int64)
    labels = rng.integers(1, 11, (B, S)) * mask
    tags = np.full((B, S), -1, dtype=np.int32)
    for b in range(B):
        tags[b, :lens[b]] = rng.integers(1, 11, lens[b])
    packed = torch.from_numpy(np.concatenate([tags, lens[:, None].astype(np.int32)], 1))
    d = DeferredTags(packed, None, S)
    want = reference_loop(labels, mask, np.where(tags < 0, 0, tags), LABEL_MAP)
    assert label_sequences(torch.from_numpy(labels), torch.from_numpy(mask), d, LABEL_MAP) == want
    assert list.__len__(d) == 0, "the packed array was used, the list was not materialised"
    assert [len(r) for r in d] == lens.tolist()  # ... and it still materialises on demand
