"""Optimizer / LR schedule of the reference trainers, for the drop-in models.

``modules/train.py::SATrainer2.multiModal_before_train`` (:894-926) builds three AdamW parameter groups by
NAME (encoder ``bert*`` at ``args.lr``; ``encoder_conv*`` / ``gates*`` at ``args.lr``; ``crf*`` / ``fc*`` at
5e-2; weight decay 1e-2 everywhere), freezes ``image_model*`` and attaches a linear warm-up / linear decay
schedule; ``bert_before_train`` (:887-892) is the text-only variant (one group, all parameters).  The same
grouping is reproduced here on top of torch's fused multi-tensor AdamW (one launch per ~25 tensors, already at
the HBM roofline: 7 fp32 streams per parameter), so a step of the reference trainer
costs what ``bench.py`` measures.

Reference quirk kept on purpose: ``projectors.*``, ``img_classifier.*`` and ``aux_img_classifier.*`` match no
group (the second group looks for the long-gone ``gates``), so the reference never updates them.
"""
from __future__ import annotations

from typing import Dict, List

import torch


def reference_param_groups(model: torch.nn.Module, lr: float, use_prefix: bool = True) -> List[Dict]:
    """The parameter groups of modules/train.py:887-916 (by parameter name, in the reference's order)."""
    named = list(model.named_parameters())
    if not use_prefix:  # bert_before_train: every parameter, one group
        return [{"params": [p for _, p in named], "lr": lr}]
    groups = [
        {"lr": lr, "weight_decay": 1e-2, "params": [p for n, p in named if "bert" in n]},
        {"lr": lr, "weight_decay": 1e-2, "params": [p for n, p in named if "encoder_conv" in n or "gates" in n]},
        {"lr": 5e-2, "weight_decay": 1e-2, "params": [p for n, p in named if "crf" in n or n.startswith("fc")]},
    ]
    return groups


def linear_schedule_with_warmup(optimizer, num_warmup_steps: float, num_training_steps: int):
    """transformers.get_linear_schedule_with_warmup restated (the reference passes a float warm-up count,
    train.py:922-924): lr factor = step / warmup while warming up, then linear decay to 0."""

    def factor(step: int) -> float:
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - num_warmup_steps)))

    return torch.optim.lr_scheduler.LambdaLR(optimizer, factor)


def build_optimizer(model: torch.nn.Module, args, train_num_steps: int):
    """-> (optimizer, scheduler) as SATrainer2.train() sets them up (train.py:574-578, 887-926)."""
    use_prefix = bool(getattr(args, "use_prefix", False))
    groups = reference_param_groups(model, args.lr, use_prefix)
    if use_prefix:
        for n, p in model.named_parameters():  # freeze resnet (:920-921)
            if "image_model" in n:
                p.requires_grad = False
    on_gpu = any(p.is_cuda for g in groups for p in g["params"])
    opt = torch.optim.AdamW(groups, lr=args.lr, fused=True) if on_gpu else torch.optim.AdamW(groups, lr=args.lr)
    sched = linear_schedule_with_warmup(opt, getattr(args, "warmup_ratio", 0.01) * train_num_steps, train_num_steps)
    return opt, sched
