"""Region-feature cache for the frozen visual front-end.

The ResNet pyramid of ``ImageModel`` is frozen (``modules/train.py:920-921``) and costs 16 GF (ResNet-50) to
46 GF (ResNet-152) per sentence -- as much as the BERT forward -- so once the encoder is fast, recomputing it
every step caps end-to-end throughput.  The task models accept the pooled pyramid ``[B, F, 2, 2]`` (F = 3840, or
960 for ResNet-18/34) directly in place of raw images, so the front-end can run ONCE per image in inference mode
and its output be cached:

    cache = RegionFeatureCache(model.image_model)           # or ImageModel(resnet_root=...)
    feats, aux = cache.extract(images, aux_imgs)             # [B,F,2,2], [B,n,F,2,2]
    out = model(..., images=feats, aux_imgs=aux)             # identical prefix to feeding the raw images

Documented deviation: the reference leaves the frozen ResNet's BatchNorm in TRAIN mode during training
(``model.train()``, train.py:579 -- batch statistics, running stats drifting); cached features use EVAL-mode
BatchNorm (running statistics), i.e. what the reference computes at dev/test time.
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional, Tuple

import torch


class RegionFeatureCache:
    def __init__(self, image_model: torch.nn.Module, dtype: torch.dtype = torch.float32):
        self.image_model = image_model
        self.dtype = dtype
        self.store: Dict[str, Tuple[torch.Tensor, Optional[torch.Tensor]]] = {}

    @torch.no_grad()
    def extract(self, images: torch.Tensor, aux_imgs: Optional[torch.Tensor] = None):
        """images [B,3,h,w], aux_imgs [B,n,3,h,w] -> (feats [B,F,2,2], aux feats [B,n,F,2,2] or None): the
        ``torch.cat(pyramid, dim=1)`` of models/bert_model.py:538-539 before its ``view(bsz, prefix_len, -1)``."""
        was_training = self.image_model.training
        self.image_model.eval()
        try:
            pyr, aux_pyr = self.image_model(images, aux_imgs)
            feats = torch.cat(pyr, dim=1).to(self.dtype)
            aux = None
            if aux_pyr is not None:
                aux = torch.stack([torch.cat(a, dim=1) for a in aux_pyr], dim=1).to(self.dtype)
        finally:
            self.image_model.train(was_training)
        return feats, aux

    def add(self, keys: Iterable[str], images: torch.Tensor, aux_imgs: Optional[torch.Tensor] = None):
        """Extract a batch and remember it per image id (CPU copies, so the cache scales with host RAM)."""
        feats, aux = self.extract(images, aux_imgs)
        for i, k in enumerate(keys):
            self.store[k] = (feats[i].cpu(), None if aux is None else aux[i].cpu())

    def batch(self, keys: Iterable[str], device) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        rows = [self.store[k] for k in keys]
        feats = torch.stack([r[0] for r in rows]).to(device, non_blocking=True)
        aux = None if rows[0][1] is None else torch.stack([r[1] for r in rows]).to(device, non_blocking=True)
        return feats, aux

    def save(self, path: str):
        torch.save(self.store, path)

    def load(self, path: str):
        self.store = torch.load(path, map_location="cpu")
        return self
