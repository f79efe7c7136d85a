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

``RegionFeatureCache(image_model, compute="bf16")`` (round 5) builds the cache in reduced precision: the trunk's eval-mode
BatchNorms are folded into their convolutions in fp32 (w' = w * gamma / sqrt(var + eps), b' = beta - mean * gamma / sqrt(var +
eps)), the folded trunk runs channels-last in bf16 on MIOpen (fp32 accumulation inside a convolution, bf16 activations between
layers), and the 2x2 region pooling of every pyramid level is done in fp32 on the level's output.  Second documented
deviation: the features carry the trunk's bf16 rounding (measured per pyramid level in tests/test_frontend.py); the default
``compute="fp32"`` is the exact eval-mode pyramid.
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional, Tuple

import torch


class _FoldedTrunk:
    """Eval-mode ResNet pyramid with BatchNorm folded into the convolutions, channels-last, in `dtype` (see the module
    docstring).  Walks the trunk as ``ImageModel.get_resnet_prompt`` does (reference: models/bert_model.py:99-111): stem, then
    layer1..layer4, each level pooled to 2 x 2 regions -- here in fp32."""

    def __init__(self, resnet: torch.nn.Module, dtype: torch.dtype):
        self.dtype = dtype
        self.stem = self._fold(resnet.conv1, resnet.bn1)
        self.stages = []
        for name in ("layer1", "layer2", "layer3", "layer4"):
            blocks = []
            for blk in getattr(resnet, name):
                convs = [self._fold(blk.conv1, blk.bn1), self._fold(blk.conv2, blk.bn2)]
                if hasattr(blk, "conv3"):
                    convs.append(self._fold(blk.conv3, blk.bn3))
                down = None if blk.downsample is None else self._fold(blk.downsample[0], blk.downsample[1])
                blocks.append((convs, down))
            self.stages.append(blocks)

    def _fold(self, conv, bn):
        scale = (bn.weight.detach().float() / torch.sqrt(bn.running_var.float() + bn.eps))
        w = (conv.weight.detach().float() * scale[:, None, None, None]).to(self.dtype).contiguous(memory_format=torch.channels_last)
        b = (bn.bias.detach().float() - bn.running_mean.float() * scale).to(self.dtype)
        return w, b, conv.stride, conv.padding

    @staticmethod
    def _conv(x, p, relu=True):
        w, b, stride, pad = p
        y = torch.nn.functional.conv2d(x, w, b, stride, pad)
        return torch.relu_(y) if relu else y

    def __call__(self, x):
        x = x.to(self.dtype).contiguous(memory_format=torch.channels_last)
        x = torch.nn.functional.max_pool2d(self._conv(x, self.stem), 3, 2, 1)
        out = []
        for blocks in self.stages:
            for convs, down in blocks:
                identity = x if down is None else self._conv(x, down, relu=False)
                h = x
                for i, c in enumerate(convs):
                    h = self._conv(h, c, relu=i + 1 < len(convs))
                x = torch.relu_(h + identity)
            k = x.size(2) // 2
            out.append(torch.nn.functional.avg_pool2d(x.float(), kernel_size=(k, k), stride=k).contiguous())
        return out


class RegionFeatureCache:
    def __init__(self, image_model: torch.nn.Module, dtype: torch.dtype = torch.float32, compute: str = "fp32"):
        if compute not in ("fp32", "bf16"):
            raise ValueError(f"compute must be 'fp32' or 'bf16', got {compute!r}")
        self.image_model = image_model
        self.dtype = dtype
        self.compute = compute
        self._folded = None  # (built on first use, from the trunk's weights at that moment: the trunk is frozen)
        self.store: Dict[str, Tuple[torch.Tensor, Optional[torch.Tensor]]] = {}

    def _pyramids(self, images, aux_imgs):
        if self.compute == "fp32":
            return self.image_model(images, aux_imgs)
        if self._folded is None:
            self._folded = _FoldedTrunk(self.image_model.resnet, torch.bfloat16)
        if aux_imgs is None:
            return self._folded(images), None
        # the main image and the aux crops of a batch as ONE batch of B * (1 + n) images (eval mode: no cross-sample coupling)
        B, n = aux_imgs.shape[:2]
        allp = self._folded(torch.cat([images[:, None], aux_imgs], 1).flatten(0, 1))
        allp = [p.view(B, 1 + n, *p.shape[1:]) for p in allp]
        return [p[:, 0] for p in allp], [[p[:, 1 + i] for p in allp] for i in range(n)]

    @torch.no_grad()
    def extract(self, images: torch.Tensor, aux_imgs: Optional[torch.Tensor] = None):
        """images [B,3,h,w], aux_imgs [B,n,3,h,w] -> (feats [B,F,2,2], aux feats [B,n,F,2,2] or None): the
        ``torch.cat(pyramid, dim=1)`` of models/bert_model.py:538-539 before its ``view(bsz, prefix_len, -1)``."""
        was_training = self.image_model.training
        self.image_model.eval()
        try:
            pyr, aux_pyr = self._pyramids(images, aux_imgs)
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
