"""Linear-chain CRF module with the public surface of ``torchcrf.CRF`` as the reference uses it
(models/bert_model.py:464 ``CRF(num_labels, batch_first=True)``, :511 ``decode``, :521
``crf(emissions, labels, mask=..., reduction='mean')``), computed by the gfx950 kernels
mtvaf_crf_nll_{fwd,bwd} / mtvaf_crf_viterbi.  Parameter names (``start_transitions``,
``end_transitions``, ``transitions``) and the uniform(-0.1, 0.1) initialisation follow pytorch-crf.
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import nn

from .. import engine, hip


class DeferredTags(list):
    """``List[List[int]]`` of Viterbi paths whose device->host copy has been ENQUEUED but not waited for.

    The reference's ``crf.decode`` returns Python lists, which forces a host sync in the middle of every training
    step (models/bert_model.py:511).  This list subclass carries the packed
    [B, S+1] int32 result (tags | length) in pinned host memory plus the copy's event and fills itself on first
    use (indexing, iteration, len, comparison, repr ...), i.e. where the trainer builds y_pred after
    ``loss.backward()`` (modules/train.py:627-647).  Values are identical to an eager ``decode``."""

    def __init__(self, packed_host: torch.Tensor, event, S: int, release=None):
        super().__init__()
        self._packed, self._event, self._S, self._release = packed_host, event, S, release

    def _fill(self):
        if self._packed is not None:
            if self._event is not None:
                self._event.synchronize()
            packed, S = self._packed, self._S
            self._packed = None
            rows, lens = packed[:, :S].tolist(), packed[:, S].tolist()
            super().extend(row[:n] for row, n in zip(rows, lens))
            self._give_back()
        return self

    def packed(self):
        """The packed [B, S+1] int32 host array (tags padded with -1 | length) after waiting for the copy, or None once
        the list has been materialised -- array consumers (``mtvaf_amd.metrics.label_sequences``) skip the Python lists."""
        if self._packed is None:
            return None
        if self._event is not None:
            self._event.synchronize()
        return self._packed.numpy()

    def _give_back(self):
        if self._release is not None:
            self._release()
            self._release = None

    def __del__(self):  # never read: the staging buffer goes back with its (possibly pending) copy event
        self._give_back()

    def __getitem__(self, i):
        self._fill()
        return super().__getitem__(i)

    def __iter__(self):
        self._fill()
        return super().__iter__()

    def __len__(self):
        self._fill()
        return super().__len__()

    def __eq__(self, other):
        self._fill()
        return list(self) == (list(other) if isinstance(other, list) else other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __repr__(self):
        self._fill()
        return super().__repr__()

    def __reduce__(self):
        return (list, (list(self),))


class CRF(nn.Module):
    def __init__(self, num_tags: int, batch_first: bool = False) -> None:
        if num_tags <= 0:
            raise ValueError(f"invalid number of tags: {num_tags}")
        if num_tags > 16:
            raise NotImplementedError("the CRF kernels keep one tag per lane with a 16-wide transition tile")
        super().__init__()
        self.num_tags = num_tags
        self.batch_first = batch_first
        self.start_transitions = nn.Parameter(torch.empty(num_tags))
        self.end_transitions = nn.Parameter(torch.empty(num_tags))
        self.transitions = nn.Parameter(torch.empty(num_tags, num_tags))
        # pinned staging buffers of decode_deferred, recycled: a fresh pinned allocation per step costs a
        # hipHostMalloc that waits for the GPU to drain (measured 185 ms per step at B=128, S=512)
        self._host_pool: list = []
        self.reset_parameters()

    def reset_parameters(self) -> None:
        nn.init.uniform_(self.start_transitions, -0.1, 0.1)
        nn.init.uniform_(self.end_transitions, -0.1, 0.1)
        nn.init.uniform_(self.transitions, -0.1, 0.1)

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(num_tags={self.num_tags})"

    def _prep(self, emissions, tags, mask):
        if emissions.dim() != 3 or emissions.size(2) != self.num_tags:
            raise ValueError(f"expected emissions [*, *, {self.num_tags}], got {tuple(emissions.shape)}")
        if not self.batch_first:
            emissions = emissions.transpose(0, 1)
            tags = tags.transpose(0, 1) if tags is not None else None
            mask = mask.transpose(0, 1) if mask is not None else None
        B, S, _ = emissions.shape
        if mask is None:
            mask = torch.ones(B, S, dtype=torch.uint8, device=emissions.device)
        mask = mask.to(torch.uint8).contiguous()
        if tags is not None:
            tags = tags.to(torch.long).contiguous()
        return emissions.float(), tags, mask

    def forward(self, emissions, tags, mask: Optional[torch.Tensor] = None, reduction: str = "sum"):
        """Log-likelihood of ``tags`` (like torchcrf).  The fused kernel produces the batch-mean NLL, so
        'mean' is exact and 'sum' is mean * B; 'none' / 'token_mean' are not on the MTVAF path."""
        emissions, tags, mask = self._prep(emissions, tags, mask)
        nll_mean = engine.CRFNLLFunction.apply(emissions, self.start_transitions, self.end_transitions,
                                               self.transitions, tags, mask)
        if reduction == "mean":
            return -nll_mean
        if reduction == "sum":
            return -nll_mean * emissions.shape[0]
        raise NotImplementedError(f"reduction={reduction!r} is not on the MTVAF path (the reference uses 'mean')")

    def nll_mean(self, emissions, tags, mask: Optional[torch.Tensor] = None):
        """``-1 * self(emissions, tags, mask=mask, reduction='mean')`` (models/bert_model.py:521) as ONE autograd node: the
        kernel's result is that quantity, so the two negations (and their two backward kernels) are not launched."""
        emissions, tags, mask = self._prep(emissions, tags, mask)
        return engine.CRFNLLFunction.apply(emissions, self.start_transitions, self.end_transitions, self.transitions,
                                           tags, mask)

    @torch.no_grad()
    def decode_packed(self, emissions, mask: Optional[torch.Tensor] = None):
        """Viterbi on device -> (tags int32 [B,S] padded with -1, lengths int32 [B]); no host sync."""
        emissions, _, mask = self._prep(emissions, None, mask)
        B, S, _ = emissions.shape
        em = emissions.contiguous()
        tags = torch.empty(B, S, dtype=torch.int32, device=em.device)
        lens = torch.empty(B, dtype=torch.int32, device=em.device)
        hip.crf_viterbi(em, mask, self.start_transitions.data, self.end_transitions.data, self.transitions.data, tags,
                        lens)
        return tags, lens

    def decode_deferred(self, emissions, mask: Optional[torch.Tensor] = None) -> DeferredTags:
        """Viterbi on device + asynchronous packed copy to pinned host memory; no host sync here."""
        tags, lens = self.decode_packed(emissions, mask)
        S = tags.shape[1]
        packed = torch.cat([tags, lens[:, None]], dim=1)
        need = packed.numel()
        buf = None
        for i, (t, ev_old) in enumerate(self._host_pool):
            if t.numel() >= need and (ev_old is None or ev_old.query()):
                buf = self._host_pool.pop(i)[0]
                break
        if buf is None:
            buf = torch.empty(max(need, 1 << 14), dtype=packed.dtype, pin_memory=True)
        host = buf[:need].view(packed.shape)
        host.copy_(packed, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        pool = self._host_pool
        return DeferredTags(host, ev, S, release=lambda: pool.append((buf, ev)) if len(pool) < 8 else None)

    def decode(self, emissions, mask: Optional[torch.Tensor] = None) -> List[List[int]]:
        tags, lens = self.decode_packed(emissions, mask)
        packed = torch.cat([tags, lens[:, None]], dim=1).cpu()  # ONE D2H copy
        S = tags.shape[1]
        return [row[:n].tolist() for row, n in zip(packed[:, :S], packed[:, S].tolist())]
