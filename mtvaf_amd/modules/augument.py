"""Cutoff augmentation for the span model -- drop-in for the reference's ``modules/augument.py::Cutoff``
(:14-159).  Same constructor and ``_training_step_with_cutoff(aug_type)`` contract:
embedding output -> span / token / dim cutoff of the embeddings and of the attention mask -> encoder again
(``BertModel.get_embedding_output`` / ``get_bert_output``, reference models/modeling_bert.py:1117-1157).

The reference draws and applies the cut sample by sample on the host (``int(tensor)`` syncs, python loops over
the batch and over the cut indices).  Here the draws are vectorised device ops (one ``torch.rand`` / ``randint``
per batch -- RNG is plumbing) and the cut is ONE fused kernel (``mtvaf_mask_mul``) whose backward is itself, so
the augmented pass adds no host sync to the step.  The cut rule is the reference's:

* span  : ``cutoff_length = int(len * ratio)``, ``start = int(u * (len - cutoff_length))``, rows and mask entries
  ``[start, start + cutoff_length)`` zeroed (:99-117);
* token : ``cutoff_length`` row indices drawn WITH replacement from ``[0, len)`` zeroed in rows and mask (:120-141);
* dim   : ``int(H * ratio)`` dimension indices drawn with replacement zeroed per sample, mask untouched (:144-159).

``len`` is the row sum of the mask that was handed in, exactly as in the reference -- i.e. it includes the prefix
slots when the prompt mask ``[B, P+S]`` is passed (models/bert_model.py:334-341).  In that case the reference's
python slicing mis-sizes its tensors whenever a cut reaches past the text axis; this implementation applies the
cut to the positions that exist on each axis instead of failing.
"""
from __future__ import annotations

import torch

from .. import engine


class Cutoff:
    def __init__(self, input_ids, token_type_ids, attention_masks, prefix_guids, args, model):
        self.input_ids = input_ids
        self.token_type_ids = token_type_ids
        self.attention_masks = attention_masks
        self.prefix_guids = prefix_guids
        self.args = args
        self.model = model

    # -- cut rules (pure index arithmetic; `u` / index draws can be injected for tests) ---------------------------
    @staticmethod
    def span_keep(input_lens, ratio, n_pos, u=None):
        """-> keep [B, n_pos] (float 0/1) for the span cut of :99-117."""
        lens = input_lens.to(torch.float32)
        cl = (lens * ratio).to(torch.int64)                         # int(input_lens[i] * ratio)
        if u is None:
            u = torch.rand(lens.shape[0], device=lens.device)
        start = (u * (input_lens - cl).to(torch.float32)).to(torch.int64)   # int(rand * (len - cutoff_length))
        pos = torch.arange(n_pos, device=lens.device)[None, :]
        cut = (pos >= start[:, None]) & (pos < (start + cl)[:, None])
        return (~cut).to(torch.float32)

    @staticmethod
    def index_keep(limit, count, n_pos, draws=None):
        """-> keep [B, n_pos]: `count[b]` indices drawn with replacement from [0, limit[b]) are zeroed (:120-159)."""
        B = limit.shape[0]
        dev = limit.device
        max_count = n_pos  # count <= limit * ratio <= n_pos
        if draws is None:
            draws = torch.rand(B, max_count, device=dev)
        idx = (draws * limit[:, None].to(torch.float32)).to(torch.int64).clamp_(max=n_pos - 1)
        live = torch.arange(max_count, device=dev)[None, :] < count[:, None]
        keep = torch.ones(B, n_pos, device=dev)
        keep.scatter_reduce_(1, idx, (~live).to(torch.float32), reduce="prod", include_self=True)
        return keep

    # -- the reference's three generators, batched ------------------------------------------------------------------
    def generate_span_cutoff_embedding(self, embeds, masks, input_lens):
        S, T = embeds.shape[1], masks.shape[1]
        keep = self.span_keep(input_lens, self.args.aug_cutoff_ratio, max(S, T))
        out = engine.MaskMulFunction.apply(embeds, keep[:, :S].contiguous().view(-1), None)
        return out, (masks * keep[:, :T].to(masks.dtype))

    def generate_token_cutoff_embedding(self, embeds, masks, input_lens):
        S, T = embeds.shape[1], masks.shape[1]
        cl = (input_lens.to(torch.float32) * self.args.aug_cutoff_ratio).to(torch.int64)
        keep = self.index_keep(input_lens, cl, max(S, T))
        out = engine.MaskMulFunction.apply(embeds, keep[:, :S].contiguous().view(-1), None)
        return out, (keep[:, :T] * masks).to(torch.int64)

    def generate_dim_cutoff_embedding(self, embeds, masks, input_lens):
        B, S, H = embeds.shape
        cl = torch.full((B,), int(H * self.args.aug_cutoff_ratio), device=embeds.device, dtype=torch.int64)
        keep = self.index_keep(torch.full((B,), H, device=embeds.device), cl, H)
        return engine.MaskMulFunction.apply(embeds, None, keep.contiguous()), masks

    def _training_step_with_cutoff(self, aug_type):
        """reference: modules/augument.py:54-77 -> (sequence_output, pooled_output) of the cut input."""
        dev = self.args.device
        input_ids = self.input_ids.to(dev)
        token_type_ids = self.token_type_ids.to(dev)
        embeds = self.model.get_embedding_output(input_ids=input_ids, token_type_ids=token_type_ids)
        masks = self.attention_masks.to(dev)
        input_lens = torch.sum(masks, dim=1)
        if aug_type == "span_cutoff":
            input_embeds, input_masks = self.generate_span_cutoff_embedding(embeds, masks, input_lens)
        elif aug_type == "token_cutoff":
            input_embeds, input_masks = self.generate_token_cutoff_embedding(embeds, masks, input_lens)
        elif aug_type == "dim_cutoff":
            input_embeds, input_masks = self.generate_dim_cutoff_embedding(embeds, masks, input_lens)
        else:
            raise NotImplementedError
        return self.model.get_bert_output(embedding_output=input_embeds, attention_mask=input_masks,
                                          past_key_values=self.prefix_guids)
