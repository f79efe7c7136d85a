"""RoBERTa encoder with per-layer visual prefix K/V, MI355X-native.

Drop-in for the reference's ``models/modeling_roberta.py`` hot-path classes (RobertaEmbeddings :70-157,
RobertaModel :~700-1020, create_position_ids_from_input_ids :1706-1719).  The encoder stack is the same
kernel path as BERT (the reference's Roberta* layer classes are copies of the Bert* ones,
modeling_roberta.py:161-580); only the embeddings differ: position ids are
``cumsum(ids != pad) * (ids != pad) + pad`` with no prefix offset, and both the word and position tables
have ``padding_idx`` (no gradient for that row).
"""
from __future__ import annotations

import torch
from torch import nn
from transformers import RobertaConfig

from .modeling_bert import BertEmbeddings, BertModel


class RobertaEmbeddings(BertEmbeddings):
    """reference: models/modeling_roberta.py:70-157"""

    roberta = True

    def __init__(self, config):
        super().__init__(config)
        self.padding_idx = config.pad_token_id
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size,
                                                padding_idx=self.padding_idx)


def create_position_ids_from_input_ids(input_ids, padding_idx, past_key_values_length=0):
    """Host-visible restatement of models/modeling_roberta.py:1706-1719 (the kernel path computes the
    same ids on device in mtvaf_roberta_position_ids)."""
    mask = input_ids.ne(padding_idx).int()
    incremental_indices = (torch.cumsum(mask, dim=1).type_as(mask) + past_key_values_length) * mask
    return incremental_indices.long() + padding_idx


class RobertaModel(BertModel):
    config_class = RobertaConfig
    base_model_prefix = "roberta"
    embeddings_class = RobertaEmbeddings

    @classmethod
    def default_config(cls, name: str):
        large = "large" in name
        return RobertaConfig(vocab_size=50265, hidden_size=1024 if large else 768,
                             num_hidden_layers=24 if large else 12, num_attention_heads=16 if large else 12,
                             intermediate_size=4096 if large else 3072, max_position_embeddings=514,
                             type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1)
