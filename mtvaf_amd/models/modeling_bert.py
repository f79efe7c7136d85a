"""BERT encoder with per-layer visual prefix K/V, MI355X-native.

Drop-in for the reference's ``models/modeling_bert.py`` classes that are on the hot path
(BertEmbeddings :165-222, BertSelfAttention :225-342, BertSelfOutput :345-356, BertAttention :359-407,
BertIntermediate :410-422, BertOutput :425-436, BertLayer :439-522, BertEncoder :525-620,
BertPooler :720-732, BertModel :943-1157): same class names, constructor arguments, ``forward``
signatures, parameter / buffer names (so checkpoints and the trainer's name-based optimizer groups
work unchanged) -- but the sub-modules are only parameter containers: the arithmetic of a whole
forward/backward runs in hand-written HIP kernels through ``mtvaf_amd.engine``.

Deliberate deviations (documented in DESIGN.md):
  * ``attentions`` is returned as ``None``: the [B,12,S,P+S] probability tensors are never
    materialised (nothing in the reference reads them: models/bert_model.py:496-506).
  * head pruning, cross-attention/decoder mode, relative position embeddings, gradient checkpointing
    and ``head_mask`` are not part of the path and raise if requested.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
from torch import nn
from transformers import BertConfig
from transformers.modeling_outputs import (BaseModelOutputWithPastAndCrossAttentions,
                                           BaseModelOutputWithPoolingAndCrossAttentions)

from .. import engine, hip


def _cfg_get(config, name, default):
    v = getattr(config, name, default)
    return default if v is None else v


class BertEmbeddings(nn.Module):
    """reference: models/modeling_bert.py:165-222"""

    roberta = False

    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=config.pad_token_id)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.position_embedding_type = _cfg_get(config, "position_embedding_type", "absolute")
        if self.position_embedding_type != "absolute":
            raise NotImplementedError("only absolute position embeddings are on the MTVAF path")
        self.register_buffer("position_ids", torch.arange(config.max_position_embeddings).expand((1, -1)))
        self.register_buffer("token_type_ids", torch.zeros(self.position_ids.size(), dtype=torch.long), persistent=False)
        self.padding_idx = config.pad_token_id if config.pad_token_id is not None else -1

    def forward(self, input_ids=None, token_type_ids=None, position_ids=None, inputs_embeds=None,
                past_key_values_length=0):
        if inputs_embeds is not None or position_ids is not None:
            raise NotImplementedError("inputs_embeds / explicit position_ids are not on the MTVAF path")
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        p = self.dropout.p if self.training else 0.0
        return engine.EmbeddingsFunction.apply(
            self.word_embeddings.weight, self.position_embeddings.weight, self.token_type_embeddings.weight,
            self.LayerNorm.weight, self.LayerNorm.bias, input_ids, token_type_ids, self.LayerNorm.eps, p, self.roberta,
            self.padding_idx)


class BertSelfAttention(nn.Module):
    """Parameter container; reference: models/modeling_bert.py:225-342 (prefix concat at :282-286)."""

    def __init__(self, config):
        super().__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError(f"The hidden size ({config.hidden_size}) is not a multiple of the number of attention "
                             f"heads ({config.num_attention_heads})")
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = config.hidden_size // config.num_attention_heads
        self.all_head_size = config.hidden_size
        if self.attention_head_size != 64:
            raise NotImplementedError("the gfx950 attention kernel is built for head_dim 64 (BERT/RoBERTa base/large)")
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)


class BertSelfOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)


class BertAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)


class BertIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)
        act = config.hidden_act
        if not (isinstance(act, str) and act == "gelu"):
            raise NotImplementedError("the fused epilogue implements the exact erf GELU of BERT/RoBERTa")


class BertOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)


class BertLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.attention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)

    def ordered_params(self) -> List[nn.Parameter]:
        # nn.Module attribute lookups are slow (16 parameters x 12 layers x 2 per step add up in the launch-bound
        # configurations): the Parameter objects are cached and the cache is dropped if the first / last one is replaced
        c = self.__dict__.get("_ordered")
        if c is not None and c[0] is self.attention.self.query._parameters["weight"] and \
                c[15] is self.output.LayerNorm._parameters["bias"]:
            return c
        a, s = self.attention, self.attention.self
        c = [s.query.weight, s.query.bias, s.key.weight, s.key.bias, s.value.weight, s.value.bias,
             a.output.dense.weight, a.output.dense.bias, a.output.LayerNorm.weight, a.output.LayerNorm.bias,
             self.intermediate.dense.weight, self.intermediate.dense.bias,
             self.output.dense.weight, self.output.dense.bias, self.output.LayerNorm.weight, self.output.LayerNorm.bias]
        self.__dict__["_ordered"] = c
        return c


class _LayerStore:
    """Flat fp32 storage of one layer's parameters (and a same-shaped gradient buffer).  The module's
    nn.Parameters are re-pointed at views of ``flat`` so that Q/K/V form one packed [3H,H] operand for
    the fused QKV GEMM while ``named_parameters()`` still exposes the reference names."""

    def __init__(self, layer: BertLayer):
        ps = layer.ordered_params()
        H = ps[0].shape[1]
        order = [0, 2, 4, 1, 3, 5] + list(range(6, 16))  # wq wk wv bq bk bv ...
        dev = ps[0].device
        n = sum(p.numel() for p in ps)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = None
        self.offsets = {}
        off = 0
        with torch.no_grad():
            for i in order:
                p = ps[i]
                v = self.flat[off:off + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                self.offsets[i] = off
                off += p.numel()
        self.H = H
        self.ptrs = [p.data_ptr() for p in ps]
        w = engine.LayerWeights()
        w.wqkv = self.flat[:3 * H * H].view(3 * H, H)
        w.bqkv = self.flat[3 * H * H:3 * H * H + 3 * H]
        (w.wo, w.bo, w.g1, w.b1, w.w1, w.bi1, w.w2, w.bi2, w.g2, w.b2) = [p.data for p in ps[6:]]
        w.wparams = (ps[0], ps[2], ps[4], ps[6], ps[10], ps[12])  # the weight matrices' Parameters (version counters)
        w._h = None
        w._pl = None
        w._st = None
        w._gst = None
        w.flat = self.flat
        self.weights = w
        self.shapes = [p.shape for p in ps]

    def valid(self, layer: BertLayer) -> bool:
        return all(p.data_ptr() == q for p, q in zip(layer.ordered_params(), self.ptrs))

    def grad_views(self):
        if self.grad is None:
            self.grad = torch.empty_like(self.flat)
            self._gv = None
        if getattr(self, "_gv", None) is None:  # the views are as persistent as the buffer: built once
            self._gv = [self.grad[self.offsets[i]:self.offsets[i] + s.numel()].view(s) for i, s in enumerate(self.shapes)]
        return self._gv

    def packed_qkv_grad(self):
        H = self.H
        return self.grad[:3 * H * H].view(3 * H, H), self.grad[3 * H * H:3 * H * H + 3 * H]


class GradSink:
    """Hands the encoder backward its parameter-gradient destinations.

    Fast path (every ``param.grad is None``, i.e. after ``zero_grad(set_to_none=True)``, and exactly ONE encoder
    node in the autograd pass): kernels write into the per-layer flat gradient buffers and autograd adopts the
    returned views without a copy, so a data-parallel hook can all-reduce one contiguous buffer per layer as soon
    as it is produced.  Otherwise fresh tensors are returned and autograd accumulates them (gradient accumulation).

    Two encoder nodes under one ``loss.backward()`` (the reference's cutoff flow runs ``model(...)`` and
    ``model(..., augument=True)`` and backpropagates the summed loss, modules/train.py:414-455) both see
    ``param.grad is None`` -- AccumulateGrad only runs once both edges have arrived -- so the flat buffer must not be
    handed to both: the second node would overwrite the views already sitting in autograd's input buffer.  Nodes are
    therefore counted when they are created (``node_created``) and the flat buffers are used only while a single one
    is outstanding; the count is cleared when the backward pass that consumed them ends."""

    def __init__(self, stores: List[_LayerStore]):
        self.stores = stores
        self.on_layer_done = None  # callable(layer_index, flat_grad_tensor) or None
        self.fast = False
        self.settle_params = False  # set by an optimizer that updates parameters from the hook (mtvaf_amd.optim.AdamW)
        self.raw_stream_hook = False  # the hook only enqueues library kernels on hip._st() (no torch stream semantics needed)
        self.token_rows = 0  # token rows of the backward pass in flight (how long a layer's backward is: optim.AdamW)
        self.optimizer = None  # the mtvaf_amd.optim.AdamW attached to this encoder, if any (GradSync re-wires through it)
        self.live_nodes = 0
        self._reset_armed = False

    def node_created(self):
        self.live_nodes += 1

    def _pass_done(self):
        self.live_nodes = 0
        self._reset_armed = False

    def acquire(self, params):
        if not self._reset_armed:  # (always called from inside a backward pass)
            self._reset_armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self._pass_done)
        self.fast = (self.live_nodes <= 1 and all(p.requires_grad for p in params)
                     and all(p.grad is None for p in params))
        if not self.fast:
            return None
        views = []
        for st in self.stores:
            views.extend(st.grad_views())
        return views

    def packed_qkv(self, li):
        return self.stores[li].packed_qkv_grad()

    def layer_done(self, li):
        if self.on_layer_done is not None:
            self.on_layer_done(li, self.stores[li].grad if self.fast else None)


class BertEncoder(nn.Module):
    """reference: models/modeling_bert.py:525-620"""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.layer = nn.ModuleList([BertLayer(config) for _ in range(config.num_hidden_layers)])
        self.gradient_checkpointing = False
        self._stores: Optional[List[_LayerStore]] = None
        self._sink: Optional[GradSink] = None

    def _prepare(self):
        if self._stores is None or not all(st.valid(l) for st, l in zip(self._stores, self.layer)):
            old = self._sink
            self._stores = [_LayerStore(l) for l in self.layer]
            self._sink = GradSink(self._stores)
            if old is not None:
                self._sink.on_layer_done, self._sink.settle_params = old.on_layer_done, old.settle_params
                self._sink.raw_stream_hook = old.raw_stream_hook
                self._sink.optimizer = old.optimizer
        return self._stores, self._sink

    @property
    def grad_sink(self) -> GradSink:
        return self._prepare()[1]

    def forward(self, hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, past_key_values=None, use_cache=None, output_attentions=False,
                output_hidden_states=False, return_dict=True, unpad_ok=False):
        """``attention_mask`` is the additive extended mask [B,1,1,T] (or [B,T]) as in the reference.  ``unpad_ok``: the
        caller does not read hidden states at masked positions (padding-free execution may zero them, engine.UNPAD)."""
        if encoder_hidden_states is not None or (head_mask is not None and any(h is not None for h in head_mask)):
            raise NotImplementedError("cross-attention / head_mask are not on the MTVAF path")
        if not hidden_states.is_cuda:
            raise RuntimeError("mtvaf_amd runs on an MI355X only (no CPU fallback): move the model and inputs to cuda")
        stores, sink = self._prepare()
        B, S, H = hidden_states.shape
        cfg = self.config
        pkv = pack_prefix(past_key_values, B, H)
        Pn = 0 if pkv is None else pkv.shape[3] // H
        if attention_mask is None:
            addmask = torch.zeros(B, Pn + S, device=hidden_states.device, dtype=torch.float32)
        else:
            addmask = attention_mask.reshape(B, -1).to(torch.float32).contiguous()
        if addmask.shape[1] != Pn + S:
            raise ValueError(f"attention mask covers {addmask.shape[1]} keys, expected prefix {Pn} + text {S}")
        tr = self.training
        ecfg = (cfg.num_attention_heads, cfg.layer_norm_eps, cfg.hidden_dropout_prob if tr else 0.0,
                cfg.attention_probs_dropout_prob if tr else 0.0, getattr(past_key_values, "ready_event", None), bool(unpad_ok))
        params = [p for l in self.layer for p in l.ordered_params()]
        outs = engine.EncoderFunction.apply(hidden_states, pkv, addmask, ecfg, [st.weights for st in stores],
                                            sink if torch.is_grad_enabled() else None, *params)
        all_hidden = (hidden_states,) + tuple(outs) if output_hidden_states else None
        if all_hidden is not None and len(outs) > 1 and outs[0].dim() == 2:
            all_hidden = _LazyHiddenStates(all_hidden, engine.LAST_PACK)  # padding-free run: intermediate states are packed
        if not return_dict:
            return tuple(v for v in [outs[-1], all_hidden] if v is not None)
        return BaseModelOutputWithPastAndCrossAttentions(last_hidden_state=outs[-1], past_key_values=None,
                                                         hidden_states=all_hidden, attentions=None,
                                                         cross_attentions=None)


class _LazyHiddenStates(tuple):
    """``hidden_states`` of a padding-free run (engine.UNPAD): the intermediate layers' outputs exist as PACKED rows of the
    unmasked tokens; an entry is scattered to [B,S,H] (zeros at masked positions, detached) when it is first read --
    the MTVAF path itself only reads ``last_hidden_state``, which is always materialised."""

    def __new__(cls, items, pack):
        obj = super().__new__(cls, items)
        obj._pack, obj._cache = pack, {}
        return obj

    def __getitem__(self, i):
        if isinstance(i, slice):
            return tuple(self[j] for j in range(*i.indices(len(self))))
        v = super().__getitem__(i)
        if v.dim() == 2:
            i = i % len(self)
            if i not in self._cache:
                self._cache[i] = self._pack.unpack(v).view(self._pack.B, self._pack.S, v.shape[1])
            v = self._cache[i]
        return v

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class PrefixKV(list):
    """The list of per-layer (key, value) tuples the reference API expects (models/bert_model.py:586-588),
    carrying the packed [L,2,B,P*H] tensor the kernels read so that no re-packing copy is needed."""

    def __init__(self, flat: torch.Tensor, num_heads: int, head_dim: int):
        L, _, B, PH = flat.shape
        P = PH // (num_heads * head_dim)
        super().__init__((flat[i, 0].view(B, num_heads, P, head_dim), flat[i, 1].view(B, num_heads, P, head_dim))
                         for i in range(L))
        self.flat = flat
        # set when the prompt generator ran on a second stream: the encoder waits for it right before the first
        # attention kernel instead of at its entry (the embeddings and the layer-0 QKV product do not need the prefix)
        self.ready_event = None


def pack_prefix(past_key_values, B: int, H: int) -> Optional[torch.Tensor]:
    """-> [L,2,B,P*H] fp32 contiguous.  ``reshape(bsz,12,-1,64)`` slabs (bert_model.py:585) are already
    head-major contiguous, so packing is a flat view/stack."""
    if past_key_values is None:
        return None
    if isinstance(past_key_values, PrefixKV):
        return past_key_values.flat
    if torch.is_tensor(past_key_values):
        return past_key_values.contiguous()
    rows = []
    for kv in past_key_values:
        k, v = kv[0], kv[1]
        if k.shape[0] != B or k.shape[1] * k.shape[3] != H:
            raise ValueError(f"prefix key shape {tuple(k.shape)} does not match batch {B} / hidden {H}")
        rows.append(torch.stack([k.reshape(B, -1), v.reshape(B, -1)]))
    return torch.stack(rows).to(torch.float32).contiguous()


class BertPooler(nn.Module):
    """reference: models/modeling_bert.py:720-732"""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.activation = nn.Tanh()

    def forward(self, hidden_states):
        first = hidden_states[:, 0].contiguous()
        return engine.LinearFunction.apply(first, self.dense.weight, self.dense.bias, True)


class BertModel(nn.Module):
    """reference: models/modeling_bert.py:943-1157"""

    config_class = BertConfig
    base_model_prefix = "bert"
    embeddings_class = BertEmbeddings

    def __init__(self, config, add_pooling_layer=True):
        super().__init__()
        self.config = config
        if _cfg_get(config, "is_decoder", False) or _cfg_get(config, "add_cross_attention", False):
            raise NotImplementedError("decoder / cross-attention mode is not on the MTVAF path")
        self.embeddings = self.embeddings_class(config)
        self.encoder = BertEncoder(config)
        self.pooler = BertPooler(config) if add_pooling_layer else None
        self.init_weights()

    # -- weight init / loading ---------------------------------------------------------------------
    def _init_weights(self, module):
        """reference: models/modeling_bert.py:816-830"""
        std = _cfg_get(self.config, "initializer_range", 0.02)
        if isinstance(module, nn.Linear):
            module.weight.data.normal_(mean=0.0, std=std)
            if module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.Embedding):
            module.weight.data.normal_(mean=0.0, std=std)
            if module.padding_idx is not None:
                module.weight.data[module.padding_idx].zero_()
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)

    def init_weights(self):
        self.apply(self._init_weights)

    @classmethod
    def from_pretrained(cls, name_or_path, *model_args, **kwargs):
        """``BertModel.from_pretrained(args.bert_name)`` as the reference calls it (models/bert_model.py:425-429): a local
        directory, or a hub NAME (``bert-base-uncased``, ``roberta-base`` ...) resolved through the local Hugging Face
        cache (``HF_HUB_CACHE`` / ``~/.cache/huggingface/hub``, or ``cache_dir=``) without network access -- the target
        machines have none, so a name that is not cached raises.  Loads ``config.json`` + ``model.safetensors`` /
        ``pytorch_model.bin``; keys may carry the ``bert.`` / ``roberta.`` prefix of task checkpoints.  With
        ``MTVAF_RANDOM_INIT=1`` a missing checkpoint yields a random-init model of the named architecture (synthetic
        benchmarks)."""
        config = kwargs.pop("config", None)
        path = cls._resolve_checkpoint_dir(str(name_or_path), kwargs.pop("cache_dir", None))
        if path is not None:
            if config is None:
                config = cls.config_class.from_pretrained(path)
            model = cls(config, *model_args)
            sd = None
            st = os.path.join(path, "model.safetensors")
            pt = os.path.join(path, "pytorch_model.bin")
            if os.path.exists(st):
                from safetensors.torch import load_file
                sd = load_file(st)
            elif os.path.exists(pt):
                sd = torch.load(pt, map_location="cpu")
            if sd is None:
                raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {path}")
            model.load_reference_state_dict(sd)
            return model
        if os.environ.get("MTVAF_RANDOM_INIT", "0") == "1":
            if config is None:
                config = cls.default_config(str(name_or_path))
            return cls(config, *model_args)
        raise FileNotFoundError(
            f"{name_or_path!r} is neither a local checkpoint directory nor a model in the local Hugging Face cache (no hub "
            "access here); set MTVAF_RANDOM_INIT=1 for a random-init model of that architecture")

    @staticmethod
    def _resolve_checkpoint_dir(name: str, cache_dir=None) -> Optional[str]:
        if os.path.isdir(name):
            return name
        try:
            from huggingface_hub import snapshot_download
            return snapshot_download(name, local_files_only=True, cache_dir=cache_dir or os.environ.get("HF_HUB_CACHE"),
                                     allow_patterns=["config.json", "model.safetensors", "pytorch_model.bin"])
        except Exception:  # not cached (LocalEntryNotFoundError), malformed repo id, hub library missing
            return None

    @classmethod
    def default_config(cls, name: str):
        large = "large" in name
        return BertConfig(hidden_size=1024 if large else 768, num_hidden_layers=24 if large else 12,
                          num_attention_heads=16 if large else 12, intermediate_size=4096 if large else 3072)

    def load_reference_state_dict(self, sd):
        own = self.state_dict()
        fixed = {}
        for k, v in sd.items():
            for pre in ("bert.", "roberta."):
                if k.startswith(pre):
                    k = k[len(pre):]
            k = k.replace("LayerNorm.gamma", "LayerNorm.weight").replace("LayerNorm.beta", "LayerNorm.bias")
            if k in own:
                fixed[k] = v
        missing = [k for k in own if k not in fixed and "position_ids" not in k]
        if any(not k.startswith("pooler.") for k in missing):
            raise KeyError(f"checkpoint lacks encoder weights: {missing[:5]} ...")
        self.load_state_dict(fixed, strict=False)

    # -- forward --------------------------------------------------------------------------------------
    def get_input_embeddings(self):
        return self.embeddings.word_embeddings

    def get_extended_attention_mask(self, attention_mask, input_shape=None, device=None):
        """(1 - mask) * -10000 in fp32, [B,1,1,T]  (models/modeling_bert.py:1064, :1134-1137)."""
        return (1.0 - attention_mask[:, None, None, :].to(torch.float32)) * -10000.0

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None,
                inputs_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None, past_key_values=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None):
        if input_ids is None:
            raise ValueError("You have to specify input_ids (inputs_embeds is not on the MTVAF path)")
        if head_mask is not None or encoder_hidden_states is not None:
            raise NotImplementedError("head_mask / encoder_hidden_states are not on the MTVAF path")
        output_hidden_states = bool(output_hidden_states)
        return_dict = True if return_dict is None else return_dict
        B, S = input_ids.shape
        if attention_mask is None:
            attention_mask = torch.ones((B, S), device=input_ids.device)  # past_key_values_length = 0 (:1050)
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        ext = self.get_extended_attention_mask(attention_mask)
        unpad_ok = bool(getattr(self, "allow_unpad", False))
        if engine.UNPAD and unpad_ok:  # padding-free execution: the packing maps are built while the host enqueues the embeddings
            engine.Packing.begin(ext.view(B, -1), ext.shape[-1] - S, S)
        emb = self.embeddings(input_ids=input_ids, token_type_ids=token_type_ids, position_ids=position_ids,
                              inputs_embeds=inputs_embeds, past_key_values_length=0)
        enc = self.encoder(emb, attention_mask=ext, past_key_values=past_key_values,
                           output_hidden_states=output_hidden_states, return_dict=True, unpad_ok=unpad_ok)
        seq = enc.last_hidden_state
        # the pooler output is consumed only by the span model's DualGCN head; TVNetSAModel2 never reads it
        # (models/bert_model.py:496-506) and sets `skip_pooler` so the [B,H]x[H,H] product is not launched
        pooled = self.pooler(seq) if (self.pooler is not None and not getattr(self, "skip_pooler", False)) else None
        if not return_dict:
            return (seq, pooled) + ((enc.hidden_states,) if enc.hidden_states is not None else ())
        return BaseModelOutputWithPoolingAndCrossAttentions(last_hidden_state=seq, pooler_output=pooled,
                                                            past_key_values=None, hidden_states=enc.hidden_states,
                                                            attentions=None, cross_attentions=None)

    def get_embedding_output(self, input_ids, token_type_ids=None, position_ids=None):
        """reference: models/modeling_bert.py:1117-1125 (Cutoff augmentation entry)"""
        assert input_ids is not None
        return self.embeddings(input_ids=input_ids, token_type_ids=token_type_ids, position_ids=position_ids)

    def get_bert_output(self, embedding_output, attention_mask=None, past_key_values=None):
        """reference: models/modeling_bert.py:1127-1157"""
        assert attention_mask.dim() == 2
        ext = self.get_extended_attention_mask(attention_mask)
        enc = self.encoder(embedding_output, attention_mask=ext, past_key_values=past_key_values, return_dict=True,
                           unpad_ok=bool(getattr(self, "allow_unpad", False)))
        seq = enc.last_hidden_state
        pooled = self.pooler(seq) if (self.pooler is not None and not getattr(self, "skip_pooler", False)) else None
        return (seq, pooled)
