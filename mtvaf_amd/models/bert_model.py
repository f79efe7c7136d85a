"""TVNetSAModel2 (CRF tagger, the live model) and TVNetSAModel (span variant) -- the MTVAF task models (BERT/RoBERTa + visual prefix + CRF tagger), MI355X-native.

Drop-in for the reference's ``models/bert_model.py::TVNetSAModel2`` (:416-588): same constructor
``(label_list, tokenizer, args, type_num=None, use_weight=False)``, same ``forward`` signature and
``TokenClassifierOutput(loss, logits=List[List[int]])`` result, same ``get_visual_prompt`` contract and
the same parameter names (``bert.*``, ``fc.*``, ``crf.*``, ``encoder_conv.{0,2}.*``, ``projectors.{i}.*``,
``img_classifier.*``, ``aux_img_classifier.{k}.*``, ``image_model.resnet.*``), so it can be handed to the
reference's ``modules/train.py::SATrainer2`` unchanged.  All arithmetic of the path
runs in the gfx950 kernels behind ``mtvaf_amd.engine``; the sub-modules are parameter containers.

Reference defects the boundary survives: undefined ``args.use_101/use_34/use_18``
flags are read with ``getattr(..., False)``; ``args.n_gpu > 1`` takes the plain loss path (one process
per GPU; the reference's DataParallelCriterion branch, bert_model.py:515-519, cannot run).
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import nn
from transformers.modeling_outputs import TokenClassifierOutput

from .. import engine
from ..modules.crf import CRF
from .modeling_bert import BertModel, PrefixKV
from .modeling_roberta import RobertaModel


def _arg(args, name, default=False):
    return getattr(args, name, default)


def _on_second_stream(fn, inputs, join=False):
    """Run the prompt generator (small, low-occupancy GEMMs and mixing kernels) on the engine's second stream so it
    overlaps the embeddings and the first QKV product; the encoder waits for the prefix right before its first
    attention kernel (``PrefixKV.ready_event``).  Autograd runs the generator's backward on the same stream, next to
    the embeddings' backward.  Small batches / CPU tensors / MTVAF_DW_STREAM=0: plain call."""
    ts = [t for t in inputs if isinstance(t, torch.Tensor)]
    if not (engine.DW_SIDE_STREAM and ts and ts[0].is_cuda and ts[0].shape[0] >= 8):  # small batches are host-bound
        return fn()
    main, side = torch.cuda.current_stream(), engine._side_stream(ts[0].device)
    side.wait_stream(main)
    _accumulate_on_producer_stream()
    with torch.cuda.stream(side):
        out = fn()
        ev = torch.cuda.Event()
        ev.record(side)
    for t in ts:
        t.record_stream(side)
    pkv = out[0] if isinstance(out, tuple) else out
    if join or not isinstance(pkv, PrefixKV):
        main.wait_event(ev)
    else:
        pkv.ready_event = ev
    return out


_warned_off = False


def _accumulate_on_producer_stream():
    """The generator's parameter gradients are PRODUCED on the second stream (autograd runs a node's backward on the stream
    its forward ran on), while their AccumulateGrad nodes were created on the default stream when the parameters were: torch
    (>= 2.9) warns about that mismatch on every backward pass.  It is intended here, and it costs no host synchronisation --
    the engine orders the accumulation behind the producer with a stream-to-stream event wait, and the encoder backward
    joins the two streams anyway -- so the warning is switched off once."""
    global _warned_off
    if not _warned_off:
        _warned_off = True
        fn = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if fn is not None:
            fn(False)


class ImageModel(nn.Module):
    """Frozen ResNet pyramid front-end (reference: models/bert_model.py:63-111).  It is UPSTREAM of the
    accelerated path and runs in plain torch (MIOpen convolutions) on the trunks of
    ``mtvaf_amd/models/resnet.py`` (torchvision's architecture and state_dict names; torchvision itself is not
    required).  ``resnet_root`` holds the reference's ``resnetNN.pth`` files; ``resnet_root="random"`` skips the
    load (synthetic runs).  Pre-extracted region features bypass it (``TVNetSAModel2.get_visual_prompt``,
    ``mtvaf_amd.features``)."""

    def __init__(self, use_152=False, use_101=False, use_34=False, use_18=False, resnet_root=None):
        super().__init__()
        from .resnet import resnet18, resnet34, resnet50, resnet101, resnet152
        name, ctor = (("resnet152", resnet152) if use_152 else ("resnet101", resnet101) if use_101 else
                      ("resnet34", resnet34) if use_34 else ("resnet18", resnet18) if use_18 else ("resnet50", resnet50))
        self.resnet = ctor()
        if resnet_root is not None and resnet_root != "random":
            self.resnet.load_state_dict(torch.load(f"{resnet_root}/{name}.pth", map_location="cpu"))

    def forward(self, x, aux_imgs=None):
        prefix_guids = self.get_resnet_prompt(x)
        if aux_imgs is not None:
            aux_imgs = aux_imgs.permute([1, 0, 2, 3, 4])
            return prefix_guids, [self.get_resnet_prompt(aux_imgs[i]) for i in range(len(aux_imgs))]
        return prefix_guids, None

    def get_resnet_prompt(self, x):
        out = []
        for name, layer in self.resnet.named_children():
            if name in ("fc", "avgpool"):
                continue
            x = layer(x)
            if "layer" in name:
                kernel = x.size(2) // 2
                out.append(nn.functional.avg_pool2d(x, kernel_size=(kernel, kernel), stride=kernel))
        return out


class _PackedLinears:
    """Views the weights of a ModuleList of equal nn.Linear(in, out) as one [n*out, in] operand."""

    def __init__(self, mods: nn.ModuleList):
        n, out, inn = len(mods), mods[0].out_features, mods[0].in_features
        dev = mods[0].weight.device
        self.w = torch.empty(n * out, inn, device=dev)
        self.b = torch.empty(n * out, device=dev)
        with torch.no_grad():
            for i, m in enumerate(mods):
                self.w[i * out:(i + 1) * out].copy_(m.weight.data)
                self.b[i * out:(i + 1) * out].copy_(m.bias.data)
                m.weight.data = self.w[i * out:(i + 1) * out]
                m.bias.data = self.b[i * out:(i + 1) * out]
        self.ptrs = [m.weight.data_ptr() for m in mods] + [m.bias.data_ptr() for m in mods]

    def valid(self, mods) -> bool:
        return self.ptrs == [m.weight.data_ptr() for m in mods] + [m.bias.data_ptr() for m in mods]


class TVNetSAModel2(nn.Module):
    def __init__(self, label_list, tokenizer, args, type_num=None, use_weight=False):
        super().__init__()
        self.args = args
        self.type_num = type_num
        self.tokenizer = tokenizer
        self.prefix_dim = _arg(args, "prefix_dim", 768)
        self.prefix_len = _arg(args, "prefix_len", 4)

        enc_cls = RobertaModel if "roberta" in args.bert_name else BertModel
        bert_config = _arg(args, "bert_config", None)  # extension: explicit config => random-init encoder
        if bert_config is not None:
            self.bert = enc_cls(bert_config)
        else:
            self.bert = enc_cls.from_pretrained(args.bert_name)
        self.bert.skip_pooler = True  # pooler_output is unused on this path
        # this head reads hidden states through the mask only (fc -> CRF with mask=attention_mask): padding-free execution
        # (engine.UNPAD) may leave zeros at masked positions.  TVNetSAModel's position softmax reads every position: no flag.
        self.bert.allow_unpad = True
        hidden = self.bert.config.hidden_size
        self.num_labels = len(label_list) + 1

        if _arg(args, "use_prefix"):
            small = _arg(args, "use_34") or _arg(args, "use_18")
            self.feat_dim = 960 if small else 3840
            if _arg(args, "resnet_root", None) is not None:
                self.image_model = ImageModel(use_152=_arg(args, "use_152"), use_101=_arg(args, "use_101"),
                                              use_34=_arg(args, "use_34"), use_18=_arg(args, "use_18"),
                                              resnet_root=args.resnet_root)
            else:
                self.image_model = None  # region features are fed directly (synthetic / cached features)
            self.encoder_conv = nn.Sequential(nn.Linear(self.feat_dim, 800), nn.Tanh(), nn.Linear(800, 4 * 2 * hidden))
            n_layers = self.bert.config.num_hidden_layers
            self.projectors = nn.ModuleList([nn.Linear(4 * hidden * 2, 4) for _ in range(n_layers)])
            self.img_dropout = nn.Dropout(0.2)
            self.img_classifier = nn.Linear(4 * 2 * hidden, 2089)
            self.aux_img_classifier = nn.ModuleList([nn.Linear(4 * 2 * hidden, 2089) for _ in range(3)])
            self._packed_proj: Optional[_PackedLinears] = None

        self.crf = CRF(self.num_labels, batch_first=True)
        self.fc = nn.Linear(hidden, self.num_labels)
        self.dropout = nn.Dropout(0.1)
        if _arg(args, "use_probe"):
            raise NotImplementedError("the structural probe (probes/) is off the hot path and its import chain is "
                                      "broken in the reference (models/bert_model.py:468-475)")

    # ------------------------------------------------------------------------------------------------
    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, labels=None, imagelabel=None,
                images=None, aux_imgs=None):
        """reference: models/bert_model.py:480-532"""
        bsz = input_ids.size(0)
        img_tag_loss = 0
        if _arg(self.args, "use_prefix"):
            prefix_guids, img_tag_loss, aux_img_tag_loss = _on_second_stream(
                lambda: self.get_visual_prompt(images, aux_imgs, imagelabel), (images, aux_imgs, imagelabel),
                join=_arg(self.args, "vao"))  # the VAO losses are consumed on the main stream right away
            img_tag_loss = img_tag_loss if _arg(self.args, "noauxloss") else img_tag_loss + sum(aux_img_tag_loss)
            prefix_len = prefix_guids[0][0].shape[2]
            prefix_mask = torch.ones((bsz, prefix_len), device=attention_mask.device, dtype=attention_mask.dtype)
            prompt_attention_mask = torch.cat((prefix_mask, attention_mask), dim=1)
        else:
            prefix_guids = None
            prompt_attention_mask = attention_mask
        bert_output = self.bert(input_ids=input_ids, attention_mask=prompt_attention_mask,
                                token_type_ids=token_type_ids, past_key_values=prefix_guids, output_attentions=True,
                                output_hidden_states=True, return_dict=True)
        sequence_output = engine.dropout(bert_output["last_hidden_state"], self.dropout.p, self.training)
        emissions = engine.LinearFunction.apply(sequence_output, self.fc.weight, self.fc.bias, False)
        mask_u8 = attention_mask.to(torch.uint8)
        # Viterbi paths: device kernel + async packed copy; the list materialises on first use (no mid-step sync).
        # One wavefront per sentence is all the parallelism Viterbi and the CRF forward algorithm have, so the two
        # run side by side (second stream) instead of back to back.
        if emissions.is_cuda and engine.DW_SIDE_STREAM and emissions.shape[0] * emissions.shape[1] >= 1024:  # host-bound below
            main, side = torch.cuda.current_stream(), engine._side_stream(emissions.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                logits = self.crf.decode_deferred(emissions.detach(), mask_u8)
                decoded = torch.cuda.Event()
                decoded.record(side)
            emissions.record_stream(side)
            mask_u8.record_stream(side)
        else:
            logits = self.crf.decode_deferred(emissions, mask_u8)
            decoded = None
        loss = None
        if labels is not None:
            loss = self.crf.nll_mean(emissions, labels, mask=mask_u8)  # = -1 * crf(..., reduction='mean'), bert_model.py:521
            extra = _arg(self.args, "alpha", 0.0) * img_tag_loss
            if torch.is_tensor(extra) or extra != 0:  # (adding the literal 0.0 of a VAO-less run is two kernels for nothing)
                loss = loss + extra
        if decoded is not None:
            # the decode reads the CRF parameters: whatever follows on the main stream (optimizer.step() updates them in
            # place) is ordered behind it here, not only by the join inside the encoder backward (frozen encoders skip it)
            torch.cuda.current_stream().wait_event(decoded)
        return TokenClassifierOutput(loss=loss, logits=logits)

    # ------------------------------------------------------------------------------------------------
    def _region_features(self, images, aux_imgs):
        """-> (feats [B,4,F], [aux feats [B,4,F]]).  Accepts raw images (through the frozen ResNet,
        bert_model.py:536-539) or pre-extracted pyramid features [B,F,2,2] / [B,4,F] (aux: [B,n,...])."""
        bsz = images.size(0)
        L = self.prefix_len if self.prefix_len == 4 else 4  # Linear(3840, .) fixes prefix_len = 4
        raw = images.dim() == 4 and images.shape[1] == 3
        if raw:
            if self.image_model is None:
                raise RuntimeError("raw images were passed but the model was built without args.resnet_root")
            pyr, aux_pyr = self.image_model(images, aux_imgs)
            feats = torch.cat(pyr, dim=1).view(bsz, L, -1)
            aux = [torch.cat(a, dim=1).view(bsz, L, -1) for a in (aux_pyr or [])]
            return feats.float(), [a.float() for a in aux]
        feats = images.reshape(bsz, L, -1).float()
        aux = []
        if aux_imgs is not None:
            aux = [aux_imgs[:, i].reshape(bsz, L, -1).float() for i in range(aux_imgs.shape[1])]
        return feats, aux

    def get_visual_prompt(self, images, aux_imgs, imagelabel):
        """reference: models/bert_model.py:534-588.  Returns (list of num_layers (K, V) [B,NH,P,64],
        img_tag_loss, [aux_img_tag_loss])."""
        feats, aux = self._region_features(images, aux_imgs)
        bsz, L, Fd = feats.shape
        cfg = self.bert.config
        hidden = cfg.hidden_size
        NI = 1 + len(aux)
        x = torch.stack([feats] + aux).reshape(NI * bsz * L, Fd)  # image-major rows
        e0, e2 = self.encoder_conv[0], self.encoder_conv[2]
        t = engine.LinearFunction.apply(x, e0.weight, e0.bias, True)
        enc = engine.LinearFunction.apply(t, e2.weight, e2.bias, False).view(NI, bsz, L, 8 * hidden)

        img_tag_loss = 0
        aux_img_tag_loss = []
        if _arg(self.args, "vao"):
            means = engine.MeanLFunction.apply(enc.view(NI * bsz, L, 8 * hidden)).view(NI, bsz, 8 * hidden)
            target = imagelabel.to(means.device)
            heads = [self.img_classifier] + list(self.aux_img_classifier)
            if NI - 1 > len(self.aux_img_classifier):
                raise ValueError("the VAO branch supports at most 3 aux images (bert_model.py:459)")
            for k in range(NI):
                m = engine.dropout(means[k], self.img_dropout.p, self.training)
                z = engine.LinearFunction.apply(m, heads[k].weight, heads[k].bias, False)
                l = engine.KLFunction.apply(z, target)
                if k == 0:
                    img_tag_loss = l
                else:
                    aux_img_tag_loss.append(l)

        if self._packed_proj is None or not self._packed_proj.valid(self.projectors):
            self._packed_proj = _PackedLinears(self.projectors)
        pp = self._packed_proj
        NL = len(self.projectors)
        proj_params = [p for m in self.projectors for p in (m.weight, m.bias)]
        pkv = engine.PromptFunction.apply(enc, pp.w, pp.b, NL, *proj_params)
        return PrefixKV(pkv, cfg.num_attention_heads, hidden // cfg.num_attention_heads), img_tag_loss, aux_img_tag_loss



# ====================================================================================================
def flatten(x):
    """reference: models/bert_model.py:113-124"""
    if x.dim() == 2:
        return x.reshape(x.shape[0] * x.shape[1])
    if x.dim() == 3:
        return x.reshape(x.shape[0] * x.shape[1], x.shape[2])
    raise Exception()


def reconstruct(x, ref):
    """reference: models/bert_model.py:127-138"""
    if x.dim() == 1:
        return x.view(ref.shape[0], ref.shape[1])
    if x.dim() == 2:
        return x.view(ref.shape[0], ref.shape[1], x.shape[1])
    raise Exception()


class TVNetSAModel(nn.Module):
    """Span-extraction variant (reference models/bert_model.py:192-414): the same prefix-fused encoder, a
    start/end extraction head (``binary_affine``) trained with distant cross entropy, and a span classifier
    (gather the span's tokens -> ``unary_affine`` self-attention pooling -> ``dense``+tanh -> ``classifier``).
    Same constructor, ``forward`` / ``extraction`` / ``classification`` / ``get_visual_prompt`` signatures and
    parameter names as the reference, so ``modules/train.py::SATrainer`` drives it unchanged.

    The reference reads the widest span (``JR``) and the valid-token count back to the host inside
    ``get_span_representation`` (:160-165); here they stay in a device-side index block
    (``mtvaf_span_index``), so a training step of this model has no host sync either.

    ``augument=True`` runs the cutoff augmentation of ``modules/augument.py`` (``mtvaf_amd.modules.augument``).
    Not built (out of scope): the GCN branches (``gcn_layer_number`` / ``num_layers`` > 0,
    whose modules are missing from the reference checkout) and the structural probe."""

    def __init__(self, label_list, tokenizer, args, type_num=None, use_weight=False):
        super().__init__()
        self.args = args
        self.type_num = type_num
        self.tokenizer = tokenizer
        self.prefix_dim = _arg(args, "prefix_dim", 768)
        self.prefix_len = _arg(args, "prefix_len", 4)
        enc_cls = RobertaModel if "roberta" in args.bert_name else BertModel
        bert_config = _arg(args, "bert_config", None)
        self.bert = enc_cls(bert_config) if bert_config is not None else enc_cls.from_pretrained(args.bert_name)
        self.bert.skip_pooler = True  # pooler_output only feeds the (unbuilt) GCN branch (:349)
        hidden = self.bert.config.hidden_size
        self.dense = nn.Linear(hidden, hidden)
        self.activation = nn.Tanh()
        self.unary_affine = nn.Linear(hidden, 1)
        self.binary_affine = nn.Linear(hidden, 2)
        self.num_labels = len(label_list) + 1
        self.classifier = nn.Linear(hidden, 4)
        if _arg(args, "use_prefix"):
            small = _arg(args, "use_34") or _arg(args, "use_18")
            self.feat_dim = 960 if small else 3840
            if _arg(args, "resnet_root", None) is not None:
                self.image_model = ImageModel(use_152=_arg(args, "use_152"), use_101=_arg(args, "use_101"),
                                              use_34=_arg(args, "use_34"), use_18=_arg(args, "use_18"),
                                              resnet_root=args.resnet_root)
            else:
                self.image_model = None
            self.encoder_conv = nn.Sequential(nn.Linear(self.feat_dim, 800), nn.Tanh(), nn.Linear(800, 4 * 2 * hidden))
            self.projectors = nn.ModuleList([nn.Linear(4 * hidden * 2, 4)
                                             for _ in range(self.bert.config.num_hidden_layers)])
            self._packed_proj: Optional[_PackedLinears] = None
        self.fc = nn.Linear(hidden, self.num_labels)  # unused by forward, kept: it is in the reference's state_dict
        self.dropout = nn.Dropout(0.1)
        if _arg(args, "gcn_layer_number", 0) > 0 or _arg(args, "num_layers", 0) > 0:
            raise NotImplementedError("the GCN branches are outside the accelerated path (their modules are not in "
                                      "the reference checkout either: models/bert_model.py:233-237)")
        if _arg(args, "use_probe"):
            raise NotImplementedError("the structural probe (probes/) is off the hot path and its import chain is "
                                      "broken in the reference (models/bert_model.py:238-245)")

    # ------------------------------------------------------------------------------------------------
    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, start_positions=None,
                end_positions=None, span_starts=None, span_ends=None, polarity_labels=None, label_masks=None,
                images=None, aux_imgs=None, valid_ids=None, adjacency_matrix=None, output_attention=False,
                augument=False, labels=None, adj_matrix=None, src_mask=None, aspect_mask=None, polaritys=None):
        """reference: models/bert_model.py:246-321 (use_probe / GCN branches excluded)."""
        bsz = input_ids.size(0)
        if _arg(self.args, "use_prefix"):
            prefix_guids = _on_second_stream(lambda: self.get_visual_prompt(images, aux_imgs), (images, aux_imgs))
            prefix_len = prefix_guids[0][0].shape[2]
            prefix_mask = torch.ones((bsz, prefix_len), device=attention_mask.device, dtype=attention_mask.dtype)
            prompt_attention_mask = torch.cat((prefix_mask, attention_mask), dim=1)
        else:
            prefix_guids = None
            prompt_attention_mask = attention_mask
        ae_logits, sequence_output = self._extract(prompt_attention_mask, input_ids, prefix_guids, token_type_ids,
                                                   augument)
        logits, ac_logits = self.classification(attention_mask=attention_mask, span_starts=span_starts,
                                                span_ends=span_ends, sequence_input=sequence_output)
        flat_polarity_labels = flatten(polarity_labels)
        flat_label_masks = flatten(label_masks).to(dtype=ac_logits.dtype)
        # :298-300  (start_loss + end_loss) / 2, both distant cross entropies in one node
        ae_loss = engine.DistantCEPairFunction.apply(ae_logits, start_positions, end_positions)
        ac_loss = engine.CrossEntropyFunction.apply(ac_logits, flat_polarity_labels)
        ac_loss = torch.sum(flat_label_masks * ac_loss) / flat_label_masks.sum()  # :303, the reference's scalar quirk
        return TokenClassifierOutput(loss=ae_loss + ac_loss, logits=logits)

    def _extract(self, prompt_attention_mask, input_ids, prefix_guids, token_type_ids, augument=False):
        if augument:
            # :333-343 -- the cut input replaces the plain pass (the reference still runs the plain encoder first
            # and throws its output away; only hidden_states[7] of it feeds the unbuilt probe)
            from ..modules.augument import Cutoff
            cutoff = Cutoff(input_ids=input_ids, token_type_ids=token_type_ids, attention_masks=prompt_attention_mask,
                            prefix_guids=prefix_guids, args=self.args, model=self.bert)
            last_hidden = cutoff._training_step_with_cutoff(self.args.aug_type)[0]
        else:
            last_hidden = self.bert(input_ids=input_ids, attention_mask=prompt_attention_mask,
                                    token_type_ids=token_type_ids, past_key_values=prefix_guids, output_attentions=True,
                                    output_hidden_states=True, return_dict=True)["last_hidden_state"]
        sequence_output = engine.dropout(last_hidden, self.dropout.p, self.training)
        ae_logits = engine.LinearFunction.apply(sequence_output, self.binary_affine.weight, self.binary_affine.bias, False)
        return ae_logits, sequence_output

    def extraction(self, prompt_attention_mask, input_ids, prefix_guids, token_type_ids, augument=False, labels=None,
                   adj_matrix=None, src_mask=None, aspect_mask=None):
        """reference: models/bert_model.py:323-361 -> (start_logits [B,S], end_logits [B,S], sequence_output)."""
        ae_logits, sequence_output = self._extract(prompt_attention_mask, input_ids, prefix_guids, token_type_ids, augument)
        return ae_logits[..., 0], ae_logits[..., 1], sequence_output

    def classification(self, span_starts, span_ends, sequence_input, attention_mask):
        """reference: models/bert_model.py:363-376 -> (logits [B,M,4], ac_logits [B*M,4])."""
        M = span_starts.shape[1]
        index = engine.hip.span_index(attention_mask.to(torch.uint8).contiguous(), span_starts.contiguous().long(),
                                      span_ends.contiguous().long())
        pooled = engine.SpanPoolFunction.apply(sequence_input, self.unary_affine.weight, self.unary_affine.bias, index, M)
        pooled = engine.LinearFunction.apply(pooled, self.dense.weight, self.dense.bias, True)
        pooled = engine.dropout(pooled, self.dropout.p, self.training)
        ac_logits = engine.LinearFunction.apply(pooled, self.classifier.weight, self.classifier.bias, False)
        return reconstruct(ac_logits, span_starts), ac_logits

    # ------------------------------------------------------------------------------------------------
    _region_features = TVNetSAModel2._region_features

    def get_visual_prompt(self, images, aux_imgs):
        """reference: models/bert_model.py:379-414 -> list of num_layers (K, V) [B,NH,P,64]."""
        feats, aux = self._region_features(images, aux_imgs)
        bsz, L, Fd = feats.shape
        cfg = self.bert.config
        hidden = cfg.hidden_size
        NI = 1 + len(aux)
        x = torch.stack([feats] + aux).reshape(NI * bsz * L, Fd)
        e0, e2 = self.encoder_conv[0], self.encoder_conv[2]
        t = engine.LinearFunction.apply(x, e0.weight, e0.bias, True)
        enc = engine.LinearFunction.apply(t, e2.weight, e2.bias, False).view(NI, bsz, L, 8 * hidden)
        if self._packed_proj is None or not self._packed_proj.valid(self.projectors):
            self._packed_proj = _PackedLinears(self.projectors)
        pp = self._packed_proj
        proj_params = [p for m in self.projectors for p in (m.weight, m.bias)]
        pkv = engine.PromptFunction.apply(enc, pp.w, pp.b, len(self.projectors), *proj_params)
        return PrefixKV(pkv, cfg.num_attention_heads, hidden // cfg.num_attention_heads)
