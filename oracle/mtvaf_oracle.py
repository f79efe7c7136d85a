"""CPU oracle for the MTVAF hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product path (``mtvaf_amd``) never does and fails loudly when its HIP
extension is missing.

This is a plain-torch, CPU, fp32 *restatement* (written for this repo, functional style, no
nn.Module) of the reference's prefix-fused BERT/RoBERTa forward pass and the TVNetSAModel2 head.
Every function cites the reference lines it follows (paths relative to the reference checkout).
Gradients come from torch autograd on this restatement.

Pinning status
--------------
* Encoder / embeddings / prompt generator / VAO loss: PINNED -- checked op-by-op and end-to-end
  against the real reference modules imported in the authoring container
  (``tests/golden/gen_golden.py``) and against the committed golden vectors
  (``tests/golden/*.npz``), see ``tests/test_oracle_golden.py``.
* Linear-chain CRF (``crf_*``): **parity unpinned** against the third-party package
  ``pytorch-crf`` (import name ``torchcrf``; the reference pins no version and ships no copy).
  It restates the published algorithm and is pinned by brute-force enumeration known-answers
  (``tests/test_oracle_crf.py``) instead.
"""
from __future__ import annotations

import itertools
import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
MASK_VALUE = -10000.0  # models/modeling_bert.py:1134-1137 (additive mask, NOT -inf)


# --------------------------------------------------------------------------------------
# Embeddings
# --------------------------------------------------------------------------------------
def roberta_position_ids(input_ids: Tensor, padding_idx: int) -> Tensor:
    """models/modeling_roberta.py:1706-1719 (create_position_ids_from_input_ids,
    past_key_values_length = 0 as hard-coded at modeling_roberta.py:909-911)."""
    mask = input_ids.ne(padding_idx).int()
    incremental = torch.cumsum(mask, dim=1).type_as(mask) * mask
    return incremental.long() + padding_idx


def embeddings(sd: Dict[str, Tensor], prefix: str, input_ids: Tensor, token_type_ids: Tensor,
               eps: float, roberta: bool = False, pad_idx: int = 0,
               position_ids: Optional[Tensor] = None) -> Tensor:
    """BertEmbeddings.forward models/modeling_bert.py:188-222 /
    RobertaEmbeddings.forward models/modeling_roberta.py:102-140 (dropout omitted: eval / p=0)."""
    B, S = input_ids.shape
    if position_ids is None:
        if roberta:
            position_ids = roberta_position_ids(input_ids, pad_idx)
        else:
            position_ids = torch.arange(S).unsqueeze(0).expand(B, S)  # :199, past length 0 (:1050)
    # nn.Embedding(..., padding_idx=config.pad_token_id) (modeling_bert.py:170, modeling_roberta.py:78,
    # :97-100): forward is a plain gather, but the padding row receives NO gradient.
    word_pad = pad_idx
    e = F.embedding(input_ids, sd[prefix + "word_embeddings.weight"], padding_idx=word_pad)
    e = e + F.embedding(token_type_ids, sd[prefix + "token_type_embeddings.weight"])
    e = e + F.embedding(position_ids, sd[prefix + "position_embeddings.weight"],
                        padding_idx=pad_idx if roberta else None)
    return F.layer_norm(e, (e.shape[-1],), sd[prefix + "LayerNorm.weight"],
                        sd[prefix + "LayerNorm.bias"], eps)


# --------------------------------------------------------------------------------------
# Encoder layer
# --------------------------------------------------------------------------------------
def prefix_self_attention(h: Tensor, ext_mask: Tensor, sd: Dict[str, Tensor], p: str, num_heads: int,
                          past_kv: Optional[Tuple[Tensor, Tensor]], return_probs: bool = False):
    """BertSelfAttention.forward models/modeling_bert.py:255-342 with the prefix branch :282-286."""
    B, S, H = h.shape
    D = H // num_heads

    def heads(x):  # transpose_for_scores :250-253
        return x.view(B, S, num_heads, D).permute(0, 2, 1, 3)

    q = heads(F.linear(h, sd[p + "query.weight"], sd[p + "query.bias"]))
    k = heads(F.linear(h, sd[p + "key.weight"], sd[p + "key.bias"]))
    v = heads(F.linear(h, sd[p + "value.weight"], sd[p + "value.bias"]))
    if past_kv is not None:
        k = torch.cat([past_kv[0], k], dim=2)  # :285
        v = torch.cat([past_kv[1], v], dim=2)  # :286
    scores = torch.matmul(q, k.transpose(-1, -2))  # :303
    scores = scores / math.sqrt(D)  # :320
    scores = scores + ext_mask  # :323
    probs = torch.softmax(scores, dim=-1)  # :325
    ctx = torch.matmul(probs, v)  # :333
    ctx = ctx.permute(0, 2, 1, 3).contiguous().view(B, S, H)  # :335-337
    return (ctx, probs) if return_probs else ctx


def bert_layer(h: Tensor, ext_mask: Tensor, sd: Dict[str, Tensor], p: str, num_heads: int, eps: float,
               past_kv: Optional[Tuple[Tensor, Tensor]]) -> Tensor:
    """BertLayer.forward models/modeling_bert.py:453-522 (BertSelfOutput :352-356,
    BertIntermediate :419-422 with erf-GELU, BertOutput :432-436); dropout omitted."""
    H = h.shape[-1]
    ctx = prefix_self_attention(h, ext_mask, sd, p + "attention.self.", num_heads, past_kv)
    a = F.linear(ctx, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"])
    h1 = F.layer_norm(a + h, (H,), sd[p + "attention.output.LayerNorm.weight"],
                      sd[p + "attention.output.LayerNorm.bias"], eps)
    inter = F.gelu(F.linear(h1, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
    f = F.linear(inter, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
    return F.layer_norm(f + h1, (H,), sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)


def extended_attention_mask(attention_mask: Tensor, dtype=torch.float32) -> Tensor:
    """models/modeling_bert.py:1064, restated by the reference itself at :1134-1137."""
    return (1.0 - attention_mask[:, None, None, :].to(dtype)) * MASK_VALUE


def bert_model(sd: Dict[str, Tensor], input_ids: Tensor, attention_mask: Tensor, token_type_ids: Tensor,
               past_key_values: Optional[Sequence[Tuple[Tensor, Tensor]]], num_layers: int, num_heads: int,
               eps: float, prefix: str = "", roberta: bool = False, pad_idx: int = 0,
               inputs_embeds_out: Optional[Tensor] = None) -> List[Tensor]:
    """BertModel.forward models/modeling_bert.py:989-1115 (RobertaModel: models/modeling_roberta.py).
    ``attention_mask`` is the FULL mask [B, P+S] (prefix part all ones, bert_model.py:490-492).
    Returns the list of 1+num_layers hidden states (BertEncoder.forward :550-600); the pooler
    (:726-732) is exposed separately as ``bert_pooler``."""
    ext = extended_attention_mask(attention_mask)
    h = embeddings(sd, prefix + "embeddings.", input_ids, token_type_ids, eps, roberta, pad_idx) \
        if inputs_embeds_out is None else inputs_embeds_out
    hs = [h]
    for i in range(num_layers):
        pkv = past_key_values[i] if past_key_values is not None else None
        h = bert_layer(h, ext, sd, f"{prefix}encoder.layer.{i}.", num_heads, eps, pkv)
        hs.append(h)
    return hs


def bert_pooler(sd: Dict[str, Tensor], last_hidden: Tensor, prefix: str = "") -> Tensor:
    """BertPooler.forward models/modeling_bert.py:726-732."""
    return torch.tanh(F.linear(last_hidden[:, 0], sd[prefix + "pooler.dense.weight"], sd[prefix + "pooler.dense.bias"]))


# --------------------------------------------------------------------------------------
# Visual prompt generator + VAO loss (TVNetSAModel2.get_visual_prompt)
# --------------------------------------------------------------------------------------
def kl_batchmean_log_softmax(logits: Tensor, target: Tensor) -> Tensor:
    """models/bert_model.py:553-554: KLDivLoss(reduction='batchmean')(softmax(z).log(), target).
    Restated with log_softmax (identical where softmax does not underflow).  0*log(0) := 0 as
    torch.nn.functional.kl_div does."""
    logp = torch.log_softmax(logits, dim=-1)
    return F.kl_div(logp, target, reduction="batchmean")


def visual_prompt(sd: Dict[str, Tensor], feats: Tensor, aux_feats: Sequence[Tensor], num_layers: int = 12,
                  num_heads: int = 12, head_dim: int = 64, vao: bool = False,
                  imagelabel: Optional[Tensor] = None):
    """TVNetSAModel2.get_visual_prompt models/bert_model.py:534-588.

    ``feats``      : [B, prefix_len(4), F] -- the ``torch.cat(pyramid, dim=1).view(bsz, prefix_len, -1)``
                     of :538 (the frozen ResNet that produces the pyramid is upstream of the path).
    ``aux_feats``  : list (n_aux) of the same for the aux crops (:539).
    Returns (list of num_layers (K, V) each [B, num_heads, P, head_dim], img_tag_loss, [aux losses]).
    Dropout(0.2) of the VAO branch omitted (eval / p = 0).
    """
    B = feats.shape[0]
    hid = num_heads * head_dim  # 768
    w0, b0 = sd["encoder_conv.0.weight"], sd["encoder_conv.0.bias"]
    w2, b2 = sd["encoder_conv.2.weight"], sd["encoder_conv.2.bias"]

    def enc(x):  # :446-454, :541-542
        return F.linear(torch.tanh(F.linear(x, w0, b0)), w2, b2)

    pg = enc(feats)  # [B,4,4*2*hid]
    apg = [enc(a) for a in aux_feats]
    split = pg.split(2 * hid, dim=-1)  # :544
    asplit = [a.split(2 * hid, dim=-1) for a in apg]  # :545

    img_tag_loss = 0
    aux_losses = []
    if vao:  # :549-563
        img_tag_loss = kl_batchmean_log_softmax(
            F.linear(pg.mean(dim=1), sd["img_classifier.weight"], sd["img_classifier.bias"]), imagelabel)
        for k, a in enumerate(apg):
            aux_losses.append(kl_batchmean_log_softmax(
                F.linear(a.mean(dim=1), sd[f"aux_img_classifier.{k}.weight"], sd[f"aux_img_classifier.{k}.bias"]),
                imagelabel))

    def mix(sp, idx):  # :567-572 / :576-580
        s = torch.stack(sp).sum(0).view(B, -1) / 4
        gate = torch.softmax(F.leaky_relu(F.linear(s, sd[f"projectors.{idx}.weight"], sd[f"projectors.{idx}.bias"])),
                             dim=-1)
        kv = torch.zeros_like(sp[0])
        for i in range(4):
            kv = kv + gate[:, i].view(-1, 1, 1) * sp[i]  # einsum('bg,blh->blh') with g == 1
        return kv

    result = []
    for idx in range(num_layers):
        kvs = [mix(split, idx)] + [mix(a, idx) for a in asplit]
        kv = torch.cat(kvs, dim=1)  # :583 [B, 4*(1+n_aux), 2*hid]
        k, v = kv.split(hid, dim=-1)  # :584
        # :585 raw row-major reinterpretation, NOT a head transpose
        result.append((k.reshape(B, num_heads, -1, head_dim).contiguous(),
                       v.reshape(B, num_heads, -1, head_dim).contiguous()))
    return result, img_tag_loss, aux_losses


# --------------------------------------------------------------------------------------
# Linear-chain CRF  (third-party `pytorch-crf`, call sites models/bert_model.py:464, :511, :521)
# Published algorithm (Lafferty et al. 2001; pytorch-crf docs): batch_first=True,
#   score(y) = start[y0] + emit[0,y0] + sum_{t>=1, mask_t} (trans[y_{t-1},y_t] + emit[t,y_t]) + end[y_last]
#   with last = sum(mask)-1; logZ by the masked forward recursion; 'mean' = mean over the batch.
# --------------------------------------------------------------------------------------
def crf_sequence_score(emissions: Tensor, tags: Tensor, mask: Tensor, start: Tensor, end: Tensor,
                       trans: Tensor) -> Tensor:
    B, S, C = emissions.shape
    maskf = mask.to(emissions.dtype)
    ar = torch.arange(B)
    score = start[tags[:, 0]] + emissions[ar, 0, tags[:, 0]]
    for t in range(1, S):
        score = score + (trans[tags[:, t - 1], tags[:, t]] + emissions[ar, t, tags[:, t]]) * maskf[:, t]
    seq_ends = mask.long().sum(dim=1) - 1
    last_tags = tags[ar, seq_ends]
    return score + end[last_tags]


def crf_log_partition(emissions: Tensor, mask: Tensor, start: Tensor, end: Tensor, trans: Tensor) -> Tensor:
    B, S, C = emissions.shape
    score = start.unsqueeze(0) + emissions[:, 0]
    for t in range(1, S):
        nxt = torch.logsumexp(score.unsqueeze(2) + trans.unsqueeze(0) + emissions[:, t].unsqueeze(1), dim=1)
        score = torch.where(mask[:, t].bool().unsqueeze(1), nxt, score)
    return torch.logsumexp(score + end.unsqueeze(0), dim=1)


def crf_log_likelihood(emissions, tags, mask, start, end, trans, reduction: str = "mean") -> Tensor:
    """`CRF.forward(emissions, tags, mask, reduction)`; reference call models/bert_model.py:521 uses
    reduction='mean' and negates the result."""
    llh = crf_sequence_score(emissions, tags, mask, start, end, trans) - \
        crf_log_partition(emissions, mask, start, end, trans)
    if reduction == "none":
        return llh
    if reduction == "sum":
        return llh.sum()
    if reduction == "mean":
        return llh.mean()
    if reduction == "token_mean":
        return llh.sum() / mask.to(emissions.dtype).sum()
    raise ValueError(reduction)


def crf_decode(emissions: Tensor, mask: Tensor, start: Tensor, end: Tensor, trans: Tensor) -> List[List[int]]:
    """`CRF.decode` (Viterbi); reference call models/bert_model.py:511.  Ties resolve to the lowest
    index (argmax/max first occurrence)."""
    B, S, C = emissions.shape
    score = start.unsqueeze(0) + emissions[:, 0]
    history = []
    for t in range(1, S):
        nxt, idx = (score.unsqueeze(2) + trans.unsqueeze(0) + emissions[:, t].unsqueeze(1)).max(dim=1)
        score = torch.where(mask[:, t].bool().unsqueeze(1), nxt, score)
        history.append(idx)
    score = score + end.unsqueeze(0)
    seq_ends = mask.long().sum(dim=1) - 1
    out = []
    for b in range(B):
        best = int(score[b].argmax())
        tags = [best]
        for hist in reversed(history[: int(seq_ends[b])]):
            best = int(hist[b][tags[-1]])
            tags.append(best)
        tags.reverse()
        out.append(tags)
    return out


def crf_bruteforce(emissions: Tensor, mask: Tensor, start: Tensor, end: Tensor, trans: Tensor):
    """Exact log-partition and argmax path by enumerating all C^L paths (contiguous masks, tiny L).
    Known-answer generator for the CRF restatement and the HIP CRF kernels."""
    B, S, C = emissions.shape
    logZ, best = [], []
    for b in range(B):
        L = int(mask[b].long().sum())
        scores = []
        paths = list(itertools.product(range(C), repeat=L))
        for path in paths:
            s = float(start[path[0]]) + float(emissions[b, 0, path[0]])
            for t in range(1, L):
                s += float(trans[path[t - 1], path[t]]) + float(emissions[b, t, path[t]])
            s += float(end[path[-1]])
            scores.append(s)
        sc = torch.tensor(scores, dtype=torch.float64)
        logZ.append(float(torch.logsumexp(sc, 0)))
        best.append(list(paths[int(sc.argmax())]))
    return logZ, best


# --------------------------------------------------------------------------------------
# Whole model: TVNetSAModel2.forward
# --------------------------------------------------------------------------------------
def tvnet2_forward(sd: Dict[str, Tensor], input_ids: Tensor, attention_mask: Tensor, token_type_ids: Tensor,
                   labels: Optional[Tensor], past_key_values, num_layers: int, num_heads: int, eps: float,
                   roberta: bool = False, pad_idx: int = 0, img_tag_loss=0.0, alpha: float = 0.0):
    """TVNetSAModel2.forward models/bert_model.py:480-532, given the visual prompt
    (``past_key_values`` from ``visual_prompt`` or synthetic).  Dropout omitted (eval / p = 0).
    Returns (loss, emissions, decoded tag lists, hidden_states)."""
    B, S = input_ids.shape
    if past_key_values is not None:
        P = past_key_values[0][0].shape[2]
        full_mask = torch.cat([torch.ones(B, P, dtype=attention_mask.dtype), attention_mask], dim=1)  # :490-492
    else:
        full_mask = attention_mask
    hs = bert_model(sd, input_ids, full_mask, token_type_ids, past_key_values, num_layers, num_heads, eps,
                    prefix="bert.", roberta=roberta, pad_idx=pad_idx)
    emissions = F.linear(hs[-1], sd["fc.weight"], sd["fc.bias"])  # :510
    crf = (sd["crf.start_transitions"], sd["crf.end_transitions"], sd["crf.transitions"])
    tags = crf_decode(emissions.detach(), attention_mask.byte(), *crf)  # :511
    loss = None
    if labels is not None:
        loss = -1 * crf_log_likelihood(emissions, labels, attention_mask.byte(), *crf, reduction="mean")  # :521
        loss = loss + alpha * img_tag_loss  # :530
    return loss, emissions, tags, hs


# --------------------------------------------------------------------------------------
# Span model TVNetSAModel (models/bert_model.py:113-376): extraction + classification heads
# --------------------------------------------------------------------------------------
def span_representation(span_starts: Tensor, span_ends: Tensor, inp: Tensor, input_mask: Tensor):
    """get_span_representation models/bert_model.py:147-170.  Spans index the FLATTENED list of valid tokens
    (flatten_emb_by_sentence :140-145); JR = widest span of the batch; positions beyond a span's width are
    masked, indices beyond the text are clipped to the last token."""
    input_mask = input_mask.to(dtype=span_starts.dtype)
    input_len = torch.sum(input_mask, dim=-1)
    word_offset = torch.cumsum(input_len, dim=0) - input_len
    s_off = (span_starts + word_offset.unsqueeze(1)).view(-1)
    e_off = (span_ends + word_offset.unsqueeze(1)).view(-1)
    width = e_off - s_off + 1
    JR = int(torch.max(width))
    B, S, H = inp.shape
    ctx = inp.reshape(B * S, H)[input_mask.reshape(B * S).nonzero().squeeze(-1), :]
    text_length = ctx.shape[0]
    idx = torch.arange(JR).unsqueeze(0) + s_off.unsqueeze(1)
    idx = torch.min(idx, (text_length - 1) * torch.ones_like(idx))
    emb = ctx[idx, :]
    span_mask = torch.arange(JR) < width.unsqueeze(-1)
    return emb, span_mask


def self_att_representation(inp: Tensor, score: Tensor, mask: Tensor) -> Tensor:
    """get_self_att_representation models/bert_model.py:172-179."""
    score = score + (1.0 - mask.to(score.dtype)) * -10000.0
    prob = torch.softmax(score, dim=-1).unsqueeze(-1)
    return torch.sum(prob * inp, dim=1)


def distant_cross_entropy(logits: Tensor, positions: Tensor) -> Tensor:
    """distant_cross_entropy models/bert_model.py:181-190 (mask=None branch, the one the model uses :298-299)."""
    logp = torch.log_softmax(logits, dim=-1)
    pos = positions.to(logp.dtype)
    return -1 * torch.mean(torch.sum(pos * logp, dim=-1) / torch.sum(pos, dim=-1))


def tvnet1_heads(sd: Dict[str, Tensor], sequence_output: Tensor, attention_mask: Tensor, span_starts: Tensor,
                 span_ends: Tensor, start_positions: Tensor, end_positions: Tensor, polarity_labels: Tensor,
                 label_masks: Tensor):
    """TVNetSAModel.extraction (:351-354) + classification (:363-376) + the loss of forward (:288-305), given the
    (dropped-out) encoder output.  Returns (tot_loss, logits [B,M,4], start_logits, end_logits)."""
    ae = F.linear(sequence_output, sd["binary_affine.weight"], sd["binary_affine.bias"])
    start_logits, end_logits = ae[..., 0], ae[..., 1]
    emb, smask = span_representation(span_starts, span_ends, sequence_output, attention_mask)
    score = F.linear(emb, sd["unary_affine.weight"], sd["unary_affine.bias"]).squeeze(-1)
    pooled = self_att_representation(emb, score, smask)
    pooled = torch.tanh(F.linear(pooled, sd["dense.weight"], sd["dense.bias"]))
    ac_logits = F.linear(pooled, sd["classifier.weight"], sd["classifier.bias"])
    B, M = span_starts.shape
    ae_loss = (distant_cross_entropy(start_logits, start_positions) + distant_cross_entropy(end_logits, end_positions)) / 2
    ac_loss = F.cross_entropy(ac_logits, polarity_labels.reshape(-1))
    flat_masks = label_masks.reshape(-1).to(ac_logits.dtype)
    ac_loss = torch.sum(flat_masks * ac_loss) / flat_masks.sum()  # reference quirk (:302-303): a scalar times the mask
    return ae_loss + ac_loss, ac_logits.view(B, M, -1), start_logits, end_logits


# --------------------------------------------------------------------------------------
# Cutoff augmentation (modules/augument.py:99-159), with the random draws passed in
# --------------------------------------------------------------------------------------
def cutoff_span(embeds: Tensor, masks: Tensor, input_lens: Tensor, ratio: float, u: Tensor):
    """generate_span_cutoff_embedding :99-117; ``u[i]`` stands for the reference's ``torch.rand(1)`` of sample i."""
    out_e, out_m = [], []
    for i in range(embeds.shape[0]):
        cutoff_length = int(input_lens[i] * ratio)
        start = int(u[i] * (input_lens[i] - cutoff_length))
        out_e.append(torch.cat((embeds[i][:start], torch.zeros([cutoff_length, embeds.shape[-1]], dtype=torch.float),
                                embeds[i][start + cutoff_length:]), dim=0))
        out_m.append(torch.cat((masks[i][:start], torch.zeros([cutoff_length], dtype=torch.long),
                                masks[i][start + cutoff_length:]), dim=0))
    return torch.stack(out_e, dim=0), torch.stack(out_m, dim=0)


def cutoff_token(embeds: Tensor, masks: Tensor, zero_index: Sequence[Tensor]):
    """generate_token_cutoff_embedding :120-141; ``zero_index[i]`` stands for ``torch.randint(len_i, (cutoff_length,))``."""
    out_e, out_m = [], []
    for i in range(embeds.shape[0]):
        tmp_mask = torch.ones(embeds[i].shape[0])
        for ind in zero_index[i]:
            tmp_mask[ind] = 0
        out_e.append(torch.mul(tmp_mask[:, None], embeds[i]))
        out_m.append(torch.mul(tmp_mask, masks[i]).type(torch.int64))
    return torch.stack(out_e, dim=0), torch.stack(out_m, dim=0)


def cutoff_dim(embeds: Tensor, masks: Tensor, zero_index: Sequence[Tensor]):
    """generate_dim_cutoff_embedding :144-159."""
    out_e = []
    for i in range(embeds.shape[0]):
        tmp_mask = torch.ones(embeds[i].shape[1])
        for ind in zero_index[i]:
            tmp_mask[ind] = 0.0
        out_e.append(torch.mul(tmp_mask, embeds[i]))
    return torch.stack(out_e, dim=0), masks
