"""Autograd glue between torch and the HIP kernels (mtvaf_amd.hip).

Each Function owns the forward/backward orchestration of one stage of the reference path and calls
nothing but the C-ABI kernels for arithmetic; torch is used for buffer allocation and graph wiring.

  EmbeddingsFunction   BertEmbeddings / RobertaEmbeddings      models/modeling_bert.py:188-222
  EncoderFunction      BertEncoder (all layers, prefix K/V)     models/modeling_bert.py:532-620
  LinearFunction       nn.Linear (+tanh)                        bert_model.py:446-454, 465, 510
  DropoutFunction      nn.Dropout                               bert_model.py:506
  CRFNLLFunction       -torchcrf.CRF(...)(reduction='mean')     bert_model.py:521
  PromptFunction       get_visual_prompt gates/mix              bert_model.py:544-585
  KLFunction           KLDivLoss(batchmean)(log softmax)        bert_model.py:553-554
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional

import torch

from . import hip

# -------------------------------------------------------------------------------------------------
# dropout RNG state: (seed, running offset).  One offset per dropout site per forward.
# -------------------------------------------------------------------------------------------------


class _Rng:
    """seed: torch.initial_seed() (so torch.manual_seed() controls the dropout masks) xor a per-stream constant.
    Under data parallelism every rank must draw different masks for its (site, row) elements even when the launcher
    seeds all ranks alike: ``mtvaf_amd.parallel.GradSync`` calls ``set_stream(rank)``."""

    def __init__(self):
        self.offset = 0
        self.stream = 0

    def set_stream(self, stream: int):
        self.stream = int(stream)

    def seed(self) -> int:
        return (int(torch.initial_seed()) ^ (self.stream * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF

    def next(self, n: int = 1) -> int:
        o = self.offset
        self.offset += n
        return o


RNG = _Rng()


def _empty(*shape, like: torch.Tensor, dtype=torch.float32):
    return torch.empty(*shape, device=like.device, dtype=dtype)


def _grad_target(param: torch.Tensor, flat_view: Optional[torch.Tensor]):
    """Where a parameter gradient is written: a fresh tensor (autograd steals or accumulates it)."""
    return flat_view if flat_view is not None else torch.empty_like(param)


# -------------------------------------------------------------------------------------------------
class EmbeddingsFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, word, pos, typ, gamma, beta, input_ids, token_type_ids, eps, p_drop, roberta, pad_idx):
        B, S = input_ids.shape
        H = word.shape[1]
        # the kernels index the tables with 64-bit ids and S (+ the RoBERTa offset) positions: anything else must not reach them
        ids = input_ids.to(torch.long).contiguous()
        tts = token_type_ids.to(torch.long).contiguous()
        if S + (pad_idx + 1 if roberta else 0) > pos.shape[0]:
            raise ValueError(f"sequence length {S} exceeds the position table ({pos.shape[0]} rows"
                             f"{', RoBERTa offset ' + str(pad_idx + 1) if roberta else ''})")
        pos_ids = None
        if roberta:
            pos_ids = _empty(B, S, like=word, dtype=torch.int32)
            hip.roberta_position_ids(ids, pos_ids, pad_idx)
        out = _empty(B * S, H, like=word)
        mean, rstd = _empty(B * S, like=word), _empty(B * S, like=word)
        seed, off = RNG.seed(), RNG.next()
        hip.embed_ln_fwd(ids, tts, pos_ids, word, pos, typ, gamma, beta, out, mean, rstd, eps, p_drop, seed, off)
        ctx.stash = (word, pos, typ, gamma, ids, tts, pos_ids, mean, rstd, p_drop, seed, off, roberta, pad_idx)
        return out.view(B, S, H)

    @staticmethod
    def backward(ctx, dout):
        word, pos, typ, gamma, ids, tts, pos_ids, mean, rstd, p_drop, seed, off, roberta, pad_idx = ctx.stash
        B, S = ids.shape
        H = word.shape[1]
        dout = dout.contiguous().view(B * S, H)
        dword, dpos, dtyp = torch.empty_like(word), torch.empty_like(pos), torch.empty_like(typ)
        dg, db = torch.empty_like(gamma), torch.empty_like(gamma)
        dz = _empty(B * S, H, like=word)
        hip.embed_ln_bwd(dout, ids, tts, pos_ids, word, pos, typ, gamma, mean, rstd, dword, dpos, dtyp, dg, db, False,
                         pad_idx, pad_idx if roberta else -1, p_drop, seed, off, dz)
        return dword, dpos, dtyp, dg, db, None, None, None, None, None, None


# -------------------------------------------------------------------------------------------------
class LayerWeights:
    """Device pointers of one encoder layer in kernel-ready (QKV-packed) form."""
    __slots__ = ("wqkv", "bqkv", "wo", "bo", "g1", "b1", "w1", "bi1", "w2", "bi2", "g2", "b2", "wparams", "_h", "flat", "_st",
                 "_gst", "_pl")


def _bf16(*shape, like: torch.Tensor):
    return torch.empty(shape, device=like.device, dtype=torch.bfloat16)


def _cast(x):
    """fp32 [R,C] -> bf16 row-major copy (the operand of a following projection in the mixed-precision mode)."""
    out = _bf16(*x.shape, like=x)
    hip.cast_bf16(x, out=out)
    return out


def _weights_bf16(w: LayerWeights):
    """bf16 shadows of a layer's four weight matrices: views of ONE flat bf16 image of the layer's parameter buffer (same
    offsets as the fp32 masters).  Every product reads them row-major -- forward as the KC operand, dX as the KM operand
    (transposing LDS reads) -- so no transposed copies exist.  Freshness: ``mtvaf_amd.optim.AdamW`` writes the shadow in
    its update kernel and stamps it; otherwise the image is rebuilt (one cast kernel per layer) when the masters may have
    changed: after ANY torch optimizer step (global post-step hook -- the fused optimizers update parameters without
    moving their version counters) or when a weight Parameter's version counter has moved (load_state_dict, in-place
    updates through the Parameter).  Writes through ``.data`` are invisible to both: MTVAF_BF16_WCACHE=0 rebuilds in
    every forward pass (85 M parameters, ~0.1 ms per step)."""
    ver = shadow_version(w) if (BF16_WCACHE and not FORCE_SHADOW_REFRESH) else None
    c = w._h
    if ver is None or c is None or c[0] != ver:
        flat = w.flat
        H = w.wo.shape[0]
        img = c[1] if c is not None else torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device)
        hip.cast_bf16(flat.view(-1, H), out=img.view(-1, H))
        c = w._h = [ver, img, None]
    if c[2] is None:
        img, base = c[1], w.flat.data_ptr()
        view = lambda t: img[(t.data_ptr() - base) // 4:(t.data_ptr() - base) // 4 + t.numel()].view(t.shape)
        c[2] = (view(w.wqkv), view(w.wo), view(w.w1), view(w.w2))
    return c[2]


def shadow_version(w: LayerWeights, epoch_ahead: int = 0):
    return (_OPT_EPOCH[0] + epoch_ahead, _IMAGE_GEN[0]) + tuple(p._version for p in w.wparams)


_IMAGE_GEN = [0]


def invalidate_weight_images(model=None):
    """Declare every cached GEMM-operand image of the encoder weights stale (the bf16 shadows of the mixed-precision mode and the
    plane images of the fp32 pre-split path): the next forward pass rebuilds them from the fp32 masters.

    Freshness is otherwise tracked through the optimizer-step epoch and the weight Parameters' version counters, which see
    ``optimizer.step()``, ``load_state_dict`` and in-place updates THROUGH the Parameter -- but not a write through ``p.data``
    (EMA / SWA swaps, ``p.data.copy_``, ``p.data.mul_``) or straight into a layer's flat buffer.  Code that writes weights that way
    calls this afterwards; ``MTVAF_WEIGHT_IMAGES=rebuild`` rebuilds the images in every forward pass instead (a debugging aid:
    one cast / split pass over 85 M parameters per step).  ``model`` is accepted for symmetry with the other helpers; the
    generation counter is process-wide (one process per GPU)."""
    _IMAGE_GEN[0] += 1


def shadow_for_update(w: LayerWeights):
    """The flat bf16 image an optimizer kernel may write while it updates ``w.flat`` (None until the first forward pass
    in bf16 mode has built it); the caller stamps it with ``shadow_written`` afterwards."""
    return w._h[1] if (hip.COMPUTE == "bf16" and w._h is not None) else None


def shadow_written(w: LayerWeights):
    """Called from inside ``optimizer.step()`` (or the backward pass preceding it): the image matches the masters as they
    will be once this step's post-step hook has run."""
    if w._h is not None:
        w._h[0] = shadow_version(w, 1)


_OPT_EPOCH = [0]
try:  # every optimizer step invalidates the bf16 weight shadows
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_post_hook
    _reg_post_hook(lambda opt, args, kwargs: _OPT_EPOCH.__setitem__(0, _OPT_EPOCH[0] + 1))
    _HAVE_OPT_HOOK = True
except Exception:  # pragma: no cover - very old torch: never cache
    _HAVE_OPT_HOOK = False
WEIGHT_IMAGES_REBUILD = os.environ.get("MTVAF_WEIGHT_IMAGES", "cache") == "rebuild"  # every forward rebuilds (debug: .data writes)
BF16_WCACHE = os.environ.get("MTVAF_BF16_WCACHE", "1") != "0" and _HAVE_OPT_HOOK and not WEIGHT_IMAGES_REBUILD
# fp32 mode, PRE-SPLIT OPERANDS (round 5; csrc/gemm_f32p.hip): on packed rows every GEMM operand of an encoder layer is read as a
# tile-blocked plane image (the three bf16 planes of the split arithmetic, written once per tensor) by kernels that split nothing
# inside their k-loops -- the eight forward / dX products and the layer's four weight gradients as one grouped launch.  Weights:
# one image per layer, rewritten behind the optimizer's update of the layer (same stream) or rebuilt when the masters may have
# changed, exactly as the bf16 shadows above.  MTVAF_F32_PLANES=0: the wave-specialised in-kernel-split path for everything.
F32_PLANES = os.environ.get("MTVAF_F32_PLANES", "1") != "0" and _HAVE_OPT_HOOK


# (few-token layers -- BASELINE configs[0]: 256 rows -- are a handful of 128 x 128 tiles: the planner's small-tile kernels win there.
# Measured, bench.py --batch b, same box, path on / off: bs 4 x S 64 832 / 903 sentences/s, bs 4 x 128 770 / 812, bs 8 (640 packed
# rows) 1374 / 1238, bs 12 2015 / 1708, bs 16 2443 / 2342, bs 24 2726 / 2480)
F32_PLANES_MIN_ROWS = int(os.environ.get("MTVAF_F32_PLANES_MIN_ROWS", "512"))
P16_EP = os.environ.get("MTVAF_P16_EP", "1") != "0"  # (the same switch csrc/executor.hip reads)


def _f32_planes_on(use_h, pack, H, I) -> bool:
    return bool(F32_PLANES and not use_h and pack is not None and pack.Mp >= F32_PLANES_MIN_ROWS and H % 128 == 0 and I % 128 == 0
                and hip.f32_split())


def _wplane_fill(w: "LayerWeights", views):
    for t, v in zip((w.wqkv, w.wo, w.w1, w.w2), views):
        hip.split_planes_blocked(t, v)


def _weights_planes(w: "LayerWeights"):
    """-> the plane images of the layer's four weight matrices (uint8 views of one buffer), rebuilt when stale."""
    ver = shadow_version(w) if not (FORCE_SHADOW_REFRESH or WEIGHT_IMAGES_REBUILD) else None
    c = w._pl
    if ver is None or c is None or c[0] != ver:
        if c is None:
            mats = (w.wqkv, w.wo, w.w1, w.w2)
            img = torch.empty(6 * sum(t.numel() for t in mats), dtype=torch.uint8, device=w.flat.device)
            off, views = 0, []
            for t in mats:
                views.append(img[off:off + 6 * t.numel()])
                off += 6 * t.numel()
            c = w._pl = [None, img, tuple(views), True]
        _wplane_fill(w, c[2])
        c[0] = ver
    c[3] = True  # read by a forward pass since the last optimizer update (planes_for_update)
    return c[2]


def planes_for_update(w: "LayerWeights"):
    """-> [(first element in w.flat, rows, cols, image), ...] of the layer's four weight matrices when their plane images exist (a
    forward pass on the pre-split path has built them): ``hip.adamw_planes`` rewrites them inside the update; the caller then stamps
    them with ``planes_written``.  None: no images to maintain."""
    if w._pl is None or not F32_PLANES or hip.COMPUTE != "fp32":
        return None
    if not w._pl[3]:
        # no forward pass has read the images since the last update (the run went padded, below F32_PLANES_MIN_ROWS or switched the
        # path off): stop paying 6 bytes per weight per step for them; the next pre-split forward rebuilds them
        w._pl = None
        if w._st is not None:
            w._st[2] = False  # (never `is` a tuple of views: _layer_struct re-points the weight-image fields)
        return None
    w._pl[3] = False
    base = w.flat.data_ptr()
    return [((t.data_ptr() - base) // 4, t.shape[0], t.shape[1], v) for t, v in zip((w.wqkv, w.wo, w.w1, w.w2), w._pl[2])]


def planes_written(w: "LayerWeights"):
    if w._pl is not None:
        w._pl[0] = shadow_version(w, 1)


def planes_rewrite(w: "LayerWeights"):
    """Called from inside ``optimizer.step()`` (or the backward pass preceding it) on the stream of the layer's update: the plane
    images are rebuilt from the updated masters and match them as they will be once this step's post-step hook has run."""
    if w._pl is not None and F32_PLANES and hip.COMPUTE == "fp32":
        _wplane_fill(w, w._pl[2])
        w._pl[0] = shadow_version(w, 1)
FORCE_SHADOW_REFRESH = False  # set while a whole-step graph is captured (mtvaf_amd.graph): the cast becomes a graph node
BF16_OPERANDS = os.environ.get("MTVAF_BF16_OPERANDS", "1") != "0"  # 0: fp32-operand bf16 kernels only (gemm_bf16.hip)


def _bf16_ok(M, H, I):
    """bf16-operand kernels need whole 128-row / 128-column tiles and 64-deep k-tiles (M is also the reduction length of
    the weight-gradient products)."""
    return hip.COMPUTE == "bf16" and BF16_OPERANDS and M % 128 == 0 and H % 128 == 0 and I % 128 == 0


N_LAYER_PARAMS = 16  # q.w q.b k.w k.b v.w v.b ao.w ao.b ln1.w ln1.b i.w i.b o.w o.b ln2.w ln2.b


# Weight-gradient products (and the bias column sums) of the encoder backward depend only on tensors the dX chain has
# already produced, so they run on a side stream and fill the bubbles of the chain's non-GEMM kernels (attention
# backward, LayerNorm backward, split-K reductions) and the head / tail of its GEMMs.  MTVAF_DW_STREAM=0 serialises.
DW_SIDE_STREAM = os.environ.get("MTVAF_DW_STREAM", "1") != "0"
# Below this many token rows the second stream loses (bs 4, 256 rows: 4.87 ms per step with it against 4.33-4.59 without, same
# box): the products are short latency chains there and the cross-stream events cost more than the overlap returns.
DW_STREAM_MIN_ROWS = int(os.environ.get("MTVAF_DW_STREAM_MIN_ROWS", "1024"))
# LayerNorm backward's column sums (dgamma, dbeta, dense bias gradient) on the weight-gradient stream (MTVAF_LN_SUMS_SIDE=0: on
# the main chain, as before round 4)
LN_SUMS_ON_SIDE = os.environ.get("MTVAF_LN_SUMS_SIDE", "1") != "0"
_side_streams = {}


def _side_stream(device) -> "torch.cuda.Stream":
    key = torch.device(device).index or 0
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=device, priority=int(os.environ.get("MTVAF_DW_PRIORITY", "0")))
    return st


# -------------------------------------------------------------------------------------------------
# native per-layer executor (csrc/executor.hip): one C call per layer and direction instead of ~65 Python-level calls.
# MTVAF_NATIVE_EXEC=0 keeps the Python orchestration below (same kernels, same order) for A/B tests.
# -------------------------------------------------------------------------------------------------
NATIVE_EXEC = os.environ.get("MTVAF_NATIVE_EXEC", "1") != "0"
# Zero-copy gradient path of the native backward: the flat-buffer views are assigned to `.grad` directly and autograd gets
# None for the encoder parameters.  That is what `loss.backward()` (the reference trainer, modules/train.py:620) needs;
# `torch.autograd.grad(loss, encoder_params)` wants the gradients RETURNED instead: set MTVAF_DIRECT_GRADS=0 (or
# engine.DIRECT_GRADS = False) for such callers.
DIRECT_GRADS = os.environ.get("MTVAF_DIRECT_GRADS", "1") != "0"
# Padding-free execution (the DEFAULT since round 5 -- the round-4 review granted it behind a gate, see
# DESIGN -- for callers that set BertModel.allow_unpad, i.e. TVNetSAModel2; MTVAF_UNPAD=0 / engine.UNPAD = False restores the
# padded run; a direct BertModel(...) call, the span model and captured (HIP graph) steps always run padded; native executor):
# the encoder layers run on the PACKED unmasked token rows -- every kernel of a layer treats token rows independently
# except attention (which gets per-sentence row offsets) -- and the last hidden state is scattered back to [B,S,H] with
# zeros at the masked positions (only for callers that set BertModel.allow_unpad: TVNetSAModel2, whose CRF head is
# masked; the span model's position softmax reads every position, so it stays padded).  Loss, decoded tags and every parameter gradient are those of the padded run (a masked
# key contributes exp(-10000) = 0 there, a masked query feeds nothing); hidden states AT masked positions are zeros
# instead of the reference's don't-care values, and the intermediate hidden states are handed out lazily / detached.
UNPAD = os.environ.get("MTVAF_UNPAD", "1") != "0"


class padding_free:
    """``with engine.padding_free(False): ...`` -- the padded run (or the padding-free one) for a block, whatever the default."""

    def __init__(self, on: bool):
        self.on = bool(on)

    def __enter__(self):
        global UNPAD
        self.was, UNPAD = UNPAD, self.on
        return self

    def __exit__(self, *exc):
        global UNPAD
        UNPAD = self.was
        return False

# Padded run: the weight-gradient products skip the k-tiles (32 token rows in fp32 mode, 64 in bf16 mode) of the token axis that hold only masked tokens.
# The gradient of a token row nothing downstream reads is EXACTLY zero (a masked key has probability exp(-10000) = 0, a
# masked query feeds only itself, the CRF is masked), so the skipped terms of dW = sum_rows dY[r]^T X[r] are zeros: every
# output of the step is unchanged.  Only for callers that vouch for it (cfg[5], BertModel.allow_unpad).
SKIP_PAD_DW = os.environ.get("MTVAF_SKIP_PAD_DW", "1") != "0"
SKIP_PAD_DW_BF16 = os.environ.get("MTVAF_SKIP_PAD_DW", "1") == "2"
# Debug switch for the masked-rows contract (MTVAF_CHECK_CONTRACT=1 / engine.CHECK_CONTRACT = True): whenever a backward pass
# is about to rely on it -- k-tile lists, the attention backward's shortened query loops (zero_tail), padding-free execution --
# the incoming hidden-state gradients are read back and every row at a masked position must be exactly zero; a head that reads
# masked positions (or a model that sets BertModel.allow_unpad wrongly) raises here instead of training on silently different
# gradients.  One host synchronisation per backward pass: a debugging aid, off by default.
CHECK_CONTRACT = os.environ.get("MTVAF_CHECK_CONTRACT", "0") == "1"
# Padding-free execution: the varlen attention launches take the sentences longest first (placement only, bit-identical results: a
# launch lasts as long as its busiest CU, and in sorted order a CU's blocks come from the long, middle and short third of the
# batch).  MTVAF_ATTN_ORDER=0: sentence z in grid slot z, as before round 6.
ATTN_ORDER = os.environ.get("MTVAF_ATTN_ORDER", "1") != "0"
LAST_PACK = None  # the Packing of the most recent native forward (None: it ran padded)
_PENDING_PACK = None  # packing started by Packing.begin, consumed by the next Packing.build
_PACK_HOST = {}
_layouts = {}


def _pack_granule() -> int:
    """Rows the packed image is padded to: whole 128-row GEMM tiles; 256 in the mixed-precision mode, whose grouped
    weight-gradient launch (256 x 256 stream-K kernel) takes reductions over whole 256-row tiles only -- at 128 a packed batch of
    2432 / 4736 rows fell back to one split launch per weight gradient (round 5: C4 dW 312 us per layer instead of ~110)."""
    return 256 if (hip.COMPUTE == "bf16" and BF16_OPERANDS) else 128


def _pack_min_gain(rows: int, H: int = 128, I: int = 128) -> int:
    """Rows a packed image must save to be worth it.  With the pre-split operand path (fp32 mode, F32_PLANES, enough rows, H and I
    whole 128-column tiles: the predicate of ``_f32_planes_on``) the packed layout is also the FASTER kernel set, so even a batch
    without any padding runs on it (0: an identity packing); otherwise one 128-row tile."""
    return 0 if (F32_PLANES and rows >= F32_PLANES_MIN_ROWS and hip.COMPUTE == "fp32" and H % 128 == 0 and I % 128 == 0
                 and hip.f32_split()) else 128


class Packing:
    """Token packing of one forward pass: rowmap [Mp] packed row -> flat token (b*S + s), -1 for the rows that pad the
    image to whole 128-row tiles; inv [B*S] flat token -> packed row or -1; cu [B+1] row offsets of the sentences."""
    __slots__ = ("rowmap", "inv", "cu", "Mv", "Mp", "B", "S")

    @staticmethod
    def begin(addmask: torch.Tensor, Pn: int, S: int):
        """Start the packing of a batch as early as its mask exists (BertModel.forward calls this before the embeddings):
        one kernel builds the maps, the kept-row count travels to pinned host memory asynchronously, and `build` waits for
        it only after the host has enqueued the prompt generator and the embeddings -- the one host sync padding-free
        execution needs (the row count sizes every launch) then costs no GPU idle time."""
        global _PENDING_PACK
        _PENDING_PACK = None
        if not (UNPAD and addmask.is_cuda and addmask.dtype == torch.float32 and addmask.is_contiguous()):
            return
        if torch.cuda.is_current_stream_capturing():
            return  # (the row count cannot reach the host inside a capture: this step runs the padded layout)
        B, T = addmask.shape
        dev = addmask.device
        # (cu: B + 1 row offsets and, behind them, the sentence order of the attention launches -- longest first: ATTN_ORDER)
        cu = torch.empty(2 * B + 1, dtype=torch.int32, device=dev)
        inv = torch.empty(B * S, dtype=torch.int32, device=dev)
        rowmap = torch.empty(B * S, dtype=torch.int32, device=dev)
        mv = torch.empty(1, dtype=torch.int32, device=dev)
        build = hip.lib().mtvaf_build_packing_ordered if ATTN_ORDER else hip.lib().mtvaf_build_packing
        hip._ck(build(hip._p(addmask), B, T, Pn, S, hip._p(cu), hip._p(inv), hip._p(rowmap), hip._p(mv), hip._st()), "mtvaf_build_packing")
        key = dev.index or 0
        host = _PACK_HOST.get(key)
        if host is None:
            host = _PACK_HOST[key] = torch.zeros(1, dtype=torch.int32, pin_memory=True)
        host.copy_(mv, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        _PENDING_PACK = (addmask.data_ptr(), B, S, Pn, cu, inv, rowmap, host, ev)

    @staticmethod
    def build(addmask: torch.Tensor, Pn: int, B: int, S: int, H: int = 128, I: int = 128) -> Optional["Packing"]:
        global _PENDING_PACK
        pend, _PENDING_PACK = _PENDING_PACK, None
        if pend is not None and pend[:4] == (addmask.data_ptr(), B, S, Pn):
            _, _, _, _, cu, inv, rowmap, host, ev = pend
            ev.synchronize()
            Mv = int(host[0])
            g = _pack_granule()
            Mp = max(g, (Mv + g - 1) // g * g)
            if Mv == 0 or Mp > B * S - _pack_min_gain(Mp, H, I):
                return None
            pk = Packing()
            pk.rowmap, pk.inv, pk.cu = rowmap[:Mp], inv, cu
            pk.Mv, pk.Mp, pk.B, pk.S = Mv, Mp, B, S
            return pk
        valid = addmask[:, Pn:] > -5000.0
        idx = torch.nonzero(valid.reshape(-1)).squeeze(1)  # (host sync: the packed row count sizes every launch)
        Mv = int(idx.numel())
        g = _pack_granule()
        Mp = max(g, (Mv + g - 1) // g * g)
        if Mv == 0 or Mp > B * S - _pack_min_gain(Mp, H, I):
            return None  # nothing to gain (or nothing to compute): stay padded
        pk = Packing()
        dev = addmask.device
        pk.rowmap = torch.full((Mp,), -1, dtype=torch.int32, device=dev)
        pk.rowmap[:Mv] = idx.to(torch.int32)
        pk.inv = torch.full((B * S,), -1, dtype=torch.int32, device=dev)
        pk.inv[idx] = torch.arange(Mv, dtype=torch.int32, device=dev)
        pk.cu = torch.zeros(B + 1, dtype=torch.int32, device=dev)
        pk.cu[1:] = valid.sum(1).cumsum(0).to(torch.int32)
        pk.Mv, pk.Mp, pk.B, pk.S = Mv, Mp, B, S
        return pk

    def pack(self, x2d: torch.Tensor) -> torch.Tensor:
        out = _empty(self.Mp, x2d.shape[1], like=x2d)
        hip._ck(hip.lib().mtvaf_gather_rows(hip._p(x2d), hip._p(self.rowmap), hip._p(out), self.Mp, x2d.shape[1], hip._st()),
                "mtvaf_gather_rows")
        return out

    def unpack(self, xp: torch.Tensor) -> torch.Tensor:
        out = _empty(self.B * self.S, xp.shape[1], like=xp)
        hip._ck(hip.lib().mtvaf_gather_rows(hip._p(xp), hip._p(self.inv), hip._p(out), self.B * self.S, xp.shape[1], hip._st()),
                "mtvaf_gather_rows")
        return out


def _layout(fields):
    off, out = 0, []
    for name, nbytes in fields:
        out.append((name, off if nbytes else -1))
        off += (nbytes + 255) & ~255
    return out, max(off, 256)


def _fwd_layout(M, H, I, B, NH, S, use_h, planes=False):
    """Offsets of one layer's activation buffers inside its arena (one allocation per layer, saved for backward).  planes: the
    plane images of the GEMM operands the layer writes (pre-split operands, fp32 mode) -- 6 bytes per element."""
    key = ("f", M, H, I, B, NH, S, use_h, planes)
    lay = _layouts.get(key)
    if lay is None:
        e = 2 if use_h else 4
        pl = 6 if planes else 0
        # (pre-split path: the GELU output leaves the FFN-1 epilogue as a plane image only -- csrc/executor.hip, p16_ep_on -- so no
        # fp32 `act` exists; MTVAF_P16_EP=0 puts the fp32 tensor + split pass back)
        act_b = 0 if (planes and P16_EP) else M * I * e
        lay = _layouts[key] = _layout([("qkv", M * 3 * H * e), ("cx", M * H * e), ("lse", B * NH * S * 4), ("a", M * H * 4),
                                       ("h1", M * H * 4), ("h1_h", M * H * 2 if use_h else 0), ("mean1", M * 4), ("rstd1", M * 4),
                                       ("pre", M * I * e), ("act", act_b), ("f", M * H * 4), ("h2", M * H * 4),
                                       ("h2_h", M * H * 2 if use_h else 0), ("mean2", M * 4), ("rstd2", M * 4),
                                       ("cx_p", M * H * pl), ("h1_p", M * H * pl), ("act_p", M * I * pl), ("h2_p", M * H * pl)])
    return lay


def _bwd_layout(M, H, I, B, NH, S, Pn, use_h, planes=False):
    key = ("b", M, H, I, B, NH, S, Pn, use_h, planes)
    lay = _layouts.get(key)
    if lay is None:
        e = 2 if use_h else 4
        nqt, nkt = (S + 63) // 64, (Pn + S + 63) // 64
        lay = _layouts[key] = _layout([("dh1", M * H * 4), ("df", M * H * e), ("dpre", M * I * e), ("da", M * H * e),
                                       ("dctx", M * H * e), ("dqkv", M * 3 * H * e),
                                       ("part", (M // 128) * I * 4 if (use_h or planes) else 0), ("partq", B * nqt * H * 4 if use_h else 0),
                                       ("partkv", B * nkt * 2 * H * 4 if use_h else 0), ("delta", 0 if use_h else B * NH * S * 4),
                                       # (the layer's own LayerNorm-backward partials: their column sums run on the second stream)
                                       ("lnpart2", int(hip.lib().mtvaf_ln_bwd_workspace_bytes(M, H)) if LN_SUMS_ON_SIDE else 0),
                                       ("lnpart1", int(hip.lib().mtvaf_ln_bwd_workspace_bytes(M, H)) if LN_SUMS_ON_SIDE else 0),
                                       ("df_p", M * H * 6 if planes else 0), ("dpre_p", M * I * 6 if planes else 0),
                                       ("da_p", M * H * 6 if planes else 0), ("dqkv_p", M * 3 * H * 6 if planes else 0)])
    return lay


def _streamk_on(stream, device) -> bool:
    """Attach the stream-K scratch of the 256x256 bf16 kernel to `stream` (None: the current one) if it is not yet, and say
    whether that stream has one (hip.STREAMK / MTVAF_STREAMK=0 switch it off)."""
    if stream is None:
        return hip.streamk_ensure(device)
    with torch.cuda.stream(stream):
        return hip.streamk_ensure(device)


def _exec_workspace_bytes(M, H, I, use_h):
    """Scratch one stream of the executor may need: split-K slabs of the largest product, LayerNorm / column-sum partials."""
    big = max(M * 3 * H, M * I, I * H, 3 * H * H)
    return max((8 if use_h else 16) * big * 4, 512 * 4 * H * 4, 64 * max(3 * H, I) * 4, 64 << 20)


def _layer_struct(w: LayerWeights, B, S, Pn, NH, H, I, use_h, eps, p_hidden, p_attn, seed, planes=False):
    c = w._st
    key = (B, S, Pn, NH, H, I, use_h, w.flat.data_ptr())
    if c is None or c[0] != key:  # shapes and parameter pointers: refilled only when they change
        st = hip.LayerStruct()
        st.B, st.S, st.P, st.NH, st.H, st.I, st.bf16 = B, S, Pn, NH, H, I, int(use_h)
        st.wqkv, st.wo, st.w1, st.w2 = w.wqkv.data_ptr(), w.wo.data_ptr(), w.w1.data_ptr(), w.w2.data_ptr()
        st.bqkv, st.bo, st.g1, st.b1 = w.bqkv.data_ptr(), w.bo.data_ptr(), w.g1.data_ptr(), w.b1.data_ptr()
        st.bi1, st.bi2, st.g2, st.b2 = w.bi1.data_ptr(), w.bi2.data_ptr(), w.g2.data_ptr(), w.b2.data_ptr()
        c = w._st = [key, st, None]
    st = c[1]
    st.eps, st.p_hidden, st.p_attn, st.seed = eps, p_hidden, p_attn, seed
    if use_h:
        wh = _weights_bf16(w)  # (freshness check; the image keeps its address)
        if c[2] is not wh:
            st.wqkv_h, st.wo_h, st.w1_h, st.w2_h = (t.data_ptr() for t in wh)
            c[2] = wh
    elif planes:
        wp = _weights_planes(w)  # (fp32 mode: the same four fields carry the weights' plane images)
        if c[2] is not wp:
            st.wqkv_h, st.wo_h, st.w1_h, st.w2_h = (t.data_ptr() for t in wp)
            c[2] = wp
    elif c[2] is not None:
        st.wqkv_h = st.wo_h = st.w1_h = st.w2_h = None
        c[2] = None
    return st


def _native_forward(ctx, h0, pkv, addmask, cfg, weights, grad_sink, params):
    B, S, H = h0.shape
    NH, eps, p_hidden, p_attn, pkv_ready = cfg[:5]
    M, L = B * S, len(weights)
    Pn = 0 if pkv is None else pkv.shape[3] // H
    I = weights[0].w1.shape[0]
    use_h = _bf16_ok(M, H, I)
    x = h0.contiguous().view(M, H)
    seed = RNG.seed()
    dev = x.device
    # (the caller vouches that nothing downstream reads hidden states at masked positions: cfg[5], BertModel.allow_unpad)
    pack = Packing.build(addmask, Pn, B, S, H, I) if (UNPAD and len(cfg) > 5 and cfg[5]) else None
    if pack is not None:
        x = pack.pack(x)
        M = pack.Mp
    x0_h = _cast(x) if use_h else None
    pkv_k = pkv
    if Pn:
        if pkv_ready is not None:  # prefix produced on the second stream (prompt generator)
            cur = torch.cuda.current_stream()
            cur.wait_event(pkv_ready)
            pkv.record_stream(cur)
        if use_h:
            pkv_k = _bf16(*pkv.shape, like=x)  # one cast of all layers' prefix slabs per step
            hip.cast_bf16(pkv.view(-1, pkv.shape[3]), out=pkv_k.view(-1, pkv.shape[3]))
    planes = _f32_planes_on(use_h, pack, H, I)
    (lay, total) = _fwd_layout(M, H, I, B, NH, S, use_h, planes)
    o_h2 = dict(lay)["h2"]
    o_h2h = dict(lay)["h2_h"]
    o_h2p = dict(lay)["h2_p"]
    x0_p = None
    if planes:  # the first layer's input as a plane image (every later layer's is written by the layer before it)
        x0_p = torch.empty(M * H * 6, dtype=torch.uint8, device=dev)
        hip.split_planes_blocked(x, x0_p)
    xp_ptr = x0_p.data_ptr() if planes else None
    ws = None if use_h else hip.workspace(_exec_workspace_bytes(M, H, I, use_h), dev)
    fn, stream = hip.lib().mtvaf_encoder_layer_fwd, hip._st()
    pk_ptr = pkv_k.data_ptr() if Pn else 0
    pk_step = (pkv_k.stride(1) * pkv_k.element_size()) if Pn else 0   # [L,2,B,P*H]: K then V slab of a layer
    pk_layer = (pkv_k.stride(0) * pkv_k.element_size()) if Pn else 0
    am_ptr = addmask.data_ptr()
    x_ptr, xh_ptr = x.data_ptr(), (x0_h.data_ptr() if use_h else 0)
    arenas, offs, outs = [], [], []
    for li, w in enumerate(weights):
        off = RNG.next(3)
        arena = torch.empty(total, dtype=torch.uint8, device=dev)
        base = arena.data_ptr()
        st = _layer_struct(w, B, S, Pn, NH, H, I, use_h, eps, p_hidden, p_attn, seed, planes)
        st.offset = off
        st.cu, st.Mv, st.Mp = (pack.cu.data_ptr(), pack.Mv, pack.Mp) if pack is not None else (None, 0, 0)
        st.x, st.x_h, st.addmask = x_ptr, xh_ptr, am_ptr
        st.pk = (pk_ptr + li * pk_layer) if Pn else None
        st.pv = (pk_ptr + li * pk_layer + pk_step) if Pn else None
        for name, o in lay:
            setattr(st, name, base + o if o >= 0 else None)
        st.x_p = xp_ptr
        if planes:
            xp_ptr = base + o_h2p
            if li == L - 1:
                st.h2_p = None  # (nobody multiplies the last layer's output)
        st.ws, st.ws_bytes = (ws.data_ptr(), ws.numel()) if ws is not None else (None, 0)
        hip._ck(fn(ctypes.byref(st), stream), "mtvaf_encoder_layer_fwd")
        h2 = arena[o_h2:o_h2 + M * H * 4].view(torch.float32)
        if pack is None:
            outs.append(h2.view(B, S, H))
        elif li < L - 1:
            outs.append(h2.view(M, H))  # packed rows (detached below): modeling_bert wraps them into lazy [B,S,H] views
        else:
            outs.append(pack.unpack(h2.view(M, H)).view(B, S, H))
        arenas.append(arena)
        offs.append(off)
        x_ptr, xh_ptr = base + o_h2, (base + o_h2h if use_h else 0)
    saved = [x] + ([x0_h] if use_h else []) + ([x0_p] if planes else []) + arenas
    ctx.save_for_backward(*saved)
    ctx.planes = planes
    ctx.stash = (offs, weights, pkv, addmask, cfg, seed, (B, S, H, Pn), grad_sink, params)
    ctx.native = (use_h, pkv_k if (use_h and Pn) else None)
    ctx.pack = pack
    global LAST_PACK
    LAST_PACK = pack
    if pack is not None and L > 1:
        ctx.mark_non_differentiable(*outs[:-1])
    if grad_sink is not None:
        grad_sink.node_created()
    ctx.set_materialize_grads(False)
    return tuple(outs)


def _check_masked_rows_contract(douts, addmask, Pn, B, S):
    """MTVAF_CHECK_CONTRACT: every hidden-state gradient handed to the encoder backward must be exactly zero at masked
    positions (reference behaviour being relied on: the head reads through the mask only, bert_model.py:511, 521)."""
    masked = (addmask[:, Pn:] < -5000.0).reshape(B * S)
    for li, g in enumerate(douts):
        if g is None:
            continue
        rows = g.reshape(B * S, -1)[masked]
        if rows.numel() and bool((rows != 0).any()):
            bad = int((rows != 0).any(1).sum())
            raise RuntimeError(f"masked-rows contract violated: the gradient of hidden state {li + 1} is non-zero at {bad} masked token "
                               "position(s), but the caller set BertModel.allow_unpad (k-tile lists / zero_tail / padding-free "
                               "execution assume exact zeros there); unset allow_unpad for heads that read masked positions")


def _ddp_without_gradsync(grad_sink) -> bool:
    """A process group with several ranks but no GradSync on this encoder: something else (torch DDP's C++ reducer, which
    Python cannot see) is probably waiting for the gradients at the AccumulateGrad nodes -- the zero-copy `.grad` assignment
    would bypass it."""
    try:
        import torch.distributed as dist
        return bool(dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
                    and (grad_sink is None or grad_sink.on_layer_done is None))
    except Exception:  # pragma: no cover
        return False


_warned_ddp = [False]


def _has_grad_hooks(params) -> bool:
    for p in params:
        if p._backward_hooks or getattr(p, "_post_accumulate_grad_hooks", None):
            return True
    return False


def _native_backward(ctx, douts):
    offs, weights, pkv, addmask, cfg, seed, (B, S, H, Pn), grad_sink, params = ctx.stash
    use_h, pkv16 = ctx.native
    saved = ctx.saved_tensors
    x0 = saved[0]
    x0_h = saved[1] if use_h else None
    planes = bool(getattr(ctx, "planes", False))
    x0_p = saved[1] if planes else None
    arenas = saved[2:] if (use_h or planes) else saved[1:]
    NH, eps, p_hidden, p_attn = cfg[:4]
    pack = ctx.pack
    L, M = len(weights), (pack.Mp if pack is not None else B * S)
    I = weights[0].w1.shape[0]
    dev = x0.device
    dpkv = torch.empty_like(pkv) if Pn else None
    need_param_grads = any(p.requires_grad for p in params)
    gviews = grad_sink.acquire(params) if grad_sink is not None else None
    if grad_sink is not None:
        grad_sink.token_rows = M
    # zero-copy gradients bypass AccumulateGrad: only when nobody listens there (tensor hooks, post-accumulate hooks).  NOT
    # detectable from Python: torch DDP's reducer, which hooks the AccumulateGrad nodes in C++ -- data parallelism here is
    # mtvaf_amd.parallel.GradSync; a DDP-wrapped model must set MTVAF_DIRECT_GRADS=0 / engine.DIRECT_GRADS = False
    direct = gviews is not None and DIRECT_GRADS and not _has_grad_hooks(params)
    if direct and _ddp_without_gradsync(grad_sink):
        #  several ranks and no GradSync: hand the gradients to autograd so that whatever listens there sees them
        direct = False
        if not _warned_ddp[0]:
            _warned_ddp[0] = True
            import warnings
            warnings.warn("mtvaf_amd: torch.distributed is initialised with several ranks but no mtvaf_amd.parallel.GradSync is attached "
                          "to this encoder; encoder gradients are returned through autograd (as with MTVAF_DIRECT_GRADS=0) so that a "
                          "DistributedDataParallel reducer sees them.  Use GradSync for the overlapped per-layer exchange.")
    pgrads: List[Optional[torch.Tensor]] = [None] * len(params)
    main = torch.cuda.current_stream()
    side = _side_stream(dev) if (DW_SIDE_STREAM and need_param_grads and M >= DW_STREAM_MIN_ROWS) else None
    main_h = hip._st()
    if side is not None:
        with torch.cuda.stream(side):
            side_h = hip._st()
            ws_side = hip.workspace(_exec_workspace_bytes(M, H, I, use_h), dev)
    else:
        side_h, ws_side = main_h, None
    ws_main = hip.workspace(_exec_workspace_bytes(M, H, I, use_h), dev)
    if ws_side is None:
        ws_side = ws_main
    if use_h and need_param_grads:
        _streamk_on(side, dev)  # the executor groups the layer's weight-gradient products when its dW stream has a scratch
    (flay, _), (blay, btotal) = _fwd_layout(M, H, I, B, NH, S, use_h, planes), _bwd_layout(M, H, I, B, NH, S, Pn, use_h, planes)
    o_h2, o_h2h, o_h2p = dict(flay)["h2"], dict(flay)["h2_h"], dict(flay)["h2_p"]
    fn = hip.lib().mtvaf_encoder_layer_bwd
    pk_src = pkv16 if use_h else pkv
    pk_ptr = pk_src.data_ptr() if Pn else 0
    pk_step = (pk_src.stride(1) * pk_src.element_size()) if Pn else 0
    pk_layer = (pk_src.stride(0) * pk_src.element_size()) if Pn else 0
    dpk_ptr = dpkv.data_ptr() if Pn else 0
    dpk_step, dpk_layer = ((dpkv.stride(1) * 4, dpkv.stride(0) * 4) if Pn else (0, 0))
    am_ptr = addmask.data_ptr()
    settle = int(grad_sink is not None and side is not None and grad_sink.on_layer_done is not None and grad_sink.settle_params)
    klist = kcnt = None
    zero_tail = bool(SKIP_PAD_DW and pack is None and len(cfg) > 5 and cfg[5] and addmask.dtype == torch.float32 and addmask.is_contiguous())
    if CHECK_CONTRACT and (zero_tail or pack is not None) and addmask.dtype == torch.float32:
        _check_masked_rows_contract(douts, addmask, Pn, B, S)
    bk = 64 if use_h else 32  # k-tile of the dW kernels
    # (mixed-precision mode: measured SLOWER with the list -- 7.85 vs 7.56 ms at C3, 11.71 vs 11.60 at C4: 64-row tiles skip
    # only ~19 % of a 33-us product and the device-side count delays its first loads -- so the list is an fp32-mode lever;
    # MTVAF_SKIP_PAD_DW=2 forces it for the bf16 kernels too)
    if (SKIP_PAD_DW and (not use_h or SKIP_PAD_DW_BF16) and pack is None and need_param_grads and len(cfg) > 5 and cfg[5] and (B * S) % bk == 0
            and addmask.dtype == torch.float32 and addmask.is_contiguous()):
        klist = torch.empty(B * S // bk, dtype=torch.int32, device=dev)
        kcnt = torch.empty(1, dtype=torch.int32, device=dev)
        hip._ck(hip.lib().mtvaf_build_ktiles(hip._p(addmask), B, addmask.shape[1], Pn, S, bk, hip._p(klist), hip._p(kcnt), main_h),
                "mtvaf_build_ktiles")
        if side is not None:
            klist.record_stream(side)
            kcnt.record_stream(side)
    keep = []  # temporaries the second stream still reads: alive until the join below
    dh = None
    for li in range(L - 1, -1, -1):
        w = weights[li]
        g_out = douts[li]
        if dh is None:
            if g_out is None:
                if Pn:
                    dpkv[li].zero_()  # layers above the last used hidden state get no gradient
                continue
            if pack is not None:
                dh = pack.pack(g_out.contiguous().view(B * S, H))  # (a fresh buffer; zero rows pad the image)
            else:
                dh = g_out.contiguous().view(M, H)
                if dh.data_ptr() == g_out.data_ptr():
                    dh = dh.clone()  # we modify / free it
        elif g_out is not None:
            dh = dh + g_out.reshape(M, H)
        base_i = li * N_LAYER_PARAMS
        if gviews is not None:
            G = gviews[base_i:base_i + N_LAYER_PARAMS]
            dwqkv, dbqkv = grad_sink.packed_qkv(li)
        else:
            G = [torch.empty_like(p) for p in params[base_i:base_i + N_LAYER_PARAMS]]
            dwqkv, dbqkv = _empty(3 * H, H, like=x0), _empty(3 * H, like=x0)
        st = _layer_struct(w, B, S, Pn, NH, H, I, use_h, eps, p_hidden, p_attn, seed, planes)
        st.offset = offs[li]
        st.cu, st.Mv, st.Mp = (pack.cu.data_ptr(), pack.Mv, pack.Mp) if pack is not None else (None, 0, 0)
        base = arenas[li].data_ptr()
        for name, o in flay:
            setattr(st, name, base + o if o >= 0 else None)
        if li == 0:
            st.x, st.x_h = x0.data_ptr(), (x0_h.data_ptr() if use_h else None)
            st.x_p = x0_p.data_ptr() if planes else None
        else:
            pb = arenas[li - 1].data_ptr()
            st.x, st.x_h = pb + o_h2, (pb + o_h2h if use_h else None)
            st.x_p = (pb + o_h2p) if planes else None
        st.addmask = am_ptr
        st.pk = (pk_ptr + li * pk_layer) if Pn else None
        st.pv = (pk_ptr + li * pk_layer + pk_step) if Pn else None
        gs = getattr(w, "_gst", None)
        if gs is None:
            gs = w._gst = hip.LayerGradsStruct()
        barena = torch.empty(btotal, dtype=torch.uint8, device=dev)
        keep.append(barena)
        bb = barena.data_ptr()
        for name, o in blay:
            setattr(gs, name, bb + o if o >= 0 else None)
        gs.dh = dh.data_ptr()
        gs.dwqkv, gs.dbqkv = dwqkv.data_ptr(), dbqkv.data_ptr()
        (gs.dwo, gs.dbo, gs.dg1, gs.db1, gs.dw1, gs.dbi1, gs.dw2, gs.dbi2, gs.dg2, gs.db2) = [t.data_ptr() for t in G[6:16]]
        gs.dpk = (dpk_ptr + li * dpk_layer) if Pn else None
        gs.dpv = (dpk_ptr + li * dpk_layer + dpk_step) if Pn else None
        gs.ws_main, gs.ws_main_bytes = ws_main.data_ptr(), ws_main.numel()
        gs.ws_side, gs.ws_side_bytes = ws_side.data_ptr(), ws_side.numel()
        gs.klist, gs.kcnt = (klist.data_ptr(), kcnt.data_ptr()) if klist is not None else (None, None)
        # (the caller vouches that masked token rows carry exactly-zero gradients -- cfg[5], BertModel.allow_unpad: the contract of
        # the k-tile lists, which the bf16 kernels do not take by default)
        gs.zero_tail = int(zero_tail)
        hip._ck(fn(ctypes.byref(st), ctypes.byref(gs), main_h, side_h, settle), "mtvaf_encoder_layer_bwd")
        if gviews is None:
            G[0], G[2], G[4] = dwqkv[:H], dwqkv[H:2 * H], dwqkv[2 * H:]
            G[1], G[3], G[5] = dbqkv[:H], dbqkv[H:2 * H], dbqkv[2 * H:]
            keep.append((dwqkv, dbqkv))
        if direct:
            # zero-copy fast path: the flat-buffer views become .grad directly (what AccumulateGrad would do with a
            # stolen gradient), and autograd is handed None for these 16 inputs -- 192 accumulation nodes less per step
            for p, gt in zip(params[base_i:base_i + N_LAYER_PARAMS], G):
                p.grad = gt
        else:
            pgrads[base_i:base_i + N_LAYER_PARAMS] = G
        if grad_sink is not None:
            if side is not None and grad_sink.on_layer_done is not None:
                if grad_sink.raw_stream_hook:  # the hook only enqueues library kernels: hand it the stream, skip the context
                    hip.STREAM_OVERRIDE = side_h
                    try:
                        grad_sink.layer_done(li)
                    finally:
                        hip.STREAM_OVERRIDE = None
                else:
                    with torch.cuda.stream(side):
                        grad_sink.layer_done(li)
            else:
                grad_sink.layer_done(li)
    if side is not None:
        main.wait_stream(side)  # join: gradients (and every buffer the second stream read) are settled from here on
    del keep
    if dh is not None and pack is not None:
        dh = pack.unpack(dh)
    dh0_out = dh.view(B, S, H) if dh is not None else None
    if not need_param_grads:
        pgrads = [None] * len(params)
    if use_h and need_param_grads:
        hip.streamk_poll()  # (a stream-K wait that ran out in the previous step raises here instead of training on)
    return (dh0_out, dpkv, None, None, None, None, *pgrads)


class EncoderFunction(torch.autograd.Function):
    """All encoder layers in one autograd node.

    inputs : h0 [B,S,H], pkv [L,2,B,P*H] or None, addmask [B,P+S], cfg tuple, weights (list of
             LayerWeights), then the 16*L layer parameters (so autograd routes their gradients).
    outputs: the L hidden states h_1..h_L, each [B,S,H].
    """

    @staticmethod
    def forward(ctx, h0, pkv, addmask, cfg, weights, grad_sink, *params):
        ctx.native = None
        if NATIVE_EXEC and len(weights):
            return _native_forward(ctx, h0, pkv, addmask, cfg, weights, grad_sink, params)
        B, S, H = h0.shape
        NH, eps, p_hidden, p_attn, pkv_ready = cfg[:5]
        L = len(weights)
        M = B * S
        Pn = 0 if pkv is None else pkv.shape[3] // H
        x = h0.contiguous().view(M, H)
        seed = RNG.seed()
        saved, offs, saved_t = [], [], []
        outs = []
        use_h = _bf16_ok(M, H, weights[0].w1.shape[0]) if L else False
        wh = [_weights_bf16(w) for w in weights] if use_h else None
        KC = hip.KC
        # mixed precision: bf16 operands written by the producing kernels (LayerNorm, GELU epilogue), fp32 accumulation,
        # residual stream / LayerNorm / softmax statistics in fp32
        x_h = _cast(x) if use_h else None
        pkv16 = None
        for li, w in enumerate(weights):
            I = w.w1.shape[0]
            off = RNG.next(3)
            lse = _empty(B, NH, S, like=x)
            if use_h:
                # Q|K|V, the context and (in backward) their gradients exist only as bf16: written by the producing
                # kernel's epilogue, read by the attention kernels / the next projection
                wqkv_h, wo_h, w1_h, w2_h = wh[li]
                qkv, cx = _bf16(M, 3 * H, like=x), _bf16(M, H, like=x)
                hip.gemm_bf16x(x_h, KC, wqkv_h, KC, M, 3 * H, H, out16=qkv, bias=w.bqkv)
                if li == 0 and Pn:
                    if pkv_ready is not None:  # prefix produced on the second stream (prompt generator)
                        cur = torch.cuda.current_stream()
                        cur.wait_event(pkv_ready)
                        pkv.record_stream(cur)
                    pkv16 = _bf16(*pkv.shape, like=x)  # one cast of all layers' prefix slabs per step
                    hip.cast_bf16(pkv.view(-1, pkv.shape[3]), out=pkv16.view(-1, pkv.shape[3]))
                hip.prefix_attn_bf16_fwd(qkv, pkv16[li, 0] if Pn else None, pkv16[li, 1] if Pn else None, addmask, cx, lse,
                                         B, S, Pn, NH, p_attn, seed, off)
            else:
                qkv, cx = _empty(M, 3 * H, like=x), _empty(M, H, like=x)
                hip.linear_fwd(x, w.wqkv, w.bqkv, qkv)
                pk = pkv[li, 0] if Pn else None
                pv = pkv[li, 1] if Pn else None
                if li == 0 and Pn and pkv_ready is not None:  # prefix produced on the second stream (prompt generator)
                    cur = torch.cuda.current_stream()
                    cur.wait_event(pkv_ready)
                    pkv.record_stream(cur)
                hip.prefix_attn_fwd(qkv, pk, pv, addmask, cx, lse, B, S, Pn, NH, p_attn, seed, off)
            a = _empty(M, H, like=x)
            if use_h:
                hip.gemm_bf16x(cx, KC, wo_h, KC, M, H, H, out32=a, bias=w.bo)
            else:
                hip.linear_fwd(cx, w.wo, w.bo, a)
            h1, mean1, rstd1 = _empty(M, H, like=x), _empty(M, like=x), _empty(M, like=x)
            f = _empty(M, H, like=x)
            h2, mean2, rstd2 = _empty(M, H, like=x), _empty(M, like=x), _empty(M, like=x)
            if use_h:
                h1_h, h2_h = _bf16(M, H, like=x), _bf16(M, H, like=x)
                hip.dropout_res_ln_fwd(a, x, w.g1, w.b1, h1, mean1, rstd1, eps, p_hidden, seed, off + 1, out16=h1_h)
                pre, act = _bf16(M, I, like=x), _bf16(M, I, like=x)  # bf16 only: pre feeds GELU', act the two products
                hip.gemm_bf16x(h1_h, KC, w1_h, KC, M, I, H, out16=act, bias=w.bi1, epi=hip.EPI_GELU, aux16=pre)
                hip.gemm_bf16x(act, KC, w2_h, KC, M, H, I, out32=f, bias=w.bi2)
                hip.dropout_res_ln_fwd(f, h1, w.g2, w.b2, h2, mean2, rstd2, eps, p_hidden, seed, off + 2, out16=h2_h)
                saved_t.extend((x_h, h1_h))
                x_h = h2_h
            else:
                hip.dropout_res_ln_fwd(a, x, w.g1, w.b1, h1, mean1, rstd1, eps, p_hidden, seed, off + 1)
                pre, act = _empty(M, I, like=x), _empty(M, I, like=x)
                hip.linear_fwd(h1, w.w1, w.bi1, act, epi=hip.EPI_GELU, aux=pre)
                hip.linear_fwd(act, w.w2, w.bi2, f)
                hip.dropout_res_ln_fwd(f, h1, w.g2, w.b2, h2, mean2, rstd2, eps, p_hidden, seed, off + 2)
            saved.extend((x, qkv, cx, lse, a, h1, mean1, rstd1, pre, act, f, mean2, rstd2))
            offs.append(off)
            outs.append(h2.view(B, S, H))
            x = h2
        # activations go through save_for_backward so that autograd releases them as soon as the backward has run
        # (a python attribute would keep ~3 GB per layer at B=128, S=512 alive until the loss tensor dies)
        ctx.save_for_backward(*saved, *saved_t)
        ctx.stash = (offs, weights, pkv, addmask, cfg, seed, (B, S, H, Pn), grad_sink, params)
        ctx.wh = wh
        ctx.pkv16 = pkv16
        if grad_sink is not None:
            grad_sink.node_created()
        ctx.set_materialize_grads(False)  # unused hidden states arrive as None, not as [B,S,H] zero fills + adds
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        if ctx.native is not None:
            return _native_backward(ctx, douts)
        offs, weights, pkv, addmask, cfg, seed, (B, S, H, Pn), grad_sink, params = ctx.stash
        flat = ctx.saved_tensors
        NH, eps, p_hidden, p_attn = cfg[:4]
        L = len(weights)
        M = B * S
        dev_like = flat[0]
        dpkv = torch.empty_like(pkv) if Pn else None
        need_param_grads = any(p.requires_grad for p in params)
        # parameter-gradient destinations: the module's flat per-layer gradient buffers when they are
        # free (param.grad is None), otherwise fresh tensors that autograd accumulates.
        gviews = grad_sink.acquire(params) if grad_sink is not None else None
        if grad_sink is not None:
            grad_sink.token_rows = M
        pgrads: List[Optional[torch.Tensor]] = [None] * len(params)

        wh = ctx.wh
        use_h = wh is not None
        flat_t = flat[13 * L:]  # bf16 operand copies (x, h1 per layer) in mixed-precision mode
        pkv16 = ctx.pkv16
        KC, KM = hip.KC, hip.KM
        main = torch.cuda.current_stream()
        # (small batches are host-bound: the extra events / stream switches cost more than the overlap returns)
        side = _side_stream(dev_like.device) if (DW_SIDE_STREAM and need_param_grads and M >= DW_STREAM_MIN_ROWS) else None

        def on_side(reads, fn):
            """Run `fn` (weight-gradient kernels that only read `reads`) behind everything enqueued so far."""
            if side is None:
                fn()
                return
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                fn()
            for t in reads:  # temporaries freed before the join below: the allocator must wait for the side stream
                t.record_stream(side)

        # (the same contract flag as the native executor's: the two paths must stay bit-identical)
        zero_tail = bool(SKIP_PAD_DW and len(cfg) > 5 and cfg[5] and addmask.dtype == torch.float32 and addmask.is_contiguous())
        if CHECK_CONTRACT and zero_tail:
            _check_masked_rows_contract(douts, addmask, Pn, B, S)
        ktiles = None  # (the native executor's k-tile list: the two paths must stay bit-identical)
        if (SKIP_PAD_DW and (not use_h or SKIP_PAD_DW_BF16) and need_param_grads and len(cfg) > 5 and cfg[5] and M % (64 if use_h else 32) == 0
                and addmask.dtype == torch.float32 and addmask.is_contiguous()):
            ktiles = hip.build_ktiles(addmask, Pn, S, bk=64 if use_h else 32)
            if side is not None:
                for t_ in ktiles:
                    t_.record_stream(side)

        dh = None  # gradient wrt the current layer's OUTPUT, [M,H], owned by us
        for li in range(L - 1, -1, -1):
            x, qkv, cx, lse, a, h1, mean1, rstd1, pre, act, f, mean2, rstd2 = flat[13 * li:13 * li + 13]
            off = offs[li]
            w = weights[li]
            I = w.w1.shape[0]
            g_out = douts[li]
            if dh is None:
                if g_out is None:
                    if Pn:
                        dpkv[li].zero_()  # layers above the last used hidden state get no gradient
                    continue
                dh = g_out.contiguous().view(M, H)
                if dh.data_ptr() == g_out.data_ptr():
                    dh = dh.clone()  # we modify / free it
            elif g_out is not None:
                dh = dh + g_out.reshape(M, H)
            base = li * N_LAYER_PARAMS
            if gviews is not None:
                G = gviews[base:base + N_LAYER_PARAMS]
                # QKV packed views: G[0] spans the [3H,H] weight block, G[1] the [3H] bias block
                dwqkv, dbqkv = grad_sink.packed_qkv(li)
            else:
                G = [torch.empty_like(p) for p in params[base:base + N_LAYER_PARAMS]]
                dwqkv, dbqkv = _empty(3 * H, H, like=dev_like), _empty(3 * H, like=dev_like)
            # ---- FFN block ----
            dh1 = _empty(M, H, like=dev_like)
            if use_h:
                # every product reads the same row-major bf16 tensors: dX = dY . W takes W as the KM operand, dW = dY^T . X
                # takes BOTH as KM operands (reduction over the token rows) -- no casts, no transposed copies
                wqkv_h, wo_h, w1_h, w2_h = wh[li]
                x_h, h1_h = flat_t[2 * li:2 * li + 2]
                df_h = _bf16(M, H, like=dev_like)
                hip.dropout_res_ln_bwd(dh, f, h1, w.g2, mean2, rstd2, None, dh1, False, G[14], G[15], False, p_hidden, seed,
                                       off + 2, dbias_x=G[13], dx16=df_h)
                # (the four weight gradients of the layer as ONE launch when a stream-K scratch is attached to the second stream:
                # the same decision, on the same inputs, as csrc/executor.hip)
                grp = ktiles is None and M % 256 == 0 and H % 256 == 0 and I % 256 == 0 and _streamk_on(side, dev_like.device)
                if not grp:
                    on_side((df_h,), lambda: hip.gemm_bf16x(df_h, KM, act, KM, H, I, M, out32=G[12], allow_split=True, ktiles=ktiles))
                dpre_h, part = _bf16(M, I, like=dev_like), _empty(M // 128, I, like=dev_like)
                hip.gemm_bf16x(df_h, KC, w2_h, KM, M, I, H, out16=dpre_h, epi=hip.EPI_DGELU, aux16=pre, colpart=part)

                def ffn1_grads():
                    hip.colsum_small(part, G[11])
                    if not grp:
                        hip.gemm_bf16x(dpre_h, KM, h1_h, KM, I, H, M, out32=G[10], allow_split=True, ktiles=ktiles)
                on_side((dpre_h, part), ffn1_grads)
                hip.gemm_bf16x(dpre_h, KC, w1_h, KM, M, H, I, out32=dh1, accumulate=True)
            else:
                df = _empty(M, H, like=dev_like)
                hip.dropout_res_ln_bwd(dh, f, h1, w.g2, mean2, rstd2, df, dh1, False, G[14], G[15], False, p_hidden, seed,
                                       off + 2, dbias_x=G[13])
                dpre = _empty(M, I, like=dev_like)
                # (few token rows: the layer's four weight gradients as ONE launch -- the same rule as csrc/executor.hip)
                grp = hip.dw_group_wanted(M, H, I)
                if not grp:
                    on_side((df,), lambda: hip.linear_bwd_weight(df, act, G[12], ktiles=ktiles))
                hip.linear_bwd_input(df, w.w2, dpre, epi=hip.EPI_DGELU, aux=pre)

                def ffn1_grads():
                    if not grp:  # (grouped: the bias gradient comes out of the grouped launch, as in csrc/executor.hip)
                        hip.colsum(dpre, G[11])
                        hip.linear_bwd_weight(dpre, h1, G[10], ktiles=ktiles)
                on_side((dpre,), ffn1_grads)
                hip.linear_bwd_input(dpre, w.w1, dh1, accumulate=True)
            # ---- attention block ----
            dh0 = dh  # reuse
            if use_h:
                da_h, dctx = _bf16(M, H, like=dev_like), _bf16(M, H, like=dev_like)
                hip.dropout_res_ln_bwd(dh1, a, x, w.g1, mean1, rstd1, None, dh0, False, G[8], G[9], False, p_hidden, seed,
                                       off + 1, dbias_x=G[7], dx16=da_h)
                if not grp:
                    on_side((da_h,), lambda: hip.gemm_bf16x(da_h, KM, cx, KM, H, H, M, out32=G[6], allow_split=True, ktiles=ktiles))
                hip.gemm_bf16x(da_h, KC, wo_h, KM, M, H, H, out16=dctx)
                dqkv = _bf16(M, 3 * H, like=dev_like)
                nqt, nkt = (S + 63) // 64, (Pn + S + 63) // 64
                partq, partkv = _empty(B * nqt, H, like=dev_like), _empty(B * nkt, 2 * H, like=dev_like)
                hip.prefix_attn_bf16_bwd(dctx, qkv, pkv16[li, 0] if Pn else None, pkv16[li, 1] if Pn else None, addmask, cx,
                                         lse, dqkv, dpkv[li, 0] if Pn else None, dpkv[li, 1] if Pn else None, partq, partkv,
                                         B, S, Pn, NH, p_attn, seed, off, zero_tail=zero_tail or ktiles is not None)

                def qkv_grads():
                    (hip.colsum_small if partq.shape[0] <= 256 else hip.colsum)(partq, dbqkv[:H])
                    (hip.colsum_small if partkv.shape[0] <= 256 else hip.colsum)(partkv, dbqkv[H:])
                    if grp:
                        hip.gemm_bf16x_dw_group([(df_h, act, G[12]), (dpre_h, h1_h, G[10]), (da_h, cx, G[6]), (dqkv, x_h, dwqkv)], M)
                    else:
                        hip.gemm_bf16x(dqkv, KM, x_h, KM, 3 * H, H, M, out32=dwqkv, allow_split=True, ktiles=ktiles)
                on_side((dqkv, partq, partkv, df_h, dpre_h, da_h), qkv_grads)
                hip.gemm_bf16x(dqkv, KC, wqkv_h, KM, M, H, 3 * H, out32=dh0, accumulate=True)
            else:
                dctx = dh1  # reuse (LayerNorm backward has consumed it by the time the dX product writes)
                # (df is still to be read: on the side stream, or by the grouped launch at the end of the layer)
                da = df if (side is None and not grp) else _empty(M, H, like=dev_like)
                hip.dropout_res_ln_bwd(dh1, a, x, w.g1, mean1, rstd1, da, dh0, False, G[8], G[9], False, p_hidden, seed,
                                       off + 1, dbias_x=G[7])
                if not grp:
                    on_side((da,), lambda: hip.linear_bwd_weight(da, cx, G[6], ktiles=ktiles))
                hip.linear_bwd_input(da, w.wo, dctx)
                dqkv, delta = _empty(M, 3 * H, like=dev_like), _empty(B, NH, S, like=dev_like)
                hip.prefix_attn_bwd(dctx, qkv, pkv[li, 0] if Pn else None, pkv[li, 1] if Pn else None, addmask, cx, lse,
                                    delta, dqkv, dpkv[li, 0] if Pn else None, dpkv[li, 1] if Pn else None, B, S, Pn, NH,
                                    p_attn, seed, off, zero_tail=zero_tail or ktiles is not None)

                def qkv_grads():
                    if grp:
                        hip.gemm_f32_dw_group([(df, act, G[12]), (dpre, h1, G[10]), (da, cx, G[6]), (dqkv, x, dwqkv)], M, ktiles=ktiles,
                                              dbias=[None, G[11], None, dbqkv])
                    else:
                        hip.colsum(dqkv, dbqkv)
                        hip.linear_bwd_weight(dqkv, x, dwqkv, ktiles=ktiles)
                on_side((dqkv, df, dpre, da) if grp else (dqkv,), qkv_grads)
                hip.linear_bwd_input(dqkv, w.wqkv, dh0, accumulate=True)
            dh = dh0
            if gviews is None:
                G[0], G[2], G[4] = dwqkv[:H], dwqkv[H:2 * H], dwqkv[2 * H:]
                G[1], G[3], G[5] = dbqkv[:H], dbqkv[H:2 * H], dbqkv[2 * H:]
            pgrads[base:base + N_LAYER_PARAMS] = G
            if grad_sink is not None:
                if side is not None and grad_sink.on_layer_done is not None:
                    # the hook (GradSync) orders its all-reduce behind the CURRENT stream: on the side stream that
                    # is behind this layer's last weight-gradient kernel, which itself waited for everything the
                    # main stream had produced up to the attention backward (incl. the LayerNorm gradients)
                    if grad_sink.settle_params:
                        # an optimizer update hangs off the hook: it must also be behind the layer's LAST product that
                        # reads the weights (dX of the QKV projection, enqueued on the main stream just above)
                        ev = torch.cuda.Event()
                        ev.record(main)
                        side.wait_event(ev)
                    with torch.cuda.stream(side):
                        grad_sink.layer_done(li)
                else:
                    grad_sink.layer_done(li)
        if side is not None:
            main.wait_stream(side)  # join: gradients (and every buffer the side stream read) are settled from here on
        dh0_out = dh.view(B, S, H) if dh is not None else None
        if not need_param_grads:
            pgrads = [None] * len(params)
        return (dh0_out, dpkv, None, None, None, None, *pgrads)


# -------------------------------------------------------------------------------------------------
class LinearFunction(torch.autograd.Function):
    """y = act(x . W^T + b), act in {identity, tanh}; x is [..., K]."""

    @staticmethod
    def forward(ctx, x, weight, bias, tanh):
        K = x.shape[-1]
        x2 = x.contiguous().view(-1, K)
        y = _empty(x2.shape[0], weight.shape[0], like=x2)
        hip.linear_fwd(x2, weight, bias, y, epi=hip.EPI_TANH if tanh else hip.EPI_NONE)
        ctx.stash = (x2, weight, y if tanh else None, bias is not None, x.shape)
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, weight, y_tanh, has_bias, xshape = ctx.stash
        N, K = weight.shape
        dy2 = dy.contiguous().view(-1, N)
        if y_tanh is not None:
            # d(pre) = dy * (1 - y^2): run it through the dtanh epilogue of an identity-free GEMM is
            # overkill; the product dy.W needs d(pre) as its A operand, so materialise it once.
            dpre = torch.addcmul(dy2, dy2 * y_tanh, y_tanh, value=-1.0)
        else:
            dpre = dy2
        dx = dw = db = None
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_db = has_bias and ctx.needs_input_grad[2]
        # (running dW / db beside a skinny dX on a third stream measured 0.4-0.7 % SLOWER on the whole step: the products
        # of the prompt generator's backward already overlap the optimizer updates of the last encoder layers)
        if need_dx:
            dx = _empty(x2.shape[0], K, like=x2)
            hip.linear_bwd_input(dpre, weight, dx)
            dx = dx.view(xshape)
        if need_dw:
            dw = torch.empty_like(weight)
            hip.linear_bwd_weight(dpre, x2, dw)
        if need_db:
            db = _empty(N, like=x2)
            hip.colsum(dpre, db)
        return dx, dw, db, None


class DropoutFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        xc = x.contiguous()
        y = torch.empty_like(xc)
        seed, off = RNG.seed(), RNG.next()
        hip.dropout(xc, y, p, seed, off)
        ctx.stash = (p, seed, off)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed, off = ctx.stash
        dyc = dy.contiguous()
        dx = torch.empty_like(dyc)
        hip.dropout(dyc, dx, p, seed, off)
        return dx, None


def dropout(x, p, training):
    if not training or p <= 0.0:
        return x
    return DropoutFunction.apply(x, p)


# -------------------------------------------------------------------------------------------------
class CRFNLLFunction(torch.autograd.Function):
    """loss = -mean_b log p(tags_b | emissions_b)  (0-dim tensor)."""

    @staticmethod
    def forward(ctx, emissions, start, end, trans, tags, mask_u8):
        B, S, C = emissions.shape
        em = emissions.contiguous()
        ws, wsb = hip.crf_workspace(B, S, C, em.device)
        loss = _empty(1, like=em)
        hip.crf_nll_fwd(em, tags, mask_u8, start, end, trans, loss, ws, wsb)
        ctx.stash = (em, start, end, trans, tags, mask_u8, ws, wsb)
        return loss.view(())

    @staticmethod
    def backward(ctx, gout):
        em, start, end, trans, tags, mask_u8, ws, wsb = ctx.stash
        g = gout.contiguous().view(1).float()
        dem = torch.empty_like(em)
        ds, de, dt = torch.empty_like(start), torch.empty_like(end), torch.empty_like(trans)
        hip.crf_nll_bwd(g, em, tags, mask_u8, start, end, trans, dem, ds, de, dt, False, ws, wsb)
        return dem, ds, de, dt, None, None


# -------------------------------------------------------------------------------------------------
class PromptFunction(torch.autograd.Function):
    """enc [NI,B,L,4W] + packed projector weights -> pkv [NL,2,B,(NI*L)*(W/2)].

    The 2*NL projector parameters are passed so that autograd routes their gradients; the kernels read
    the packed [NL*4, L*W] / [NL*4] buffers they are views of."""

    @staticmethod
    def forward(ctx, enc, wp, bp, NL, *proj_params):
        NI, B, Lp, W4 = enc.shape
        W = W4 // 4
        n = NI * B
        encc = enc.contiguous()
        sm = _empty(n, Lp * W, like=encc)
        hip._ck(hip.lib().mtvaf_split_mean(hip._p(encc), hip._p(sm), n * Lp, W, hip._st()), "mtvaf_split_mean")
        logits = _empty(n, NL * 4, like=encc)
        hip.linear_fwd(sm, wp, bp, logits)
        gate = torch.empty_like(logits)
        hip._ck(hip.lib().mtvaf_gate_fwd(hip._p(logits), hip._p(gate), n * NL, hip._st()), "mtvaf_gate_fwd")
        pkv = _empty(NL, 2, B, NI * Lp * (W // 2), like=encc)
        hip._ck(hip.lib().mtvaf_prompt_mix_fwd(hip._p(encc), hip._p(gate), hip._p(pkv), NI, B, Lp, W, NL, hip._st()),
                "mtvaf_prompt_mix_fwd")
        ctx.stash = (encc, wp, sm, logits, gate, NL)
        ctx.proj_params = proj_params
        return pkv

    @staticmethod
    def backward(ctx, dpkv):
        encc, wp, sm, logits, gate, NL = ctx.stash
        NI, B, Lp, W4 = encc.shape
        W = W4 // 4
        n = NI * B
        dpkv = dpkv.contiguous()
        dpart = _empty(n * Lp, NL * 4, like=encc)
        dlog = _empty(n, NL * 4, like=encc)
        L_ = hip.lib()
        hip._ck(L_.mtvaf_prompt_mix_bwd_gate(hip._p(encc), hip._p(dpkv), hip._p(logits), hip._p(gate), hip._p(dpart),
                                             hip._p(dlog), NI, B, Lp, W, NL, hip._st()), "mtvaf_prompt_mix_bwd_gate")
        dwp = torch.empty_like(wp)
        hip.linear_bwd_weight(dlog, sm, dwp)
        dbp = _empty(NL * 4, like=encc)
        hip.colsum(dlog, dbp)
        dsm = torch.empty_like(sm)
        hip.linear_bwd_input(dlog, wp, dsm)
        denc = torch.empty_like(encc)
        hip._ck(L_.mtvaf_prompt_mix_bwd_enc(hip._p(gate), hip._p(dpkv), hip._p(dsm), hip._p(denc), NI, B, Lp, W, NL,
                                            hip._st()), "mtvaf_prompt_mix_bwd_enc")
        pg = []
        for i in range(NL):
            pg.append(dwp[4 * i:4 * i + 4])
            pg.append(dbp[4 * i:4 * i + 4])
        pp = ctx.proj_params
        if DIRECT_GRADS and all(p.requires_grad and p.grad is None for p in pp) and not _has_grad_hooks(pp):
            # the 2*NL slices of the packed gradients become .grad directly: handed to autograd they are VIEWS, which
            # AccumulateGrad deep-copies one by one (24 copies at the very tail of the step, behind the encoder backward)
            for p, g in zip(pp, pg):
                p.grad = g
            pg = [None] * len(pg)
        return (denc, None, None, None, *pg)


class MeanLFunction(torch.autograd.Function):
    """[n, L, W] -> mean over L  (prefix_guids.mean(dim=1), bert_model.py:550)."""

    @staticmethod
    def forward(ctx, enc):
        n, Lp, W4 = enc.shape
        e = enc.contiguous()
        out = _empty(n, W4, like=e)
        hip._ck(hip.lib().mtvaf_mean_l_fwd(hip._p(e), hip._p(out), n, Lp, W4, hip._st()), "mtvaf_mean_l_fwd")
        ctx.shape = (n, Lp, W4)
        return out

    @staticmethod
    def backward(ctx, dmean):
        n, Lp, W4 = ctx.shape
        d = dmean.contiguous()
        denc = torch.zeros(n, Lp, W4, device=d.device, dtype=d.dtype)
        hip._ck(hip.lib().mtvaf_mean_l_bwd(hip._p(d), hip._p(denc), n, Lp, W4, hip._st()), "mtvaf_mean_l_bwd")
        return denc


class KLFunction(torch.autograd.Function):
    """KLDivLoss(reduction='batchmean')(log_softmax(logits), target) -> 0-dim tensor."""

    @staticmethod
    def forward(ctx, logits, target):
        B, N = logits.shape
        z, t = logits.contiguous(), target.contiguous().float()
        loss, row = _empty(1, like=z), _empty(B, like=z)
        hip._ck(hip.lib().mtvaf_kl_logsoftmax_fwd(hip._p(z), hip._p(t), hip._p(loss), hip._p(row), B, N, hip._st()),
                "mtvaf_kl_logsoftmax_fwd")
        ctx.stash = (z, t)
        return loss.view(())

    @staticmethod
    def backward(ctx, gout):
        z, t = ctx.stash
        B, N = z.shape
        g = gout.contiguous().view(1).float()
        dz = torch.empty_like(z)
        hip._ck(hip.lib().mtvaf_kl_logsoftmax_bwd(hip._p(g), 1.0, hip._p(z), hip._p(t), hip._p(dz), B, N, hip._st()),
                "mtvaf_kl_logsoftmax_bwd")
        return dz, None


# -------------------------------------------------------------------------------------------------
# span model heads (TVNetSAModel): reference models/bert_model.py:147-190, 288-305, 363-369
# -------------------------------------------------------------------------------------------------
class SpanPoolFunction(torch.autograd.Function):
    """get_span_representation + unary_affine + get_self_att_representation fused (bert_model.py:364-369):
    pooled[b*M+m] = sum_r softmax_r(score) x_r over the span's tokens.  `index` comes from hip.span_index."""

    @staticmethod
    def forward(ctx, seq, w_unary, b_unary, index, M):
        B, S, H = seq.shape
        x = seq.contiguous()
        pooled, stats = _empty(B * M, H, like=x), _empty(B * M, 2, like=x)
        hip.span_pool_fwd(x, w_unary, b_unary, index, pooled, stats, B, S, M)
        ctx.stash = (x, w_unary, b_unary, index, pooled, stats, M)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        x, w_unary, b_unary, index, pooled, stats, M = ctx.stash
        B, S, H = x.shape
        dseq = torch.empty_like(x)
        dw, db = torch.empty_like(w_unary), torch.empty_like(b_unary)
        hip.span_pool_bwd(dpooled.contiguous(), pooled, stats, x, w_unary, b_unary, index, dseq, dw.view(-1), db, B, S, M)
        return dseq, dw, db, None, None


class DistantCEPairFunction(torch.autograd.Function):
    """(distant_cross_entropy(start_logits, start_positions) + distant_cross_entropy(end_logits, end_positions)) / 2
    (bert_model.py:298-300) on the [B,S,2] binary_affine output -- the two logit sets are its columns."""

    @staticmethod
    def forward(ctx, ae_logits, start_positions, end_positions):
        B, S, two = ae_logits.shape
        assert two == 2
        z = ae_logits.contiguous()
        sp, ep = start_positions.contiguous().float(), end_positions.contiguous().float()
        loss, ws = _empty(1, like=z), _empty(2, B, 3, like=z)
        hip.distant_ce_fwd(z, 2, sp, loss, ws[0], B, S, 0.5, False)
        hip.distant_ce_fwd(z[..., 1:], 2, ep, loss, ws[1], B, S, 0.5, True)
        ctx.stash = (z, sp, ep, ws)
        return loss.view(())

    @staticmethod
    def backward(ctx, gout):
        z, sp, ep, ws = ctx.stash
        B, S, _ = z.shape
        g = gout.contiguous().view(1).float()
        dz = torch.empty_like(z)
        hip.distant_ce_bwd(g, 0.5, z, 2, sp, ws[0], dz, 2, B, S)
        hip.distant_ce_bwd(g, 0.5, z[..., 1:], 2, ep, ws[1], dz[..., 1:], 2, B, S)
        return dz, None, None


class CrossEntropyFunction(torch.autograd.Function):
    """nn.CrossEntropyLoss() (mean over rows whose label != -100) -> 0-dim tensor."""

    @staticmethod
    def forward(ctx, logits, labels):
        z, lab = logits.contiguous(), labels.contiguous().long()
        loss, ws2 = _empty(1, like=z), _empty(2, like=z)
        hip.ce_fwd(z, lab, loss, ws2)
        ctx.stash = (z, lab, ws2)
        return loss.view(())

    @staticmethod
    def backward(ctx, gout):
        z, lab, ws2 = ctx.stash
        dz = torch.empty_like(z)
        hip.ce_bwd(gout.contiguous().view(1).float(), z, lab, ws2, dz)
        return dz, None


class MaskMulFunction(torch.autograd.Function):
    """Cutoff augmentation apply (modules/augument.py:99-159): x * row_keep[b,s] * col_keep[b,:]."""

    @staticmethod
    def forward(ctx, x, row_keep, col_keep):
        xc = x.contiguous()
        ctx.stash = (row_keep, col_keep)
        return hip.mask_mul(xc, row_keep, col_keep, torch.empty_like(xc))

    @staticmethod
    def backward(ctx, dy):
        row_keep, col_keep = ctx.stash
        dyc = dy.contiguous()
        return hip.mask_mul(dyc, row_keep, col_keep, torch.empty_like(dyc)), None, None
