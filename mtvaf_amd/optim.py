"""Optimizer / LR schedule of the reference trainers, for the drop-in models.

``AdamW`` below is the hand-written optimizer of the path: torch.optim.AdamW semantics on the HIP kernels of
``csrc/optim.hip`` -- ONE launch per encoder layer over the layer's flat parameter / gradient / moment buffers, one
multi-tensor launch per 48 remaining tensors -- and, with ``overlap=True``, the per-layer updates are enqueued from
inside the backward pass (behind the layer's gradient all-reduce on the communication stream under ``GradSync``, behind
its weight-gradient kernels on the second stream otherwise), so the HBM-bound update runs next to the MFMA-bound rest of
the backward pass instead of after it.

``modules/train.py::SATrainer2.multiModal_before_train`` (:894-926) builds three AdamW parameter groups by
NAME (encoder ``bert*`` at ``args.lr``; ``encoder_conv*`` / ``gates*`` at ``args.lr``; ``crf*`` / ``fc*`` at
5e-2; weight decay 1e-2 everywhere), freezes ``image_model*`` and attaches a linear warm-up / linear decay
schedule; ``bert_before_train`` (:887-892) is the text-only variant (one group, all parameters).  The same
grouping is reproduced here (``reference_param_groups`` / ``build_optimizer``) on top of ``AdamW`` below, so a step of
the reference trainer costs what ``bench.py`` measures.

Reference quirk kept on purpose: ``projectors.*``, ``img_classifier.*`` and ``aux_img_classifier.*`` match no
group (the second group looks for the long-gone ``gates``), so the reference never updates them.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import os

import torch

from . import hip


class AdamW(torch.optim.Optimizer):
    """torch.optim.AdamW (decoupled weight decay, bias correction, no amsgrad) on the gfx950 kernels.

    ``attach(model)`` (or ``model=`` at construction) lets the optimizer see the encoder's per-layer flat buffers
    (``BertEncoder._stores``): a layer whose 16 parameters sit in one parameter group is updated by ONE launch.

    ``overlap=True``: layers are updated from inside ``loss.backward()`` as soon as their gradients are final (after
    the all-reduce under data parallelism).  Contract: every backward pass is followed by exactly one ``step()`` before
    the next backward (the reference trainer's flow with gradient_accumulation_steps = 1, modules/train.py:620-625);
    a second backward without a ``step()`` raises -- with or without ``zero_grad`` in between, so gradient accumulation
    (gradient_accumulation_steps > 1, train.py:616-625) cannot silently train the encoder on one micro-batch: use
    ``overlap=False`` for it (``build_optimizer`` does).  ``step()`` then only updates what is left (embeddings, heads,
    prompt generator) and layers that could not take the fast path.

    Checkpoints: ``state_dict()`` / ``load_state_dict()`` are torch.optim's; the flat per-layer moment buffers are
    rebuilt from the loaded per-parameter ``exp_avg`` / ``exp_avg_sq`` / ``step`` entries on the first update."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 model: Optional[torch.nn.Module] = None, overlap: bool = False, grad_sync=None):
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0) or weight_decay < 0.0:
            raise ValueError("invalid AdamW hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.overlap = overlap
        self._encoder = None
        self._layer_group: Dict[int, Optional[dict]] = {}
        self._early = set()
        self._param_layer: Dict[int, int] = {}  # id(parameter) -> encoder layer whose flat state its entries are views of
        self._background_ok = True
        self.suspended = False    # True: the backward hook does nothing (backward passes that are not followed by step())
        if model is not None:
            self.attach(model, grad_sync)

    # -- wiring -----------------------------------------------------------------------------------------------
    def attach(self, model: torch.nn.Module, grad_sync=None):
        enc = getattr(getattr(model, "bert", model), "encoder", None)
        if enc is None or not hasattr(enc, "grad_sink"):
            raise ValueError("AdamW.attach needs a mtvaf_amd model (TVNetSAModel2 / TVNetSAModel / BertModel)")
        self._encoder = enc
        self._map_layers()
        # under GradSync the hook runs on the communication stream, where a slow update would delay the next layer's
        # all-reduce: full width there
        self._background_ok = grad_sync is None
        sink = enc.grad_sink
        sink.optimizer = self  # (a GradSync constructed later re-wires the hook through this, parallel.py)
        if self.overlap:
            sink.settle_params = True  # the hook's stream must also be behind the layer's last dX product (engine.py)
            if grad_sync is not None:
                grad_sync.adopt_optimizer(self)
            else:
                sink.on_layer_done = self._sink_hook
                sink.raw_stream_hook = True  # the update is one library launch on hip._st()
        return self

    def _sink_hook(self, li: int, flat):
        """GradSink.on_layer_done without data parallelism.  flat is None: the pass could not use the flat gradient buffers
        (some .grad pre-existed: gradient accumulation, or two encoder nodes under one backward) -- that layer is left to
        step(); but if an earlier pass has ALREADY updated layers from inside its backward, this pass accumulates on top of
        weights that moved mid-accumulation: refuse."""
        self.layer_pass_check(li, flat)
        if flat is not None:
            self._early_layer_update(li)

    def layer_pass_check(self, li: int, flat):
        if self.suspended or not self.overlap:
            return
        if flat is None and self._early:
            raise RuntimeError("mtvaf_amd.optim.AdamW(overlap=True): a second backward pass ran before optimizer.step() "
                               "(gradient accumulation needs overlap=False)")

    def _map_layers(self):
        group_of = {id(p): g for g in self.param_groups for p in g["params"]}
        self._layer_group = {}
        for li, layer in enumerate(self._encoder.layer):
            gs = {id(group_of.get(id(p))) for p in layer.ordered_params()}
            g0 = group_of.get(id(layer.ordered_params()[0]))
            self._layer_group[li] = g0 if (len(gs) == 1 and g0 is not None) else None

    def _layer_state(self, li: int, store):
        """Flat moment buffers of layer li; the per-parameter state entries are views of them."""
        key = ("layer", li)
        st = self.__dict__.setdefault("_flat_state", {}).get(key)
        if st is None or st["m"].numel() != store.flat.numel() or st["m"].device != store.flat.device:
            st = {"m": torch.zeros_like(store.flat), "v": torch.zeros_like(store.flat), "step": 0,
                  "step_t": torch.zeros((), dtype=torch.float32)}
            self._flat_state[key] = st
            steps = set()
            for i, p in enumerate(self._encoder.layer[li].ordered_params()):
                off, n = store.offsets[i], p.numel()
                ps = self.state[p]
                m_v, v_v = st["m"][off:off + n].view(p.shape), st["v"][off:off + n].view(p.shape)
                if "exp_avg" in ps:  # per-parameter state that exists already (load_state_dict, or earlier per-tensor updates)
                    m_v.copy_(ps["exp_avg"])
                    v_v.copy_(ps["exp_avg_sq"])
                    steps.add(int(ps.get("step", 0)))
                ps["exp_avg"], ps["exp_avg_sq"] = m_v, v_v
                ps["step"] = st["step_t"]  # (torch.optim keeps `step` as a tensor too)
                self._param_layer[id(p)] = li
            if len(steps) > 1:
                raise RuntimeError(f"AdamW: the parameters of encoder layer {li} carry different step counts {sorted(steps)}")
            if steps:
                st["step"] = steps.pop()
                st["step_t"].fill_(st["step"])
        return st

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # the flat per-layer buffers are rebuilt from the loaded per-parameter entries by the next _layer_state()
        self.__dict__["_flat_state"] = {}
        self._param_layer = {}
        self._early = set()

    # an update enqueued from inside the backward pass runs beside MFMA-bound products: on 128 blocks it trickles under
    # them (csrc/optim.hip); below this many token rows a layer's backward is shorter than such an update and it would
    # only pile up behind the pass
    BACKGROUND_MIN_ROWS = int(os.environ.get("MTVAF_ADAMW_BG_MIN_ROWS", "2048"))
    BACKGROUND_BLOCKS = int(os.environ.get("MTVAF_ADAMW_BG_BLOCKS", "128"))
    # (the pre-split path's second stream is lighter -- one grouped launch per layer -- and the update also writes the weights' plane
    # images: 256 blocks measured best there, 3257 - 3260 sentences/s against 3237 - 3238 at 128, 3220 - 3225 at 512, 3128 - 3133 at 64)
    BACKGROUND_BLOCKS_PLANES = int(os.environ.get("MTVAF_ADAMW_BG_BLOCKS", "256"))
    # (bf16 mode below 4096 token rows -- C3: a layer's backward pass is 260 us, shorter than a 128-block trickle of its update --
    # 256 blocks: 5.52 - 5.54 -> 5.41 - 5.43 ms median step, same box; at 4864 rows (C4) 128 / 192 / 256 measure equal)
    BACKGROUND_BLOCKS_SHORT = int(os.environ.get("MTVAF_ADAMW_BG_BLOCKS", "256"))
    BACKGROUND_SHORT_ROWS = 4096

    def _update_layer_flat(self, li: int, group: dict, store, background: bool = False, rows: int = 0):
        st = self._layer_state(li, store)
        st["step"] += 1
        st["step_t"].fill_(st["step"])  # the 16 per-parameter state entries share this 0-dim tensor
        b1, b2 = group["betas"]
        from . import engine
        shadow = engine.shadow_for_update(store.weights)  # bf16 compute mode: the GEMM operand image, written in the same pass
        segs = engine.planes_for_update(store.weights) if shadow is None else None
        if segs is not None and store.flat.numel() % 4 == 0:
            # fp32 mode, pre-split operands: the weights' plane images are rewritten by the update kernel itself
            hip.adamw_planes(store.flat, store.grad, st["m"], st["v"], float(group["lr"]), b1, b2, group["eps"], group["weight_decay"],
                             st["step"], segs, max_blocks=self.BACKGROUND_BLOCKS_PLANES if background else 0)
            engine.planes_written(store.weights)
            return
        hip.adamw(store.flat, store.grad, st["m"], st["v"], float(group["lr"]), b1, b2, group["eps"], group["weight_decay"],
                  st["step"], p_bf16=shadow,
                  max_blocks=(self.BACKGROUND_BLOCKS_SHORT if 0 < rows < self.BACKGROUND_SHORT_ROWS else self.BACKGROUND_BLOCKS) if background else 0)
        if shadow is not None:
            engine.shadow_written(store.weights)
        engine.planes_rewrite(store.weights)  # (images exist but the fused form does not apply: rebuilt behind the update)

    def _early_layer_update(self, li: int):
        """Called from inside the backward pass (current stream: the one the layer's gradients are final on)."""
        group = self._layer_group.get(li)
        stores = self._encoder._stores
        if self.suspended or group is None or stores is None or stores[li].grad is None:
            return
        if li in self._early:
            raise RuntimeError("mtvaf_amd.optim.AdamW(overlap=True): a second backward pass ran before optimizer.step()")
        # (the sink that is calling us: `encoder.grad_sink` would re-validate all twelve layer stores on each of the twelve calls --
        # 2304 pointer comparisons per step, 0.4 ms of host time in the launch-bound configurations)
        rows = getattr(self._encoder._sink, "token_rows", 0)
        # (layer 0 is the last one of the pass: nothing left to hide behind)
        background = self._background_ok and li > 0 and rows >= self.BACKGROUND_MIN_ROWS
        self._update_layer_flat(li, group, stores[li], background=background, rows=rows)
        self._early.add(li)

    # -- the step ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        done = set()
        enc = self._encoder
        if enc is not None and enc._stores is not None:
            if len(self._layer_group) != len(enc.layer):
                self._map_layers()
            for li, layer in enumerate(enc.layer):
                ps = layer.ordered_params()
                if li in self._early:
                    done.update(id(p) for p in ps)
                    continue
                group, store = self._layer_group.get(li), enc._stores[li]
                if group is None or store.grad is None or not store.valid(layer):
                    continue
                lo = store.grad.data_ptr()
                if all(p.grad is not None and p.grad.data_ptr() == lo + 4 * store.offsets[i] and p.grad.is_contiguous()
                       for i, p in enumerate(ps)):
                    self._update_layer_flat(li, group, store)
                    done.update(id(p) for p in ps)
        self._early = set()
        touched = set()  # layers updated tensor by tensor this step: ONE step counter per layer, shared with the flat path
        flat_state = self.__dict__.get("_flat_state", {})
        for group in self.param_groups:
            b1, b2 = group["betas"]
            buckets: Dict[int, list] = {}
            for p in group["params"]:
                if p.grad is None or id(p) in done:
                    continue
                if p.grad.is_sparse or p.dtype != torch.float32 or not p.is_cuda:
                    raise RuntimeError("mtvaf_amd.optim.AdamW updates dense fp32 parameters on the MI355X only")
                stt = self.state[p]
                if "exp_avg" not in stt:
                    stt["exp_avg"], stt["exp_avg_sq"], stt["step"] = torch.zeros_like(p), torch.zeros_like(p), 0
                li = self._param_layer.get(id(p))
                fst = flat_state.get(("layer", li)) if li is not None else None
                if fst is not None and stt["step"] is fst["step_t"]:
                    # moments live in the layer's flat buffers (views): this tensor-by-tensor update advances the layer's counter
                    step_no = fst["step"] + 1
                    touched.add(li)
                else:
                    if torch.is_tensor(stt["step"]):
                        stt["step"] = int(stt["step"])
                    stt["step"] += 1
                    step_no = stt["step"]
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if not p.is_contiguous():
                    raise RuntimeError("non-contiguous parameter")
                buckets.setdefault(step_no, []).append((p, g, stt["exp_avg"], stt["exp_avg_sq"]))
            for step_no, items in buckets.items():
                hip.adamw_multi([i[0] for i in items], [i[1] for i in items], [i[2] for i in items], [i[3] for i in items],
                                float(group["lr"]), b1, b2, group["eps"], group["weight_decay"], step_no)
        for li in touched:
            fst = flat_state[("layer", li)]
            fst["step"] += 1
            fst["step_t"].fill_(fst["step"])
        return loss


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
    if on_gpu:
        # the path's own optimizer kernels.  Updates from inside the backward pass only when every backward is followed by a
        # step (gradient_accumulation_steps == 1: train.py:616-625 steps every `accum` backward passes)
        accum = int(getattr(args, "gradient_accumulation_steps", 1) or 1)
        overlap = bool(getattr(args, "overlap_optimizer", True)) and accum == 1
        opt = AdamW(groups, lr=args.lr, model=model, overlap=overlap, grad_sync=getattr(args, "grad_sync", None))
    else:
        opt = torch.optim.AdamW(groups, lr=args.lr)
    sched = linear_schedule_with_warmup(opt, getattr(args, "warmup_ratio", 0.01) * train_num_steps, train_num_steps)
    return opt, sched
