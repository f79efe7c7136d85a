"""Whole-step HIP graph for the launch-bound configurations.

At small batches (BASELINE configs[0]: bs 4, S 64 -- ~0.13 TFLOP per step) and in the bf16 mode at bs 32 the step is
bound by the HOST: ~450 kernel launches, each behind a Python-level call, while the GPU needs a fraction of that time.
``GraphedTrainStep`` captures ``model(**batch)`` + ``loss.backward()`` ONCE into a HIP graph (all kernels of the path
launch on torch's current stream, so they are captured like torch's own; the engine's second stream joins the capture
through its events) and replays it per step: one launch from the host.

What makes the path capturable:
  * dropout -- kernel arguments are frozen by a capture, so host-fed (seed, offset) pairs would repeat the same masks.
    The library folds a DEVICE-side epoch word into every mask (``mtvaf_rng_set_epoch_ptr``); the first node of the graph
    bumps it, every replay draws fresh masks, the forward and backward kernels of a replay read the same value;
  * decoded tags -- the Viterbi kernel and its packed device->host copy are graph nodes writing a pinned buffer owned by
    this object; each call hands out a fresh ``DeferredTags`` over it (read it before the next call);
  * bf16 weight images -- rebuilt by a cast inside the graph (one kernel per layer), so ANY optimizer may update the fp32
    masters between replays;
  * gradients live at fixed addresses (the encoder's flat per-layer buffers, the graph's private pool for the rest);
    ``param.grad`` is re-attached after every replay, so ``optimizer.zero_grad(set_to_none=True)`` stays legal.

The optimizer step stays eager (its learning rate changes per step): use ``mtvaf_amd.optim.AdamW(..., overlap=False)``
or any torch optimizer.  Shapes are frozen: one object per (batch, seq_len) -- the reference's last, smaller batch of an
epoch (modules/train.py:596-650) runs through the plain eager path.

    step = GraphedTrainStep(model, batch)      # batch: dict of forward() keyword tensors on the device
    out = step(**batch2)                        # same shapes; TokenClassifierOutput(loss, logits) as model(**batch2)
    optimizer.step(); optimizer.zero_grad(set_to_none=True)
"""
from __future__ import annotations

from typing import Dict

import torch
from transformers.modeling_outputs import TokenClassifierOutput

from . import engine, hip
from .modules.crf import DeferredTags


class GraphedTrainStep:
    def __init__(self, model: torch.nn.Module, batch: Dict[str, torch.Tensor], warmup: int = 3):
        tens = {k: v for k, v in batch.items() if torch.is_tensor(v)}
        if not tens or not all(v.is_cuda for v in tens.values()):
            raise RuntimeError("GraphedTrainStep needs the batch on the MI355X (no CPU fallback)")
        if "labels" not in tens:
            raise ValueError("a training step needs labels (loss.backward() is part of the graph)")
        # padding-free execution (engine.UNPAD) reads the packed row count on the host once per step, which a capture cannot
        # contain: it steps aside -- the warm-up passes and the captured step run the padded layout (same loss, tags and
        # parameter gradients; tests/test_unpad_gpu.py), the switch is restored for eager steps afterwards
        unpad_was, engine.UNPAD = engine.UNPAD, False
        try:
            self._build(model, batch, tens, warmup)
        finally:
            engine.UNPAD = unpad_was

    def _build(self, model, batch, tens, warmup):
        enc = getattr(getattr(model, "bert", model), "encoder", None)
        sink = getattr(enc, "_sink", None)
        if sink is not None and sink.on_layer_done is not None:
            # a backward-pass hook is attached: mtvaf_amd.optim.AdamW(overlap=True) would apply real updates during the
            # warm-up passes below (and raise on the second), GradSync would enqueue collectives into the capture
            raise RuntimeError("GraphedTrainStep: the encoder has a backward hook attached (AdamW(overlap=True) or GradSync); "
                               "build the optimizer with overlap=False and capture single-GPU steps only")
        self.model = model
        self.static = {k: v.clone() for k, v in tens.items()}
        self.const = {k: v for k, v in batch.items() if not torch.is_tensor(v)}
        dev = next(iter(tens.values())).device
        self.epoch = torch.zeros(1, dtype=torch.int64, device=dev)
        hip._ck(hip.lib().mtvaf_rng_set_epoch_ptr(self.epoch.data_ptr()), "mtvaf_rng_set_epoch_ptr")
        crf = getattr(model, "crf", None)
        B, S = self.static["input_ids"].shape
        if crf is not None:  # the staging buffer of the decoded tags: first in the pool, never waited for during capture
            self._tag_buf = torch.empty(max(B * (S + 1), 1 << 14), dtype=torch.int32, pin_memory=True)
        params = [p for p in model.parameters() if p.requires_grad]

        def one_step():
            if crf is not None:
                crf._host_pool.insert(0, (self._tag_buf, None))
            out = model(**self.static, **self.const)
            out.loss.backward()
            return out

        # warm-up on a side stream (allocator, lazily built buffers, kernel attributes), as torch's graph recipe does
        cur = torch.cuda.current_stream()
        s = torch.cuda.Stream()
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            for _ in range(max(1, warmup)):
                for p in params:
                    p.grad = None
                out = one_step()
                if isinstance(out.logits, DeferredTags):
                    out.logits._release = None  # the buffer is ours, not the pool's
                del out
        cur.wait_stream(s)
        torch.cuda.synchronize()
        if crf is not None:
            crf._host_pool[:] = [e for e in crf._host_pool if e[0] is not self._tag_buf]
        for p in params:
            p.grad = None
        self.graph = torch.cuda.CUDAGraph()
        engine.FORCE_SHADOW_REFRESH = True
        try:
            with torch.cuda.graph(self.graph):
                hip._ck(hip.lib().mtvaf_rng_epoch_advance(self.epoch.data_ptr(), hip._st()), "mtvaf_rng_epoch_advance")
                out = one_step()
        finally:
            engine.FORCE_SHADOW_REFRESH = False
        self.loss = out.loss.detach()
        self._S = S
        self._tags_host = None
        if isinstance(out.logits, DeferredTags):
            self._tags_host = out.logits._packed
            out.logits._packed, out.logits._release = None, None  # never filled from the capture-time event
        self.grads = [(p, p.grad) for p in params if p.grad is not None]
        if crf is not None:
            crf._host_pool[:] = [e for e in crf._host_pool if e[0] is not self._tag_buf]

    def set_epoch(self, value: int):
        """Dropout epoch of the NEXT call is value + 1 (tests: reproduce a replay's masks)."""
        self.epoch.fill_(int(value))

    def __call__(self, **batch) -> TokenClassifierOutput:
        if self.epoch is None:
            raise RuntimeError("GraphedTrainStep was closed")
        for k, v in batch.items():
            if torch.is_tensor(v):
                dst = self.static.get(k)
                if dst is None or dst.shape != v.shape:
                    raise ValueError(f"GraphedTrainStep was captured for {k} of shape "
                                     f"{None if dst is None else tuple(dst.shape)}, got {tuple(v.shape)}: use the eager model")
                dst.copy_(v, non_blocking=True)
        self.graph.replay()
        for p, g in self.grads:
            p.grad = g
        logits = None
        if self._tags_host is not None:
            ev = torch.cuda.Event()
            ev.record()
            logits = DeferredTags(self._tags_host, ev, self._S)
        return TokenClassifierOutput(loss=self.loss, logits=logits)

    def close(self):
        """Unregister the device-side dropout epoch (the word is owned by this object: the library must not keep its
        address once the object dies)."""
        if getattr(self, "epoch", None) is not None:
            hip._ck(hip.lib().mtvaf_rng_set_epoch_ptr(None), "mtvaf_rng_set_epoch_ptr")
            self.epoch = None
            # stream-K scratches attached during the capture live in the graph's memory pool: the library must not keep
            # their addresses either (eager streams re-attach theirs on the next mixed-precision backward)
            hip.streamk_detach_all()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
