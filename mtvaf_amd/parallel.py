"""Data-parallel gradient synchronisation: one process per GPU, RCCL all-reduce over xGMI.

The reference never wraps its model in DistributedDataParallel (MTVAF_training.py:305-309 initialises a
process group and a DistributedSampler and stops there), so this is new functionality whose semantic
oracle is "single-process gradients on the concatenated global batch": per-rank losses are batch means,
gradients are averaged (sum / world_size).

Design for the 8-GPU xGMI mesh: the encoder backward (``engine.EncoderFunction``) writes each layer's
parameter gradients into ONE contiguous fp32 buffer (28 MB for BERT-base) and reports it through
``GradSink.on_layer_done`` as soon as that layer's kernels are enqueued, layer 11 first.  This hook
launches the all-reduce of that buffer on a side stream, so communication overlaps the remaining
backward; everything else (embeddings, head, prompt generator: ~125 MB, dominated by the 94 MB word
table) is reduced in one flat bucket when autograd finishes.  No gradient is copied on the fast path.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, model: torch.nn.Module, process_group=None, force: bool = False, big_numel: int = 1 << 20):
        if not dist.is_initialized():
            raise RuntimeError("GradSync needs an initialised process group (backend 'nccl' = RCCL on ROCm)")
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        # RCCL averages inside the collective; gloo (CPU tests, and the 2-rank-on-one-GPU test) has no AVG
        self._avg_op = dist.get_backend(process_group) == "nccl"
        self.encoder = model.bert.encoder if hasattr(model, "bert") else model.encoder
        self.encoder.grad_sink.on_layer_done = self._layer_done
        self._enc_param_ids = {id(p) for l in self.encoder.layer for p in l.ordered_params()}
        self._comm = torch.cuda.Stream() if torch.cuda.is_available() else None
        self._pending: List = []
        self._slow_layers: List[int] = []
        self._armed = False
        self.enabled = True
        self.force = force  # run the collectives even with world_size == 1 (single-GPU test of the N > 1 path)
        # Large non-encoder gradients (94 MB word table, encoder_conv weights) are reduced the moment autograd has
        # accumulated them, so e.g. the word-table all-reduce overlaps the prompt generator's backward instead of
        # sitting in the un-overlapped tail bucket.
        self._early_done = set()
        for p in model.parameters():
            if id(p) not in self._enc_param_ids and p.requires_grad and p.numel() >= big_numel:
                p.register_post_accumulate_grad_hook(self._param_ready)

    # -- called from inside EncoderFunction.backward, newest layer first ---------------------------------
    def _layer_done(self, li: int, flat_grad: Optional[torch.Tensor]):
        if not self.enabled or (self.world == 1 and not self.force):
            return
        self._arm()
        if flat_grad is None:  # gradient-accumulation fallback: reduce the .grad tensors at the end
            self._slow_layers.append(li)
            return
        if self._comm is None:  # CPU / gloo (tests)
            self._pending.append((dist.all_reduce(flat_grad, group=self.group, async_op=True), flat_grad))
            return
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self._comm):
            self._comm.wait_event(ev)
            self._allreduce_mean(flat_grad)

    def _allreduce_mean(self, t: torch.Tensor):
        if self._avg_op:
            dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(t, group=self.group)
            t.mul_(1.0 / self.world)

    def _arm(self):
        if not self._armed:
            self._armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finish)

    def _param_ready(self, p):
        if not self.enabled or (self.world == 1 and not self.force) or p.grad is None:
            return
        self._arm()
        g = p.grad
        if self._comm is None:
            self._pending.append((dist.all_reduce(g, group=self.group, async_op=True), g))
        else:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self._comm):
                self._comm.wait_event(ev)
                self._allreduce_mean(g)
        self._early_done.add(id(p))

    # -- runs once when the autograd pass is complete ---------------------------------------------------------
    def _finish(self):
        self._armed = False
        rest = [p for p in self.model.parameters()
                if p.grad is not None and (id(p) not in self._enc_param_ids) and (id(p) not in self._early_done)]
        self._early_done = set()
        # a layer whose gradients autograd copied instead of adopting (its .grad does not alias the flat
        # buffer that was reduced) is reduced again from its .grad tensors -- correctness never depends on
        # the zero-copy fast path
        stores = getattr(self.encoder, "_stores", None) or []
        for li, st in enumerate(stores):
            if li in self._slow_layers or st.grad is None:
                continue
            g = self.encoder.layer[li].intermediate.dense.weight.grad
            lo = st.grad.data_ptr()
            if g is not None and not (lo <= g.data_ptr() < lo + st.grad.numel() * 4):
                self._slow_layers.append(li)
        for li in self._slow_layers:
            rest.extend(p for p in self.encoder.layer[li].ordered_params() if p.grad is not None)
        self._slow_layers = []
        if self._comm is not None:
            self._comm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm):
                self._reduce_bucket(rest)
            torch.cuda.current_stream().wait_stream(self._comm)
        else:
            for work, buf in self._pending:
                work.wait()
                buf.mul_(1.0 / self.world)
            self._pending = []
            self._reduce_bucket(rest)

    def _reduce_bucket(self, params):
        if not params:
            return
        grads = [p.grad for p in params]
        flat = torch._utils._flatten_dense_tensors(grads)
        self._allreduce_mean(flat)
        for g, s in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
            g.copy_(s)
