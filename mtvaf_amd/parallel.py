"""Data-parallel gradient synchronisation: one process per GPU, RCCL collectives over xGMI.

The reference never wraps its model in DistributedDataParallel (MTVAF_training.py:305-309 initialises a
process group and a DistributedSampler and stops there), so this is new functionality whose semantic
oracle is "single-process gradients on the concatenated global batch": per-rank losses are batch means,
gradients are averaged (sum / world_size).

Design for the 8-GPU xGMI mesh: the encoder backward (``engine.EncoderFunction``) writes each layer's
parameter gradients into ONE contiguous fp32 buffer (28 MB for BERT-base) and reports it through
``GradSink.on_layer_done`` as soon as that layer's kernels are enqueued, layer 11 first.  This hook
launches the all-reduce of that buffer on a communication stream, so communication overlaps the remaining
backward; large parameters outside the encoder (94 MB word table, ``encoder_conv``) are reduced the moment
autograd has accumulated them; the small rest goes through ONE persistent flat bucket when autograd
finishes.  No gradient is copied on the fast path.

Two wire formats:

* fp32 (default in fp32 compute mode): RCCL ``all_reduce(AVG)`` of the flat buffer in place.
* bf16 (``compress="bf16"``, default in bf16 compute mode -- BASELINE configs 3-4): a direct exchange shaped for the
  fully connected xGMI mesh instead of a ring: every rank packs its fp32 gradients to bf16, ``all_to_all`` sends
  chunk r to rank r over the dedicated link, each rank sums the ``world`` chunks it received IN FP32 (rank order,
  deterministic), scales by 1/world, rounds ONCE to bf16 and ``all_gather``s the reduced chunk; the result is
  unpacked into the fp32 gradient buffer.  Half the bytes of the fp32 ring on every link, one rounding per phase
  instead of one per ring hop, and every rank ends with bit-identical gradients.  Pack / reduce / unpack are HIP
  kernels (``mtvaf_grad_pack_bf16`` / ``_reduce_bf16`` / ``_unpack_bf16`` in ``csrc/optim.hip``) on the communication stream.
  ``layer_buckets=k`` sends the encoder layers in k exchanges instead of one per layer (3 layers = 85 MB per exchange at
  k = 4 for BERT-base): a layer's packed gradients land in its slice of ONE send buffer and the all_to_all / sum /
  all_gather run once per bucket, when the bucket's last layer is done -- 2 k collectives per step where the per-layer form
  issues 24, so that an 8-GPU step is not bound by collective launch latency.

Dropout under data parallelism: ``engine.RNG`` derives its seed from ``torch.initial_seed()``; launchers that seed
every rank alike would make all ranks draw the same masks for the same (site, row).  GradSync therefore folds the
rank into the dropout seed (``engine.RNG.set_stream(rank)``) unless ``seed_per_rank=False``.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, model: torch.nn.Module, process_group=None, force: bool = False, big_numel: int = 1 << 20,
                 compress: Optional[str] = "auto", seed_per_rank: bool = True, layer_buckets: Optional[int] = None):
        if not dist.is_initialized():
            raise RuntimeError("GradSync needs an initialised process group (backend 'nccl' = RCCL on ROCm)")
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        # RCCL averages inside the collective; gloo (CPU tests, and the 2-rank-on-one-GPU test) has no AVG
        self._avg_op = dist.get_backend(process_group) == "nccl"
        if compress == "auto":
            from . import hip
            compress = "bf16" if hip.COMPUTE == "bf16" else None
        if compress not in (None, "bf16"):
            raise ValueError(f"compress must be None, 'bf16' or 'auto', got {compress!r}")
        self.compress = compress
        self.encoder = model.bert.encoder if hasattr(model, "bert") else model.encoder
        # bf16 wire: encoder layers per exchange.  Layers finish from the last to the first; bucket b = a run of consecutive
        # layers, exchanged when all of them have reported (any order).  None / fp32 wire: one collective per layer.
        L = len(self.encoder.layer)
        self.layer_buckets = None
        self._bucket_of, self._bucket_size, self._bucket_acc = {}, {}, {}
        if layer_buckets is not None and compress == "bf16":
            nb = max(1, min(int(layer_buckets), L))
            self.layer_buckets = nb
            for li in range(L):
                b = (L - 1 - li) * nb // L  # bucket 0 = the layers that finish first
                self._bucket_of[li] = b
                self._bucket_size[b] = self._bucket_size.get(b, 0) + 1
        sink = self.encoder.grad_sink
        sink.on_layer_done = self._layer_done
        # this hook records events and switches to the communication stream with torch's stream context: it must be
        # called under torch.cuda.stream(side), never through the raw-stream shortcut an optimizer attached EARLIER may
        # have asked for (engine._native_backward; otherwise the all-reduce could start before the layer's dW kernels end)
        sink.raw_stream_hook = False
        self._enc_param_ids = {id(p) for l in self.encoder.layer for p in l.ordered_params()}
        self._comm = torch.cuda.Stream() if torch.cuda.is_available() else None
        self._pending: List = []
        self._slow_layers: List[int] = []
        self._fast_layers: List[int] = []
        self._armed = False
        self.enabled = True
        self.force = force  # run the collectives even with world_size == 1 (single-GPU test of the N > 1 path)
        self._bufs = {}     # persistent staging: tail bucket, bf16 send / receive / shard
        self.after_layer_reduced = None  # callable(layer_index) run on the comm stream behind a layer's all-reduce
        self.before_layer = None         # callable(layer_index, flat_grad or None): the optimizer's contract check
        # timing = True: every collective is bracketed by events on the communication stream and the join at the end of the
        # backward pass by one event on each stream; take_timing() -> ms of communication-stream work and the part of it
        # the backward pass did not cover.  Off in timed regions (an event pair per collective is not free).
        self.timing = False
        self._tev: List = []
        self._ttail: List = []
        opt = getattr(sink, "optimizer", None)
        if opt is not None and getattr(opt, "overlap", False):  # AdamW(overlap=True) attached before this object existed
            self.adopt_optimizer(opt)
        # Large non-encoder gradients (94 MB word table, encoder_conv weights) are reduced the moment autograd has
        # accumulated them, so e.g. the word-table all-reduce overlaps the prompt generator's backward instead of
        # sitting in the un-overlapped tail bucket.
        self._early_done = set()
        for p in model.parameters():
            if id(p) not in self._enc_param_ids and p.requires_grad and p.numel() >= big_numel:
                p.register_post_accumulate_grad_hook(self._param_ready)
        if seed_per_rank:
            from . import engine
            engine.RNG.set_stream(self.rank)

    def adopt_optimizer(self, opt):
        """mtvaf_amd.optim.AdamW(overlap=True): its per-layer update runs on the communication stream right behind the
        layer's reduction (full width: the stream must not wait), whichever of the two objects was constructed first."""
        self.after_layer_reduced = opt._early_layer_update
        self.before_layer = opt.layer_pass_check
        opt._background_ok = False
        sink = self.encoder.grad_sink
        sink.on_layer_done = self._layer_done
        sink.raw_stream_hook = False

    def _timed(self, fn, *a):
        if not self.timing or self._comm is None:
            return fn(*a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a)
        e1.record()
        self._tev.append((e0, e1))
        return r

    def take_timing(self):
        """-> {"comm_stream_ms": time the communication stream spent in exchanges (+ the optimizer updates queued behind
        them), "exposed_tail_ms": time from the end of the backward pass's own kernels to the end of the last exchange},
        summed over the passes since the last call.  Synchronises the device."""
        if self._comm is None:
            return None
        torch.cuda.synchronize()
        comm = sum(a.elapsed_time(b) for a, b in self._tev)
        tail = sum(max(0.0, a.elapsed_time(b)) for a, b in self._ttail)
        n = len(self._ttail)
        self._tev, self._ttail = [], []
        return {"comm_stream_ms": comm, "exposed_tail_ms": tail, "passes": n}

    # -- wire formats ------------------------------------------------------------------------------------------
    def _buf(self, name: str, numel: int, dtype, device) -> torch.Tensor:
        b = self._bufs.get(name)
        if b is None or b.numel() < numel or b.device != device:
            b = self._bufs[name] = torch.zeros(numel, dtype=dtype, device=device)
        return b

    def _allreduce_mean(self, t: torch.Tensor):
        """In-place mean over the ranks of the flat fp32 tensor `t` (enqueued on the current stream)."""
        if self.compress == "bf16":
            return self._allreduce_mean_bf16(t)
        if self._avg_op:
            dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(t, group=self.group)
            t.mul_(1.0 / self.world)

    def _allreduce_mean_bf16(self, t: torch.Tensor):
        W, n = self.world, t.numel()
        chunk = ((n + W - 1) // W + 7) // 8 * 8  # 16-byte aligned chunks
        send = self._buf("send", W * chunk, torch.bfloat16, t.device)[:W * chunk]
        recv = self._buf("recv", W * chunk, torch.bfloat16, t.device)[:W * chunk]
        shard = self._buf("shard", chunk, torch.bfloat16, t.device)[:chunk]
        flat = t.view(-1)
        if t.is_cuda:
            from . import hip
            hip.grad_pack_bf16(flat, send, n, W * chunk)
        else:  # gloo / CPU: protocol test harness only
            send[:n].copy_(flat)
            send[n:].zero_()
        dist.all_to_all_single(recv, send, group=self.group)
        if t.is_cuda:
            hip.grad_reduce_bf16(recv, shard, W, chunk, 1.0 / W)
        else:
            shard.copy_(recv.view(W, chunk).float().sum(0).mul_(1.0 / W))
        dist.all_gather_into_tensor(send, shard, group=self.group)
        if t.is_cuda:
            hip.grad_unpack_bf16(send, flat, n)
        else:
            flat.copy_(send[:n])

    def _allreduce_mean_bf16_multi(self, tensors):
        """One exchange for several flat fp32 tensors (the layers of a bucket): tensor i is packed into its slice of the send
        buffer (slices start on 16-byte boundaries; the gaps and the tail are zeros), ONE all_to_all / fp32 sum / all_gather
        over the concatenation, every tensor unpacked from its slice.  Same arithmetic per element as the single-tensor form
        (the sum over ranks in rank order, one rounding), so bucketed and per-layer exchanges agree bit for bit."""
        if len(tensors) == 1:
            return self._allreduce_mean_bf16(tensors[0])
        W = self.world
        offs, off = [], 0
        for t in tensors:
            offs.append(off)
            off += (t.numel() + 7) // 8 * 8
        chunk = ((off + W - 1) // W + 7) // 8 * 8
        dev = tensors[0].device
        send = self._buf("send", W * chunk, torch.bfloat16, dev)[:W * chunk]
        recv = self._buf("recv", W * chunk, torch.bfloat16, dev)[:W * chunk]
        shard = self._buf("shard", chunk, torch.bfloat16, dev)[:chunk]
        ends = offs[1:] + [W * chunk]
        cuda = tensors[0].is_cuda
        if cuda:
            from . import hip
        for t, o, e in zip(tensors, offs, ends):
            n = t.numel()
            if cuda:
                hip.grad_pack_bf16(t.view(-1), send[o:], n, e - o)
            else:
                send[o:o + n].copy_(t.view(-1))
                send[o + n:e].zero_()
        dist.all_to_all_single(recv, send, group=self.group)
        if cuda:
            hip.grad_reduce_bf16(recv, shard, W, chunk, 1.0 / W)
        else:
            shard.copy_(recv.view(W, chunk).float().sum(0).mul_(1.0 / W))
        dist.all_gather_into_tensor(send, shard, group=self.group)
        for t, o in zip(tensors, offs):
            if cuda:
                hip.grad_unpack_bf16(send[o:], t.view(-1), t.numel())
            else:
                t.view(-1).copy_(send[o:o + t.numel()])

    def _bucket_ready(self, li: int, flat_grad: Optional[torch.Tensor]):
        """-> the (layer, flat gradient) list to exchange now, or None while the layer's bucket is still filling."""
        if self.layer_buckets is None:
            return [(li, flat_grad)]
        b = self._bucket_of[li]
        acc = self._bucket_acc.setdefault(b, [])
        acc.append((li, flat_grad))
        if len(acc) < self._bucket_size[b]:
            return None
        del self._bucket_acc[b]
        return acc

    # -- called from inside EncoderFunction.backward, newest layer first ---------------------------------
    def _layer_done(self, li: int, flat_grad: Optional[torch.Tensor]):
        if self.before_layer is not None:
            self.before_layer(li, flat_grad)
        if not self.enabled or (self.world == 1 and not self.force):
            return
        self._arm()
        ready = self._bucket_ready(li, flat_grad)
        if flat_grad is None:  # gradient-accumulation fallback: reduce the .grad tensors at the end
            self._slow_layers.append(li)
        else:
            self._fast_layers.append(li)
        if ready is None:
            return
        fast = [(l, g) for l, g in ready if g is not None]
        if not fast:
            return
        if self._comm is None:  # CPU / gloo (tests)
            if self.compress is None:
                for l, g in fast:
                    self._pending.append((dist.all_reduce(g, group=self.group, async_op=True), g))
            else:
                self._allreduce_mean_bf16_multi([g for _, g in fast])
            if self.after_layer_reduced is not None:
                for l, _ in fast:
                    self._pending.append((None, l))
            return
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self._comm):
            self._comm.wait_event(ev)
            if self.compress is None:
                for _, g in fast:
                    self._timed(self._allreduce_mean, g)
            else:
                self._timed(self._allreduce_mean_bf16_multi, [g for _, g in fast])
            if self.after_layer_reduced is not None:
                for l, _ in fast:
                    self.after_layer_reduced(l)

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
            if self.compress is None:
                self._pending.append((dist.all_reduce(g, group=self.group, async_op=True), g))
            else:
                self._allreduce_mean(g)
        else:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self._comm):
                self._comm.wait_event(ev)
                self._timed(self._allreduce_mean, g)
        self._early_done.add(id(p))

    # -- runs once when the autograd pass is complete ---------------------------------------------------------
    def _finish(self):
        self._armed = False
        if self._bucket_acc:  # a bucket that never filled (a layer did not report in this pass): exchange what it holds
            pairs = [(l, g) for acc in self._bucket_acc.values() for l, g in acc if g is not None]
            self._bucket_acc = {}
            if pairs:
                left = [g for _, g in pairs]
                if self._comm is None:
                    self._allreduce_mean_bf16_multi(left)
                    if self.after_layer_reduced is not None:
                        self._pending.extend((None, l) for l, _ in pairs)
                else:
                    ev = torch.cuda.Event()
                    ev.record()
                    with torch.cuda.stream(self._comm):
                        self._comm.wait_event(ev)
                        self._timed(self._allreduce_mean_bf16_multi, left)
                        if self.after_layer_reduced is not None:
                            for l, _ in pairs:
                                self.after_layer_reduced(l)
        rest = [p for p in self.model.parameters()
                if p.grad is not None and (id(p) not in self._enc_param_ids) and (id(p) not in self._early_done)]
        self._early_done = set()
        # A layer whose flat buffer was all-reduced but whose .grad tensors do NOT alias it (autograd cloned the returned
        # views instead of adopting them, e.g. because a tensor hook kept a reference): the clone ran on the main stream
        # while the ring was still writing partial sums into the flat buffer, so the .grad values may be torn.  They
        # are NOT reduced again; once the communication stream has been joined the (correctly reduced) flat buffer is
        # copied over them.  The fast path is only taken when every .grad was None and a single encoder node was in the
        # pass, so .grad holds nothing but that clone.
        stores = getattr(self.encoder, "_stores", None) or []
        copied = []
        for li in self._fast_layers:
            st = stores[li]
            lo, hi = st.grad.data_ptr(), st.grad.data_ptr() + st.grad.numel() * st.grad.element_size()
            ps = self.encoder.layer[li].ordered_params()
            if any(p.grad is not None and not (lo <= p.grad.data_ptr() < hi) for p in ps):
                copied.append(li)
        for li in self._slow_layers:
            rest.extend(p for p in self.encoder.layer[li].ordered_params() if p.grad is not None)
        self._slow_layers, self._fast_layers = [], []
        if self._comm is not None:
            if self.timing:
                t_main = torch.cuda.Event(enable_timing=True)
                t_main.record()  # the backward pass's own kernels end here
            self._comm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm):
                self._timed(self._reduce_bucket, rest)
                if self.timing:
                    t_comm = torch.cuda.Event(enable_timing=True)
                    t_comm.record()
                    self._ttail.append((t_main, t_comm))
            torch.cuda.current_stream().wait_stream(self._comm)
        else:
            hooks = []
            for work, buf in self._pending:
                if work is None:
                    hooks.append(buf)
                    continue
                work.wait()
                buf.mul_(1.0 / self.world)
            self._pending = []
            for li in hooks:
                self.after_layer_reduced(li)
            self._reduce_bucket(rest)
        for li in copied:
            views = stores[li].grad_views()
            with torch.no_grad():
                for p, v in zip(self.encoder.layer[li].ordered_params(), views):
                    if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                        p.grad.copy_(v)

    def _reduce_bucket(self, params):
        """The small non-encoder rest (position / type tables, LayerNorms, fc, crf, projectors ...: ~3 MB) through one
        PERSISTENT flat buffer: a gather (torch.cat into the buffer), the collective, a multi-tensor scatter back."""
        if not params:
            return
        grads = [p.grad for p in params]
        n = sum(g.numel() for g in grads)
        flat = self._buf("tail", n, torch.float32, grads[0].device)[:n]
        torch.cat([g.reshape(-1) for g in grads], out=flat)
        self._allreduce_mean(flat)
        views, off = [], 0
        for g in grads:
            views.append(flat[off:off + g.numel()].view(g.shape))
            off += g.numel()
        torch._foreach_copy_(grads, views)
