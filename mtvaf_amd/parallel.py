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

Collective order (round 5).  Collectives of one process group match by ISSUE ORDER, so every rank must issue the same
sequence whatever its own autograd pass looked like.  The encoder layers report in a fixed order; everything else follows a
PLAN: the list of large parameters reduced on their own (in the order their gradients became ready on rank 0 in the first
pass) and the list of small parameters of the tail bucket, both agreed on by all ranks at the end of the first pass (one
all_reduce(MAX) of a presence mask, one broadcast of rank 0's order).  A planned parameter whose gradient is None on THIS rank
in some later pass (a rank whose batch has no images: no gradient for ``encoder_conv``) is exchanged as zeros -- its rank
contributes nothing to the mean -- and a large parameter is only issued once every planned parameter before it has been, the
rest being flushed in plan order when the pass ends: no rank ever waits in a collective the others do not issue.  A gradient
that appears for a parameter OUTSIDE the plan raises on the rank that sees it before anything is issued for it.
``check_every=N`` (or MTVAF_CHECK_COLLECTIVES=N) hashes the (kind, numel) sequence of each pass and compares it across the
ranks every N passes; a mismatch raises on every rank.

Dropout under data parallelism: ``engine.RNG`` derives its seed from ``torch.initial_seed()``; launchers that seed
every rank alike would make all ranks draw the same masks for the same (site, row).  GradSync therefore folds the
rank into the dropout seed (``engine.RNG.set_stream(rank)``) unless ``seed_per_rank=False``.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


def balanced_shards(lengths, world: int):
    """Deal the sentences of a global batch to `world` ranks so that every rank gets the same NUMBER of sentences (the remainder
    of len(lengths) / world is dropped from the short end, as DistributedSampler(drop_last=True) does) and nearly the same number
    of TOKEN ROWS: sorted by length (longest first, ties by index: deterministic on every rank without communication) and dealt in
    snake order (0 .. W-1, W-1 .. 0, ...).  -> list of `world` index lists.

    Why: under padding-free execution (engine.UNPAD, the default) a rank's step time follows its packed row count, not B x S, so
    per-rank batches drawn independently differ by ~10 % in work (ragged U{16..128} at bs 32: 2377 +- 250 rows) and the slowest
    rank sets the step -- SURVEY 8e assumed "identical per-rank work (fixed S padding)".  The reference shards with a plain
    DistributedSampler (MTVAF_training.py:328-331); this is the length-aware replacement for it."""
    n = (len(lengths) // world) * world
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))[:n]
    shards = [[] for _ in range(world)]
    for j, i in enumerate(order):
        r, c = divmod(j, world)
        shards[c if r % 2 == 0 else world - 1 - c].append(i)
    return shards


class GradSync:
    def __init__(self, model: torch.nn.Module, process_group=None, force: bool = False, big_numel: int = 1 << 20,
                 compress: Optional[str] = "auto", seed_per_rank: bool = True, layer_buckets: Optional[int] = None,
                 check_every: Optional[int] = None):
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
        if layer_buckets is not None and compress is None and int(layer_buckets) < L:
            import warnings
            warnings.warn(f"GradSync: layer_buckets={layer_buckets} applies to the bf16 wire only; the fp32 wire (all_reduce in "
                          f"place, no staging buffer) exchanges one buffer per layer: {L} collectives per pass", stacklevel=2)
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
        self._outer = [p for p in model.parameters() if id(p) not in self._enc_param_ids and p.requires_grad]
        self._outer_idx = {id(p): i for i, p in enumerate(self._outer)}
        self._big_numel = big_numel
        # the plan (module docstring): None until the first synchronised pass has ended
        self._plan: Optional[List] = None   # schedule keys ("L", bucket or layer) / ("P", index into _outer), in issue order
        self._plan_pos = {}
        self._plan_small: List[int] = []    # indices into _outer: the tail bucket's members, in parameter order
        self._next = 0                      # position in _plan of the next item to issue in this pass
        self._ready = {}                    # key -> issue closure of items that wait for their predecessors (or for the plan)
        self._first_order: List = []        # first pass: the order in which items became ready on this rank
        self._issued_first = set()
        if check_every is None:
            import os
            check_every = int(os.environ.get("MTVAF_CHECK_COLLECTIVES", "0")) or None
        self.check_every = check_every
        self._seq: List = []        # (kind, numel) of every collective issued in this pass
        self._passes = 0
        for p in self._outer:
            if p.numel() >= big_numel:
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
        self._seq.append(("ar", t.numel()))
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
            self._seq.append(("ar", tensors[0].numel()))
            return self._allreduce_mean_bf16(tensors[0])
        self._seq.append(("a2a", sum(t.numel() for t in tensors)))
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

    # -- the ordered schedule (module docstring: "Collective order") ------------------------------------------------
    def _submit(self, key, issue):
        """`issue()` enqueues the collective(s) of schedule item `key` -- ("L", bucket or layer) or ("P", index into _outer).
        First pass: layer exchanges go out at once (every rank's encoder reports its layers in the same order), large
        parameters wait for the plan (_finish); the order in which items became ready is noted for the plan.  Later passes:
        an item goes out once every planned item before it has; the rest is flushed, in plan order, when the pass ends."""
        if self._plan is None:
            self._first_order.append(key)
            if key[0] == "L":
                issue()
            else:
                self._ready[key] = issue
            return
        if key not in self._plan_pos:
            if key[0] == "P" and key[1] in self._plan_small:
                return  # (planned into the tail bucket: reduced there)
            what = (f"parameter #{key[1]} of shape {tuple(self._outer[key[1]].shape)}" if key[0] == "P" else f"encoder exchange {key[1]}")
            raise RuntimeError(f"GradSync: {what} received a gradient on rank {self.rank} but had none on any rank when the "
                               "exchange plan was made (first pass); call GradSync.replan() on every rank when the set of "
                               "trained parameters changes")
        self._ready[key] = issue
        self._drain()

    def _drain(self):
        while self._next < len(self._plan) and self._plan[self._next] in self._ready:
            self._ready.pop(self._plan[self._next])()
            self._next += 1

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
        key = ("L", self._bucket_of[li] if self.layer_buckets is not None else li)
        self._submit(key, self._layer_issue(fast))

    def _layer_issue(self, fast):
        """-> closure that enqueues the exchange of the (layer, flat gradient) pairs `fast`, ordered behind everything the
        CURRENT stream holds now (the event is recorded here, at hook time, also when the closure runs later)."""
        if self._comm is None:  # CPU / gloo (tests)
            def issue():
                if self.compress is None:
                    for l, g in fast:
                        self._seq.append(("ar", g.numel()))
                        self._pending.append((dist.all_reduce(g, group=self.group, async_op=True), g))
                else:
                    self._allreduce_mean_bf16_multi([g for _, g in fast])
                if self.after_layer_reduced is not None:
                    for l, _ in fast:
                        self._pending.append((None, l))
            return issue
        ev = torch.cuda.Event()
        ev.record()

        def issue():
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
        return issue

    def _arm(self):
        if not self._armed:
            self._armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finish)

    def _param_issue(self, p, g):
        """-> closure that enqueues the collective of one large parameter's gradient `g` (event recorded now, as above)."""
        if self._comm is None:
            def issue():
                if self.compress is None:
                    self._seq.append(("ar", g.numel()))
                    self._pending.append((dist.all_reduce(g, group=self.group, async_op=True), g))
                else:
                    self._allreduce_mean(g)
            return issue
        ev = torch.cuda.Event()
        ev.record()

        def issue():
            with torch.cuda.stream(self._comm):
                self._comm.wait_event(ev)
                self._timed(self._allreduce_mean, g)
        return issue

    def _param_ready(self, p):
        if not self.enabled or (self.world == 1 and not self.force) or p.grad is None:
            return
        self._arm()
        self._submit(("P", self._outer_idx[id(p)]), self._param_issue(p, p.grad))

    def replan(self):
        """Forget the plan: the next pass agrees on a new one (call on EVERY rank, e.g. after switching heads on or off)."""
        self._plan = None
        self._plan_pos, self._plan_small, self._first_order = {}, [], []

    def _make_plan(self):
        """End of the first pass: the ranks agree on which parameters outside the encoder take part (the union of those with a
        gradient on any rank) and on the order of the schedule (rank 0's ready order of layer exchanges and large parameters;
        large parameters rank 0 did not see go last, by index)."""
        n = len(self._outer)
        cdev = self._outer[0].device if (n and self._comm is not None) else torch.device("cpu")
        have = torch.tensor([0 if p.grad is None else 1 for p in self._outer] + [0], dtype=torch.int32, device=cdev)
        dist.all_reduce(have, op=dist.ReduceOp.MAX, group=self.group)
        L = len(self.encoder.layer)
        order = torch.full((n + L + 1,), -1, dtype=torch.int32, device=cdev)  # codes: layer exchange b -> b, parameter i -> L + i
        if self.rank == 0 and self._first_order:
            codes = [k[1] if k[0] == "L" else L + k[1] for k in self._first_order]
            order[:len(codes)] = torch.tensor(codes, dtype=torch.int32, device=cdev)
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        dist.broadcast(order, src=src, group=self.group)
        have_l = have.tolist()[:n]
        big = [i for i in range(n) if have_l[i] and self._outer[i].numel() >= self._big_numel]
        plan = []
        for c in order.tolist():
            if c < 0:
                continue
            key = ("L", c) if c < L else ("P", c - L)
            if key[0] == "L" or key[1] in big:
                plan.append(key)
        plan += [("P", i) for i in big if ("P", i) not in plan]
        self._plan = plan
        self._plan_pos = {k: j for j, k in enumerate(plan)}
        self._plan_small = [i for i in range(n) if have_l[i] and self._outer[i].numel() < self._big_numel]
        self._first_order = []
        # what this first pass has issued already: its layer exchanges (they are not replayed by the flush below)
        self._issued_first = {k for k in plan if k[0] == "L"}

    def _check_sequence(self):
        """Debug aid: every rank must have issued the same (kind, numel) sequence in this pass."""
        import hashlib
        h = int.from_bytes(hashlib.sha256(repr(self._seq).encode()).digest()[:7], "little")
        cdev = self._outer[0].device if (self._comm is not None and self._outer) else torch.device("cpu")
        mine = torch.tensor([h, len(self._seq)], dtype=torch.int64, device=cdev)
        allh = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(allh, mine, group=self.group)
        vals = [tuple(t.tolist()) for t in allh]
        if len(set(vals)) != 1:
            raise RuntimeError(f"GradSync: the ranks issued different collective sequences in pass {self._passes}: "
                               f"(hash, count) per rank = {vals}; this rank's sequence = {self._seq}")

    # -- runs once when the autograd pass is complete ---------------------------------------------------------
    def _finish(self):
        self._armed = False
        if self._bucket_acc:
            # a bucket that never filled (a layer did not report in this pass): what it holds goes THROUGH THE SCHEDULE like a
            # full bucket -- issued directly here it could overtake planned items that still wait for a predecessor on this
            # rank but not on another (a large parameter whose gradient is None here), and the ranks' collective sequences
            # would differ
            left = sorted(self._bucket_acc.items())
            self._bucket_acc = {}
            for b, acc in left:
                pairs = [(l, g) for l, g in acc if g is not None]
                if pairs:
                    self._submit(("L", b), self._layer_issue(pairs))
        first = self._plan is None
        if first:
            self._make_plan()
        # flush, in plan order, what this pass has not issued yet: items that waited for a predecessor, and large parameters whose
        # gradient is None on THIS rank (exchanged as zeros: this rank adds nothing to the mean).  A layer exchange that did not
        # report in this pass (every rank alike: gradient accumulation sends those layers through the tail) is passed over.
        flush = []
        zero_filled: List[int] = []  # planned parameters whose gradient is None on THIS rank in this pass (indices into _outer)
        for key in self._plan[self._next:]:
            if first and key in self._issued_first:
                continue
            fn = self._ready.pop(key, None)
            if fn is None and key[0] == "P":
                p = self._outer[key[1]]
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                    zero_filled.append(key[1])
                fn = self._param_issue(p, p.grad)
            if fn is not None:
                flush.append(fn)
        self._next, self._ready, self._issued_first = 0, {}, set()
        for i in self._plan_small:
            p = self._outer[i]
            if p.grad is None:
                p.grad = torch.zeros_like(p)
                zero_filled.append(i)
        planned = set(self._plan_small) | {k[1] for k in self._plan if k[0] == "P"}
        # presence flags ride in the tail bucket: one float per planned parameter, 1 where this rank had a real gradient.  Their
        # mean is > 0 iff SOME rank had one; a parameter nobody had a gradient for gets `.grad = None` back (below), so that the
        # optimizer skips it as it would in a single-process run instead of applying weight decay and momentum to a zero gradient.
        planned_order = sorted(planned)
        zf = set(zero_filled)
        flags = [0.0 if i in zf else 1.0 for i in planned_order]
        unplanned = [i for i, p in enumerate(self._outer) if p.grad is not None and i not in planned]
        if unplanned:
            raise RuntimeError(f"GradSync: parameters {unplanned} (indices outside the encoder) received a gradient on rank {self.rank} "
                               "but are not part of the exchange plan made in the first pass; call GradSync.replan() on every rank")
        rest = [self._outer[i] for i in self._plan_small]
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
        flags_out = None
        if self._comm is not None:
            if self.timing:
                t_main = torch.cuda.Event(enable_timing=True)
                t_main.record()  # the backward pass's own kernels end here
            self._comm.wait_stream(torch.cuda.current_stream())
            for fn in flush:
                fn()
            with torch.cuda.stream(self._comm):
                flags_out = self._timed(self._reduce_bucket, rest, flags)
                if self.timing:
                    t_comm = torch.cuda.Event(enable_timing=True)
                    t_comm.record()
                    self._ttail.append((t_main, t_comm))
            torch.cuda.current_stream().wait_stream(self._comm)
        else:
            for fn in flush:
                fn()
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
            flags_out = self._reduce_bucket(rest, flags)
        if zero_filled and flags_out is not None:
            # (only a rank that filled zeros in has anything to undo; the read synchronises with the tail exchange)
            seen = flags_out.tolist()
            for j, i in enumerate(planned_order):
                if i in zf and seen[j] <= 0.0:
                    self._outer[i].grad = None
        for li in copied:
            views = stores[li].grad_views()
            with torch.no_grad():
                for p, v in zip(self.encoder.layer[li].ordered_params(), views):
                    if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                        p.grad.copy_(v)
        self._passes += 1
        seq_check = self.check_every and self._passes % self.check_every == 0
        if seq_check:
            self._check_sequence()
        self._seq = []

    def _reduce_bucket(self, params, flags=None):
        """The small non-encoder rest (position / type tables, LayerNorms, fc, crf, projectors ...: ~3 MB) through one
        PERSISTENT flat buffer: a gather (torch.cat into the buffer), the collective, a multi-tensor scatter back.  `flags`
        (one float per planned parameter, the same count on every rank) travel behind the gradients; -> their means."""
        nf = len(flags) if flags else 0
        if not params and not nf:
            return None
        grads = [p.grad for p in params]
        n = sum(g.numel() for g in grads)
        dev = grads[0].device if grads else (self._outer[0].device if self._comm is not None else torch.device("cpu"))
        flat = self._buf("tail", n + nf, torch.float32, dev)[:n + nf]
        if grads:
            torch.cat([g.reshape(-1) for g in grads], out=flat[:n])
        if nf:
            flat[n:].copy_(torch.tensor(flags, dtype=torch.float32), non_blocking=True)
        self._allreduce_mean(flat)
        views, off = [], 0
        for g in grads:
            views.append(flat[off:off + g.numel()].view(g.shape))
            off += g.numel()
        if grads:
            torch._foreach_copy_(grads, views)
        return flat[n:] if nf else None
