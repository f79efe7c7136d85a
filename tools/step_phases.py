"""Where a training step spends its time ON THE GPU's CLOCK, without a profiler in the process: HIP events recorded on the main
stream at the phase boundaries of the bench step (model hooks, tensor gradient hooks), averaged over the timed steps of a pipelined
run (no host sync inside a step other than the product path's own).

    python tools/step_phases.py [--batch 32 --seq 128 --dtype fp32 --model bert --steps 20]

phases: head of the forward pass (prompt generator, packing, embeddings) | encoder forward | classifier + CRF + start of backward()
(CRF / classifier gradients) | encoder backward | tail of backward() (embedding and prompt-generator gradients) | optimizer.step()
+ zero_grad.
"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--seq", type=int, default=128)
ap.add_argument("--dtype", default="fp32")
ap.add_argument("--model", default="bert")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--padded", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
from mtvaf_amd import engine, hip  # noqa: E402
from mtvaf_amd.optim import AdamW  # noqa: E402
hip.lib()
hip.set_compute_dtype(a.dtype)
engine.UNPAD = not a.padded
model, cfg = bench.build_model(dev, a.model, a.seq)
model.train()
opt = AdamW([p for p in model.parameters() if p.requires_grad], lr=3e-5, weight_decay=1e-2, model=model, overlap=True)
ids, mask, tt, labels, feats, aux = bench.synthetic_batch(a.batch, a.seq, 8, cfg.vocab_size, 1234, dev)

NAMES = ["step start", "encoder forward starts", "encoder forward done", "forward() returned", "encoder backward starts",
         "encoder backward done", "backward() returned", "step() + zero_grad returned"]
cur = {}


def mark(i):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    cur[i] = e


def first_grad_tensor(o):
    if torch.is_tensor(o):
        return o if o.requires_grad else None
    if isinstance(o, (tuple, list)):
        for x in o:
            t = first_grad_tensor(x)
            if t is not None:
                return t
    if hasattr(o, "values"):
        for x in o.values():
            t = first_grad_tensor(x)
            if t is not None:
                return t
    return None


enc = model.bert.encoder


def pre(m, args, kwargs):
    mark(1)
    t = first_grad_tensor(list(args) + list(kwargs.values()))
    if t is not None:
        t.register_hook(lambda g: mark(5))


def post(m, args, kwargs, out):
    mark(2)
    t = first_grad_tensor(out)
    if t is not None:
        t.register_hook(lambda g: mark(4))


enc.register_forward_pre_hook(pre, with_kwargs=True)
enc.register_forward_hook(post, with_kwargs=True)
rows = []
for i in range(a.steps + 5):
    mark(0)
    out = model(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, imagelabel=None, images=feats, aux_imgs=aux)
    mark(3)
    out.loss.backward()
    mark(6)
    opt.step()
    opt.zero_grad(set_to_none=True)
    mark(7)
    assert len(out.logits) == a.batch
    if i >= 5:
        rows.append(dict(cur))
    cur = {}
torch.cuda.synchronize()
tot = statistics.mean(r[0].elapsed_time(r[7]) for r in rows)
print(f"{a.model} {a.dtype} bs {a.batch} S {a.seq}{' padded' if a.padded else ''}: {tot:.3f} ms per step on the main stream "
      f"({1e3 * a.batch / tot:.0f} sentences/s), {len(rows)} steps")
for k in range(7):
    if all(k in r and k + 1 in r for r in rows):
        d = statistics.mean(r[k].elapsed_time(r[k + 1]) for r in rows)
        print(f"  {NAMES[k]:28s} -> {NAMES[k + 1]:30s} {d:7.3f} ms  {100 * d / tot:5.1f} %")
    else:
        print(f"  {NAMES[k]} -> {NAMES[k + 1]}: hook did not fire")
