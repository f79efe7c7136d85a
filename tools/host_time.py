"""Host time to ENQUEUE the phases of one training step of the bench workload (no synchronisation inside the loop) against
the GPU's time per step: is the step GPU-bound, and how far ahead does the host run?

    python tools/host_time.py [steps]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mtvaf_amd.optim import AdamW  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
model, cfg = bench.build_model(dev, "bert", 128)
model.train()
opt = AdamW([p for p in model.parameters() if p.requires_grad], lr=3e-5, weight_decay=1e-2, model=model, overlap=True)
ids, mask, tt, labels, feats, aux = bench.synthetic_batch(32, 128, 8, cfg.vocab_size, 1234, dev)
kw = dict(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, imagelabel=None, images=feats, aux_imgs=aux)
acc = [0.0] * 5


def step(timed):
    t = [time.perf_counter()]
    out = model(**kw); t.append(time.perf_counter())
    out.loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    opt.zero_grad(set_to_none=True); t.append(time.perf_counter())
    n = len(out.logits); t.append(time.perf_counter())  # waits for this step's Viterbi copy (the trainer reads the tags)
    assert n == 32
    if timed:
        for i in range(5):
            acc[i] += t[i + 1] - t[i]


for _ in range(5):
    step(False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step(True)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
names = ["forward", "backward", "optimizer.step", "zero_grad", "wait for the tags"]
print(f"{steps} steps: {1e3 * t_all / steps:.3f} ms per step on the GPU's clock; host loop {1e3 * t_host / steps:.3f} ms per step, of which")
for n, a in zip(names, acc):
    print(f"  {n:18s} {1e3 * a / steps:7.3f} ms")
print(f"  host work without the wait: {1e3 * sum(acc[:4]) / steps:.3f} ms")
