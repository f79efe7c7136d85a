"""Which host lines issue the device-to-device copies (and other small torch kernels) of one training step at the bench
workload: torch.profiler with Python stacks, grouped by the innermost frame inside this repository.

    python tools/find_copies.py [fp32|bf16] [bert|roberta] [batch] [seq] [aux]            # -> stdout
"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mtvaf_amd.optim import AdamW  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
arch = sys.argv[2] if len(sys.argv) > 2 else "bert"
BS = int(sys.argv[3]) if len(sys.argv) > 3 else 32
SEQ = int(sys.argv[4]) if len(sys.argv) > 4 else 128
AUX = int(sys.argv[5]) if len(sys.argv) > 5 else 8
from mtvaf_amd import hip  # noqa: E402
hip.set_compute_dtype(dtype)
dev = torch.device("cuda:0")
model, cfg = bench.build_model(dev, arch, SEQ)
model.train()
opt = AdamW([p for p in model.parameters() if p.requires_grad], lr=3e-5, weight_decay=1e-2, model=model, overlap=True)
ids, mask, tt, labels, feats, aux = bench.synthetic_batch(BS, SEQ, AUX, cfg.vocab_size, 1234, dev)
kw = dict(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, imagelabel=None, images=feats, aux_imgs=aux)


def step():
    out = model(**kw)
    out.loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    return out


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
by = collections.Counter()
dur = collections.Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
        continue
    if ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue  # nested op
    if ev.device_time_total <= 0:
        continue  # (views, allocations: no device work)
    where = "?"
    for fr in ev.stack or []:
        if root in fr and "tools/find_copies" not in fr:
            where = fr.replace(root + "/", "")
            break
    by[(ev.name, where)] += 1
    dur[(ev.name, where)] += ev.device_time_total
print(f"== {dtype} {arch} bs {BS} S {SEQ}: top-level aten ops with device work in ONE training step, by the innermost repository frame ==")
print(f"   {sum(by.values())} ops, {sum(dur.values()):.1f} us of device time")
for (name, where), n in by.most_common(80):
    print(f"{n:4d} x {name:16s} {dur[(name, where)]:8.1f} us  {where}")
