#!/usr/bin/env python
"""Which host-side ops launch the small torch kernels (fills, copies, adds) inside one training step?
torch.profiler with stacks around 3 steps of the bench workload; prints, per aten op, the python frames
(mtvaf_amd / bench) that called it.  Used to hunt glue launches; not part of the measured path."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = "cuda"
    model, cfg = bench.build_model(dev)
    model.train()
    ids, mask, tt, labels, feats, aux = bench.synthetic_batch(32, 128, 8, cfg.vocab_size, 0, dev)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-5, fused=True)

    def step():
        out = model(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, images=feats, aux_imgs=aux)
        out.loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return out

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    want = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::cat", "aten::sum",
            "aten::div", "aten::addcmul", "aten::to", "aten::_to_copy", "aten::ones_like", "aten::zeros")
    agg = collections.Counter()
    for ev in prof.events():
        if ev.name in want and ev.device_time_total > 0:
            frames = [f for f in (ev.stack or []) if "mtvaf_amd" in f or "bench.py" in f or "step_trace" in f or "optim" in f]
            agg[(ev.name, frames[0] if frames else "<autograd/other>")] += 1
    for (name, fr), n in sorted(agg.items(), key=lambda kv: -kv[1]):
        print(f"{n / 3:6.1f}/step  {name:18s} {fr}")


if __name__ == "__main__":
    main()
