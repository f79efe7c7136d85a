#!/usr/bin/env python
"""Wall-clock forward / backward split of one bench step with allocator statistics (are device mallocs / frees
happening inside steady-state steps?).  usage: step_breakdown.py [batch] [seq]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = "cuda"
model, cfg = bench.build_model(dev, "bert", S)
model.train()
ids, mask, tt, labels, feats, aux = bench.synthetic_batch(B, S, 8, cfg.vocab_size, 0, dev)
opt = torch.optim.AdamW(model.parameters(), lr=3e-5, fused=True) if len(sys.argv) > 3 and sys.argv[3] == "opt" else None
for i in range(6):
    st0 = torch.cuda.memory_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = model(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, images=feats, aux_imgs=aux)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    out.loss.backward()
    t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    if opt is not None:
        opt.step()
        opt.zero_grad(set_to_none=True)
    else:
        for p in model.parameters():
            p.grad = None
    t5 = time.perf_counter(); torch.cuda.synchronize(); t6 = time.perf_counter()
    n = len(out.logits)
    st1 = torch.cuda.memory_stats()
    d = lambda k: st1.get(k, 0) - st0.get(k, 0)
    print(f"step {i}: fwd enqueue {1e3 * (t1 - t0):7.1f} ms, fwd done {1e3 * (t2 - t0):7.1f} | bwd enqueue {1e3 * (t3 - t2):7.1f}, "
          f"bwd done {1e3 * (t4 - t2):7.1f} | opt enqueue {1e3 * (t5 - t4):6.1f}, done {1e3 * (t6 - t4):6.1f} | device mallocs {d('num_device_alloc')}, frees {d('num_device_free')}, "
          f"retries {d('num_alloc_retries')}, reserved {st1['reserved_bytes.all.current'] / 2**30:.1f} GiB, "
          f"peak alloc {st1['allocated_bytes.all.peak'] / 2**30:.1f} GiB", flush=True)
