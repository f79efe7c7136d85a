"""Where does the HOST spend its time enqueueing one default training step (bs 32 / S 128, padding-free, pre-split path)?
Wall-clock segments without any synchronisation of our own (forward incl. the packing wait, backward, optimizer), then a cProfile
of 20 steps.

    python tools/host_time_probe.py [steps] [fp32|bf16] [bert|roberta] [batch] [seq]
"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mtvaf_amd.optim import AdamW  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp32"      # fp32 | bf16
arch = sys.argv[3] if len(sys.argv) > 3 else "bert"        # bert | roberta
BS = int(sys.argv[4]) if len(sys.argv) > 4 else 32
SEQ = int(sys.argv[5]) if len(sys.argv) > 5 else 128
from mtvaf_amd import hip  # noqa: E402
hip.set_compute_dtype(dtype)
dev = torch.device("cuda:0")
torch.manual_seed(1)
model, cfg = bench.build_model(dev, arch, SEQ)
model.train()
opt = AdamW([p for p in model.parameters() if p.requires_grad], lr=3e-5, weight_decay=1e-2, model=model, overlap=True)
ids, mask, tt, labels, feats, aux = bench.synthetic_batch(BS, SEQ, 8, cfg.vocab_size, 1234, dev)
kw = dict(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, imagelabel=None, images=feats, aux_imgs=aux)
seg = [0.0, 0.0, 0.0, 0.0]


def step(timed=False):
    t0 = time.perf_counter()
    out = model(**kw)
    t1 = time.perf_counter()
    out.loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    t4 = time.perf_counter()
    if timed:
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            seg[i] += d


for _ in range(8):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step(True)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
print(f"step {ms:.3f} ms; host: forward (incl. the wait for the packed row count) {seg[0] / steps * 1e3:.3f}, backward "
      f"{seg[1] / steps * 1e3:.3f}, optimizer.step {seg[2] / steps * 1e3:.3f}, zero_grad {seg[3] / steps * 1e3:.3f} ms", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(45)

# the encoder's backward pass runs on the autograd engine's thread: profile it from inside
from mtvaf_amd import engine  # noqa: E402
_nb = engine._native_backward
pr2 = cProfile.Profile()


def _profiled(*a, **k):
    return pr2.runcall(_nb, *a, **k)


engine._native_backward = _profiled
for _ in range(20):
    step()
torch.cuda.synchronize()
engine._native_backward = _nb
print("== inside engine._native_backward (autograd thread), 20 steps ==")
st2 = pstats.Stats(pr2, stream=sys.stdout)
st2.sort_stats("tottime").print_stats(25)
st2.sort_stats("cumulative").print_stats(25)
