"""Which torch-level ops of one training step launch device copies / torch kernels (everything that is not a library
kernel)?   python tools/find_copies.py [--dtype bf16] [--batch 32] [--seq 128]"""
import argparse, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mtvaf_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="fp32")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--seq", type=int, default=128)
ap.add_argument("--aux", type=int, default=8)
a = ap.parse_args()
dev = "cuda:0"
hip.lib(); hip.set_compute_dtype(a.dtype)
model, cfg = bench.build_model(dev, "bert", a.seq)
model.train()
from mtvaf_amd.optim import AdamW
opt = AdamW([p for p in model.parameters() if p.requires_grad], lr=3e-5, weight_decay=1e-2, model=model, overlap=True)
ids, mask, tt, labels, feats, aux = bench.synthetic_batch(a.batch, a.seq, a.aux, cfg.vocab_size, 1234, dev, False)

def step():
    out = model(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, imagelabel=None, images=feats, aux_imgs=aux)
    out.loss.backward()
    opt.step(); opt.zero_grad(set_to_none=True)
    return out

for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.cpu_parent is None or (
            ev.cpu_parent is not None and not ev.cpu_parent.name.startswith("aten::") and ev.name.startswith("aten::")):
        stack = [s for s in (ev.stack or []) if "mtvaf_amd" in s or "bench" in s or "tools/" in s]
        agg[(ev.name, stack[0] if stack else "?")] += 1
for (name, where), n in sorted(agg.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{n:4d}  {name:28s} {where}")
print()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
