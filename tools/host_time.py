"""Host enqueue time vs GPU time of a training step: is a configuration launch-bound?
    python tools/host_time.py [--dtype bf16] [--batch 32] [--seq 128] [--aux 8]
Enqueues N steps WITHOUT reading the tags (no host sync), measures the Python time to enqueue them and the GPU time to
drain them; also a cProfile of the enqueue loop."""
import argparse, cProfile, os, pstats, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="fp32"); ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--seq", type=int, default=128); ap.add_argument("--aux", type=int, default=8)
ap.add_argument("--profile", action="store_true")
a = ap.parse_args()
from mtvaf_amd import hip
from mtvaf_amd.optim import AdamW
hip.set_compute_dtype(a.dtype)
dev = "cuda"
model, cfg = bench.build_model(dev, "bert", a.seq)
model.train()
ids, mask, tt, labels, feats, aux = bench.synthetic_batch(a.batch, a.seq, a.aux, cfg.vocab_size, 0, dev)
opt = AdamW([p for p in model.parameters() if p.requires_grad], lr=3e-5, model=model, overlap=True)
def step():
    out = model(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, images=feats, aux_imgs=aux)
    out.loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
    return out
for _ in range(5): step()
torch.cuda.synchronize()
N = 10
t0 = time.perf_counter()
outs = [step() for _ in range(N)]
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{a.dtype} B={a.batch} S={a.seq}: host enqueue {1e3 * (t1 - t0) / N:.2f} ms/step, drained after {1e3 * (t2 - t0) / N:.2f} ms/step "
      f"({'HOST-bound' if (t2 - t1) < 0.1 * (t1 - t0) else 'GPU-bound'})")
if a.profile:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10): step()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
