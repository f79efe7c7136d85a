"""Where the HOST time of a training step goes: wall time inside each autograd Function's forward / backward (the
backward runs on autograd's device thread, invisible to cProfile), the optimizer step, and the remainder.
    python tools/host_breakdown.py [--dtype bf16] [--batch 32] [--seq 128] [--aux 8]"""
import argparse, collections, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="fp32"); ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--seq", type=int, default=128); ap.add_argument("--aux", type=int, default=8)
a = ap.parse_args()
from mtvaf_amd import engine, hip
from mtvaf_amd.optim import AdamW
hip.set_compute_dtype(a.dtype)
acc = collections.defaultdict(float)
def wrap(cls, name):
    f = getattr(cls, name)
    def g(*args, **kw):
        t = time.perf_counter()
        try:
            return f(*args, **kw)
        finally:
            acc[f"{cls.__name__}.{name}"] += time.perf_counter() - t
    setattr(cls, name, staticmethod(g))
for cls in (engine.EncoderFunction, engine.LinearFunction, engine.PromptFunction, engine.CRFNLLFunction, engine.EmbeddingsFunction,
            engine.DropoutFunction):
    wrap(cls, "forward"); wrap(cls, "backward")
dev = "cuda"
model, cfg = bench.build_model(dev, "bert", a.seq)
model.train()
ids, mask, tt, labels, feats, aux = bench.synthetic_batch(a.batch, a.seq, a.aux, cfg.vocab_size, 0, dev)
opt = AdamW([p for p in model.parameters() if p.requires_grad], lr=3e-5, model=model, overlap=True)
tm = collections.defaultdict(float)
def step():
    t0 = time.perf_counter()
    out = model(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, images=feats, aux_imgs=aux)
    t1 = time.perf_counter()
    out.loss.backward()
    t2 = time.perf_counter()
    opt.step(); opt.zero_grad(set_to_none=True)
    t3 = time.perf_counter()
    tm["forward"] += t1 - t0; tm["backward"] += t2 - t1; tm["optimizer"] += t3 - t2
for _ in range(5): step()
torch.cuda.synchronize(); acc.clear(); tm.clear()
N = 10
for _ in range(N): step()
torch.cuda.synchronize()
print(f"{a.dtype} B={a.batch} S={a.seq}: per step, ms")
for k, v in tm.items(): print(f"  {k:28s} {1e3 * v / N:7.3f}")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print(f"    {k:34s} {1e3 * v / N:7.3f}")
