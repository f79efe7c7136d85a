import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
B, S = 128, 512
dev = "cuda"
model, cfg = bench.build_model(dev, "bert", S)
model.train()
ids, mask, tt, labels, feats, aux = bench.synthetic_batch(B, S, 8, cfg.vocab_size, 1234, dev)
opt = torch.optim.AdamW(model.parameters(), lr=3e-5, fused=True)
def step(read_tags=True):
    out = model(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, imagelabel=None, images=feats, aux_imgs=aux)
    out.loss.backward()
    opt.step(); opt.zero_grad(set_to_none=True)
    if read_tags:
        assert len(out.logits) == B
    return out
for name, kw, sync in (("bench loop", {}, False), ("no tag read", {"read_tags": False}, False), ("sync each step", {}, True)):
    for _ in range(4): step(**kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4):
        out = step(**kw)
        if sync: torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(f"{name:16s}: {(time.perf_counter() - t0) / 4 * 1e3:7.1f} ms/step, peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
