"""Does the one host synchronisation of padding-free execution (the packed row count sizes every launch: engine.Packing.build)
cost GPU idle time at the headline shape?  Timing probe only: mode `reuse` hands every step the Packing of the first one (the
bench feeds the same mask each step) without the map kernel, the copy of the count and the wait -- the host then runs ahead of
the GPU as in the padded layout.  `host` = wall time of the loop before the final synchronisation (what the host needs to
enqueue a step).

    python tools/pack_sync_probe.py [steps]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mtvaf_amd import engine  # noqa: E402
from mtvaf_amd.optim import AdamW  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
_begin, _build = engine.Packing.begin, engine.Packing.build


def run(mode):
    torch.manual_seed(1)
    model, cfg = bench.build_model(dev, "bert", 128)
    model.train()
    opt = AdamW([p for p in model.parameters() if p.requires_grad], lr=3e-5, weight_decay=1e-2, model=model, overlap=True)
    ids, mask, tt, labels, feats, aux = bench.synthetic_batch(32, 128, 8, cfg.vocab_size, 1234, dev)
    kw = dict(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, imagelabel=None, images=feats, aux_imgs=aux)
    cache = {}

    def begin(addmask, Pn, S):
        if "pk" not in cache:
            return _begin(addmask, Pn, S)

    def build(addmask, Pn, B, S):
        if "pk" not in cache:
            cache["pk"] = _build(addmask, Pn, B, S)
        return cache["pk"]

    if mode == "reuse":
        engine.Packing.begin, engine.Packing.build = staticmethod(begin), staticmethod(build)
    else:
        engine.Packing.begin, engine.Packing.build = staticmethod(_begin), staticmethod(_build)

    def step():
        out = model(**kw)
        out.loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)

    for _ in range(8):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    host = (time.perf_counter() - t0) / steps * 1e3
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{mode:6s}: {ms:7.3f} ms / step = {32 / ms * 1e3:7.1f} sentences / s   (host loop {host:6.3f} ms / step)", flush=True)
    del model, opt
    torch.cuda.empty_cache()


for m in ("sync", "reuse", "sync", "reuse"):
    run(m)
