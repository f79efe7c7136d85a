"""Would the per-layer AdamW updates cost less beside the NEXT step's forward pass than inside the backward pass?
Timing probe only: mode `emulate` enqueues optimizer.step() on a second stream behind the backward pass and lets the next forward
start at once -- the updates then race with the forward's reads of the weights (results are meaningless), but the machine sees
the work mix a deferred update would produce.  Modes: overlap (updates from inside backward: the product path), serial (all of
step() behind backward on the main stream), emulate.

    python tools/defer_probe.py [steps]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mtvaf_amd.optim import AdamW  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")


def run(mode):
    torch.manual_seed(1)
    model, cfg = bench.build_model(dev, "bert", 128)
    model.train()
    opt = AdamW([p for p in model.parameters() if p.requires_grad], lr=3e-5, weight_decay=1e-2, model=model, overlap=(mode == "overlap"))
    ids, mask, tt, labels, feats, aux = bench.synthetic_batch(32, 128, 8, cfg.vocab_size, 1234, dev)
    kw = dict(input_ids=ids, attention_mask=mask, token_type_ids=tt, labels=labels, imagelabel=None, images=feats, aux_imgs=aux)
    side = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream()

    def step():
        out = model(**kw)
        out.loss.backward()
        if mode == "emulate":
            side.wait_stream(main)
            with torch.cuda.stream(side):
                opt.step()
                opt.zero_grad(set_to_none=True)
        else:
            opt.step()
            opt.zero_grad(set_to_none=True)

    for _ in range(8):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    main.wait_stream(side)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{mode:8s}: {ms:7.3f} ms / step = {32 / ms * 1e3:7.1f} sentences / s", flush=True)
    del model, opt
    torch.cuda.empty_cache()


for m in ("overlap", "serial", "emulate", "overlap", "emulate"):
    run(m)
