"""Per-step cost of the CRF kernels: time vs sequence length at fixed batch (slope = one time step of the serial
recursion, intercept = launch + staging + epilogue).   python tools/crf_bench.py [B] [C]"""
import sys
import torch
sys.path[:0] = ["."]
from mtvaf_amd import hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
C = int(sys.argv[2]) if len(sys.argv) > 2 else 11
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
rows = []
for S in (32, 64, 128, 256, 512):
    em = torch.randn(B, S, C, generator=g).to(dev)
    tags = torch.randint(0, C, (B, S), generator=g).to(dev)
    mask = torch.ones(B, S, dtype=torch.uint8, device=dev)
    start, end, trans = (torch.rand(n, generator=g).sub(0.5).to(dev) for n in ((C,), (C,), (C, C)))
    ws, wsb = hip.crf_workspace(B, S, C, dev)
    loss = torch.empty(1, device=dev)
    dem = torch.empty(B, S, C, device=dev)
    ds, de, dt = (torch.zeros(n, device=dev) for n in ((C,), (C,), (C, C)))
    tg, ln = torch.empty(B, S, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev)
    fns = {"fwd": lambda: hip.crf_nll_fwd(em, tags, mask, start, end, trans, loss, ws, wsb),
           "bwd": lambda: hip.crf_nll_bwd(None, em, tags, mask, start, end, trans, dem, ds, de, dt, False, ws, wsb),
           "viterbi": lambda: hip.crf_viterbi(em, mask, start, end, trans, tg, ln)}
    row = {"S": S}
    for name, fn in fns.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        row[name] = e0.elapsed_time(e1) * 1e3 / n  # us per call (fwd / bwd include their small second kernel)
    rows.append(row)
    print(row, flush=True)
for name in ("fwd", "bwd", "viterbi"):
    a, b = rows[2], rows[4]
    slope = (b[name] - a[name]) / (b["S"] - a["S"])
    print(f"{name}: {slope * 1e3:.1f} ns per step, intercept {a[name] - slope * a['S']:.1f} us (back-to-back launches)")
