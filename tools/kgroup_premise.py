"""Premise check for an intra-block split-K (two 4-wave k-groups sharing one LDS ring): how much faster is the MAIN kernel
of a one-tile-per-CU product when two blocks per CU split the reduction (2-stage ring, split-K 2: the reduce launch and
the slab traffic are what the intra-block variant would not pay) than one 3-stage block per CU?  Kernel time only
(library profiler: HIP events around the main kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mtvaf_amd import hip
dev = "cuda"
M = 4096
shapes = [("wo fwd", 0, 0, M, 768, 768), ("ffn2 fwd", 0, 0, M, 768, 3072), ("qkv dX", 0, 1, M, 768, 2304), ("ffn1 dX", 0, 1, M, 768, 3072)]
for name, la, lb, m, n, k in shapes:
    a = torch.randn(m, k, device=dev)
    b = torch.randn(n, k, device=dev) if lb == 0 else torch.randn(k, n, device=dev)
    c = torch.empty(m, n, device=dev)
    for cfg, sp in ((9, 1), (12, 1), (12, 2), (9, 2)):
        for _ in range(3):
            hip.gemm(a, la, b, lb, c, m, n, k, cfg=cfg, splits=sp, allow_split=True)
        torch.cuda.synchronize()
        hip.prof_start(256)
        for _ in range(20):
            hip.gemm(a, la, b, lb, c, m, n, k, cfg=cfg, splits=sp, allow_split=True)
        recs = hip.prof_stop(256)
        us = 1e3 * sum(ms for _, ms in recs) / len(recs)
        print(f"{name:9s} [{m}x{n}x{k}] cfg {hip.TILE_NAMES[cfg]:14s} splits {recs[0][0]['splits']}: {us:7.1f} us main kernel  {2.0 * m * n * k / us / 1e6:6.1f} TF", flush=True)
