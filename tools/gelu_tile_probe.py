"""FFN-1 forward (+ GELU) and FFN-2 dX (x GELU') at M tokens: every DMA tile configuration, main kernel time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mtvaf_amd import hip
dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H, I = 768, 3072
x = torch.randn(M, H, device=dev); w1 = torch.randn(I, H, device=dev) * 0.02; b1 = torch.zeros(I, device=dev)
pre = torch.empty(M, I, device=dev); act = torch.empty(M, I, device=dev)
df = torch.randn(M, H, device=dev); w2 = torch.randn(H, I, device=dev) * 0.02; dpre = torch.empty(M, I, device=dev)
cases = [("ffn1 fwd+gelu", lambda cfg: hip.gemm(x, 0, w1, 0, act, M, I, H, bias=b1, epi=hip.EPI_GELU, aux=pre, cfg=cfg)),
         ("ffn2 dX+dgelu", lambda cfg: hip.gemm(df, 0, w2, 1, dpre, M, I, H, epi=hip.EPI_DGELU, aux=pre, cfg=cfg)),
         ("qkv-like fwd  ", lambda cfg: hip.gemm(x, 0, w1, 0, act, M, I, H, bias=b1, cfg=cfg))]
for name, fn in cases:
    for cfg in (9, 10, 12, 13, -1):
        for _ in range(3):
            fn(cfg)
        torch.cuda.synchronize()
        hip.prof_start(256)
        for _ in range(20):
            fn(cfg)
        recs = hip.prof_stop(256)
        us = 1e3 * sum(ms for _, ms in recs) / len(recs)
        k = recs[0][0]
        print(f"{name} [{M}x{I}x{H}] cfg {hip.TILE_NAMES.get(k['cfg'], k['cfg']):14s}{' (auto)' if cfg < 0 else '       '}: {us:7.1f} us  {2.0 * M * I * H / us / 1e6:6.1f} TF", flush=True)
