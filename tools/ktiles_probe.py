"""List-mode dW products on the bench batch's mask: main-kernel time per split count / tile configuration."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mtvaf_amd import hip
dev = "cuda"
B, S, P = 32, 128, 36
ids, mask, tt, labels, feats, aux = bench.synthetic_batch(B, S, 8, 30522, 1234, dev, False)
addmask = torch.cat([torch.zeros(B, P, device=dev), (1 - mask.float()) * -10000.0], 1).contiguous()
klist, kcnt = hip.build_ktiles(addmask, P, S)
print("listed k-tiles:", int(kcnt.item()), "of", B * S // 32)
M = B * S
valid = mask.view(-1).bool()
for name, NO, KI in (("ffn2 dW", 768, 3072), ("ffn1 dW", 3072, 768), ("qkv dW", 2304, 768), ("wo dW", 768, 768)):
    dy = torch.randn(M, NO, device=dev) * valid[:, None]
    x = torch.randn(M, KI, device=dev)
    out = torch.empty(NO, KI, device=dev)
    for cfg in (-1, 9, 12, 14, 15):
        for sp in (-1, 2, 3, 4, 6, 8):
            for _ in range(3):
                hip.gemm_ktiles(dy, x, out, NO, KI, M, klist, kcnt, cfg=cfg, splits=sp)
            torch.cuda.synchronize()
            hip.prof_start(256)
            for _ in range(10):
                hip.gemm_ktiles(dy, x, out, NO, KI, M, klist, kcnt, cfg=cfg, splits=sp)
            recs = hip.prof_stop(256)
            us = 1e3 * sum(ms for _, ms in recs) / len(recs)
            k = recs[0][0]
            fl = 2.0 * NO * KI * 32 * int(kcnt.item())
            print(f"{name} [{NO}x{KI}] cfg {hip.TILE_NAMES.get(k['cfg'], k['cfg']):14s} splits {k['splits']:2d}{' (auto)' if sp < 0 and cfg < 0 else ''}: {us:7.1f} us {fl / us / 1e6:6.1f} TF", flush=True)
