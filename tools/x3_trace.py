"""Where the wave-specialised split kernel's time goes, per wave and k-tile (mtvaf_f32x3_trace: block 0 stamps the shader
clock around every tile barrier).  Consumers (waves 0-3): [2] arrive at the barrier, [3] leave it.  Producers (4-7): [0] step
start, [1] planes stored, [2] next loads requested = arrive, [3] leave.

    python tools/x3_trace.py [M N K] [cfg] [la lb]     # la / lb: 0 = KC (reduction index contiguous), 1 = KM (k-major)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mtvaf_amd import hip  # noqa: E402

dev = "cuda"
args = [int(a) for a in sys.argv[1:]]
M, N, K = (args + [4096, 3072, 768])[:3] if len(args) >= 3 else (4096, 3072, 768)
cfg = args[3] if len(args) > 3 else 5
la, lb = (args[4], args[5]) if len(args) > 5 else (0, 0)
a = torch.randn(*((K, M) if la else (M, K)), device=dev)
b = torch.randn(*((K, N) if lb else (N, K)), device=dev)
out = torch.empty(M, N, device=dev)
run = lambda: hip.gemm(a, la, b, lb, out, M, N, K, compute="fp32x3", cfg=cfg)
for _ in range(3):
    run()
buf = torch.zeros(8 * 64 * 4 + 17, dtype=torch.int64, device=dev)
hip.lib().mtvaf_f32x3_trace(hip._p(buf))
run()
torch.cuda.synchronize()
hip.lib().mtvaf_f32x3_trace(None)
t = buf.cpu()
t0 = int(t[8 * 64 * 4])
st = (t[:8 * 64 * 4].view(8, 64, 4) - t0).clamp_min(0)
nk = min(K // 32, 64)
print(f"[{M}x{N}x{K}] cfg {cfg}: block 0, cycles since block start; nk = {nk}")
print("kt   | consumers: arrive/leave (wait)                         | producers: start/stored/arrive/leave")
for kt in list(range(min(nk, 6))) + list(range(max(6, nk - 3), nk)):
    c = "  ".join(f"{int(st[w, kt, 2]):6d}/{int(st[w, kt, 3]):6d}" for w in range(4))
    p_ = "  ".join(f"{int(st[w, kt, 0]):6d}/{int(st[w, kt, 1]):6d}/{int(st[w, kt, 2]):6d}/{int(st[w, kt, 3]):6d}" for w in range(4, 8))
    print(f"{kt:3d}  | {c} | {p_}")
lo, hi = 2, nk - 1
cw = (st[:4, lo:hi, 3] - st[:4, lo:hi, 2]).double().mean()
pw = (st[4:, lo:hi, 3] - st[4:, lo:hi, 2]).double().mean()
step = (st[:4, hi - 1, 3] - st[:4, lo, 3]).double().mean() / (hi - 1 - lo)
pstage = (st[4:, lo:hi, 1] - st[4:, lo:hi, 0]).double().mean()
pload = (st[4:, lo:hi, 2] - st[4:, lo:hi, 1]).double().mean()
print(f"steady state: {float(step):.0f} cycles per k-tile; consumers wait {float(cw):.0f} at the barrier; producers: staging {float(pstage):.0f}, "
      f"requests {float(pload):.0f}, barrier wait {float(pw):.0f}")
ends = t[8 * 64 * 4 + 1:8 * 64 * 4 + 9] - t0
fin = t[8 * 64 * 4 + 9:8 * 64 * 4 + 17] - t0
print("k-loop over at      :", [int(x) for x in ends])
print("stores issued at    :", [int(x) for x in fin])
print(f"first barrier left at: {int(st[0, 0, 3])} (prologue + first k-tile)")
