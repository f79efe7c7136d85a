"""Per-step rate and fixed cost of the 256 x 256 bf16 kernel (csrc/gemm_bf16p.hip) by operand layout: ONE round of 256 tiles
(M = N = 4096), fp32 result straight from the registers, K swept -- the slope of time over K / 64 is the in-loop time of a
256 x 256 x 64 step, the intercept the launch's fixed cost (start, pipeline fill, epilogue).

    python tools/p256_kscan.py
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mtvaf_amd import hip
dev = "cuda"


def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3


def bf(*shape):
    return (torch.rand(*shape, device=dev) * 2 - 1).to(torch.bfloat16)


M = N = 4096
out = torch.empty(M, N, device=dev)
Ks = [768, 1536, 3072, 6144, 12288, 24576]
for la, lb, name in ((0, 0, "KC x KC (forward)"), (0, 1, "KC x KM (dX)"), (1, 1, "KM x KM (dW)")):
    res = []
    for K in Ks:
        a = bf(M, K) if la == 0 else bf(K, M)
        b = bf(N, K) if lb == 0 else bf(K, N)
        us = t(lambda: hip.gemm_bf16x(a, la, b, lb, M, N, K, out32=out, tile=5, splits=1))
        res.append(us)
        del a, b
    # least squares over the four longest reductions
    xs = [k / 64 for k in Ks[2:]]
    ys = res[2:]
    n = len(xs); mx = sum(xs) / n; my = sum(ys) / n
    slope = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
    icpt = my - slope * mx
    print(f"{name:20s} " + "  ".join(f"K={k}: {u:7.1f} us ({2.0 * M * N * k / u / 1e6:6.0f} TF)" for k, u in zip(Ks, res)), flush=True)
    print(f"{'':20s} per 256x256x64 step {slope:.3f} us ({2 * 256 * 256 * 64 * 256 / slope / 1e6:.0f} TF in-loop), fixed {icpt:.1f} us; "
          f"first two points: {(res[1] - res[0]) / 12:.3f} us per step", flush=True)
