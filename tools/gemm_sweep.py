#!/usr/bin/env python
"""Times every tile configuration (and split-K factor for dW) of mtvaf_gemm_f32 on the GEMM shapes of one
encoder layer at the benchmark size.  Used to tune the launcher's plan table (run on the MI355X box)."""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mtvaf_amd import hip  # noqa: E402


def time_call(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=4096)
    ap.add_argument("--cfgs", default="0,1,2,4,5,6,7,8")
    ap.add_argument("--compute", default="fp32")
    ap.add_argument("--zeros", action="store_true", help="zero operands (DVFS check: MI355X_MICROARCH.md give-back)")
    ap.add_argument("--splits", default="", help="split-K factors tried on EVERY product (few-token sizes: --M 256)")
    a = ap.parse_args()
    M, H, I = a.M, 768, 3072
    dev = "cuda"
    cfgs = [int(c) for c in a.cfgs.split(",")]
    g = torch.Generator(device=dev).manual_seed(0)
    R = (lambda *s: torch.zeros(*s, device=dev)) if a.zeros else (lambda *s: torch.randn(*s, device=dev, generator=g))
    shapes = [  # name, la, lb, M, N, K, epi, split
        ("qkv_fwd", 0, 0, M, 3 * H, H, 0, 0), ("ao_fwd", 0, 0, M, H, H, 0, 0), ("ffn1_fwd", 0, 0, M, I, H, 1, 0),
        ("ffn2_fwd", 0, 0, M, H, I, 0, 0),
        ("qkv_dx", 0, 1, M, H, 3 * H, 0, 0), ("ao_dx", 0, 1, M, H, H, 0, 0), ("ffn1_dx", 0, 1, M, H, I, 0, 0),
        ("ffn2_dx", 0, 1, M, I, H, 3, 0),
        ("qkv_dw", 1, 1, 3 * H, H, M, 0, 1), ("ao_dw", 1, 1, H, H, M, 0, 1), ("ffn1_dw", 1, 1, I, H, M, 0, 1),
        ("ffn2_dw", 1, 1, H, I, M, 0, 1)]
    out = {}
    slist = [int(x) for x in a.splits.split(",")] if a.splits else None
    for name, la, lb, m, n, k, epi, split in shapes:
        A = R(k, m) if la else R(m, k)
        B = R(k, n) if lb else R(n, k)
        C = torch.empty(m, n, device=dev)
        aux = R(m, n) if epi else None
        bias = R(n) if not split else None
        best = None
        rows = []
        for cfg in cfgs:
            for s in (slist if slist else ([1] if not split else [1, 2, 3, 4, 6, 8, 16])):
                try:
                    us = time_call(lambda: hip.gemm(A, la, B, lb, C, m, n, k, bias=bias, epi=epi, aux=aux,
                                                    allow_split=bool(split) or bool(slist), cfg=cfg, splits=s, compute=a.compute))
                except RuntimeError as e:
                    continue
                tf = 2.0 * m * n * k / us / 1e6
                rows.append((cfg, s, round(us, 1), round(tf, 1)))
                if best is None or us < best[2]:
                    best = (cfg, s, us, tf)
        auto = hip.gemm_plan(m, n, k, bool(split) or bool(slist), la, lb, epi)
        us_auto = time_call(lambda: hip.gemm(A, la, B, lb, C, m, n, k, bias=bias, epi=epi, aux=aux,
                                             allow_split=bool(split) or bool(slist), compute=a.compute))
        out[name] = {"best": best, "auto": auto, "rows": rows, "auto_us": us_auto}
        print(f"{name:9s} M={m:5d} N={n:5d} K={k:5d} best cfg={hip.TILE_NAMES[best[0]]:11s} splits={best[1]:2d} "
              f"{best[2]:7.1f} us {best[3]:6.1f} TF | auto cfg={hip.TILE_NAMES[auto[0]]} s={auto[1]} {us_auto:.1f} us | " +
              " ".join(f"{c}/{s}:{tf}" for c, s, us, tf in rows), flush=True)
    tot = sum(v["best"][2] for v in out.values())
    fl = sum(2.0 * m * n * k for _, _, _, m, n, k, _, _ in shapes)
    print(f"sum of auto: {sum(v['auto_us'] for v in out.values()):.1f} us per layer")
    print(f"sum of best: {tot:.1f} us per layer -> {12 * tot / 1e3:.2f} ms per step, {fl / tot / 1e6:.1f} TF average")


if __name__ == "__main__":
    main()
