"""fp32 GEMM by three-way bf16 operand splitting (mtvaf_gemm_f32x3, csrc/gemm_f32x3.hip) against the fp32-MFMA kernels
(mtvaf_gemm_f32) on the product shapes of the headline configuration: time, fp32-equivalent TFLOP/s and the error of both
against the fp64 product (max |err| / max |ref| and the rms ratio).

    python tools/f32x3_bench.py [M] [--tiles]    # default 4096; --tiles: also each split tile forced (128x128, 128x96) per product
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mtvaf_amd import hip
dev = "cuda"
H, I = 768, 3072


def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    tiles = "--tiles" in sys.argv
    M = int(args[0]) if args else 4096
    g = torch.Generator(device=dev).manual_seed(1)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g) * torch.exp2(torch.randint(-6, 6, (s[0], 1), device=dev, generator=g).float())
    x, x3 = rn(M, H), rn(M, I)
    wq, wo, w1, w2 = rn(3 * H, H), rn(H, H), rn(I, H), rn(H, I)
    dy, dy3, dyq = rn(M, H), rn(M, I), rn(M, 3 * H)
    KC, KM = hip.KC, hip.KM
    cases = [("qkv fwd", x, KC, wq, KC, (M, 3 * H, H)), ("wo fwd", x, KC, wo, KC, (M, H, H)), ("ffn1 fwd", x, KC, w1, KC, (M, I, H)),
             ("ffn2 fwd", x3, KC, w2, KC, (M, H, I)), ("ffn2 dX", dy, KC, w2, KM, (M, I, H)), ("ffn1 dX", dy3, KC, w1, KM, (M, H, I)),
             ("wo dX", dy, KC, wo, KM, (M, H, H)), ("qkv dX", dyq, KC, wq, KM, (M, H, 3 * H)),
             ("ffn2 dW", dy, KM, x3, KM, (H, I, M)), ("ffn1 dW", dy3, KM, x, KM, (I, H, M)), ("wo dW", dy, KM, x, KM, (H, H, M)),
             ("qkv dW", dyq, KM, x, KM, (3 * H, H, M))]
    tot = {"fp32": 0.0, "fp32x3": 0.0}
    flops = 0.0
    for name, a, la, b, lb, (m, n, k) in cases:
        out = torch.empty(m, n, device=dev)
        A = (a if la == KC else a.t()).double()
        B = (b.t() if lb == KC else b).double()
        ref = A @ B
        line = f"{name:9s} [{m:5d}x{n:5d}x{k:5d}]"
        fl = 2.0 * m * n * k
        flops += fl
        for mode in ("fp32", "fp32x3"):
            hip.f32_split(mode == "fp32x3")  # ("fp32" = the fp32 MFMA pipe: the library default is the split)
            run = lambda: hip.gemm(a, la, b, lb, out, m, n, k, allow_split=True, compute=mode)
            run()
            err = (out.double() - ref)
            emax = float(err.abs().max() / ref.abs().max())
            erms = float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
            us = t(run)
            tot[mode] += us
            line += f" | {mode:7s} {us:6.1f} us {fl / us / 1e6:5.1f} TF err {emax:.1e}/{erms:.1e}"
        hip.f32_split(True)
        if tiles:
            for cfg in (5, 6, 4):
                if n % {5: 128, 6: 96, 4: 64}[cfg] or m % 128:
                    continue
                for sp in ((-1,) if la == KC else (-1, 2, 3, 4, 6, 8)):
                    run = lambda: hip.gemm(a, la, b, lb, out, m, n, k, allow_split=True, compute="fp32x3", cfg=cfg, splits=sp)
                    run()
                    us = t(run)
                    tname = {5: "128x128", 6: "128x96", 4: "128x64"}[cfg]
                    line += f" | {tname}{'' if sp < 0 else '/s' + str(sp)} {us:6.1f}"
        print(line, flush=True)
    print(f"M={M}: one layer's 12 products: fp32 pipe {tot['fp32']:.0f} us ({flops / tot['fp32'] / 1e6:.0f} TF), split {tot['fp32x3']:.0f} us "
          f"({flops / tot['fp32x3'] / 1e6:.0f} TF fp32-equivalent)")


if __name__ == "__main__":
    main()
