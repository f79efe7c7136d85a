"""bf16-operand GEMM (mtvaf_gemm_bf16x): rate of every product shape of the path in its real operand layout
(forward KCxKC, dX KCxKM, dW KMxKM + split-K), per tile / ring depth, at one or more token counts.

    python tools/bf16x_bench.py [M ...] [--splits]   # default 4096 8192 65536; --splits: also split-K 2 / 3 / 4 of the fp32-output products
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


def bf(*shape):
    return (torch.randn(*shape, device=dev) * 0.5).to(torch.bfloat16)


def main():
    Ms = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [4096, 8192, 65536]
    for M in Ms:
        x, x3, w_qkv, w_o, w_1, w_2 = bf(M, H), bf(M, I), bf(3 * H, H), bf(H, H), bf(I, H), bf(H, I)
        dy, dy3, dyq = bf(M, H), bf(M, I), bf(M, 3 * H)
        cases = [  # name, a, la, b, lb, (m, n, k), kwargs
            ("qkv fwd", x, 0, w_qkv, 0, (M, 3 * H, H), {}), ("wo fwd", x, 0, w_o, 0, (M, H, H), {}),
            ("ffn1 fwd+gelu", x, 0, w_1, 0, (M, I, H), {"gelu": 1}), ("ffn2 fwd", x3, 0, w_2, 0, (M, H, I), {}),
            ("ffn2 dX+dgelu", dy, 0, w_2, 1, (M, I, H), {"dgelu": 1}), ("ffn1 dX", dy3, 0, w_1, 1, (M, H, I), {}),
            ("wo dX", dy, 0, w_o, 1, (M, H, H), {}), ("qkv dX", dyq, 0, w_qkv, 1, (M, H, 3 * H), {}),
            ("ffn2 dW", dy, 1, x3, 1, (H, I, M), {"split": 1}), ("ffn1 dW", dy3, 1, x, 1, (I, H, M), {"split": 1}),
            ("wo dW", dy, 1, x, 1, (H, H, M), {"split": 1}), ("qkv dW", dyq, 1, x, 1, (3 * H, H, M), {"split": 1})]
        tot_us, tot_fl = 0.0, 0.0
        for name, a, la, b, lb, (m, n, k), kw in cases:
            out32 = torch.empty(m, n, device=dev)
            out16 = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
            aux = bf(m, n)
            bias = torch.randn(n, device=dev)
            best = None
            for tile in (1, 2, 3, 4):
                if (tile == 1 and n % 96) or (tile == 3 and (la == 1 or m % 256 or kw.get("dgelu"))) or (tile == 4 and (la == 1 or m % 256 or n % 192)):
                    continue
                for stages in (2, 3):
                    if (tile == 3 and stages > 3) or (tile == 2 and stages > 4) or (tile == 4 and stages != 2):
                        continue
                    def run():
                        if kw.get("split"):
                            hip.gemm_bf16x(a, la, b, lb, m, n, k, out32=out32, allow_split=True, tile=tile, stages=stages)
                        elif kw.get("gelu"):
                            hip.gemm_bf16x(a, la, b, lb, m, n, k, out16=out16, bias=bias, epi=hip.EPI_GELU, aux16=aux, tile=tile, stages=stages)
                        elif kw.get("dgelu"):
                            hip.gemm_bf16x(a, la, b, lb, m, n, k, out16=out16, epi=hip.EPI_DGELU, aux16=aux, tile=tile, stages=stages)
                        else:
                            hip.gemm_bf16x(a, la, b, lb, m, n, k, out32=out32, bias=bias, tile=tile, stages=stages)
                    us = t(run)
                    tf = 2.0 * m * n * k / us / 1e6
                    print(f"M={M:6d} {name:14s} [{m:5d}x{n:5d}x{k:5d}] tile {['', '128x96 ', '128x128', '256x128', '256x192'][tile]} stages {stages}: {us:8.1f} us {tf:7.1f} TF", flush=True)
                    if best is None or us < best[0]:
                        best = (us, tile, stages)
            if "--splits" in sys.argv and la == 0 and not (kw.get("gelu") or kw.get("dgelu")):
                # fp32-output forward / dX products: the deterministic split-K (slabs + ordered reduction) per tile
                for tile in (1, 2):
                    if tile == 1 and n % 96:
                        continue
                    for stages in (2, 3):
                        for sp in (2, 3, 4):
                            run = lambda: hip.gemm_bf16x(a, la, b, lb, m, n, k, out32=out32, bias=bias, allow_split=True, tile=tile, stages=stages, splits=sp)
                            us = t(run)
                            print(f"M={M:6d} {name:14s} [{m:5d}x{n:5d}x{k:5d}] tile {['', '128x96 ', '128x128'][tile]} stages {stages} splits {sp}: {us:8.1f} us "
                                  f"{2.0 * m * n * k / us / 1e6:7.1f} TF", flush=True)
                            if us < best[0]:
                                best = (us, tile, stages, sp)
            us_auto = t(lambda: (hip.gemm_bf16x(a, la, b, lb, m, n, k, out32=out32, allow_split=bool(kw.get("split")))
                                 if not (kw.get("gelu") or kw.get("dgelu")) else
                                 hip.gemm_bf16x(a, la, b, lb, m, n, k, out16=out16, bias=bias if kw.get("gelu") else None,
                                                epi=hip.EPI_GELU if kw.get("gelu") else hip.EPI_DGELU, aux16=aux)))
            print(f"    -> best {best[0]:.1f} us (tile {best[1]}, stages {best[2]}{', splits ' + str(best[3]) if len(best) > 3 else ''}); auto {us_auto:.1f} us", flush=True)
            tot_us += us_auto
            tot_fl += 2.0 * m * n * k
        print(f"M={M}: one layer's 12 products {tot_us:.0f} us, {tot_fl / tot_us / 1e6:.0f} TF average (auto plan)\n", flush=True)


if __name__ == "__main__":
    main()
