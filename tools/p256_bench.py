"""The 256x256 eight-phase bf16 GEMM (mtvaf_gemm_bf16x tile 5, csrc/gemm_bf16p.hip) against the planner's choice among the
older tiles, on every product shape of the path in its real operand layout, per split count.

    python tools/p256_bench.py [M ...]         # default 4096 8192 65536
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
    return (torch.rand(*shape, device=dev) * 2 - 1).to(torch.bfloat16)  # uniform [-1, 1): the guide's reference data


def main():
    Ms = [int(a) for a in sys.argv[1:]] or [4096, 8192, 65536]
    for M in Ms:
        x, x3, w_qkv, w_o, w_1, w_2 = bf(M, H), bf(M, I), bf(3 * H, H), bf(H, H), bf(I, H), bf(H, I)
        dy, dy3, dyq = bf(M, H), bf(M, I), bf(M, 3 * H)
        cases = [  # name, a, la, b, lb, (m, n, k), kind
            ("qkv fwd", x, 0, w_qkv, 0, (M, 3 * H, H), "b16"), ("wo fwd", x, 0, w_o, 0, (M, H, H), "f32"),
            ("ffn1 fwd+gelu", x, 0, w_1, 0, (M, I, H), "gelu"), ("ffn2 fwd", x3, 0, w_2, 0, (M, H, I), "f32"),
            ("ffn2 dX+dgelu", dy, 0, w_2, 1, (M, I, H), "dgelu"), ("ffn1 dX", dy3, 0, w_1, 1, (M, H, I), "f32"),
            ("wo dX", dy, 0, w_o, 1, (M, H, H), "b16"), ("qkv dX", dyq, 0, w_qkv, 1, (M, H, 3 * H), "acc"),
            ("ffn2 dW", dy, 1, x3, 1, (H, I, M), "split"), ("ffn1 dW", dy3, 1, x, 1, (I, H, M), "split"),
            ("wo dW", dy, 1, x, 1, (H, H, M), "split"), ("qkv dW", dyq, 1, x, 1, (3 * H, H, M), "split")]
        tot_old, tot_new, tot_best, tot_fl, tot_sk = 0.0, 0.0, 0.0, 0.0, 0.0
        for name, a, la, b, lb, (m, n, k), kind in cases:
            out32 = torch.empty(m, n, device=dev)
            out16 = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
            aux = bf(m, n)
            bias = torch.randn(n, device=dev)
            part = torch.empty(m // 128, n, device=dev)

            def run(tile, splits=-1):
                if kind == "split":
                    hip.gemm_bf16x(a, la, b, lb, m, n, k, out32=out32, allow_split=True, tile=tile, splits=splits)
                elif kind == "gelu":
                    hip.gemm_bf16x(a, la, b, lb, m, n, k, out16=out16, bias=bias, epi=hip.EPI_GELU, aux16=aux, tile=tile)
                elif kind == "dgelu":
                    hip.gemm_bf16x(a, la, b, lb, m, n, k, out16=out16, epi=hip.EPI_DGELU, aux16=aux, colpart=part if tile in (0, 5) else None, tile=tile)
                elif kind == "b16":
                    hip.gemm_bf16x(a, la, b, lb, m, n, k, out16=out16, bias=bias, tile=tile)
                elif kind == "acc":
                    hip.gemm_bf16x(a, la, b, lb, m, n, k, out32=out32, accumulate=True, allow_split=True, tile=tile, splits=splits)
                else:
                    hip.gemm_bf16x(a, la, b, lb, m, n, k, out32=out32, bias=bias, allow_split=True, tile=tile, splits=splits)
            fl = 2.0 * m * n * k
            us_old = t(lambda: run(0))
            res = []
            tiles = (m // 256) * (n // 256)
            cand = [1] if kind in ("gelu", "dgelu", "b16") else sorted({1, 2, 3, 4, 6, 8, max(1, 256 // tiles), max(1, 512 // tiles)})
            for sp in cand:
                if k // 64 < sp or sp > 8:
                    continue
                res.append((t(lambda: run(5, sp)), sp))
            hip.streamk_ensure(dev)
            us_sk = t(lambda: run(6))
            us_new, sp_new = min(res)
            print(f"M={M:6d} {name:14s} [{m:5d}x{n:5d}x{k:5d}] auto(old) {us_old:7.1f} us {fl / us_old / 1e6:7.1f} TF | 256x256: "
                  + "  ".join(f"s{sp} {us:6.1f}" for us, sp in res) + f" | best s{sp_new} {fl / us_new / 1e6:7.1f} TF | stream-K {us_sk:6.1f} us {fl / us_sk / 1e6:7.1f} TF", flush=True)
            tot_old += us_old; tot_new += us_new; tot_best += min(us_old, us_new, us_sk); tot_fl += fl; tot_sk += us_sk
        # the four weight gradients as one grouped stream-K launch
        items = [(dy, x3, torch.empty(H, I, device=dev)), (dy3, x, torch.empty(I, H, device=dev)),
                 (dy, x, torch.empty(H, H, device=dev)), (dyq, x, torch.empty(3 * H, H, device=dev))]
        us_g = t(lambda: hip.gemm_bf16x_dw_group(items, M))
        fl_g = 2.0 * M * (H * I * 2 + H * H + 3 * H * H)
        print(f"M={M:6d} four dW as ONE stream-K group: {us_g:7.1f} us {fl_g / us_g / 1e6:7.1f} TF; stream-K everywhere {tot_sk:.0f} us", flush=True)
        print(f"M={M}: one layer's 12 products: old auto {tot_old:.0f} us ({tot_fl / tot_old / 1e6:.0f} TF), 256x256 everywhere "
              f"{tot_new:.0f} us ({tot_fl / tot_new / 1e6:.0f} TF), best of both {tot_best:.0f} us ({tot_fl / tot_best / 1e6:.0f} TF)\n", flush=True)


if __name__ == "__main__":
    main()
