"""Pre-split-operand GEMM (mtvaf_gemm_f32p, csrc/gemm_f32p.hip) against the wave-specialised split kernel (mtvaf_gemm_f32 in
split mode) on the forward products of a layer: time, fp32-equivalent TFLOP/s, error against the fp64 product, the cost of the
split passes, the timing-only ablations (no MFMAs / no requests / no fragment reads) and a k-tile trace of block 0.

    python tools/f32p_bench.py [M ...]        # default 4096 2432
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


def trace(run, nk):
    buf = torch.zeros(8 * 64 * 2 + 17, dtype=torch.int64, device=dev)
    hip._ck(hip.lib().mtvaf_f32p_trace(hip._p(buf)), "trace")
    run(); torch.cuda.synchronize()
    hip._ck(hip.lib().mtvaf_f32p_trace(None), "trace")
    b = buf.cpu().tolist()
    t0 = b[8 * 64 * 2]
    st = lambda w, k, j: b[(w * 64 + k) * 2 + j] - t0
    n = min(nk, 64)
    if n < 8:
        return "(too few k-tiles)"
    per = (st(0, n - 2, 1) - st(0, 3, 1)) / (n - 5)
    cw = sum(st(w, k, 1) - st(w, k, 0) for w in range(4) for k in range(3, n - 1)) / (4 * (n - 4))
    dw = sum(st(w, k, 1) - st(w, k, 0) for w in range(4, 8) for k in range(3, n - 1)) / (4 * (n - 4))
    end = [b[8 * 64 * 2 + 1 + w] - t0 for w in range(8)]
    done = [b[8 * 64 * 2 + 9 + w] - t0 for w in range(8)]
    return (f"{per:.0f} ticks per k-tile; at the barrier: MFMA waves wait {cw:.0f}, DMA waves wait {dw:.0f} (incl. vmcnt); first tile barrier left at "
            f"{st(0, 0, 1)}, k-loop over at {max(end)}, stores issued at {max(done)}")


def main():
    Ms = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [4096, 2432]
    g = torch.Generator(device=dev).manual_seed(1)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g) * torch.exp2(torch.randint(-6, 6, (s[0], 1), device=dev, generator=g).float())
    for M in Ms:
        x, x3 = rn(M, H), rn(M, I)
        wq, wo, w1, w2 = rn(3 * H, H), rn(H, H), rn(I, H), rn(H, I)
        tot_ws = tot_p = 0.0
        for name, a, b in (("qkv fwd", x, wq), ("wo fwd", x, wo), ("ffn1 fwd", x, w1), ("ffn2 fwd", x3, w2)):
            m, k = a.shape
            n = b.shape[0]
            out = torch.empty(m, n, device=dev)
            ref = a.double() @ b.double().t()
            fl = 2.0 * m * n * k
            hip.f32_split(True)
            run_ws = lambda: hip.gemm(a, hip.KC, b, hip.KC, out, m, n, k, allow_split=True)
            run_ws()
            e_ws = float((out.double() - ref).abs().max() / ref.abs().max())
            us_ws = t(run_ws)
            line = f"M={M} {name:9s} [{m:5d}x{n:5d}x{k:5d}] | ws {us_ws:6.1f} us {fl / us_ws / 1e6:5.1f} TF err {e_ws:.1e}"
            for blocked in (True, False):
                pa, pb = hip.Planes(a, blocked), hip.Planes(b, blocked)
                best = None
                for sp in (1, 2):
                    if sp == 2 and (m // 128) * (n // 128) > 200:
                        continue
                    run_p = lambda: hip.gemm_planes(pa, pb, out, splits=sp)
                    out.zero_(); run_p()
                    e_p = float((out.double() - ref).abs().max() / ref.abs().max())
                    us_p = t(run_p)
                    line += f" | P16{'b' if blocked else 'n'}{'/s2' if sp == 2 else ''} {us_p:6.1f} us {fl / us_p / 1e6:5.1f} TF err {e_p:.1e}"
                    best = us_p if best is None else min(best, us_p)
                if blocked:
                    tot_p += best
                    for ab, tag in ((1, "noMFMA"), (2, "noDMA"), (4, "noRD"), (6, "MFMAonly"), (5, "DMAonly")):
                        line += f" {tag} {t(lambda: hip.gemm_planes(pa, pb, out, ablate=ab)):.1f}"
                    line += f" | split A {t(lambda: pa.refresh(a)):.1f} B {t(lambda: pb.refresh(b)):.1f}"
                    print(line, flush=True)
                    print("      trace: " + trace(lambda: hip.gemm_planes(pa, pb, out), k // 32), flush=True)
                    line = "      natural-layout planes:"
            print(line, flush=True)
            tot_ws += us_ws
        print(f"M={M}: four forward products: ws {tot_ws:.0f} us, P16 (blocked planes, best split) {tot_p:.0f} us", flush=True)
        # the dX products: A = dY [M, N_w] (k-contiguous), B = W [N_w, K_w] with the reduction index as its row (natural planes)
        dy, dy3, dyq = rn(M, H), rn(M, I), rn(M, 3 * H)
        tot_ws = tot_p = 0.0
        for name, a, w in (("ffn2 dX", dy, w2), ("ffn1 dX", dy3, w1), ("wo dX", dy, wo), ("qkv dX", dyq, wq)):
            m, k = a.shape
            n = w.shape[1]
            out = torch.empty(m, n, device=dev)
            ref = a.double() @ w.double()
            fl = 2.0 * m * n * k
            run_ws = lambda: hip.gemm(a, hip.KC, w, hip.KM, out, m, n, k, allow_split=True)
            run_ws()
            e_ws = float((out.double() - ref).abs().max() / ref.abs().max())
            us_ws = t(run_ws)
            pa, pb = hip.Planes(a, True), hip.Planes(w, False)
            line = f"M={M} {name:9s} [{m:5d}x{n:5d}x{k:5d}] | ws {us_ws:6.1f} us {fl / us_ws / 1e6:5.1f} TF err {e_ws:.1e}"
            best = None
            for sp in (1, 2):
                if sp == 2 and (m // 128) * (n // 128) > 200:
                    continue
                run_p = lambda: hip.gemm_planes(pa, pb, out, splits=sp, layout_b=hip.KM)
                out.zero_(); run_p()
                e_p = float((out.double() - ref).abs().max() / ref.abs().max())
                us_p = t(run_p)
                line += f" | P16{'/s2' if sp == 2 else ''} {us_p:6.1f} us {fl / us_p / 1e6:5.1f} TF err {e_p:.1e}"
                best = us_p if best is None else min(best, us_p)
            print(line, flush=True)
            print("      trace: " + trace(lambda: hip.gemm_planes(pa, pb, out, layout_b=hip.KM), k // 32), flush=True)
            tot_ws += us_ws
            tot_p += best
        print(f"M={M}: four dX products: ws {tot_ws:.0f} us, P16 (k-major B from natural planes, best split) {tot_p:.0f} us", flush=True)
        # the weight gradients: out [Mo, No] = dY [M, Mo]^T . X [M, No], both k-major (natural planes); one launch per product here
        # (the product path groups a layer's four into one launch of the wave-specialised kernel: 198 - 204 us at 2432 rows)
        tot_ws = tot_p = 0.0
        for name, a, b in (("ffn2 dW", dy, x3), ("ffn1 dW", dy3, x), ("wo dW", dy, x), ("qkv dW", dyq, x)):
            k, m = a.shape
            n = b.shape[1]
            out = torch.empty(m, n, device=dev)
            ref = a.double().t() @ b.double()
            fl = 2.0 * m * n * k
            run_ws = lambda: hip.gemm(a, hip.KM, b, hip.KM, out, m, n, k, allow_split=True)
            run_ws()
            e_ws = float((out.double() - ref).abs().max() / ref.abs().max())
            us_ws = t(run_ws)
            pa, pb = hip.Planes(a, False), hip.Planes(b, False)
            line = f"M={M} {name:9s} [{m:5d}x{n:5d}x{k:5d}] | ws {us_ws:6.1f} us {fl / us_ws / 1e6:5.1f} TF err {e_ws:.1e}"
            best = None
            for sp in (1, 2, 3, 4, 6):
                if (k // 32) // sp < 4:
                    continue
                run_p = lambda: hip.gemm_planes(pa, pb, out, splits=sp, layout_a=hip.KM, layout_b=hip.KM)
                out.zero_(); run_p()
                e_p = float((out.double() - ref).abs().max() / ref.abs().max())
                us_p = t(run_p)
                line += f" | P16/s{sp} {us_p:6.1f} us {fl / us_p / 1e6:5.1f} TF err {e_p:.1e}"
                best = us_p if best is None else min(best, us_p)
            print(line, flush=True)
            tot_ws += us_ws
            tot_p += best
        print(f"M={M}: four weight-gradient products, one launch each: ws {tot_ws:.0f} us, P16 (best split) {tot_p:.0f} us", flush=True)
        # ... and the four in ONE unsplit launch: the wave-specialised kernel's grouped launch (the product path; here without its bias
        # sums) against the P16 kernel's (mtvaf_gemm_f32p_dw_group), natural and tile-blocked images
        prods = [(dy, x3), (dy3, x), (dy, x), (dyq, x)]
        outs = [torch.empty(a.shape[1], b.shape[1], device=dev) for a, b in prods]
        us_ws = t(lambda: hip.gemm_f32_dw_group([(a, b, o) for (a, b), o in zip(prods, outs)], M))
        fl = sum(2.0 * a.shape[1] * b.shape[1] * M for a, b in prods)
        line = f"M={M} grouped dW (432 tiles) | ws {us_ws:6.1f} us {fl / us_ws / 1e6:5.1f} TF"
        for blocked in (False, True):
            items = [(hip.Planes(a, blocked), hip.Planes(b, blocked), o) for (a, b), o in zip(prods, outs)]
            run_p = lambda: hip.gemm_planes_dw_group(items)
            run_p()
            err = max(float((o.double() - a.double().t() @ b.double()).abs().max() / (a.double().t() @ b.double()).abs().max()) for (a, b), o in zip(prods, outs))
            us_p = t(run_p)
            line += f" | P16 {'blocked' if blocked else 'natural'} images {us_p:6.1f} us {fl / us_p / 1e6:5.1f} TF err {err:.1e}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
