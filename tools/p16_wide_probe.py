"""The pre-split fp32 GEMM on its 128 x 256 tile (csrc/gemm_f32pw.hip) against the 128 x 128 tile (csrc/gemm_f32p.hip): per product of
an encoder layer at a packed row count -- same bits?  time per launch (HIP events, 30 launches)?

    python tools/p16_wide_probe.py [rows ...]          # default 2432 4096
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mtvaf_amd import hip  # noqa: E402

dev = "cuda"
H, I = 768, 3072


def timed(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def both(fn, out_of):
    res = []
    for wide in (False, True):
        hip.f32p_wide(wide)
        o = out_of()
        fn(o)
        torch.cuda.synchronize()
        res.append((o, timed(lambda: fn(o))))
    hip.f32p_wide(False)
    (o0, t0), (o1, t1) = res
    same = all(torch.equal(x, y) for x, y in zip(o0, o1)) if isinstance(o0, (list, tuple)) else torch.equal(o0, o1)
    return same, t0, t1


for M in [int(a) for a in sys.argv[1:]] or [2432, 4096]:
    g = torch.Generator(device="cpu").manual_seed(M)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    print(f"== {M} rows ==")
    tot = [0.0, 0.0]
    # forward: A [M, K] . W [N, K]^T
    for name, N, K, splits in (("qkv fwd", 3 * H, H, 1), ("wo fwd", H, H, 2), ("ffn1 fwd", I, H, 1), ("ffn2 fwd", H, I, 2)):
        pa, pb = hip.Planes(rnd(M, K), True), hip.Planes(rnd(N, K), True)
        bias = rnd(N)
        same, t0, t1 = both(lambda o: hip.gemm_planes(pa, pb, o, bias=bias, splits=splits), lambda: torch.full((M, N), float("nan"), device=dev))
        fl = 2.0 * M * N * K
        print(f"{name:10s} [{M}x{N}x{K}] s{splits}: 128x128 {t0:7.1f} us {fl / t0 * 1e-6:6.1f} TF | 128x256 {t1:7.1f} us {fl / t1 * 1e-6:6.1f} TF | {'same bits' if same else 'DIFFERENT'}")
        if N == H:  # 57 wide tiles: what twice the slabs give (other bits: another summation order)
            hip.f32p_wide(True)
            o = torch.empty(M, N, device=dev)
            t2 = timed(lambda: hip.gemm_planes(pa, pb, o, bias=bias, splits=2 * splits))
            hip.f32p_wide(False)
            print(f"{'':10s} 128x256 with {2 * splits} slabs: {t2:7.1f} us")
        tot[0] += t0
        tot[1] += t1
    # dX: dY [M, K] . W [K, N] (k-major B from a tile-blocked image)
    for name, N, K, splits in (("ffn2 dX", I, H, 1), ("ffn1 dX", H, I, 2), ("wo dX", H, H, 2), ("qkv dX", H, 3 * H, 2)):
        pa, pb = hip.Planes(rnd(M, K), True), hip.Planes(rnd(K, N), True)
        same, t0, t1 = both(lambda o: hip.gemm_planes(pa, pb, o, splits=splits, layout_b=hip.KM), lambda: torch.full((M, N), float("nan"), device=dev))
        fl = 2.0 * M * N * K
        print(f"{name:10s} [{M}x{N}x{K}] s{splits}: 128x128 {t0:7.1f} us {fl / t0 * 1e-6:6.1f} TF | 128x256 {t1:7.1f} us {fl / t1 * 1e-6:6.1f} TF | {'same bits' if same else 'DIFFERENT'}")
        if N == H:
            hip.f32p_wide(True)
            o = torch.empty(M, N, device=dev)
            t2 = timed(lambda: hip.gemm_planes(pa, pb, o, splits=2 * splits, layout_b=hip.KM))
            hip.f32p_wide(False)
            print(f"{'':10s} 128x256 with {2 * splits} slabs: {t2:7.1f} us")
        tot[0] += t0
        tot[1] += t1
    # GELU epilogue with the plane image + GELU' with column partials
    pa, pb = hip.Planes(rnd(M, H), True), hip.Planes(rnd(I, H), True)
    bias = rnd(I)

    def ep(o):
        hip.gemm_planes_ep(pa, pb, o[0], bias=bias, epi=hip.EPI_GELU, aux=o[1])
    mk = lambda: (hip.Planes(torch.empty(M, I, device=dev), True, fill=False), torch.full((M, I), float("nan"), device=dev))
    res = []
    for wide in (False, True):
        hip.f32p_wide(wide)
        o = mk()
        o[0].img.fill_(float("nan"))
        ep(o)
        res.append((o, timed(lambda: ep(o))))
    hip.f32p_wide(False)
    same = torch.equal(res[0][0][0].img.view(torch.int16), res[1][0][0].img.view(torch.int16)) and torch.equal(res[0][0][1], res[1][0][1])
    print(f"ffn1 fwd GELU -> plane image: 128x128 {res[0][1]:7.1f} us | 128x256 {res[1][1]:7.1f} us | {'same bits' if same else 'DIFFERENT'}")
    pre = res[0][0][1]
    pd, pw = hip.Planes(rnd(M, H), True), hip.Planes(rnd(H, I), True)

    def ep2(o):
        hip.gemm_planes_ep(pd, pw, o[0], epi=hip.EPI_DGELU, aux=pre, colpart=o[1], layout_b=hip.KM)
    mk2 = lambda: (hip.Planes(torch.empty(M, I, device=dev), True, fill=False), torch.full((M // 128, I), float("nan"), device=dev))
    res = []
    for wide in (False, True):
        hip.f32p_wide(wide)
        o = mk2()
        o[0].img.fill_(float("nan"))
        ep2(o)
        res.append((o, timed(lambda: ep2(o))))
    hip.f32p_wide(False)
    same = torch.equal(res[0][0][0].img.view(torch.int16), res[1][0][0].img.view(torch.int16)) and torch.equal(res[0][0][1], res[1][0][1])
    print(f"ffn2 dX GELU' -> plane image + column partials: 128x128 {res[0][1]:7.1f} us | 128x256 {res[1][1]:7.1f} us | {'same bits' if same else 'DIFFERENT'}")
    # the four weight gradients of a layer as one launch: dY [M, out]^T . X [M, in]
    shapes = ((3 * H, H), (H, H), (I, H), (H, I))
    items = [(hip.Planes(rnd(M, o_), True), hip.Planes(rnd(M, i_), True)) for o_, i_ in shapes]

    def dw(o):
        hip.gemm_planes_dw_group([(a, b, c) for (a, b), c in zip(items, o)])
    same, t0, t1 = both(dw, lambda: [torch.full(s, float("nan"), device=dev) for s in shapes])
    fl = sum(2.0 * M * o_ * i_ for o_, i_ in shapes)
    print(f"grouped dW (four products, one launch): 128x128 {t0:7.1f} us {fl / t0 * 1e-6:6.1f} TF | 128x256 {t1:7.1f} us {fl / t1 * 1e-6:6.1f} TF | {'same bits' if same else 'DIFFERENT'}")
    print(f"eight forward / dX products: 128x128 {tot[0]:7.1f} us | 128x256 {tot[1]:7.1f} us")

# ---- where the time goes (forward products, blocked images): ablations and the k-tile trace of block 0, both tiles
if os.environ.get("PROBE_ABL", "1") != "0":
    from f32p_bench import trace  # noqa: E402
    for M in [int(a) for a in sys.argv[1:]] or [2432, 4096]:
        g = torch.Generator(device="cpu").manual_seed(M + 1)
        rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
        for name, N, K, lb in (("qkv fwd", 3 * H, H, hip.KC), ("ffn1 fwd", I, H, hip.KC), ("ffn2 fwd", H, I, hip.KC), ("ffn2 dX", I, H, hip.KM)):
            pa = hip.Planes(rnd(M, K), True)
            pb = hip.Planes(rnd(N, K) if lb == hip.KC else rnd(K, N), True)
            out = torch.empty(M, N, device=dev)
            for wide in (False, True):
                hip.f32p_wide(wide)
                line = f"M={M} {name:9s} {'128x256' if wide else '128x128'}: all {timed(lambda: hip.gemm_planes(pa, pb, out, layout_b=lb)):6.1f}"
                if lb == hip.KC:
                    for ab, tag in ((1, "noMFMA"), (2, "noDMA"), (4, "noRD"), (6, "MFMAonly"), (5, "DMAonly")):
                        line += f" {tag} {timed(lambda: hip.gemm_planes(pa, pb, out, ablate=ab)):.1f}"
                print(line + " | " + trace(lambda: hip.gemm_planes(pa, pb, out, layout_b=lb), K // 32), flush=True)
            hip.f32p_wide(False)
