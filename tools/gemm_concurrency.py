"""Do two co-resident GEMM kernels per CU raise MFMA utilisation?  Runs the dX and dW products of one encoder
layer (FFN-2, FFN-1, attention-output, QKV) serially on one stream and concurrently on two streams (dW on the
side stream), with the planner's tiles and with the 2-stage LDS-DMA tiles (57-67 KB LDS: two different kernels
fit on one CU)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mtvaf_amd import hip
dev = "cuda"
M, H, I = 4096, 768, 3072
g = torch.Generator(device=dev).manual_seed(0)
R = lambda *s: torch.randn(*s, device=dev, generator=g)
# (dy [M,N], w [N,K], x [M,K]) per linear: N = out features, K = in features
lin = {"ffn2": (R(M, H), R(H, I), R(M, I)), "ffn1": (R(M, I), R(I, H), R(M, H)), "ao": (R(M, H), R(H, H), R(M, H)),
       "qkv": (R(M, 3 * H), R(3 * H, H), R(M, H))}
outs = {k: (torch.empty_like(x), torch.empty_like(w)) for k, (dy, w, x) in lin.items()}
s2 = torch.cuda.Stream()
ws2 = torch.empty(256 << 20, dtype=torch.uint8, device=dev)


def dx(k, cfg):
    dy, w, x = lin[k]
    hip.gemm(dy, hip.KC, w, hip.KM, outs[k][0], M, w.shape[1], w.shape[0], cfg=cfg)


def dw(k, cfg, splits):
    dy, w, x = lin[k]
    hip.gemm(dy, hip.KM, x, hip.KM, outs[k][1], w.shape[0], w.shape[1], M, allow_split=True, cfg=cfg, splits=splits)


def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def run(plan, concurrent):
    """plan: {name: (dx cfg, dw cfg, dw splits)}"""
    def fn():
        main = torch.cuda.current_stream()
        for k in ("ffn2", "ffn1", "ao", "qkv"):
            cx, cw, sw = plan[k]
            if concurrent:
                s2.wait_stream(main)
                with torch.cuda.stream(s2):
                    old = hip._ws.get(0); hip._ws[0] = ws2
                    dw(k, cw, sw)
                    hip._ws[0] = old
                dx(k, cx)
            else:
                dx(k, cx); dw(k, cw, sw)
        if concurrent:
            main.wait_stream(s2)
    return fn


fl = sum(2.0 * M * w.numel() * 2 for _, w, _ in lin.values())
auto = {k: (-1, -1, -1) for k in lin}
two = {"ffn2": (13, 12, 4), "ffn1": (12, 12, 4), "ao": (12, 12, 4), "qkv": (12, 12, 6)}
two_b = {"ffn2": (13, 12, 2), "ffn1": (12, 12, 2), "ao": (12, 12, 4), "qkv": (12, 12, 3)}
for name, plan, conc in (("serial auto", auto, False), ("concurrent auto", auto, True), ("serial 2-stage", two, False),
                         ("concurrent 2-stage", two, True), ("concurrent 2-stage, fewer splits", two_b, True)):
    t = timeit(run(plan, conc))
    print(f"{name:34s}: {t:8.1f} us per layer-backward GEMM set  {fl / t / 1e6:6.1f} TF", flush=True)
