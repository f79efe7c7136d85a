"""Do two co-resident GEMM kernels per CU (2 waves/SIMD) raise MFMA utilisation?  Runs the dX (LDS-DMA kernel,
84 KB LDS) and dW (register-staged kernel, 64.5 KB LDS) products of one FFN layer serially on one stream and
concurrently on two streams."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mtvaf_amd import hip
dev = "cuda"
M, H, I = 4096, 768, 3072
dy = torch.randn(M, H, device=dev); w2 = torch.randn(H, I, device=dev); act = torch.randn(M, I, device=dev)
dpre = torch.empty(M, I, device=dev); dw2 = torch.empty(H, I, device=dev)
dpre_in = torch.randn(M, I, device=dev); w1 = torch.randn(I, H, device=dev); h1 = torch.randn(M, H, device=dev)
dh1 = torch.empty(M, H, device=dev); dw1 = torch.empty(I, H, device=dev)
s2 = torch.cuda.Stream()
ws2 = torch.empty(64 << 20, dtype=torch.uint8, device=dev)

def dx_a(): hip.linear_bwd_input(dy, w2, dpre)            # [M,I] = dy[M,H] . w2[H,I]   (K = 768)
def dw_a(): hip.linear_bwd_weight(dy, act, dw2)           # [H,I] = dy^T act            (K = 4096, split)
def dx_b(): hip.linear_bwd_input(dpre_in, w1, dh1)        # [M,H] = dpre[M,I] . w1[I,H] (K = 3072)
def dw_b(): hip.linear_bwd_weight(dpre_in, h1, dw1)       # [I,H]

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

def serial(): dx_a(); dw_a(); dx_b(); dw_b()

def concurrent():
    main = torch.cuda.current_stream()
    s2.wait_stream(main)
    with torch.cuda.stream(s2):
        # side stream needs its own split-K workspace
        old = hip._ws.get(0); hip._ws[0] = ws2
        dw_a(); dw_b()
        hip._ws[0] = old
    dx_a(); dx_b()
    main.wait_stream(s2)

fl = 2.0 * M * H * I * 4
ts, tc = timeit(serial), timeit(concurrent)
print(f"serial     : {ts:8.1f} us  {fl/ts/1e6:6.1f} TF")
print(f"concurrent : {tc:8.1f} us  {fl/tc/1e6:6.1f} TF   speed-up {ts/tc:.3f}x")
