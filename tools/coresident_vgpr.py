"""How many registers may a streaming kernel use and still run INSIDE the CUs a 228-register GEMM block occupies?  The kernels of
tools/micro/vgpr_fit.hip (NV float4 loads in flight per thread: 12 .. 66 VGPRs) alone and beside the grouped weight gradients of a
layer; run under rocprofv3 --kernel-trace:  PROBE=tools/coresident_vgpr.py bash tools/coresident_run.sh"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mtvaf_amd import hip
dev = "cuda"
H, I, M = 768, 3072, 2432
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "libvgpr_fit.so"))
lib.vgpr_fit_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
x, x3 = rn(M, H), rn(M, I)
dy, dy3, dyq = rn(M, H), rn(M, I), rn(M, 3 * H)
prods = [(dy, x3), (dy3, x), (dy, x), (dyq, x)]
outs = [torch.empty(a.shape[1], b.shape[1], device=dev) for a, b in prods]
items = [(hip.Planes(a, True), hip.Planes(b, True), o) for (a, b), o in zip(prods, outs)]
s2 = torch.cuda.Stream()
n4 = M * H // 4
src = rn(14, M, H)
dst = torch.empty(M, H, device=dev)
DELAY = 4000


def launch(nv):
    rc = lib.vgpr_fit_launch(nv, src.data_ptr(), dst.data_ptr(), n4, n4, (n4 + 255) // 256, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


for nv in (2, 4, 5, 6, 7, 8, 9, 10, 14):
    for _ in range(8):  # alone
        launch(nv)
    torch.cuda.synchronize()
    main = torch.cuda.current_stream()
    for _ in range(10):  # beside the GEMM launches
        torch.cuda.synchronize()
        s2.wait_stream(main)
        with torch.cuda.stream(s2):
            for _ in range(3): hip.gemm_planes_dw_group(items)
        torch.cuda._sleep(DELAY)
        launch(nv)
    torch.cuda.synchronize()
print("done")
