"""Three products on the all-planes split GEMM (csrc/gemm_f32p.hip) and on the wave-specialised kernel for a rocprofv3 --pmc pass:
FFN-1 forward [4096 x 3072 x 768] (KC x KC), FFN-2 dX [4096 x 3072 x 768] (KC x KM), FFN-1 dW [3072 x 768 x 4096] (KM x KM)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mtvaf_amd import hip
dev = "cuda"
M, H, I = 4096, 768, 3072
x, w1, w2, dy, dy3 = (torch.randn(*s, device=dev) for s in ((M, H), (I, H), (H, I), (M, H), (M, I)))
out, dw = torch.empty(M, I, device=dev), torch.empty(I, H, device=dev)
px, pw1, pw2, pdy, pdy3 = (hip.Planes(t) for t in (x, w1, w2, dy, dy3))
for _ in range(6):
    hip.gemm_planes(px.img[0], 0, H, px.stride, pw1.img[0], 0, H, pw1.stride, out, M, I, H, tile_n=128)
    hip.gemm_planes(pdy.img[0], 0, H, pdy.stride, pw2.img[0], 1, I, pw2.stride, out, M, I, H, tile_n=128)
    hip.gemm_planes(pdy3.img[0], 1, I, pdy3.stride, px.img[0], 1, H, px.stride, dw, I, H, M, tile_n=128, allow_split=True, splits=3)
    hip.gemm(x, 0, w1, 0, out, M, I, H, compute="fp32x3", cfg=5)
    hip.gemm(dy, 0, w2, 1, out, M, I, H, compute="fp32x3", cfg=5)
torch.cuda.synchronize()
print("done")
