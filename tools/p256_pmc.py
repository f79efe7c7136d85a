"""One product shape on the 256x256 eight-phase bf16 kernel, tile per block, for a rocprofv3 --pmc pass:
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES -- python3 tools/p256_pmc.py
FFN-2 forward [65536 x 768 x 3072] (48 k-tiles per tile, 3 rounds of 256 tiles; plain fp32 result: no epilogue arithmetic) and
the weight gradient [768 x 3072 x 65536] over 7 splits, uniform random operands."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mtvaf_amd import hip
dev = "cuda"
M, H, I = 65536, 768, 3072
bf = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).to(torch.bfloat16)
x3, w2, dy = bf(M, I), bf(H, I), bf(M, H)
out, bias = torch.empty(M, H, device=dev), torch.randn(H, device=dev)
dw = torch.empty(H, I, device=dev)
for _ in range(10):
    hip.gemm_bf16x(x3, hip.KC, w2, hip.KC, M, H, I, out32=out, bias=bias, tile=5)
    hip.gemm_bf16x(dy, hip.KM, x3, hip.KM, H, I, M, out32=dw, allow_split=True, tile=5, splits=7)
torch.cuda.synchronize()
print("done")
