"""Two product shapes on the split-fp32 GEMM (csrc/gemm_f32x3.hip: the wave-specialised 128x128 kernel) for a rocprofv3 --pmc pass:
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -- python3 tools/f32x3_pmc.py
FFN-1 forward [4096 x 3072 x 768] (KC x KC) and its weight gradient [3072 x 768 x 4096] (KM x KM)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mtvaf_amd import hip
dev = "cuda"
M, H, I = 4096, 768, 3072
x, w1, dy3 = torch.randn(M, H, device=dev), torch.randn(I, H, device=dev), torch.randn(M, I, device=dev)
out, dw = torch.empty(M, I, device=dev), torch.empty(I, H, device=dev)
for _ in range(10):
    hip.gemm(x, hip.KC, w1, hip.KC, out, M, I, H, compute="fp32x3")
    hip.gemm(dy3, hip.KM, x, hip.KM, dw, I, H, M, allow_split=True, compute="fp32x3")
torch.cuda.synchronize()
print("done")
