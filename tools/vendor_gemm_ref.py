"""Reference point only (NOT used by the product path): what the vendor fp32 GEMM (hipBLASLt/rocBLAS through
torch.mm) reaches on the shapes of one encoder layer, next to mtvaf_gemm_f32."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mtvaf_amd import hip
from tools.gemm_sweep import time_call
torch.backends.cuda.matmul.allow_tf32 = False
dev = "cuda"
M, H, I = 4096, 768, 3072
for name, m, n, k in [("qkv_fwd", M, 3 * H, H), ("ao_fwd", M, H, H), ("ffn1_fwd", M, I, H), ("ffn2_fwd", M, H, I)]:
    x, w = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev)
    out = torch.empty(m, n, device=dev)
    us_v = time_call(lambda: torch.mm(x, w.t(), out=out), iters=30)
    us_m = time_call(lambda: hip.gemm(x, 0, w, 0, out, m, n, k), iters=30)
    print(f"{name:9s} nt  vendor {us_v:7.1f} us {2.0*m*n*k/us_v/1e6:6.1f} TF | mtvaf {us_m:7.1f} us {2.0*m*n*k/us_m/1e6:6.1f} TF")
for name, m, n, k in [("ffn1_dw", I, H, M), ("ao_dw", H, H, M)]:
    dy, x = torch.randn(k, m, device=dev), torch.randn(k, n, device=dev)
    out = torch.empty(m, n, device=dev)
    us_v = time_call(lambda: torch.mm(dy.t(), x, out=out), iters=30)
    us_m = time_call(lambda: hip.gemm(dy, 1, x, 1, out, m, n, k, allow_split=True), iters=30)
    print(f"{name:9s} tn  vendor {us_v:7.1f} us {2.0*m*n*k/us_v/1e6:6.1f} TF | mtvaf {us_m:7.1f} us {2.0*m*n*k/us_m/1e6:6.1f} TF")
