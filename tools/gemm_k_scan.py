import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mtvaf_amd import hip
from tools.gemm_sweep import time_call
dev = "cuda"
M, N = 4096, 768
for K in (768, 3072, 12288, 49152):
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
    for cfg in (9, 12):
        us = time_call(lambda: hip.gemm(A, 0, B, 0, C, M, N, K, cfg=cfg), iters=10)
        print(f"K={K:6d} cfg={hip.TILE_NAMES[cfg]:13s} {us:9.1f} us  {2.0*M*N*K/us/1e6:6.1f} TF  ({us/(K/32):.3f} us per k-tile)")
