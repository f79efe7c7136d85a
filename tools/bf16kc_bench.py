"""bf16-operand DMA GEMM (mtvaf_gemm_bf16kc): correctness against fp32 products of the rounded operands, and rate."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mtvaf_amd import hip
dev = "cuda"
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator(device=dev).manual_seed(0)
for name, M, N, K, split in [("qkv_fwd", 4096, 2304, 768, 0), ("ao_fwd", 4096, 768, 768, 0), ("ffn1_fwd", 4096, 3072, 768, 0),
                             ("ffn2_fwd", 4096, 768, 3072, 0), ("ffn_dw", 3072, 768, 4096, 1), ("ao_dw", 768, 768, 4096, 1),
                             ("qkv_dw", 2304, 768, 4096, 1), ("big", 65536, 768, 3072, 0)]:
    A = torch.randn(M, K, device=dev, generator=g); B = torch.randn(N, K, device=dev, generator=g)
    Ah, Bh = torch.empty(M, K, dtype=torch.bfloat16, device=dev), torch.empty(N, K, dtype=torch.bfloat16, device=dev)
    hip.cast_bf16(A, out=Ah); hip.cast_bf16(B, out=Bh)
    assert torch.equal(Ah, A.to(torch.bfloat16)) and torch.equal(Bh, B.to(torch.bfloat16))
    bias = torch.randn(N, device=dev, generator=g)
    C = torch.empty(M, N, device=dev)
    hip.gemm_bf16kc(Ah, Bh, C, bias=None if split else bias, allow_split=bool(split))
    ref = Ah.float() @ Bh.float().t() + (0 if split else bias)
    err = float((C - ref).abs().max()) / float(ref.abs().max())
    us = t(lambda: hip.gemm_bf16kc(Ah, Bh, C, bias=None if split else bias, allow_split=bool(split)))
    usc = t(lambda: hip.cast_bf16(A, out=Ah))
    print(f"{name:9s} M={M:6d} N={N:5d} K={K:5d}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF   rel err {err:.1e}   cast A {usc:6.1f} us "
          f"({A.numel() * 6 / usc / 1e6:.2f} TB/s)", flush=True)
# transposed cast
X = torch.randn(4096, 3072, device=dev); XT = torch.empty(3072, 4096, dtype=torch.bfloat16, device=dev); Xh = torch.empty(4096, 3072, dtype=torch.bfloat16, device=dev)
hip.cast_bf16(X, out=Xh, out_t=XT)
assert torch.equal(XT, X.t().contiguous().to(torch.bfloat16)) and torch.equal(Xh, X.to(torch.bfloat16))
print(f"cast + transpose 4096x3072: {t(lambda: hip.cast_bf16(X, out=Xh, out_t=XT)):6.1f} us")
