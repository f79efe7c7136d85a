"""Would two half-batch chains on two streams hide the per-launch fixed cost of the dependent GEMM chain?
One encoder layer's four forward products (QKV, Wo, FFN-1 + GELU, FFN-2), 12 layers deep, each product reading the
previous one's output: (a) M tokens on one stream, (b) two chains of M/2 tokens on two streams, (c) four of M/4.
    python tools/chain_split_probe.py [M] [dtype]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mtvaf_amd import hip
dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H, I, L = 768, 3072, 12
w = [dict(qkv=torch.randn(3 * H, H, device=dev) * 0.02, o=torch.randn(H, H, device=dev) * 0.02,
          f1=torch.randn(I, H, device=dev) * 0.02, f2=torch.randn(H, I, device=dev) * 0.02) for _ in range(L)]
bias = dict(qkv=torch.zeros(3 * H, device=dev), o=torch.zeros(H, device=dev), f1=torch.zeros(I, device=dev), f2=torch.zeros(H, device=dev))


def make(m):
    return dict(x=torch.randn(m, H, device=dev), qkv=torch.empty(m, 3 * H, device=dev), a=torch.empty(m, H, device=dev),
                pre=torch.empty(m, I, device=dev), act=torch.empty(m, I, device=dev), y=torch.empty(m, H, device=dev))


def chain(t, m):
    for l in range(L):
        hip.gemm(t["x"], 0, w[l]["qkv"], 0, t["qkv"], m, 3 * H, H, bias=bias["qkv"])
        hip.gemm(t["qkv"], 0, w[l]["o"], 0, t["a"], m, H, H, bias=bias["o"], lda=3 * H)  # (stand-in for attention + Wo)
        hip.gemm(t["a"], 0, w[l]["f1"], 0, t["act"], m, I, H, bias=bias["f1"], epi=hip.EPI_GELU, aux=t["pre"])
        hip.gemm(t["act"], 0, w[l]["f2"], 0, t["x"], m, H, I, bias=bias["f2"])


def run(parts):
    m = M // parts
    ts = [make(m) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    def once():
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for s, t in zip(streams, ts):
            with torch.cuda.stream(s):
                chain(t, m)
        for s in streams:
            cur.wait_stream(s)
    for _ in range(2):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n):
        once()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * M * (3 * H * H + H * H + 2 * H * I) * L
    print(f"M={M} in {parts} chain(s) of {m}: {ms:.3f} ms per 12-layer forward chain, {fl / ms / 1e9:.1f} TFLOP/s", flush=True)


for parts in (1, 2, 4):
    run(parts)
