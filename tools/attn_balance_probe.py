"""Is the fp32 varlen attention launch bound by the IMBALANCE of its blocks (a launch lasts as long as its busiest CU)?  The
bench's ragged batch (lengths of bench.synthetic_batch, seed 1234) against batches of the SAME total token count with equal
lengths, and against the work-proportional time of a full-length batch.

    python tools/attn_balance_probe.py
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mtvaf_amd import hip
dev = "cuda"
NH, H, S, P, B = 12, 768, 128, 36, 32


def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3


def run(lens, tag):
    lens = [int(x) for x in lens]
    Mv = sum(lens)
    Mp = (Mv + 127) // 128 * 128
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
    qkv = torch.zeros(Mp, 3 * H, device=dev)
    qkv[:Mv] = torch.randn(Mv, 3 * H, device=dev) * 0.5
    pk, pv = torch.randn(B, P * H, device=dev) * 0.02, torch.randn(B, P * H, device=dev) * 0.02
    ctx, lse = torch.empty(Mp, H, device=dev), torch.zeros(B, NH, S, device=dev)
    dctx = torch.zeros(Mp, H, device=dev)
    dctx[:Mv] = torch.randn(Mv, H, device=dev)
    delta, dqkv = torch.zeros(B, NH, S, device=dev), torch.empty(Mp, 3 * H, device=dev)
    dpk, dpv = torch.empty_like(pk), torch.empty_like(pv)
    fwd = lambda: hip.prefix_attn_varlen_fwd(qkv, pk, pv, cu, Mp - Mv, ctx, lse, B, S, P, NH, 0.1, 1234, 7)
    bwd = lambda: hip.prefix_attn_varlen_bwd(dctx, qkv, pk, pv, cu, Mp - Mv, ctx, lse, delta, dqkv, dpk, dpv, B, S, P, NH, 0.1, 1234, 7)
    tf = t(fwd)
    tb = t(bwd)
    flops = sum(4 * n * (n + P) * H for n in lens)
    units = sum(((n + 63) // 64) * ((n + P + 63) // 64) for n in lens)
    print(f"{tag:34s} tokens {Mv:5d}  flops {flops / 1e9:6.3f} G  (q-tile x key-tile) units {units:4d}  max per sentence {max(((n + 63) // 64) * ((n + P + 63) // 64) for n in lens)}"
          f"  forward {tf:6.1f} us  backward {tb:6.1f} us", flush=True)


g = torch.Generator().manual_seed(1234)
lens = torch.randint(16, S + 1, (B,), generator=g)
lens[0] = S
lens = lens.tolist()
tot = sum(lens)
run(lens, "bench batch (ragged 16..128)")
run(sorted(lens, reverse=True), "the same, longest first")
eq = [tot // B + (1 if i < tot % B else 0) for i in range(B)]
run(eq, "equal lengths, same tokens")
run([64] * B, "all 64")
run([92] * B, "all 92")
run([128] * B, "all 128 (full length)")
run([28] * B, "all 28 (one key tile)")
