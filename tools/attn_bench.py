"""fp32 prefix attention (mtvaf_prefix_attn_fwd / _bwd) at the headline shape: time per layer, padded launch with ragged masks.

    python tools/attn_bench.py [B S P]
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mtvaf_amd import hip
dev = "cuda"


def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3


def main():
    a = [int(x) for x in sys.argv[1:]]
    B, S, P = (a + [32, 128, 36])[:3] if len(a) >= 3 else (32, 128, 36)
    NH, H = 12, 768
    g = torch.Generator().manual_seed(1)
    lens = torch.randint(16, S + 1, (B,), generator=g); lens[0] = S
    mask = (torch.arange(S)[None, :] < lens[:, None]).float()
    addmask = torch.cat([torch.zeros(B, P), (1 - mask) * -10000.0], 1).to(dev).contiguous()
    qkv = torch.randn(B * S, 3 * H, device=dev) * 0.5
    pk, pv = torch.randn(B, P * H, device=dev) * 0.02, torch.randn(B, P * H, device=dev) * 0.02
    ctx, lse = torch.empty(B * S, H, device=dev), torch.empty(B, NH, S, device=dev)
    dctx = torch.randn(B * S, H, device=dev) * mask.reshape(-1, 1).to(dev)
    delta, dqkv = torch.empty(B, NH, S, device=dev), torch.empty(B * S, 3 * H, device=dev)
    dpk, dpv = torch.empty_like(pk), torch.empty_like(pv)
    fwd = lambda: hip.prefix_attn_fwd(qkv, pk, pv, addmask, ctx, lse, B, S, P, NH, 0.1, 1234, 7)
    fwd()
    for zt in (False, True):
        bwd = lambda: hip.prefix_attn_bwd(dctx, qkv, pk, pv, addmask, ctx, lse, delta, dqkv, dpk, dpv, B, S, P, NH, 0.1, 1234, 7, zero_tail=zt)
        print(f"B={B} S={S} P={P}: forward {t(fwd):.1f} us, backward (zero_tail={zt}) {t(bwd):.1f} us", flush=True)


if __name__ == "__main__":
    main()
