"""Times the prompt-mix kernels at the bench shape (NI = 9 images, B = 32, L = 4, W = 1536, NL = 12)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mtvaf_amd.hip import _ck, _p, _st, lib
NI, B, L, W, NL = 9, 32, 4, 1536, 12
dev = "cuda"
enc = torch.randn(NI, B, L, 4 * W, device=dev); gate = torch.rand(NI * B, NL * 4, device=dev)
pkv = torch.empty(NL, 2, B, NI * L * (W // 2), device=dev); dpkv = torch.randn_like(pkv)
dsm = torch.randn(NI * B, L * W, device=dev); denc = torch.empty_like(enc)
dpart = torch.empty(NI * B * L, NL * 4, device=dev); dlog = torch.empty(NI * B, NL * 4, device=dev); logits = torch.randn(NI * B, NL * 4, device=dev)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
mb = (enc.numel() + pkv.numel()) * 4 / 1e6
us = t(lambda: _ck(lib().mtvaf_prompt_mix_fwd(_p(enc), _p(gate), _p(pkv), NI, B, L, W, NL, _st()), "f"))
print(f"mix fwd     : {us:7.1f} us  {mb / us:6.2f} TB/s ({mb:.0f} MB)")
us = t(lambda: _ck(lib().mtvaf_prompt_mix_bwd_enc(_p(gate), _p(dpkv), _p(dsm), _p(denc), NI, B, L, W, NL, _st()), "b"))
print(f"mix bwd enc : {us:7.1f} us  {(mb + dsm.numel() * 4 / 1e6) / us:6.2f} TB/s")
us = t(lambda: _ck(lib().mtvaf_prompt_mix_bwd_gate(_p(enc), _p(dpkv), _p(logits), _p(gate), _p(dpart), _p(dlog), NI, B, L, W, NL, _st()), "g"))
print(f"mix bwd gate: {us:7.1f} us  {mb / us:6.2f} TB/s")
