import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mtvaf_amd import hip
DEV = "cuda"
L_ = hip.lib()
B, S, Pn, NH, p = 5, 100, 36, 4, 0.1
H = NH * 64
lens = [S, 1, 53, 20, 77]
Mv = sum(lens); Mp = (Mv + 127) // 128 * 128 + 128
cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=DEV)
g = torch.Generator().manual_seed(1)
qkv = torch.zeros(Mp, 3 * H, device=DEV); qkv[:Mv] = torch.randn(Mv, 3 * H, generator=g).to(DEV)
pk, pv = torch.randn(B, Pn * H, generator=g).to(DEV), torch.randn(B, Pn * H, generator=g).to(DEV)
c1 = torch.full((Mp, H), float("nan"), device=DEV); l1 = torch.zeros(B, NH, S, device=DEV)
img = hip.Planes(c1, True, fill=False); img.img.view(torch.int16).fill_(0x7777)
hip._ck(L_.mtvaf_prefix_attn_varlen_fwd_planes(hip._p(qkv), hip._p(pk), hip._p(pv), hip._p(cu), Mp - Mv, hip._p(c1), hip._p(l1), B, S, Pn, NH, 64, p, 11, 5, hip._p(img.img), Mp, hip._st()), "f")
want = hip.Planes(c1, True)
a = img.img.view(torch.int16).view(H // 32, 3, Mp, 32); b = want.img.view(torch.int16).view(H // 32, 3, Mp, 32)
bad = (a != b)
print("bad elements", int(bad.sum()), "of", bad.numel())
idx = bad.nonzero()
print(idx[:10])
print("unwritten (0x7777):", int((a == 0x7777).sum()))
rows_bad = bad.any(3).any(1).any(0).nonzero().flatten()
print("rows with differences:", rows_bad[:20].tolist(), len(rows_bad))
planes_bad = bad.any(3).any(2).any(0)
print("planes", planes_bad.tolist())
