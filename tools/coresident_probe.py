"""Which HBM-bound kernels of the backward pass run INSIDE the CUs a one-round GEMM launch occupies?  The grouped weight gradients
of a layer (216 tiles of 128 x 256, 228 registers x 2 waves per SIMD, 144 KiB of LDS: one block per CU on 216 of 256 CUs) run on
a side stream; beside them one small kernel at a time on the main stream, timed by events: alone, and launched ~40 us after the
GEMMs have taken their CUs.  A kernel that fits into what the GEMM block leaves (<= 48 registers per lane, <= 16 KiB of LDS)
should keep (most of) its own duration; one that does not is confined to the 40 free CUs.

    python tools/coresident_probe.py [M, default 2432]
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mtvaf_amd import hip
dev = "cuda"
H, I = 768, 3072
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2432
DELAY = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
x, x3 = rn(M, H), rn(M, I)
dy, dy3, dyq = rn(M, H), rn(M, I), rn(M, 3 * H)
prods = [(dy, x3), (dy3, x), (dy, x), (dyq, x)]
outs = [torch.empty(a.shape[1], b.shape[1], device=dev) for a, b in prods]
items = [(hip.Planes(a, True), hip.Planes(b, True), o) for (a, b), o in zip(prods, outs)]
s2 = torch.cuda.Stream()

# the small kernels
a_, b_, c_ = rn(M, H), rn(M, H), torch.empty(M, H, device=dev)
gamma, beta = rn(H), rn(H)
mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
dx, dres = torch.empty(M, H, device=dev), torch.empty(M, H, device=dev)
dgamma, dbeta, dbias = torch.empty(H, device=dev), torch.empty(H, device=dev), torch.empty(H, device=dev)
n = 7_087_872
p, gr, m_, v_ = (torch.zeros(n, device=dev) for _ in range(4))
slab = rn(2, M, H)
small = {
    "dropout_kernel (22 regs)": lambda: hip.dropout(a_, c_, 0.1, 1, 0),
    "torch add (aten)": lambda: torch.add(a_, b_, out=c_),
    "ln_fwd (82-112 regs)": lambda: hip.dropout_res_ln_fwd(a_, b_, gamma, beta, c_, mean, rstd, 1e-12, 0.1, 1, 0),
    "ln_bwd + finish (lean: 44 regs; MTVAF_LN_LEAN=0: 169 regs, 16 KiB)": lambda: hip.dropout_res_ln_bwd(a_, b_, c_, gamma, mean, rstd, dx, dres, False, dgamma, dbeta, False, 0.1, 1, 0, dbias_x=dbias),
    "adamw (51 regs), 7.1 M parameters": lambda: hip.adamw(p, gr, m_, v_, 1e-5, 0.9, 0.999, 1e-8, 0.01, 1),
    "splitk reduce (torch sum of 2 slabs)": lambda: torch.sum(slab, 0, out=c_),
}


def ev():
    return torch.cuda.Event(enable_timing=True)


def alone(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


def beside(fn, it=10, ngemm=3, delay_cycles=DELAY):
    main = torch.cuda.current_stream()
    tot = 0.0
    for _ in range(it):
        torch.cuda.synchronize()
        s2.wait_stream(main)
        with torch.cuda.stream(s2):
            for _ in range(ngemm): hip.gemm_planes_dw_group(items)
        torch.cuda._sleep(delay_cycles)   # ~ 40 us at 100 MHz of the sleep counter
        e0, e1 = ev(), ev()
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1) * 1e3
    return tot / it


t_g = alone(lambda: hip.gemm_planes_dw_group(items))
print(f"M = {M}: grouped weight gradients alone {t_g:.1f} us per launch", flush=True)
e0, e1 = ev(), ev()
torch.cuda._sleep(DELAY); torch.cuda.synchronize(); e0.record(); torch.cuda._sleep(DELAY); e1.record(); torch.cuda.synchronize()
print(f"the delay in front of the small kernel: {e0.elapsed_time(e1) * 1e3:.1f} us")
for name, fn in small.items():
    print(f"  {name:40s} alone {alone(fn):7.1f} us | beside the grouped weight gradients {beside(fn):7.1f} us", flush=True)
