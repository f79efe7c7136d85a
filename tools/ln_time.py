import os, sys, torch
sys.path.insert(0, os.getcwd())
from mtvaf_amd import hip
dev = "cuda"
for M in (2432, 4096):
    H = 768
    g = torch.Generator(device=dev).manual_seed(1)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    dout, x, res, gamma = r(M, H), r(M, H), r(M, H), r(H)
    mean, rstd = r(M), r(M).abs() + 0.5
    dx, dres = torch.empty(M, H, device=dev), torch.empty(M, H, device=dev)
    dg, db, dbx = torch.empty(H, device=dev), torch.empty(H, device=dev), torch.empty(H, device=dev)
    f = lambda: hip.dropout_res_ln_bwd(dout, x, res, gamma, mean, rstd, dx, dres, False, dg, db, False, 0.1, 7, 3, dbias_x=dbx)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    both = e0.elapsed_time(e1) / 50 * 1e3
    part = torch.empty(int(hip.lib().mtvaf_ln_bwd_workspace_bytes(M, H)) // 4, device=dev)
    rows = lambda: hip._ck(hip.lib().mtvaf_dropout_res_ln_bwd_rows(hip._p(dout), hip._p(x), hip._p(res), hip._p(gamma), hip._p(mean), hip._p(rstd),
                                                                   hip._p(dx), hip._p(dres), 0, M, H, 0.1, 7, 3, hip._p(part), None, hip._st()), "rows")
    for _ in range(5): rows()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50): rows()
    e1.record(); torch.cuda.synchronize()
    print(M, "rows: ln_bwd + column sums", round(both, 2), "us; the row kernel alone", round(e0.elapsed_time(e1) / 50 * 1e3, 2), "us")
