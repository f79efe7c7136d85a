import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
B, S, Pn, NH, H = (int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (32, 128, 36, 12, 768)))
dev = "cuda"
qkv = torch.randn(B * S, 3 * H, device=dev); pk = torch.randn(B, Pn * H, device=dev); pv = torch.randn(B, Pn * H, device=dev)
am = torch.zeros(B, Pn + S, device=dev); ctx = torch.empty(B * S, H, device=dev); lse = torch.empty(B, NH, S, device=dev)
P = ctypes.c_void_p
for v in range(5):
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"attn_v{v}.so")
    if not os.path.exists(path):
        continue
    lib = ctypes.CDLL(path)
    fn = lib.mtvaf_prefix_attn_fwd
    fn.argtypes = [P, P, P, P, P, P] + [ctypes.c_int] * 5 + [ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64, P]
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: fn(qkv.data_ptr(), pk.data_ptr(), pv.data_ptr(), am.data_ptr(), ctx.data_ptr(), lse.data_ptr(), B, S, Pn, NH, 64, 0.1, 1, 2, st)
    for _ in range(5): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): call()
    e1.record(); torch.cuda.synchronize()
    print(f"variant {v}: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us", flush=True)
