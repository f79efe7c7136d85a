"""Adversarial accuracy cases of the split-fp32 GEMM (csrc/gemm_f32x3.hip) against the fp64 product, beside the fp32 MFMA
pipe on the same operands (the numbers behind tests/test_ops_gpu.py::test_gemm_f32_split_adversarial).

    python tools/f32x3_adversarial.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from mtvaf_amd import hip  # noqa: E402
from x3_cases import CASES, operands  # noqa: E402

DEV = "cuda:0"
for case in CASES:
    for la, lb in ((0, 0), (0, 1), (1, 1)):
        M, N, K = 256, 384, 512
        A, B = operands(case, M, N, K, seed=5 + la + 2 * lb)
        ref = A.double() @ B.double()
        mag = A.double().abs() @ B.double().abs()
        a = (A if la == 0 else A.t().contiguous()).to(DEV)
        b = (B.t().contiguous() if lb == 0 else B).to(DEV)
        o_nat, o_spl = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
        hip.f32_split(False)
        hip.gemm(a, la, b, lb, o_nat, M, N, K)
        hip.f32_split(True)
        hip.gemm(a, la, b, lb, o_spl, M, N, K, compute="fp32x3", cfg=5)
        en = (o_nat.double().cpu() - ref).abs() / mag
        es = (o_spl.double().cpu() - ref).abs() / mag
        fin = bool(torch.isfinite(o_spl).all()), bool(torch.isfinite(o_nat).all())
        print(f"{case:22s} la{la} lb{lb}: split max {float(es.max()):.3e} rms {float(es.pow(2).mean().sqrt()):.3e} | pipe max "
              f"{float(en.max()):.3e} rms {float(en.pow(2).mean().sqrt()):.3e} | 2^-24 = {2.0 ** -24:.3e}  finite {fin}  "
              f"|ref| max {float(ref.abs().max()):.3e} mag max {float(mag.max()):.3e}", flush=True)
