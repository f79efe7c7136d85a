import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(params=["split", "pipe"])
def f32_arith(request):
    """fp32 GEMM arithmetic of the library for one test: "split" = the default (six exact bf16 partial products of three-way
    split fp32 operands, csrc/gemm_f32x3.hip), "pipe" = the fp32 MFMA pipe forced (MTVAF_F32_SPLIT=0).  The parity tests
    that anchor the headline run in both; every other -m gpu test runs the default."""
    from mtvaf_amd import hip
    was = hip.f32_split()
    hip.f32_split(request.param == "split")
    try:
        yield request.param
    finally:
        hip.f32_split(was)
