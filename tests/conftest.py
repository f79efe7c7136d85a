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


@pytest.fixture(params=["unpad", "padded"])
def pad_mode(request):
    """Execution layout of `TVNetSAModel2` for one test: "unpad" = the default since round 5 (padding-free: the encoder layers
    run on the packed unmasked token rows, mtvaf_amd.engine.UNPAD), "padded" = MTVAF_UNPAD=0 (every [B, S] row computed, the
    reference's layout).  The parity tests that anchor the headline run in both, crossed with `f32_arith`; batches too small
    to pack (fewer than one 128-row tile to save) run padded in either mode."""
    from mtvaf_amd import engine
    with engine.padding_free(request.param == "unpad"):
        yield request.param
