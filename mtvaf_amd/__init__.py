"""mtvaf_amd -- MI355X-native (gfx950) implementation of MTVAF's prefix-fused BERT/RoBERTa hot path.

The arithmetic lives in hand-written HIP kernels behind a C ABI (``include/mtvaf_hip.h``,
``mtvaf_amd/csrc``); the Python modules under ``mtvaf_amd.models`` keep the reference's nn.Module
surface so they drop into ``MTVAF_training.py`` / ``modules/train.py``.
"""
__version__ = "0.1.0"


def set_compute_dtype(dtype: str) -> None:
    """Process-wide arithmetic of the dense projections: "fp32" (default, the reference's precision) or "bf16"
    (BASELINE configs 3-4: operands rounded to bf16 inside the GEMM kernels, fp32 accumulation, fp32 master
    weights / activations / LayerNorm / softmax statistics).  Also settable with MTVAF_COMPUTE_DTYPE."""
    from . import hip
    hip.set_compute_dtype(dtype)


import os as _os

if _os.environ.get("MTVAF_COMPUTE_DTYPE"):
    set_compute_dtype(_os.environ["MTVAF_COMPUTE_DTYPE"])
