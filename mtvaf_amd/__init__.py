"""mtvaf_amd -- MI355X-native (gfx950) implementation of MTVAF's prefix-fused BERT/RoBERTa hot path.

The arithmetic lives in hand-written HIP kernels behind a C ABI (``include/mtvaf_hip.h``,
``mtvaf_amd/csrc``); the Python modules under ``mtvaf_amd.models`` keep the reference's nn.Module
surface so they drop into ``MTVAF_training.py`` / ``modules/train.py``.
"""
__version__ = "0.1.0"
