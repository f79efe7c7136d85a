"""hip.kernel_symbol: the launch profiler's (cfg, layouts, fast) keys name the template instantiations rocprofv3 reports --
`bench.py` joins its HIP-event timings with `profiles/pmc_gemm*.json` through these strings (no GPU needed)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_kernel_symbols_of_every_gemm_family():
    from mtvaf_amd import hip
    assert hip.kernel_symbol(12, 1, 1, 2 + 8) == "gemm_f32_dma_kernel<128, 96, 4, 1, true, true, 2, true>"
    assert hip.kernel_symbol(9, 0, 1, 2) == "gemm_f32_dma_kernel<128, 96, 4, 1, false, true, 3, false>"
    assert hip.kernel_symbol(6, 0, 0, 2) == "gemm_f32_kernel<128, 96, 4, 1, 32, false, false, true, true>"
    assert hip.kernel_symbol(105, 0, 1, 2) == "gemm_bf16_kernel<128, 128, 2, 2, false, true, true>"
    # split-fp32 kernels (csrc/gemm_f32x3.hip)
    assert hip.kernel_symbol(225, 0, 0, 2) == "gemm_f32x3_ws_kernel<false, false, false, 128, false>"
    assert hip.kernel_symbol(225, 1, 1, 2 + 8) == "gemm_f32x3_ws_kernel<true, true, true, 128, false>"
    assert hip.kernel_symbol(226, 0, 1, 2) == "gemm_f32x3_ws_kernel<false, true, false, 96, false>"
    assert hip.kernel_symbol(206, 0, 0, 2) == "gemm_f32x3_kernel<128, 96, 4, 1, false, false, false, 32>"
    assert hip.kernel_symbol(205, 0, 1, 2) == "gemm_f32x3_kernel<128, 128, 2, 2, false, true, false, 32>"
    assert hip.kernel_symbol(203, 1, 1, 2) == "gemm_f32x3_kernel<64, 64, 2, 2, true, true, false, 32>"
    # pre-split operands (csrc/gemm_f32p.hip): forward, dX, the grouped weight gradients
    assert hip.kernel_symbol(400, 0, 0, 2) == "gemm_f32p16_kernel<0, false, false, false, false>"
    assert hip.kernel_symbol(408, 0, 1, 2) == "gemm_f32p16_kernel<0, false, true, false, false>"
    assert hip.kernel_symbol(428, 1, 1, 2) == "gemm_f32p16_kernel<0, false, true, true, true>"
    # bf16-operand kernels: the 256 x 256 eight-phase kernel and the rings
    assert hip.kernel_symbol(300 + 64 + 4 + 8 + 128, 1, 1, 2) == "gemm_bf16_p256_kernel<true, true, true>"
    assert hip.kernel_symbol(300 + 1 + 8, 0, 1, 2) == "gemm_bf16x_kernel<128, 128, 2, 2, false, true, 2, false>"
    # the grouped weight-gradient launch of the fp32 LDS-DMA kernel has a key of its own (1000 + tile)
    assert hip.kernel_symbol(1012, 1, 1, 2 + 8) == "gemm_f32_dma_group_kernel<128, 96, 4, 1, 2, true>"
    assert hip.kernel_symbol(1225, 1, 1, 2 + 8) == "gemm_f32x3_ws_kernel<true, true, true, 128, true>"
    assert hip.kernel_symbol(1225, 1, 1, 2) == "gemm_f32x3_ws_kernel<true, true, false, 128, true>"
