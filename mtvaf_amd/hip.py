"""ctypes binding of the C-ABI shared library (include/mtvaf_hip.h) + thin tensor-level wrappers.

The product path has NO fallback: if the library is missing or a call fails, a RuntimeError is raised.
PyTorch only supplies device memory (``tensor.data_ptr()``) and the current HIP stream.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_double, c_float, c_int, c_int64, c_long, c_size_t, c_uint64, c_void_p
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MTVAF_LIB: another build of the library (an A/B variant made by `MTVAF_LIBDIR=... MTVAF_EXTRA_FLAGS=-D... python -m mtvaf_amd.build`:
# ablation builds never overwrite the product library)
LIB_PATH = os.environ.get("MTVAF_LIB") or os.path.join(_HERE, "lib", "libmtvaf_hip.so")

_ERR = {-1: "bad shape", -2: "bad alignment", -3: "bad argument", -4: "workspace too small"}

P, I, L, F, U64, SZ = c_void_p, c_int, c_long, c_float, c_uint64, c_size_t

_SIGS = {
    "mtvaf_version": (c_int, []),
    "mtvaf_encoder_layer_fwd": (c_int, [P, P]),
    "mtvaf_encoder_layer_bwd": (c_int, [P, P, P, P, I]),
    "mtvaf_layer_struct_bytes": (SZ, [I]),
    "mtvaf_rng_set_epoch_ptr": (c_int, [P]),
    "mtvaf_rng_epoch_advance": (c_int, [P, P]),
    "mtvaf_device_cus": (c_int, []),
    "mtvaf_gemm_f32_workspace_bytes": (SZ, [I, I, I, I]),
    "mtvaf_prof_start": (c_int, [I]),
    "mtvaf_prof_stop": (c_int, [P, P, P, I]),
    "mtvaf_gemm_f32_plan": (c_int, [I, I, I, I, I, I, I, P, P]),
    "mtvaf_gemm_f32": (c_int, [I, I, P, I, P, I, P, I, I, I, I, P, I, P, I, I, I, P, SZ, I, I, P]),
    "mtvaf_gemm_f32x3": (c_int, [I, I, P, I, P, I, P, I, I, I, I, P, I, P, I, I, I, P, SZ, I, I, P]),
    "mtvaf_f32_split": (c_int, [I]),
    "mtvaf_f32x3_trace": (c_int, [P]),
    "mtvaf_gemm_f32_slabs": (c_int, [I, I, P, I, P, I, P, I, I, I, I, P, I, P, SZ, P, P]),
    "mtvaf_dropout_res_ln_fwd_slabs": (c_int, [P, I, P, P, P, P, P, P, P, P, I, I, F, F, U64, U64, P, P]),
    "mtvaf_dropout_res_ln_bwd_rows_slabs": (c_int, [P, P, I, P, P, P, P, P, P, P, I, I, I, F, U64, U64, P, P, P]),
    "mtvaf_dropout_res_ln_fwd_planes": (c_int, [P, I, P, P, P, P, P, P, P, P, I, I, F, F, U64, U64, P, P]),
    "mtvaf_dropout_res_ln_bwd_rows_planes": (c_int, [P, P, I, P, P, P, P, P, P, P, I, I, I, F, U64, U64, P, P, P]),
    "mtvaf_f32_split_planes": (c_int, [P, P, I, I, I, L, L, L, P]),
    "mtvaf_gemm_f32p": (c_int, [I, P, L, L, L, L, I, P, L, L, L, L, P, I, I, I, I, P, I, P, I, I, I, P, SZ, I, P]),
    "mtvaf_f32p_trace": (c_int, [P]),
    "mtvaf_f32p_wide": (c_int, [I]),
    "mtvaf_gemm_f32p_dw_group": (c_int, [I, P, P, P, P, P, P, P, I, P]),
    "mtvaf_gemm_f32p_dw_group_colsum": (c_int, [I, P, P, P, P, P, P, P, I, I, P, P, P, P, P, P]),
    "mtvaf_gemm_f32p_slabs": (c_int, [I, P, L, L, L, L, I, P, L, L, L, L, P, I, I, I, I, P, I, I, P, SZ, P, P]),
    "mtvaf_gemm_f32p_ep": (c_int, [I, P, L, L, L, L, I, P, L, L, L, L, P, I, P, P, I, I, I, P, I, P, I, I, P]),
    "mtvaf_gemm_bf16": (c_int, [I, I, P, I, P, I, P, I, I, I, I, P, I, P, I, I, I, P, SZ, I, I, P]),
    "mtvaf_prefix_attn_fwd": (c_int, [P, P, P, P, P, P, I, I, I, I, I, F, U64, U64, P]),
    "mtvaf_prefix_attn_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, U64, U64, P]),
    "mtvaf_prefix_attn_bwd_tail": (c_int, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, U64, U64, I, P]),
    "mtvaf_prefix_attn_varlen_fwd": (c_int, [P, P, P, P, I, P, P, I, I, I, I, I, F, U64, U64, P]),
    "mtvaf_prefix_attn_varlen_bwd": (c_int, [P, P, P, P, P, I, P, P, P, P, P, P, I, I, I, I, I, F, U64, U64, P]),
    "mtvaf_prefix_attn_varlen_fwd_planes": (c_int, [P, P, P, P, I, P, P, I, I, I, I, I, F, U64, U64, P, I, P]),
    "mtvaf_prefix_attn_varlen_bwd_planes": (c_int, [P, P, P, P, P, I, P, P, P, P, P, P, I, I, I, I, I, F, U64, U64, P, I, P]),
    "mtvaf_prefix_attn_bf16_varlen_fwd": (c_int, [P, P, P, P, I, P, P, I, I, I, I, I, F, U64, U64, P]),
    "mtvaf_prefix_attn_bf16_varlen_bwd": (c_int, [P, P, P, P, P, I, P, P, P, P, P, P, P, I, I, I, I, I, F, U64, U64, P]),
    "mtvaf_gather_rows": (c_int, [P, P, P, I, I, P]),
    "mtvaf_build_packing": (c_int, [P, I, I, I, I, P, P, P, P, P]),
    "mtvaf_build_packing_ordered": (c_int, [P, I, I, I, I, P, P, P, P, P]),
    "mtvaf_build_ktiles": (c_int, [P, I, I, I, I, I, P, P, P]),
    "mtvaf_gemm_f32_ktiles": (c_int, [I, I, P, I, P, I, P, I, I, I, I, P, I, P, I, I, I, P, SZ, I, I, P, P, P]),
    "mtvaf_zero_f32": (c_int, [P, L, P]),
    "mtvaf_prefix_attn_bf16_fwd": (c_int, [P, P, P, P, P, P, I, I, I, I, I, F, U64, U64, P]),
    "mtvaf_prefix_attn_bf16_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, U64, U64, P]),
    "mtvaf_prefix_attn_bf16_bwd_tail": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, U64, U64, I, P]),
    "mtvaf_ln_bwd_workspace_bytes": (SZ, [I, I]),
    "mtvaf_roberta_position_ids": (c_int, [P, P, I, I, I, P]),
    "mtvaf_embed_ln_fwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, F, F, U64, U64, P, P]),
    "mtvaf_embed_ln_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, F, U64,
                                   U64, P, P, SZ, P]),
    "mtvaf_dropout_res_ln_fwd": (c_int, [P, P, P, P, P, P, P, I, I, F, F, U64, U64, P, P]),
    "mtvaf_dropout_res_ln_bwd": (c_int, [P, P, P, P, P, P, P, P, I, P, P, P, I, I, I, F, U64, U64, P, SZ, P, P]),
    "mtvaf_dropout_res_ln_bwd_rows": (c_int, [P, P, P, P, P, P, P, P, I, I, I, F, U64, U64, P, P, P]),
    "mtvaf_dropout_res_ln_bwd_finish": (c_int, [P, I, I, P, P, P, I, P]),
    "mtvaf_colsum_workspace_bytes": (SZ, [I, I]),
    "mtvaf_colsum": (c_int, [P, I, I, I, P, I, P, SZ, P]),
    "mtvaf_dropout": (c_int, [P, P, L, F, U64, U64, P]),
    "mtvaf_crf_workspace_bytes": (SZ, [I, I, I]),
    "mtvaf_crf_nll_fwd": (c_int, [P, P, P, P, P, P, P, I, I, I, P, SZ, P]),
    "mtvaf_crf_nll_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P, SZ, P]),
    "mtvaf_crf_viterbi": (c_int, [P, P, P, P, P, P, P, I, I, I, P]),
    "mtvaf_split_mean": (c_int, [P, P, L, I, P]),
    "mtvaf_gate_fwd": (c_int, [P, P, L, P]),
    "mtvaf_prompt_mix_fwd": (c_int, [P, P, P, I, I, I, I, I, P]),
    "mtvaf_prompt_mix_bwd_gate": (c_int, [P, P, P, P, P, P, I, I, I, I, I, P]),
    "mtvaf_prompt_mix_bwd_enc": (c_int, [P, P, P, P, I, I, I, I, I, P]),
    "mtvaf_kl_logsoftmax_fwd": (c_int, [P, P, P, P, I, I, P]),
    "mtvaf_kl_logsoftmax_bwd": (c_int, [P, F, P, P, P, I, I, P]),
    "mtvaf_mean_l_fwd": (c_int, [P, P, L, I, I, P]),
    "mtvaf_mean_l_bwd": (c_int, [P, P, L, I, I, P]),
    "mtvaf_span_index_ints": (SZ, [I, I, I]),
    "mtvaf_span_index": (c_int, [P, P, P, P, I, I, I, P]),
    "mtvaf_span_pool_fwd": (c_int, [P, P, P, P, P, P, I, I, I, I, P]),
    "mtvaf_span_pool_bwd_workspace_bytes": (SZ, [I, I, I, I]),
    "mtvaf_span_pool_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, P, SZ, P]),
    "mtvaf_distant_ce_fwd": (c_int, [P, I, P, P, P, I, I, F, I, P]),
    "mtvaf_distant_ce_bwd": (c_int, [P, F, P, I, P, P, P, I, I, I, P]),
    "mtvaf_ce_fwd": (c_int, [P, P, P, P, I, I, P]),
    "mtvaf_ce_bwd": (c_int, [P, P, P, P, P, I, I, P]),
    "mtvaf_mask_mul": (c_int, [P, P, P, P, I, I, I, P]),
    "mtvaf_gemm_bf16x": (c_int, [I, I, P, I, P, I, P, I, P, I, I, I, I, P, I, P, I, I, P, I, P, SZ, I, I, I, P]),
    "mtvaf_gemm_bf16x_ktiles": (c_int, [I, I, P, I, P, I, P, I, P, I, I, I, I, P, I, P, I, I, P, I, P, SZ, I, I, I, P, P, P]),
    "mtvaf_colsum_small": (c_int, [P, I, I, P, I, P]),
    "mtvaf_embed_scatter_mode": (c_int, [I]),
    "mtvaf_embed_ln_bwd_workspace_bytes": (SZ, [I, I, I, I]),
    "mtvaf_gemm_f32_dw_group": (c_int, [I, P, P, P, P, P, P, P, P, I, P, P, P, SZ, I, P]),
    "mtvaf_gemm_f32_dw_group_bias": (c_int, [I, P, P, P, P, P, P, P, P, I, P, P, P, P, SZ, I, P]),
    "mtvaf_gemm_f32_dw_group_workspace_bytes": (SZ, [I, P, P, I, I]),
    "mtvaf_dw_group_rows": (c_int, [I]),
    "mtvaf_dw_group_wanted": (c_int, [I, I, I]),
    "mtvaf_streamk_attach": (c_int, [P, SZ, P]),
    "mtvaf_streamk_scratch_bytes": (SZ, [I]),
    "mtvaf_streamk_attached": (c_int, [P]),
    "mtvaf_gemm_bf16x_dw_group": (c_int, [I, P, P, P, P, P, P, P, P, I, P]),
    "mtvaf_cast_bf16": (c_int, [P, I, P, I, P, I, I, I, P]),
    "mtvaf_adamw": (c_int, [P, P, P, P, L, F, c_double, c_double, F, F, F, F, F, P, I, P]),
    "mtvaf_adamw_planes": (c_int, [P, P, P, P, L, F, c_double, c_double, F, F, F, F, F, I, P, P, P, P, I, P]),
    "mtvaf_adamw_multi": (c_int, [I, P, P, P, P, P, F, c_double, c_double, F, F, F, F, F, P]),
    "mtvaf_grad_pack_bf16": (c_int, [P, P, L, L, P]),
    "mtvaf_grad_reduce_bf16": (c_int, [P, P, I, L, F, P]),
    "mtvaf_grad_unpack_bf16": (c_int, [P, P, L, P]),
}

class LayerStruct(ctypes.Structure):
    """mtvaf_layer_t (include/mtvaf_hip.h): one encoder layer's shapes, parameters, input and activation buffers."""
    _fields_ = ([(n, c_int) for n in ("B", "S", "P", "NH", "H", "I", "bf16")] +
                [(n, c_float) for n in ("eps", "p_hidden", "p_attn")] + [("seed", c_uint64), ("offset", c_uint64)] +
                [(n, c_void_p) for n in ("wqkv", "wo", "w1", "w2", "wqkv_h", "wo_h", "w1_h", "w2_h", "bqkv", "bo", "g1", "b1",
                                         "bi1", "bi2", "g2", "b2", "x", "x_h", "pk", "pv", "addmask", "qkv", "cx", "lse", "a",
                                         "h1", "h1_h", "mean1", "rstd1", "pre", "act", "f", "h2", "h2_h", "mean2", "rstd2",
                                         "ws")] + [("ws_bytes", c_size_t), ("cu", c_void_p), ("Mv", c_int), ("Mp", c_int)] +
                [(n, c_void_p) for n in ("x_p", "cx_p", "h1_p", "act_p", "h2_p")])


class LayerGradsStruct(ctypes.Structure):
    """mtvaf_layer_grads_t: gradient in / out, temporaries, parameter-gradient destinations, per-stream scratch."""
    _fields_ = ([(n, c_void_p) for n in ("dh", "dh1", "df", "dpre", "da", "dctx", "dqkv", "part", "partq", "partkv", "delta",
                                         "dwqkv", "dbqkv", "dwo", "dbo", "dg1", "db1", "dw1", "dbi1", "dw2", "dbi2", "dg2",
                                         "db2", "dpk", "dpv", "ws_main")] + [("ws_main_bytes", c_size_t), ("ws_side", c_void_p),
                                                                            ("ws_side_bytes", c_size_t), ("klist", c_void_p),
                                                                            ("kcnt", c_void_p), ("zero_tail", c_int),
                                                                            ("lnpart2", c_void_p), ("lnpart1", c_void_p)] +
                [(n, c_void_p) for n in ("df_p", "dpre_p", "da_p", "dqkv_p")])


_lib = None


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"mtvaf_amd: HIP extension {LIB_PATH} not found -- run `python -m mtvaf_amd.build` "
                "(there is no CPU fallback for the product path)")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        if (l.mtvaf_layer_struct_bytes(0), l.mtvaf_layer_struct_bytes(1)) != (ctypes.sizeof(LayerStruct), ctypes.sizeof(LayerGradsStruct)):
            raise RuntimeError("mtvaf_amd: the ctypes mirrors of mtvaf_layer_t / mtvaf_layer_grads_t do not match the library")
        _lib = l
    return _lib


def exported_symbols():
    return sorted(_SIGS)


def _p(t: Optional[torch.Tensor]):
    """Device pointer of a tensor handed to a kernel.  A host tensor here would be dereferenced by the GPU (a memory fault that
    takes the process -- and on a shared host possibly more -- down): there is no CPU fallback, so it is an error, loudly."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"mtvaf_amd: a {t.device} tensor reached a HIP kernel -- the path has no CPU fallback; move the "
                           "model and its inputs to the GPU")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


STREAM_OVERRIDE = None  # raw hipStream_t: set by the native executor around hooks that only enqueue library kernels


def _st():
    """hipStream_t of torch's current stream on the current device (one C call: torch.cuda.current_stream() builds a
    Stream object and costs ~8 us, which adds up over ~130 launches per step in the launch-bound configurations)."""
    if STREAM_OVERRIDE is not None:
        return STREAM_OVERRIDE
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _ck(rc: int, name: str):
    if rc != 0:
        msg = _ERR.get(rc, f"hipError {rc}" if rc > 0 else f"error {rc}")
        raise RuntimeError(f"{name} failed: {msg}")


def _f32(*ts):
    for t in ts:
        if t is not None:
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), (t.device, t.dtype, t.is_contiguous())


# -------------------------------------------------------------------------------------------------
# workspace: one growing scratch buffer per device (the library never allocates)
# -------------------------------------------------------------------------------------------------
_ws = {}


def workspace(nbytes: int, device) -> torch.Tensor:
    """Scratch for split-K slabs / column-sum partials, one buffer per (device, stream): kernels launched on a side
    stream (the weight-gradient products of the encoder backward) must not share slabs with the main stream."""
    key = (torch.device(device).index or 0, _st() if _lib is not None else 0)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


PROFILE = None  # set to a list to record (key, start_event, stop_event) around every GEMM call (incl. reduce)


def prof_start(capacity: int = 4096):
    """Start the in-library launch profiler (HIP events around each main GEMM kernel, on its stream)."""
    _ck(lib().mtvaf_prof_start(capacity), "mtvaf_prof_start")


def prof_stop(capacity: int = 4096):
    """-> list of (key dict, ms).  Synchronises the recorded events."""
    n = ctypes.c_int(0)
    keys = (ctypes.c_int * (8 * capacity))()
    ms = (ctypes.c_float * capacity)()
    _ck(lib().mtvaf_prof_stop(ctypes.byref(n), keys, ms, capacity), "mtvaf_prof_stop")
    out = []
    for i in range(n.value):
        k = keys[8 * i:8 * i + 8]
        out.append((dict(cfg=k[0], la=k[1], lb=k[2], fast=k[3], M=k[4], N=k[5], K=k[6], splits=k[7]), ms[i]))
    return out


def kernel_symbol(cfg, la, lb, fast):
    """The template instantiation rocprofv3 reports for a (tile cfg, layouts, fast) launch."""
    dims = {0: (128, 128, 2, 2, 16), 1: (128, 96, 4, 1, 16), 2: (128, 288, 4, 1, 16), 3: (64, 64, 2, 2, 16),
            4: (128, 64, 4, 1, 16), 5: (128, 128, 2, 2, 32), 6: (128, 96, 4, 1, 32), 7: (128, 192, 2, 2, 16),
            8: (128, 192, 2, 2, 32), 9: (128, 96, 4, 1), 10: (128, 128, 2, 2), 11: (128, 192, 2, 2),
            12: (128, 96, 4, 1), 13: (128, 128, 2, 2), 14: (128, 64, 4, 1), 15: (128, 64, 4, 1),
            16: (64, 64, 2, 2), 17: (64, 64, 2, 2)}.get(cfg)
    b = lambda x: "true" if x else "false"
    klist, fast = bool(fast & 8), fast & 7  # (+8: the launch walked a k-tile list)
    if cfg == 1225:  # the grouped weight-gradient launch of the wave-specialised split kernel (mtvaf_gemm_f32_dw_group)
        return f"gemm_f32x3_ws_kernel<true, true, {b(klist)}, 128, true>"
    if cfg >= 1000:  # the grouped weight-gradient launch of the fp32 LDS-DMA kernel (mtvaf_gemm_f32_dw_group)
        return f"gemm_f32_dma_group_kernel<128, 96, 4, 1, 2, {b(klist)}>"
    if 400 <= cfg < 500:  # pre-split operands (csrc/gemm_f32p.hip): gemm_f32p16_kernel<ABL, TRACE, B_KM, A_KM, GROUP>
        c = cfg - 400
        if c & 32:  # the 128 x 256 tile (csrc/gemm_f32pw.hip): gemm_f32p16w_kernel<B_KM, A_KM, GROUP, ABL, TRACE>
            return f"gemm_f32p16w_kernel<{b(c & 8)}, {b(c & 4)}, {b(c & 16)}, 0, false, {6 if c & 64 else 8}>"
        return f"gemm_f32p16_kernel<0, false, {b(c & 8)}, {b(c & 4)}, {b(c & 16)}>"
    if cfg >= 300:  # gemm_bf16x_kernel<BM, BN, WM, WN, A_KM, B_KM, NSTAGE, KLIST>
        c = cfg - 300
        if c & 64:  # gemm_bf16_p256_kernel<A_KM, B_KM, SK> (+128: the in-launch-combine form, +256: a grouped launch)
            return f"gemm_bf16_p256_kernel<{b(c & 4)}, {b(c & 8)}, {b(c & 128)}>"
        t = "256, 192, 4, 2" if c & 32 else ("256, 128, 4, 2" if c & 16 else ("128, 128, 2, 2" if c & 1 else "128, 96, 4, 1"))
        return f"gemm_bf16x_kernel<{t}, {b(c & 4)}, {b(c & 8)}, {3 if c & 2 else 2}, {b(klist)}>"
    if cfg >= 200:  # split-fp32 kernels (csrc/gemm_f32x3.hip): +20 the wave-specialised kernel
        if cfg in (224, 225, 226):
            bn = {224: 64, 225: 128, 226: 96}[cfg]
            return f"gemm_f32x3_ws_kernel<{b(la)}, {b(lb)}, {b(klist)}, {bn}, false>"
        d = (64, 64, 2, 2) if cfg == 203 else ((128, 96, 4, 1) if cfg == 206 else (128, 128, 2, 2))
        return f"gemm_f32x3_kernel<{d[0]}, {d[1]}, {d[2]}, {d[3]}, {b(la)}, {b(lb)}, {b(klist)}, 32>"
    if cfg >= 100:
        d = {106: (128, 96, 4, 1), 105: (128, 128, 2, 2), 103: (64, 64, 2, 2)}[cfg]
        return f"gemm_bf16_kernel<{d[0]}, {d[1]}, {d[2]}, {d[3]}, {b(la)}, {b(lb)}, {b(fast == 2)}>"
    if cfg >= 9:
        return (f"gemm_f32_dma_kernel<{dims[0]}, {dims[1]}, {dims[2]}, {dims[3]}, {b(la)}, {b(lb)}, "
                f"{2 if cfg in (12, 13, 15, 17) else 3}, {b(klist)}>")
    return (f"gemm_f32_kernel<{dims[0]}, {dims[1]}, {dims[2]}, {dims[3]}, {dims[4]}, {b(la)}, {b(lb)}, "
            f"{b(fast >= 1)}, {b(fast == 2)}>")
TILE_NAMES = {0: "128x128x16", 1: "128x96x16", 2: "128x288x16", 3: "64x64x16", 4: "128x64x16", 5: "128x128x32",
              6: "128x96x32", 7: "128x192x16", 8: "128x192x32", 9: "128x96x32dma", 10: "128x128x32dma",
              11: "128x192x32dma", 12: "128x96x32dma2", 13: "128x128x32dma2", 14: "128x64x32dma", 15: "128x64x32dma2",
              16: "64x64x32dma", 17: "64x64x32dma2"}


def gemm_plan(M, N, K, allow_split, layout_a=0, layout_b=0, epi=0):
    c, s = ctypes.c_int(0), ctypes.c_int(0)
    _ck(lib().mtvaf_gemm_f32_plan(layout_a, layout_b, M, N, K, epi, int(allow_split), ctypes.byref(c), ctypes.byref(s)),
        "mtvaf_gemm_f32_plan")
    return c.value, s.value


KC, KM = 0, 1
EPI_NONE, EPI_GELU, EPI_TANH, EPI_DGELU, EPI_DTANH = 0, 1, 2, 3, 4
# GEMM arithmetic: "fp32" (v_mfma_f32_32x32x2_f32) or "bf16" (operands rounded to bf16 while staged, fp32
# accumulation; buffers stay fp32).  Set per call or process-wide through set_compute_dtype().
COMPUTE = "fp32"


def f32_split(on=None) -> bool:
    """fp32-mode GEMMs on the bf16 matrix pipe by three-way operand splitting (csrc/gemm_f32x3.hip: six exact bf16 partial
    products per fp32 product, fp32 accumulate; operands and results stay fp32).  on = True / False switches every
    mtvaf_gemm_f32 / _ktiles call of the process (Python orchestration and native executor alike); None queries.
    Default: ON (MTVAF_F32_SPLIT=0 in the environment keeps the fp32 MFMA pipe)."""
    return bool(lib().mtvaf_f32_split(-1 if on is None else int(bool(on))))


def f32p_wide(mask=None) -> int:
    """The pre-split GEMM's tiles (csrc/gemm_f32pw.hip, round 6): a mask of the products that may leave the 128 x 128 tile -- 1 forward
    (128 x 256, or 128 x 192 without a plane-image result), 2 dX, 4 weight gradients; a launch takes the cheapest admitted tile by the
    library's rounds x tile-time estimate; 8 = never 128 x 128 where another tile can serve (tests), 16 = never 128 x 192; True = 15,
    False = 0, None queries.  Placement only: the kernels agree bit for bit.  Default: MTVAF_P16_WIDE (7 if unset)."""
    if mask is True:
        mask = 15
    return int(lib().mtvaf_f32p_wide(-1 if mask is None else int(mask)))


class Planes:
    """Plane image of an fp32 matrix [rows, cols] (csrc/gemm_f32p.hip): the three bf16 planes of the split, `blocked` =
    [k-tile][plane][rows][32] (every 1-KiB request of the kernel reads contiguous memory) or natural [3][rows][cols]."""
    __slots__ = ("img", "s_plane", "s_row", "s_kt", "rows", "cols")

    def __init__(self, x: torch.Tensor, blocked: bool = True, fill: bool = True):
        rows, cols = x.shape
        self.rows, self.cols = rows, cols
        self.img = torch.empty(3 * rows * cols, dtype=torch.bfloat16, device=x.device)
        if blocked:
            self.s_plane, self.s_row, self.s_kt = rows * 64, 64, 3 * rows * 64
        else:
            self.s_plane, self.s_row, self.s_kt = rows * cols * 2, cols * 2, 64
        if fill:  # (False: an image some kernel is about to write; x only gives the shape)
            self.refresh(x)

    def refresh(self, x: torch.Tensor):
        _ck(lib().mtvaf_f32_split_planes(_p(x), _p(self.img), self.rows, self.cols, x.stride(0), self.s_plane, self.s_row, self.s_kt,
                                         _st()), "mtvaf_f32_split_planes")


def gemm_planes_ep(a: "Planes", b: "Planes", out_planes: "Planes", out=None, bias=None, epi=EPI_NONE, aux=None, colpart=None, layout_b=KC):
    """mtvaf_gemm_f32p_ep: the product of two plane images written as the (tile-blocked) plane image `out_planes` -- and as fp32 `out`
    too unless None; colpart [M / 128, N]: per-tile column sums."""
    if layout_b == KM:
        M, N, K = a.rows, b.cols, a.cols
        bs = _km_strides(b)
    else:
        M, N, K = a.rows, b.rows, a.cols
        bs = (b.s_plane, b.s_row, b.s_kt, 0)
    assert (out_planes.rows, out_planes.cols) == (M, N) and out_planes.s_row == 64, "the result image is tile-blocked"
    _ck(lib().mtvaf_gemm_f32p_ep(KC, _p(a.img), a.s_plane, a.s_row, a.s_kt, 0, layout_b, _p(b.img), bs[0], bs[1], bs[2], bs[3], _p(out),
                                 out.stride(0) if out is not None else 0, _p(out_planes.img), _p(colpart), M, N, K, _p(bias), epi, _p(aux),
                                 aux.stride(0) if aux is not None else 0, 0, _st()), "mtvaf_gemm_f32p_ep")
    return out_planes


def gemm_planes(a: Planes, b: Planes, out, bias=None, epi=EPI_NONE, aux=None, accumulate=False, splits=1, ablate=0, layout_b=KC,
                layout_a=KC):
    """out[M,N] = A[M,K] . B[N,K]^T (layout_b = KC) or A[M,K] . B[K,N] (layout_b = KM: `b` is the plane image of the [K, N]
    matrix, natural or tile-blocked) from the plane images of both operands (mtvaf_gemm_f32p)."""
    if layout_a == KM:  # weight gradients: out[M,N] = A[K,M]^T . B[K,N] (natural or tile-blocked images)
        assert layout_b == KM and a.rows == b.rows
        M, N, K = a.cols, b.cols, a.rows
        as_, bs = _km_strides(a), _km_strides(b)
    elif layout_b == KM:
        M, N, K = a.rows, b.cols, a.cols
        assert b.rows == K, "k-major B: the plane image of the [K, N] matrix"
        as_ = (a.s_plane, a.s_row, a.s_kt, 0)
        bs = _km_strides(b)
    else:
        M, N, K = a.rows, b.rows, a.cols
        as_ = (a.s_plane, a.s_row, a.s_kt, 0)
        bs = (b.s_plane, b.s_row, b.s_kt, 0)
    ws, wsb = None, 0
    if splits > 1:
        wsb = splits * M * N * 4
        ws = workspace(wsb, out.device)
    _ck(lib().mtvaf_gemm_f32p(layout_a, _p(a.img), as_[0], as_[1], as_[2], as_[3], layout_b, _p(b.img), bs[0], bs[1], bs[2], bs[3], _p(out),
                              out.stride(0), M, N, K,
                              _p(bias), epi, _p(aux), aux.stride(0) if aux is not None else 0, int(accumulate), splits, _p(ws), wsb,
                              ablate, _st()), "mtvaf_gemm_f32p")
    return out


def split_planes_blocked(x: torch.Tensor, out: torch.Tensor):
    """fp32 [rows, cols] (contiguous) -> its tile-blocked plane image [cols / 32][3][rows][32] bf16 in `out` (6 rows cols bytes)."""
    rows, cols = x.shape
    _ck(lib().mtvaf_f32_split_planes(_p(x), _p(out), rows, cols, x.stride(0), rows * 64, 64, 3 * rows * 64, _st()), "mtvaf_f32_split_planes")
    return out


def _km_strides(pl: "Planes"):
    """(plane, k-row, k-tile, 128-column block) byte strides of a plane image read K-MAJOR (its rows are the reduction index)."""
    if pl.s_row == pl.cols * 2:  # natural [3][rows][cols]
        return (pl.s_plane, pl.s_row, 32 * pl.s_row, 256)
    return (pl.s_plane, 64, 32 * 64, 4 * pl.s_kt)  # tile-blocked [cols / 32][3][rows][32]: a 32-column block is pl.s_kt apart


def gemm_planes_dw_group(items, colsum=None):
    """items: up to four (a: Planes of dY [K, M], b: Planes of X [K, N], out [M, N] fp32): out = dY^T . X for each, one unsplit launch
    (mtvaf_gemm_f32p_dw_group; natural or tile-blocked images).  colsum = (src fp32 [K, C], dst [C]): dst = column sums of src, as
    extra blocks of the same launch; or a list of up to eight such pairs."""
    n = len(items)
    K = items[0][0].rows
    assert all(a.rows == K and b.rows == K for a, b, _ in items)
    vp = lambda ts: (ctypes.c_void_p * n)(*[_p(t) for t in ts])
    ia = lambda xs: (ctypes.c_int * n)(*xs)
    st = []
    for a, b, _ in items:
        st += list(_km_strides(a)) + list(_km_strides(b))
    if colsum is not None:
        jobs = [colsum] if isinstance(colsum, tuple) else list(colsum)
        nj = len(jobs)
        pj = lambda ts: (ctypes.c_void_p * nj)(*[_p(t) for t in ts])
        ij = lambda xs: (ctypes.c_int * nj)(*xs)
        _ck(lib().mtvaf_gemm_f32p_dw_group_colsum(n, vp([a.img for a, _, _ in items]), vp([b.img for _, b, _ in items]),
                                                  (ctypes.c_long * (8 * n))(*st), vp([o for _, _, o in items]),
                                                  ia([o.stride(0) for _, _, o in items]), ia([a.cols for a, _, _ in items]),
                                                  ia([b.cols for _, b, _ in items]), K, nj, pj([s_ for s_, _ in jobs]),
                                                  ij([s_.shape[0] for s_, _ in jobs]), ij([s_.shape[1] for s_, _ in jobs]),
                                                  ij([s_.stride(0) for s_, _ in jobs]), pj([d for _, d in jobs]), _st()),
            "mtvaf_gemm_f32p_dw_group_colsum")
        return
    _ck(lib().mtvaf_gemm_f32p_dw_group(n, vp([a.img for a, _, _ in items]), vp([b.img for _, b, _ in items]), (ctypes.c_long * (8 * n))(*st),
                                       vp([o for _, _, o in items]), ia([o.stride(0) for _, _, o in items]), ia([a.cols for a, _, _ in items]),
                                       ia([b.cols for _, b, _ in items]), K, _st()), "mtvaf_gemm_f32p_dw_group")


def set_compute_dtype(dtype: str):
    global COMPUTE
    if dtype not in ("fp32", "bf16"):
        raise ValueError(dtype)
    COMPUTE = dtype


def gemm(a: torch.Tensor, layout_a: int, b: torch.Tensor, layout_b: int, out: torch.Tensor, M: int, N: int, K: int,
         bias: Optional[torch.Tensor] = None, epi: int = EPI_NONE, aux: Optional[torch.Tensor] = None,
         accumulate: bool = False, allow_split: bool = False, lda: Optional[int] = None, ldb: Optional[int] = None,
         ldc: Optional[int] = None, cfg: int = -1, splits: int = -1, compute: Optional[str] = None):
    """out[M,N] = opA[M,K] . opB[K,N] (+bias, epilogue).  KC: reduction index contiguous; KM: k-major."""
    _f32(a, b, out, bias, aux)
    lda = a.stride(0) if lda is None else lda
    ldb = b.stride(0) if ldb is None else ldb
    ldc = out.stride(0) if ldc is None else ldc
    ws, wsb = None, 0
    if allow_split:
        wsb = lib().mtvaf_gemm_f32_workspace_bytes(M, N, K, 1)
        ws = workspace(wsb, out.device)
    prof = PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    mode = compute or COMPUTE
    fn = lib().mtvaf_gemm_bf16 if mode == "bf16" else (lib().mtvaf_gemm_f32x3 if mode == "fp32x3" else lib().mtvaf_gemm_f32)
    _ck(fn(layout_a, layout_b, _p(a), lda, _p(b), ldb, _p(out), ldc, M, N, K, _p(bias), epi, _p(aux),
           aux.stride(0) if aux is not None else 0, int(accumulate), int(allow_split), _p(ws), wsb, cfg, splits, _st()),
        "mtvaf_gemm")
    if prof is not None:
        e1.record()
        prof.append(((layout_a, layout_b, M, N, K, epi, int(allow_split)), e0, e1))
    return out


def gemm_ktiles(a, b, out, M, N, K, klist, kcnt, accumulate=False, cfg=-1, splits=-1):
    """out[M,N] (+)= a[K,M]^T . b[K,N] over the listed 32-row k-tiles only (a is exactly zero elsewhere): klist / kcnt int32."""
    wsb = lib().mtvaf_gemm_f32_workspace_bytes(M, N, K, 1)
    ws = workspace(wsb, out.device)
    _ck(lib().mtvaf_gemm_f32_ktiles(KM, KM, _p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), M, N, K, None,
                                    EPI_NONE, None, 0, int(accumulate), 1, _p(ws), wsb, cfg, splits, _p(klist), _p(kcnt), _st()),
        "mtvaf_gemm_f32_ktiles")
    return out


def build_ktiles(addmask, Pn, S, bk=32):
    B, T = addmask.shape
    klist = torch.empty(B * S // bk, dtype=torch.int32, device=addmask.device)
    kcnt = torch.empty(1, dtype=torch.int32, device=addmask.device)
    _ck(lib().mtvaf_build_ktiles(_p(addmask), B, T, Pn, S, bk, _p(klist), _p(kcnt), _st()), "mtvaf_build_ktiles")
    return klist, kcnt


def linear_fwd(x, w, bias, out, epi=EPI_NONE, aux=None):
    """out[M,N] = x[M,K] . w[N,K]^T + bias  (nn.Linear)."""
    M, K = x.shape
    N = w.shape[0]
    # every epilogue may follow the deterministic split-K (few output tiles, long K: e.g. the projector logits
    # [288, 48] x K = 6144, every product of a bs-4 step); the planner only splits when the tile grid underfills the chip
    return gemm(x, KC, w, KC, out, M, N, K, bias=bias, epi=epi, aux=aux, allow_split=True)


def linear_bwd_input(dy, w, dx, accumulate=False, epi=EPI_NONE, aux=None):
    """dx[M,K] (+)= dy[M,N] . w[N,K]"""
    M, N = dy.shape
    K = w.shape[1]
    return gemm(dy, KC, w, KM, dx, M, K, N, accumulate=accumulate, epi=epi, aux=aux,
                allow_split=True)


def linear_bwd_weight(dy, x, dw, accumulate=False, ktiles=None):
    """dw[N,K] (+)= dy[M,N]^T . x[M,K]   (deterministic split-K over M).  ktiles = (klist, kcnt): dy is exactly zero outside
    the listed 32-row k-tiles of the token axis (mtvaf_build_ktiles): the reduction skips the others."""
    M, N = dy.shape
    K = x.shape[1]
    if ktiles is not None and COMPUTE == "fp32":
        return gemm_ktiles(dy, x, dw, N, K, M, ktiles[0], ktiles[1], accumulate=accumulate)
    return gemm(dy, KM, x, KM, dw, N, K, M, accumulate=accumulate, allow_split=True)


def colsum(x, out, accumulate=False):
    rows, cols = x.shape
    wsb = lib().mtvaf_colsum_workspace_bytes(rows, cols)
    ws = workspace(wsb, x.device)
    _ck(lib().mtvaf_colsum(_p(x), rows, cols, x.stride(0), _p(out), int(accumulate), _p(ws), wsb, _st()), "mtvaf_colsum")
    return out


def dropout(x, out, p, seed, offset):
    _ck(lib().mtvaf_dropout(_p(x), _p(out), x.numel(), float(p), seed, offset, _st()), "mtvaf_dropout")
    return out


def embed_ln_fwd(ids, tts, pos_ids, word, pos, typ, gamma, beta, out, mean, rstd, eps, p, seed, offset, out16=None):
    B, S = ids.shape
    H = word.shape[1]
    _ck(lib().mtvaf_embed_ln_fwd(_p(ids), _p(tts), _p(pos_ids), _p(word), _p(pos), _p(typ), _p(gamma), _p(beta), _p(out),
                                 _p(mean), _p(rstd), B, S, H, float(eps), float(p), seed, offset, _p(out16), _st()),
        "mtvaf_embed_ln_fwd")


def embed_ln_bwd(dout, ids, tts, pos_ids, word, pos, typ, gamma, mean, rstd, dword, dpos, dtype, dgamma, dbeta,
                 accumulate, word_pad, pos_pad, p, seed, offset, dz_ws):
    B, S = ids.shape
    H = word.shape[1]
    wsb = lib().mtvaf_embed_ln_bwd_workspace_bytes(B * S, H, word.shape[0], pos.shape[0])
    ws = workspace(wsb, dout.device)
    _ck(lib().mtvaf_embed_ln_bwd(_p(dout), _p(ids), _p(tts), _p(pos_ids), _p(word), _p(pos), _p(typ), _p(gamma), _p(mean),
                                 _p(rstd), _p(dword), _p(dpos), _p(dtype), _p(dgamma), _p(dbeta), int(accumulate), B, S, H,
                                 word.shape[0], pos.shape[0], typ.shape[0], word_pad, pos_pad, float(p), seed, offset,
                                 _p(dz_ws), _p(ws), wsb, _st()), "mtvaf_embed_ln_bwd")


def roberta_position_ids(ids, out, pad_idx):
    B, S = ids.shape
    _ck(lib().mtvaf_roberta_position_ids(_p(ids), _p(out), B, S, pad_idx, _st()), "mtvaf_roberta_position_ids")
    return out


def dropout_res_ln_fwd(x, res, gamma, beta, out, mean, rstd, eps, p, seed, offset, out16=None):
    M, H = x.shape
    _ck(lib().mtvaf_dropout_res_ln_fwd(_p(x), _p(res), _p(gamma), _p(beta), _p(out), _p(mean), _p(rstd), M, H, float(eps),
                                       float(p), seed, offset, _p(out16), _st()), "mtvaf_dropout_res_ln_fwd")


def dropout_res_ln_bwd(dout, x, res, gamma, mean, rstd, dx, dres, dres_accumulate, dgamma, dbeta, accumulate, p, seed,
                       offset, dbias_x=None, dx16=None):
    M, H = x.shape
    wsb = lib().mtvaf_ln_bwd_workspace_bytes(M, H)
    ws = workspace(wsb, x.device)
    _ck(lib().mtvaf_dropout_res_ln_bwd(_p(dout), _p(x), _p(res), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dres),
                                       int(dres_accumulate), _p(dgamma), _p(dbeta), _p(dbias_x), int(accumulate), M, H,
                                       float(p), seed,
                                       offset, _p(ws), wsb, _p(dx16), _st()), "mtvaf_dropout_res_ln_bwd")


def prefix_attn_fwd(qkv, pk, pv, addmask, ctx, lse, B, S, Pn, NH, p, seed, offset):
    _ck(lib().mtvaf_prefix_attn_fwd(_p(qkv), _p(pk), _p(pv), _p(addmask), _p(ctx), _p(lse), B, S, Pn, NH, 64, float(p),
                                    seed, offset, _st()), "mtvaf_prefix_attn_fwd")


def prefix_attn_bwd(dctx, qkv, pk, pv, addmask, ctx, lse, delta, dqkv, dpk, dpv, B, S, Pn, NH, p, seed, offset, zero_tail=False):
    """zero_tail: the caller vouches that dctx is exactly zero for the queries behind each sentence's last unmasked position
    (the k-tile-list contract): same bits, the query loops stop there."""
    _ck(lib().mtvaf_prefix_attn_bwd_tail(_p(dctx), _p(qkv), _p(pk), _p(pv), _p(addmask), _p(ctx), _p(lse), _p(delta), _p(dqkv),
                                         _p(dpk), _p(dpv), B, S, Pn, NH, 64, float(p), seed, offset, int(bool(zero_tail)), _st()),
        "mtvaf_prefix_attn_bwd_tail")


def prefix_attn_varlen_fwd(qkv, pk, pv, cu, pad_rows, ctx, lse, B, S, Pn, NH, p, seed, offset):
    """PACKED token rows: cu [B+1] int32 row offsets; the pad_rows rows behind the last sentence are zero-filled."""
    _ck(lib().mtvaf_prefix_attn_varlen_fwd(_p(qkv), _p(pk), _p(pv), _p(cu), int(pad_rows), _p(ctx), _p(lse), B, S, Pn, NH, 64,
                                           float(p), seed, offset, _st()), "mtvaf_prefix_attn_varlen_fwd")


def prefix_attn_varlen_bwd(dctx, qkv, pk, pv, cu, pad_rows, ctx, lse, delta, dqkv, dpk, dpv, B, S, Pn, NH, p, seed, offset):
    _ck(lib().mtvaf_prefix_attn_varlen_bwd(_p(dctx), _p(qkv), _p(pk), _p(pv), _p(cu), int(pad_rows), _p(ctx), _p(lse), _p(delta),
                                           _p(dqkv), _p(dpk), _p(dpv), B, S, Pn, NH, 64, float(p), seed, offset, _st()),
        "mtvaf_prefix_attn_varlen_bwd")


def prefix_attn_bf16_varlen_fwd(qkv16, pk16, pv16, cu, pad_rows, ctx16, lse, B, S, Pn, NH, p, seed, offset):
    _ck(lib().mtvaf_prefix_attn_bf16_varlen_fwd(_p(qkv16), _p(pk16), _p(pv16), _p(cu), int(pad_rows), _p(ctx16), _p(lse), B, S, Pn,
                                                NH, 64, float(p), seed, offset, _st()), "mtvaf_prefix_attn_bf16_varlen_fwd")


def prefix_attn_bf16_varlen_bwd(dctx16, qkv16, pk16, pv16, cu, pad_rows, ctx16, lse, dqkv16, dpk, dpv, partq, partkv, B, S, Pn, NH,
                                p, seed, offset):
    _ck(lib().mtvaf_prefix_attn_bf16_varlen_bwd(_p(dctx16), _p(qkv16), _p(pk16), _p(pv16), _p(cu), int(pad_rows), _p(ctx16),
                                                _p(lse), _p(dqkv16), _p(dpk), _p(dpv), _p(partq), _p(partkv), B, S, Pn, NH, 64,
                                                float(p), seed, offset, _st()), "mtvaf_prefix_attn_bf16_varlen_bwd")


def prefix_attn_bf16_fwd(qkv16, pk16, pv16, addmask, ctx16, lse, B, S, Pn, NH, p, seed, offset):
    _ck(lib().mtvaf_prefix_attn_bf16_fwd(_p(qkv16), _p(pk16), _p(pv16), _p(addmask), _p(ctx16), _p(lse), B, S, Pn, NH, 64,
                                         float(p), seed, offset, _st()), "mtvaf_prefix_attn_bf16_fwd")


def prefix_attn_bf16_bwd(dctx16, qkv16, pk16, pv16, addmask, ctx16, lse, dqkv16, dpk, dpv, partq, partkv, B, S, Pn, NH, p,
                         seed, offset, zero_tail=False):
    """partq [B*ceil(S/64), H], partkv [B*ceil((Pn+S)/64), 2H]: per-block column sums of dQ and dK|dV (QKV bias gradient).
    zero_tail: as prefix_attn_bwd."""
    _ck(lib().mtvaf_prefix_attn_bf16_bwd_tail(_p(dctx16), _p(qkv16), _p(pk16), _p(pv16), _p(addmask), _p(ctx16), _p(lse),
                                              _p(dqkv16), _p(dpk), _p(dpv), _p(partq), _p(partkv), B, S, Pn, NH, 64, float(p),
                                              seed, offset, int(bool(zero_tail)), _st()), "mtvaf_prefix_attn_bf16_bwd_tail")


def crf_workspace(B, S, C, device):
    n = lib().mtvaf_crf_workspace_bytes(B, S, C)
    return torch.empty(n, dtype=torch.uint8, device=device), n


def crf_nll_fwd(em, tags, mask_u8, start, end, trans, loss, ws, wsb):
    B, S, C = em.shape
    _ck(lib().mtvaf_crf_nll_fwd(_p(em), _p(tags), _p(mask_u8), _p(start), _p(end), _p(trans), _p(loss), B, S, C, _p(ws), wsb,
                                _st()), "mtvaf_crf_nll_fwd")


def crf_nll_bwd(gout, em, tags, mask_u8, start, end, trans, dem, dstart, dend, dtrans, accumulate, ws, wsb):
    B, S, C = em.shape
    _ck(lib().mtvaf_crf_nll_bwd(_p(gout), _p(em), _p(tags), _p(mask_u8), _p(start), _p(end), _p(trans), _p(dem), _p(dstart),
                                _p(dend), _p(dtrans), int(accumulate), B, S, C, _p(ws), wsb, _st()), "mtvaf_crf_nll_bwd")


def crf_viterbi(em, mask_u8, start, end, trans, tags_out, lens_out):
    B, S, C = em.shape
    _ck(lib().mtvaf_crf_viterbi(_p(em), _p(mask_u8), _p(start), _p(end), _p(trans), _p(tags_out), _p(lens_out), B, S, C,
                                _st()), "mtvaf_crf_viterbi")


# ---- span model heads (csrc/span.hip) -----------------------------------------------------------------------------
def span_index(mask_u8, span_starts, span_ends):
    B, S = mask_u8.shape
    M = span_starts.shape[1]
    index = torch.empty(lib().mtvaf_span_index_ints(B, S, M), dtype=torch.int32, device=mask_u8.device)
    _ck(lib().mtvaf_span_index(_p(mask_u8), _p(span_starts), _p(span_ends), _p(index), B, S, M, _st()), "mtvaf_span_index")
    return index


def span_pool_fwd(seq, w_unary, b_unary, index, pooled, stats, B, S, M):
    H = seq.shape[-1]
    _ck(lib().mtvaf_span_pool_fwd(_p(seq), _p(w_unary), _p(b_unary), _p(index), _p(pooled), _p(stats), B, S, M, H, _st()),
        "mtvaf_span_pool_fwd")


def span_pool_bwd(dpooled, pooled, stats, seq, w_unary, b_unary, index, dseq, dw, db, B, S, M):
    """dseq overwritten; dw [H] and db [1] = column sums of the per-span partials (deterministic)."""
    H = seq.shape[-1]
    wsb = lib().mtvaf_span_pool_bwd_workspace_bytes(B, S, M, H)
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=seq.device)  # private: the colsum below uses the shared one
    dwp, dbp = ctypes.c_void_p(), ctypes.c_void_p()
    _ck(lib().mtvaf_span_pool_bwd(_p(dpooled), _p(pooled), _p(stats), _p(seq), _p(w_unary), _p(b_unary), _p(index), _p(dseq),
                                  ctypes.byref(dwp), ctypes.byref(dbp), B, S, M, H, _p(ws), wsb, _st()), "mtvaf_span_pool_bwd")
    off_w = (dwp.value - ws.data_ptr()) // 4
    off_b = (dbp.value - ws.data_ptr()) // 4
    NS = B * M
    colsum(ws[off_w:off_w + NS * H].view(NS, H), dw)
    colsum(ws[off_b:off_b + NS].view(NS, 1), db)


def distant_ce_fwd(logits, ld, positions, loss, row_ws, B, S, scale, accumulate):
    _ck(lib().mtvaf_distant_ce_fwd(_p(logits), ld, _p(positions), _p(loss), _p(row_ws), B, S, float(scale), int(accumulate),
                                   _st()), "mtvaf_distant_ce_fwd")


def distant_ce_bwd(gout, scale, logits, ld, positions, row_ws, dlogits, ldd, B, S):
    _ck(lib().mtvaf_distant_ce_bwd(_p(gout), float(scale), _p(logits), ld, _p(positions), _p(row_ws), _p(dlogits), ldd, B, S,
                                   _st()), "mtvaf_distant_ce_bwd")


def ce_fwd(logits, labels, loss, ws2):
    N, C = logits.shape
    _ck(lib().mtvaf_ce_fwd(_p(logits), _p(labels), _p(loss), _p(ws2), N, C, _st()), "mtvaf_ce_fwd")


def ce_bwd(gout, logits, labels, ws2, dlogits):
    N, C = logits.shape
    _ck(lib().mtvaf_ce_bwd(_p(gout), _p(logits), _p(labels), _p(ws2), _p(dlogits), N, C, _st()), "mtvaf_ce_bwd")


def mask_mul(x, row_keep, col_keep, out):
    B, S, H = x.shape
    _ck(lib().mtvaf_mask_mul(_p(x), _p(row_keep), _p(col_keep), _p(out), B, S, H, _st()), "mtvaf_mask_mul")
    return out


# ---- bf16-operand GEMM (csrc/gemm_bf16x.hip) ----------------------------------------------------------------------
def cast_bf16(x, out=None, out_t=None):
    """x [R,C] fp32 -> out [R,C] bf16 and / or out_t [C,R] bf16 (either may be None)."""
    R, C = x.shape
    _ck(lib().mtvaf_cast_bf16(_p(x), x.stride(0), _p(out), out.stride(0) if out is not None else 0, _p(out_t),
                              out_t.stride(0) if out_t is not None else 0, R, C, _st()), "mtvaf_cast_bf16")


def gemm_bf16x(a, layout_a, b, layout_b, M, N, K, out32=None, out16=None, bias=None, epi=EPI_NONE, aux16=None,
               accumulate=False, colpart=None, allow_split=False, tile=0, splits=-1, stages=0, ktiles=None):
    """out[M,N] = opA[M,K] . opB[K,N], bf16 operands, fp32 accumulation.  KC: a is [M,K] / b is [N,K]; KM: a is [K,M] /
    b is [K,N] (the same row-major tensors read in their other role).  out32 fp32 and / or out16 bf16; aux16: bf16
    pre-activation (written by EPI_GELU, read by EPI_DGELU); colpart [M/128, N] fp32 per-tile column sums of the result."""
    ws, wsb = None, 0
    if allow_split:
        wsb = 8 * M * N * 4
        ws = workspace(wsb, a.device)
    if ktiles is not None:  # (a, KM) is exactly zero outside the listed 64-row k-tiles (mtvaf_build_ktiles, bk = 64)
        _ck(lib().mtvaf_gemm_bf16x_ktiles(layout_a, layout_b, _p(a), a.stride(0), _p(b), b.stride(0), _p(out32),
                                          out32.stride(0) if out32 is not None else 0, _p(out16),
                                          out16.stride(0) if out16 is not None else 0, M, N, K, _p(bias), epi, _p(aux16),
                                          aux16.stride(0) if aux16 is not None else 0, int(accumulate), _p(colpart),
                                          int(allow_split), _p(ws), wsb, tile, splits, stages, _p(ktiles[0]), _p(ktiles[1]),
                                          _st()), "mtvaf_gemm_bf16x_ktiles")
        return
    _ck(lib().mtvaf_gemm_bf16x(layout_a, layout_b, _p(a), a.stride(0), _p(b), b.stride(0), _p(out32),
                               out32.stride(0) if out32 is not None else 0, _p(out16),
                               out16.stride(0) if out16 is not None else 0, M, N, K, _p(bias), epi, _p(aux16),
                               aux16.stride(0) if aux16 is not None else 0, int(accumulate), _p(colpart), int(allow_split),
                               _p(ws), wsb, tile, splits, stages, _st()), "mtvaf_gemm_bf16x")


_sk_scratch = {}
STREAMK = os.environ.get("MTVAF_STREAMK", "1") != "0"


def streamk_ensure(device) -> bool:
    """Attach (once per device and stream) the scratch of the stream-K launches of the 256x256 bf16 kernel to the CURRENT
    stream: 4 KiB of flag words + one 256-KiB slab per CU, zero-initialised, owned here.  MTVAF_STREAMK=0 / STREAMK = False:
    no scratch is attached and the library keeps to its tile-per-block / split-K launches."""
    if not STREAMK:
        return False
    key = (torch.device(device).index or 0, _st())
    if key not in _sk_scratch:
        nbytes = int(lib().mtvaf_streamk_scratch_bytes(256))
        # zero-filled ON the stream it is attached to: every launch that uses it is ordered behind the fill (no host sync --
        # this may run inside a HIP-graph capture, where the fill simply becomes the graph's first node)
        buf = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        if lib().mtvaf_streamk_attach(_p(buf), nbytes, key[1]) != 0:  # (the library serves 8 streams)
            _sk_scratch[key] = None
        else:
            _sk_scratch[key] = buf
    return _sk_scratch[key] is not None


def streamk_detach_all():
    for (_, st), buf in list(_sk_scratch.items()):
        if buf is not None:
            _ck(lib().mtvaf_streamk_attach(None, 0, st), "mtvaf_streamk_attach")
    _sk_scratch.clear()
    _sk_poll.clear()


def streamk_error(device) -> int:
    """The error word of the current stream's scratch (non-zero: a bounded wait inside a stream-K launch ran out)."""
    buf = _sk_scratch.get((torch.device(device).index or 0, _st()))
    return 0 if buf is None else int(buf[4092:4096].view(torch.int32).item())


def streamk_errors() -> int:
    """Sum of the error words of every attached scratch (all streams); synchronises.  Non-zero: some finisher's bounded wait
    ran out and its tile is wrong -- never observed; bench.py and the tests assert it."""
    torch.cuda.synchronize()
    return sum(int(buf[4092:4096].view(torch.int32).item()) for buf in _sk_scratch.values() if buf is not None)


_sk_poll = {}


def streamk_poll():
    """Asynchronous check of the stream-K error words, for once-per-step callers (the encoder backward in bf16 mode): raises if
    the copy enqueued by the PREVIOUS call has landed non-zero, then enqueues a fresh 4-byte copy of every attached scratch's
    error word to pinned memory on the current stream.  No host synchronisation: a timed-out wait (wrong weight gradients in
    that launch) surfaces one step later instead of never; the header is re-zeroed so that later launches do not read the
    stale flags the failed launch left behind.  Inside a HIP-graph capture it does nothing (pinned allocations and host
    reads would invalidate the capture; bench.py and the graph tests call streamk_errors() instead)."""
    if torch.cuda.is_current_stream_capturing():
        return
    for key, buf in _sk_scratch.items():
        if buf is None:
            continue
        ent = _sk_poll.get(key)
        if ent is not None:
            host, ev = ent
            if ev.query() and int(host.item()) != 0:
                buf[:4096].zero_()
                host.zero_()
                raise RuntimeError("a stream-K launch of the bf16 GEMM timed out waiting for a contribution: the weight gradients of "
                                   "that step are wrong (mtvaf_amd.hip.streamk_poll); flags and error word were reset")
        else:
            host = torch.zeros(1, dtype=torch.int32).pin_memory()
            ev = torch.cuda.Event()
            _sk_poll[key] = (host, ev)
        host, ev = _sk_poll[key]
        host.copy_(buf[4092:4096].view(torch.int32), non_blocking=True)
        ev.record()


def dw_group_rows(rows: int = -1) -> int:
    """fp32 mode: layers of at most this many token rows send their four weight-gradient products as one grouped launch
    (csrc/executor.hip: mtvaf_dw_group_rows; default 1024, MTVAF_DW_GROUP_ROWS).  rows >= 0 sets it."""
    return lib().mtvaf_dw_group_rows(rows)


def dw_group_wanted(rows: int, H: int, I: int) -> bool:
    """fp32 mode: does a layer of `rows` token rows send its four weight gradients as one grouped launch?  (The rule of
    csrc/executor.hip: few-token layers on the fp32 pipe's grouped ring; under the split arithmetic also longer ones, unsplit,
    on the split kernel's GROUP form.)"""
    return bool(lib().mtvaf_dw_group_wanted(rows, H, I))


def gemm_f32_dw_group(items, K, ktiles=None, splits=-1, dbias=None):
    """items: up to four (a [K,M] fp32, b [K,N] fp32, out [M,N] fp32): out = a^T . b for each, ONE launch (the 128x96 LDS-DMA
    kernel, or -- split arithmetic, K > 1024, whole 128x128 tiles -- the GROUP form of the wave-specialised split kernel) + one
    ordered slab reduction per product when the reduction is split.  dbias: list of len(items) tensors / None -- dbias[i] [M]
    <- column sums of a over its K rows (the bias gradient that goes with the weight gradient)."""
    n = len(items)
    vp = lambda ts: (ctypes.c_void_p * n)(*[_p(t) for t in ts])
    ia = lambda xs: (ctypes.c_int * n)(*xs)
    As, Bs, Cs = [i[0] for i in items], [i[1] for i in items], [i[2] for i in items]
    Ms, Ns = ia([t.shape[1] for t in As]), ia([t.shape[1] for t in Bs])
    wsb = int(lib().mtvaf_gemm_f32_dw_group_workspace_bytes(n, Ms, Ns, K, splits))  # the library's own plan, as the executor's call
    if dbias is not None:
        wsb = max(wsb, max(int(lib().mtvaf_colsum_workspace_bytes(K, t.shape[1])) for t in As))
    ws = workspace(max(wsb, 16), As[0].device)
    kl, kc = (ktiles if ktiles is not None else (None, None))
    if dbias is None:
        _ck(lib().mtvaf_gemm_f32_dw_group(n, vp(As), ia([t.stride(0) for t in As]), vp(Bs), ia([t.stride(0) for t in Bs]), vp(Cs),
                                          ia([t.stride(0) for t in Cs]), Ms, Ns,
                                          K, _p(kl), _p(kc), _p(ws), ws.numel(), splits, _st()), "mtvaf_gemm_f32_dw_group")
    else:
        assert len(dbias) == n
        _f32(*[t for t in dbias if t is not None])
        _ck(lib().mtvaf_gemm_f32_dw_group_bias(n, vp(As), ia([t.stride(0) for t in As]), vp(Bs), ia([t.stride(0) for t in Bs]), vp(Cs),
                                               ia([t.stride(0) for t in Cs]), Ms, Ns, K, _p(kl), _p(kc), vp(dbias), _p(ws), ws.numel(),
                                               splits, _st()), "mtvaf_gemm_f32_dw_group_bias")


def gemm_bf16x_dw_group(items, K):
    """items: up to four (a [K,M] bf16, b [K,N] bf16, out [M,N] fp32): out = a^T . b for each, ONE stream-K launch."""
    n = len(items)
    vp = lambda ts: (ctypes.c_void_p * n)(*[_p(t) for t in ts])
    ia = lambda xs: (ctypes.c_int * n)(*xs)
    As, Bs, Cs = [i[0] for i in items], [i[1] for i in items], [i[2] for i in items]
    streamk_ensure(As[0].device)
    _ck(lib().mtvaf_gemm_bf16x_dw_group(n, vp(As), ia([t.stride(0) for t in As]), vp(Bs), ia([t.stride(0) for t in Bs]), vp(Cs),
                                        ia([t.stride(0) for t in Cs]), ia([t.shape[1] for t in As]), ia([t.shape[1] for t in Bs]),
                                        K, _st()), "mtvaf_gemm_bf16x_dw_group")


def colsum_small(part, out, accumulate=False):
    rows, cols = part.shape
    _ck(lib().mtvaf_colsum_small(_p(part), rows, cols, _p(out), int(accumulate), _st()), "mtvaf_colsum_small")
    return out


# ---- optimizer / gradient wire format (csrc/optim.hip) ------------------------------------------------------------
def adamw_planes(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, segs, grad_scale=1.0, max_blocks=0):
    """adamw over a flat buffer that also rewrites the plane images of the matrices inside it: segs = [(begin element, rows, cols,
    image uint8 tensor), ...] (at most four)."""
    _f32(p, g, m, v)
    n = len(segs)
    bc1 = 1.0 - beta1 ** step
    bc2_sqrt = (1.0 - beta2 ** step) ** 0.5
    _ck(lib().mtvaf_adamw_planes(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                                 float(bc1), float(bc2_sqrt), float(grad_scale), n, (ctypes.c_long * n)(*[s[0] for s in segs]),
                                 (ctypes.c_int * n)(*[s[1] for s in segs]), (ctypes.c_int * n)(*[s[2] for s in segs]),
                                 (ctypes.c_void_p * n)(*[_p(s[3]) for s in segs]), int(max_blocks), _st()), "mtvaf_adamw_planes")


def adamw(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, p_bf16=None, max_blocks=0):
    """In-place AdamW update of one flat fp32 tensor (torch.optim.AdamW semantics); `step` is the 1-based step count;
    max_blocks > 0: a background update on that many blocks (runs under MFMA-bound kernels without starving them)."""
    n = p.numel()
    bc1, bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
    _ck(lib().mtvaf_adamw(_p(p), _p(g), _p(m), _p(v), n, lr, beta1, beta2, eps, weight_decay, bc1, bc2 ** 0.5, grad_scale,
                          _p(p_bf16), int(max_blocks), _st()), "mtvaf_adamw")


def adamw_multi(ps, gs, ms, vs, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    """One launch per 48 tensors of a parameter group sharing hyper-parameters and step count."""
    k = len(ps)
    if k == 0:
        return
    arr = lambda ts: (ctypes.c_void_p * k)(*[_p(t) for t in ts])
    ns = (ctypes.c_long * k)(*[t.numel() for t in ps])
    bc1, bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
    _ck(lib().mtvaf_adamw_multi(k, arr(ps), arr(gs), arr(ms), arr(vs), ns, lr, beta1, beta2, eps, weight_decay, bc1,
                                bc2 ** 0.5, grad_scale, _st()), "mtvaf_adamw_multi")


def grad_pack_bf16(src, dst, n, npad):
    _ck(lib().mtvaf_grad_pack_bf16(_p(src), _p(dst), n, npad, _st()), "mtvaf_grad_pack_bf16")


def grad_reduce_bf16(recv, out, world, chunk, scale):
    _ck(lib().mtvaf_grad_reduce_bf16(_p(recv), _p(out), world, chunk, scale, _st()), "mtvaf_grad_reduce_bf16")


def grad_unpack_bf16(src, dst, n):
    _ck(lib().mtvaf_grad_unpack_bf16(_p(src), _p(dst), n, _st()), "mtvaf_grad_unpack_bf16")
