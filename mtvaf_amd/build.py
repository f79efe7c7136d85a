"""Builds the HIP/C-ABI shared library for gfx950 in-tree (mtvaf_amd/lib/libmtvaf_hip.so).

    python -m mtvaf_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the
repository snapshot, so it must be built before `gpurun`.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# MTVAF_LIBDIR: build into another directory (A/B variants with MTVAF_EXTRA_FLAGS; load them with MTVAF_LIB=<dir>/libmtvaf_hip.so).
# The flags are part of every object's stamp, so a variant can never be mistaken for the product build.
LIBDIR = os.environ.get("MTVAF_LIBDIR") or os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmtvaf_hip.so")
SOURCES = ["gemm.hip", "gemm_bf16.hip", "gemm_bf16x.hip", "gemm_bf16p.hip", "gemm_f32x3.hip", "gemm_f32p.hip", "gemm_f32pw.hip", "attention.hip", "attention_f32s.hip", "attention_bf16.hip", "rowops.hip", "crf.hip", "prompt.hip", "span.hip", "optim.hip", "executor.hip", "runtime.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast"] + os.environ.get("MTVAF_EXTRA_FLAGS", "").split()
# The attention kernels read their MFMA results with VALU code every 16 products (softmax, dS): keeping the
# accumulators in architectural VGPRs saves ~200 v_accvgpr moves per key tile (gfx950 has one unified file).
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "attention_f32s.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               # the operand split writes its residuals as scalar subtractions: the SLP vectoriser would pair them into
               # v_pk_add_f32, which issues slower beside MFMAs (csrc/gemm_f32x3.hip, resid2)
               "gemm_f32x3.hip": ["-fno-slp-vectorize"]}


def _headers_digest() -> bytes:
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".h"):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.digest()


def _src_digest(src: str, hdr: bytes) -> str:
    h = hashlib.sha256()
    h.update(hdr)
    h.update(open(os.path.join(CSRC, src), "rb").read())
    h.update(" ".join(FLAGS + EXTRA_FLAGS.get(src, [])).encode())
    return h.hexdigest()


def _digest() -> str:
    hdr = _headers_digest()
    return hashlib.sha256("".join(_src_digest(s, hdr) for s in SOURCES).encode()).hexdigest()


def build_library(force: bool = False, verbose: bool = True) -> str:
    """Compiles the sources whose text / headers / flags changed (one stamp per object) and links the library."""
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "build.stamp")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdr = _headers_digest()

    def compile_one(src):
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        ostamp, odig = obj + ".stamp", _src_digest(src, hdr)
        if not force and os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read().strip() == odig:
            return obj
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[mtvaf build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        with open(ostamp, "w") as f:
            f.write(odig)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print("[mtvaf build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
