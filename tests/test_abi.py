"""CPU-side ABI checks: the library builds for gfx950, loads, and exports every symbol that
include/mtvaf_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "mtvaf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mtvaf_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_header_symbols():
    from mtvaf_amd.build import build_library
    path = build_library(verbose=False)
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mtvaf_hip.h but not exported"
    lib.mtvaf_version.restype = ctypes.c_int
    assert lib.mtvaf_version() >= 100


def test_binding_matches_header():
    from mtvaf_amd import hip
    assert set(hip.exported_symbols()) == set(_declared())
    hip.lib()  # binds every signature; raises if a symbol is missing


def test_header_compiles_as_c():
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write('#include "mtvaf_hip.h"\nint main(void){return MTVAF_OK;}\n')
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", src, "-o",
                        os.path.join(d, "t.o")], check=True)


def test_split_fp32_is_the_library_default_and_the_environment_switches_it_off():
    """mtvaf_f32_split: ON by default (the arithmetic bench.py times and every default-mode parity test runs), MTVAF_F32_SPLIT=0
    keeps the fp32 MFMA pipe, the setter overrides both (host-side switch: no GPU call)."""
    import subprocess
    import sys
    from mtvaf_amd.build import build_library
    path = build_library(verbose=False)
    code = (f"import ctypes; l = ctypes.CDLL({path!r}); print(l.mtvaf_f32_split(-1), l.mtvaf_f32_split(0), l.mtvaf_f32_split(-1), "
            "l.mtvaf_f32_split(1), l.mtvaf_f32_split(-1))")
    env = {k: v for k, v in os.environ.items() if k != "MTVAF_F32_SPLIT"}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["1", "0", "0", "1", "1"], out
    out = subprocess.run([sys.executable, "-c", code], env=dict(env, MTVAF_F32_SPLIT="0"), capture_output=True, text=True,
                         check=True).stdout.split()
    assert out[0] == "0", out


def test_grouped_weight_gradient_rule_is_host_logic_of_the_library():
    """mtvaf_dw_group_wanted(rows, H, I): the one rule the native executor and the Python orchestration ask before sending a
    layer's four weight gradients as ONE launch -- few-token layers on the fp32 pipe's grouped ring (<= mtvaf_dw_group_rows, H and
    I multiples of 96 and 128), longer ones on the split kernel's GROUP form only under the split arithmetic (whole 128 x 128
    tiles); MTVAF_X3_DW_GROUP=0 keeps one launch per product.  No GPU call."""
    import subprocess
    import sys
    from mtvaf_amd.build import build_library
    path = build_library(verbose=False)
    code = (f"import ctypes; l = ctypes.CDLL({path!r}); w = l.mtvaf_dw_group_wanted; "
            "print(w(256, 768, 3072), w(1024, 768, 3072), w(4096, 768, 3072), w(4100, 768, 3072), w(4096, 800, 3072), w(4096, 1024, 4096), "
            "w(512, 1024, 4096), l.mtvaf_f32_split(0), w(4096, 768, 3072), w(256, 768, 3072))")
    env = {k: v for k, v in os.environ.items() if k not in ("MTVAF_F32_SPLIT", "MTVAF_X3_DW_GROUP", "MTVAF_DW_GROUP_ROWS")}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    #               C1   1024 rows  headline  rows % 32  H % 128  BERT-large  its few-token layers (1024 % 96 != 0)  pipe: ->  no  yes
    assert out == ["1", "1", "1", "0", "0", "1", "0", "0", "0", "1"], out
    out = subprocess.run([sys.executable, "-c", code], env=dict(env, MTVAF_X3_DW_GROUP="0"), capture_output=True, text=True,
                         check=True).stdout.split()
    assert out[:3] == ["1", "1", "0"], out
