"""CPU: the C-ABI library loads, exports every symbol include/garlic_hip.h declares, and fails
loudly (no CPU fallback) where there is no HIP device."""
import ctypes as C
import os
import re

import pytest

from garlic_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "garlic_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(garlic_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(abi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(abi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert abi.lib().garlic_hip_abi_version() == abi.ABI_VERSION == 8


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    n = C.c_int32(-1)
    rc = abi.lib().garlic_hip_device_count(C.byref(n))
    assert rc != abi.OK or n.value == 0
    with pytest.raises(abi.GarlicError) as e:
        abi.Context(0)
    assert e.value.code == abi.ERR_HIP
    assert "no CPU path" in str(e.value) or "HIP" in str(e.value)


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under garlic_amd/ or include/ may name it."""
    for base in ("garlic_amd", "include"):
        for dirpath, _dirs, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".c", ".inc")) or f == "Makefile":
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "oracle" not in text.lower() or f == "__init__.py", os.path.join(dirpath, f)


def test_every_switch_has_a_test():
    """every getenv("GARLIC_...") of the library and the host tool is exercised by a test (tests/test_gpu_switches.py runs a
    soak slice under each kernel-selecting switch) or is one of the documented debugging aids"""
    import glob
    import re
    src = "".join(open(f).read() for f in glob.glob(os.path.join(ROOT, "garlic_amd", "csrc", "*.h*")) +
                  glob.glob(os.path.join(ROOT, "garlic_amd", "host", "*.cpp")))
    switches = set(re.findall(r'getenv\("(GARLIC_[A-Z0-9_]+)"\)', src))
    tests = "".join(open(f).read() for f in glob.glob(os.path.join(ROOT, "tests", "*.py")))
    debugging_aids = {"GARLIC_TRACE",            # per-item time stamps of the chain kernels into a file
                      "GARLIC_WORKERS",          # host tool: parser threads
                      "GARLIC_ALLOC_POOL_GB"}    # cap of the score-buffer pool (include/garlic_hip.h)
    missing = sorted(s for s in switches - debugging_aids if s not in tests)
    assert not missing, missing
