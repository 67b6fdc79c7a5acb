"""The C-ABI library loads and exports every symbol include/jrender_hip.h declares; without a
GPU it fails loudly (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest

from jrender_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "jrender_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(jr_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_exported_and_bound():
    names = _declared()
    assert len(names) >= 25
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
        assert n in _ffi.SIGNATURES, "no ctypes signature for: " + n
    assert set(_ffi.SIGNATURES) <= set(names)


def test_version_and_error_string():
    lib = _ffi.load()
    assert b"gfx950" in lib.jr_version()


def test_no_cpu_fallback():
    if _ffi_has_gpu():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="jrender_hip|no HIP device"):
        _ffi.Context(0)


def _ffi_has_gpu():
    try:
        return _ffi.device_count() > 0
    except RuntimeError:
        return False


def test_product_does_not_import_oracle():
    # the shipped package must never reach into oracle/ (parity claims depend on it)
    pkg = os.path.join(ROOT, "jrender_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(d, f)
                assert "softras_oracle" not in src and "libsoftras_ref" not in src, os.path.join(d, f)
