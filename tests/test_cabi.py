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


def _prototypes():
    """name -> list of parameter kinds ('ptr', 'int', 'float', 'size') parsed from the header's prototypes."""
    hdr = open(os.path.join(ROOT, "include", "jrender_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(jr_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        name, params = m.group(1), " ".join(m.group(2).split())
        kinds = []
        for prm in ([] if params in ("", "void") else params.split(",")):
            prm = prm.strip()
            if "*" in prm or "[" in prm:             # arrays decay to pointers
                kinds.append("ptr")
            elif re.match(r"(const )?(size_t|uint64_t|unsigned long long)\b", prm):
                kinds.append("size")
            elif re.match(r"(const )?float\b", prm):
                kinds.append("float")
            elif re.match(r"(const )?double\b", prm):
                kinds.append("double")
            elif re.match(r"(const )?(int|int32_t|uint32_t|unsigned)\b", prm):
                kinds.append("int")
            else:
                kinds.append("?" + prm)
        out[name] = kinds
    return out


def test_ctypes_signatures_match_the_header_prototypes():
    """Every bound function passes as many arguments as its prototype takes, pointers where the header has pointers,
    floats where it has floats (a miscounted pointer list only shows up as a TypeError on the GPU box otherwise)."""
    C = ctypes
    kind_of = {C.c_void_p: "ptr", C.c_char_p: "ptr", C.c_int: "int", C.c_uint: "int", C.c_uint32: "int", C.c_int32: "int",
               C.c_float: "float", C.c_double: "double", C.c_size_t: "size", C.c_uint64: "size", C.c_ulonglong: "size"}
    protos = _prototypes()
    for name, (_, argtypes) in _ffi.SIGNATURES.items():
        assert name in protos, name
        got = ["ptr" if (isinstance(t, type) and issubclass(t, (C._Pointer, C.Array))) else kind_of.get(t, "?%r" % (t,))
               for t in argtypes]
        assert got == protos[name], (name, got, protos[name])


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


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/jrender_hip.h must be consumable by a C99 compiler (cgo / JNI / any FFI generator) and a C
    program must link against the library and call it (no GPU needed for jr_version / jr_last_error)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "jrender_amd", "csrc")
    src = tmp_path / "t.c"
    src.write_text('#include <stdio.h>\n#include <string.h>\n#include "jrender_hip.h"\n'
                   'int main(void) {\n'
                   '    jr_ctx* c = 0;\n'
                   '    if (strstr(jr_version(), "gfx950") == 0) return 2;\n'
                   '    /* invalid arguments are rejected with a message, without touching a device */\n'
                   '    if (jr_softras_forward(c, 0, 0, 0, 0, 0, 0, 1, 1, 1, 8, 1, 1.f, 100.f, 1e-3f, 1e-5f, 2, 9.2f, 1e-4f, 1, 2, 0, 1, 0) == 0) return 3;\n'
                   '    if (strlen(jr_last_error()) == 0) return 4;\n'
                   '    printf("%s\\n", jr_version());\n'
                   '    return 0;\n}\n')
    exe = tmp_path / "t"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(root, "include"),
                    str(src), "-o", str(exe), "-L", lib_dir, "-ljrender_hip", "-Wl,-rpath," + lib_dir],
                   check=True, capture_output=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stderr)
    assert "gfx950" in out.stdout


def test_library_is_not_older_than_its_sources():
    """The GPU box runs the prebuilt in-tree library: a library older than csrc/ means the measured / tested code is
    not the code in the tree (`python -m jrender_amd._build`, or `__graft_entry__.build()`, rebuilds it)."""
    from jrender_amd import _build
    deps = [os.path.join(_build.CSRC, f) for f in _build.SOURCES + _build.HEADERS]
    t = os.path.getmtime(_ffi.LIB_PATH)
    newer = [os.path.relpath(d, ROOT) for d in deps if os.path.getmtime(d) > t]
    assert not newer, "libjrender_hip.so is older than %s" % newer
