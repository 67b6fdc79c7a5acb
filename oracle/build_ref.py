#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — builds oracle/_ref/libsoftras_ref.so.

Compiles the *reference's own* SoftRas kernel arithmetic for the host CPU,
from the sources where they lie under /root/reference (nothing is copied into
the repository; the extracted strings go to the git-ignored oracle/_ref/).

Recipe:
  1. put a stub ``jittor`` package on sys.path whose ``code(...)`` records its
     keyword arguments instead of JIT-compiling CUDA;
  2. import jrender/renderer/dr/softras/cuda/soft_rasterize.py by file path and
     call forward_soft_rasterize / backward_soft_rasterize with dummy Vars —
     this yields the ``cuda_header`` strings (the kernels, SRK:12-458 and
     SRK:975-1362);
  3. write them to oracle/_ref/srk_{fwd,bwd}.inc and compile
     oracle/ref_driver.cpp (our launch glue) against oracle/ref_shim/ with
     ``g++ -O2 -ffp-contract=off -fno-fast-math -fopenmp``.

One textual patch is applied, and recorded here because it defines the oracle:
``backward_sample_texture`` (SRK:1154-1174) returns an UNINITIALISED local for
every texel that is not the sampled one (undefined behaviour inherited from
official SoftRas; the evident intent is 0).  We initialise that local to 0.

The result only exists where /root/reference exists (the build container);
the .so then travels to the GPU box with the repo snapshot.
"""
import importlib.util
import os
import subprocess
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("JRENDER_REFERENCE", "/root/reference")
SRK = os.path.join(REF_ROOT, "jrender/renderer/dr/softras/cuda/soft_rasterize.py")
N3K = os.path.join(REF_ROOT, "jrender/renderer/dr/n3mr/cuda/rasterize.py")
N3_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libn3mr_ref.so")
OUT_DIR = os.path.join(HERE, "_ref")
LIB = os.path.join(OUT_DIR, "libsoftras_ref.so")

CXXFLAGS = ["-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fopenmp",
            "-fPIC", "-shared", "-fpermissive", "-w"]


class _Var:
    """Dummy jittor.Var: only .shape / .dtype are touched by the op wrappers."""

    def __init__(self, shape, dtype="float32"):
        self.shape, self.dtype = tuple(shape), dtype


def _extract():
    captured = []
    stub = types.ModuleType("jittor")

    def code(shapes, dtypes, inputs, **kw):
        captured.append(kw)
        return [_Var(s, d) for s, d in zip(shapes, dtypes)]

    stub.code = code
    saved = sys.modules.get("jittor")
    sys.modules["jittor"] = stub
    try:
        spec = importlib.util.spec_from_file_location("_ref_srk", SRK)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        B, NF, T, IS, K = 1, 1, 1, 4, 16
        fv, tex = _Var((B, NF, 9)), _Var((B, NF, T, 3))
        info, aggr = _Var((B, NF, 27)), _Var((B, 2, IS, IS))
        rgba, ids = _Var((B, 4, IS, IS)), _Var((B, K, IS, IS), "int32")
        scal = (IS, 1, 100, 1e-3, 1e-5, 2, 9.21, 1e-4, 1, 2, 0, 1)
        mod.forward_soft_rasterize(fv, tex, info, aggr, rgba, ids, *scal)
        mod.backward_soft_rasterize(fv, tex, rgba, info, aggr, ids, fv, tex, rgba, *scal)
    finally:
        if saved is None:
            del sys.modules["jittor"]
        else:
            sys.modules["jittor"] = saved
    fwd, bwd = captured[0]["cuda_header"], captured[1]["cuda_header"]
    needle = "scalar_t grad_texture_k;"
    if bwd.count(needle) != 1:
        raise RuntimeError("reference backward_sample_texture changed; re-check the UB patch")
    bwd = bwd.replace(needle, "scalar_t grad_texture_k = 0;")
    return fwd, bwd


def _extract_n3mr():
    """The five `cuda_header` strings of the NMR ops (N3K), same stub-jittor recipe."""
    captured = []
    stub = types.ModuleType("jittor")

    def code(*args, **kw):
        captured.append(kw)
        shapes = args[0] if args and isinstance(args[0], (list, tuple)) and args[0] and isinstance(args[0][0], tuple) else [(1,)]
        return [_Var(s) for s in shapes] if isinstance(shapes, list) else _Var(shapes)

    stub.code = code
    stub.empty = lambda shape, dtype="float32": _Var(shape, dtype)
    saved = sys.modules.get("jittor")
    sys.modules["jittor"] = stub
    try:
        spec = importlib.util.spec_from_file_location("_ref_n3k", N3K)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        B, NF, IS, TS = 1, 1, 4, 2
        faces = _Var((B, NF, 3, 3))
        m1, m3, m8, m9 = _Var((B, IS, IS), "int32"), _Var((B, IS, IS, 3)), _Var((B, IS, IS, 8)), _Var((B, IS, IS, 3, 3))
        tex = _Var((B, NF, TS, TS, TS, 3))
        mod.forward_face_index_map(faces, m1, m3, m1, m9, faces, IS, 0.1, 100, 1, 1, 1)
        mod.forward_texture_sampling(faces, tex, m1, m3, m1, m3, m8, m8, IS, 1e-3)
        mod.backward_pixel_map(faces, m1, m3, m1, m3, m1, faces, IS, 1e-3, 1, 1)
        mod.backward_textures(m1, m8, m8, m3, tex, NF)
        mod.backward_depth_map(faces, m1, m1, m9, m3, m1, faces, IS)
    finally:
        if saved is None:
            del sys.modules["jittor"]
        else:
            sys.modules["jittor"] = saved
    if len(captured) != 5:
        raise RuntimeError("expected 5 jt.code sites in the n3mr kernels, got %d" % len(captured))
    return [c["cuda_header"] for c in captured]


def build_n3mr(force=False):
    """Build (or reuse) oracle/_ref/libn3mr_ref.so; None when the reference tree is not mounted."""
    if not os.path.exists(N3K):
        return N3_LIB if os.path.exists(N3_LIB) else None
    deps = [N3K, os.path.join(HERE, "ref_n3mr_driver.cpp"), os.path.join(HERE, "ref_shim/cuda_runtime.h"),
            os.path.abspath(__file__)]
    if (not force and os.path.exists(N3_LIB)
            and os.path.getmtime(N3_LIB) >= max(os.path.getmtime(d) for d in deps)):
        return N3_LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    names = ["n3k_fwd_index", "n3k_fwd_tex", "n3k_bwd_pix", "n3k_bwd_tex", "n3k_bwd_depth"]
    for name, src in zip(names, _extract_n3mr()):
        with open(os.path.join(OUT_DIR, name + ".inc"), "w") as f:
            f.write(src)
    tmp = tempfile.mktemp(suffix=".so", dir=OUT_DIR)
    flags = [f for f in CXXFLAGS if f != "-fopenmp"]
    cmd = ["g++", *flags, "-I", HERE, "-I", os.path.join(HERE, "ref_shim"),
           os.path.join(HERE, "ref_n3mr_driver.cpp"), "-o", tmp]
    subprocess.check_call(cmd, cwd=HERE)
    os.replace(tmp, N3_LIB)
    return N3_LIB


LTX = os.path.join(REF_ROOT, "jrender/io/utils/load_textures.py")
LTX_LIB = os.path.join(OUT_DIR, "libtextures_ref.so")


def _extract_textures():
    """`cuda_header` of the softras texel sampler and of the n3mr sampler for every
    (texture_wrapping, use_bilinear) pair (the reference substitutes both into the source text)."""
    captured = []
    stub = types.ModuleType("jittor")

    def code(shape, dtype, inputs, **kw):
        captured.append(kw)
        return _Var(shape, dtype)

    stub.code = code
    saved = sys.modules.get("jittor")
    sys.modules["jittor"] = stub
    try:
        spec = importlib.util.spec_from_file_location("_ref_ltx", LTX)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        img, faces, upd = _Var((4, 4, 3)), _Var((1, 3, 2)), _Var((1,), "int32")
        mod._load_textures_for_softras(img, faces, _Var((1, 4, 3)), upd)
        out = {"ltx_softras": captured[-1]["cuda_header"]}
        for w in range(4):
            for b in range(2):
                mod._load_textures_for_n3mr(img, faces, _Var((1, 2, 2, 2, 3)), upd, w, b)
                out["ltx_n3mr_%d_%d" % (w, b)] = captured[-1]["cuda_header"]
    finally:
        if saved is None:
            del sys.modules["jittor"]
        else:
            sys.modules["jittor"] = saved
    return out


def build_textures(force=False):
    """Build (or reuse) oracle/_ref/libtextures_ref.so; None when the reference tree is not mounted."""
    if not os.path.exists(LTX):
        return LTX_LIB if os.path.exists(LTX_LIB) else None
    deps = [LTX, os.path.join(HERE, "ref_textures_driver.cpp"), os.path.join(HERE, "ref_shim/cuda_runtime.h"),
            os.path.abspath(__file__)]
    if (not force and os.path.exists(LTX_LIB)
            and os.path.getmtime(LTX_LIB) >= max(os.path.getmtime(d) for d in deps)):
        return LTX_LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    for name, src in _extract_textures().items():
        with open(os.path.join(OUT_DIR, name + ".inc"), "w") as f:
            f.write(src)
    tmp = tempfile.mktemp(suffix=".so", dir=OUT_DIR)
    flags = [f for f in CXXFLAGS if f != "-fopenmp"]
    subprocess.check_call(["g++", *flags, "-I", HERE, "-I", os.path.join(HERE, "ref_shim"),
                           os.path.join(HERE, "ref_textures_driver.cpp"), "-o", tmp], cwd=HERE)
    os.replace(tmp, LTX_LIB)
    return LTX_LIB


def build(force=False):
    """Build (or reuse) oracle/_ref/libsoftras_ref.so.  Returns its path, or
    None when the reference tree is not mounted (e.g. on the GPU box)."""
    if not os.path.exists(SRK):
        return LIB if os.path.exists(LIB) else None
    deps = [SRK, os.path.join(HERE, "ref_driver.cpp"), os.path.join(HERE, "ref_shim/cuda_runtime.h"),
            os.path.abspath(__file__)]
    if (not force and os.path.exists(LIB)
            and os.path.getmtime(LIB) >= max(os.path.getmtime(d) for d in deps)):
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    fwd, bwd = _extract()
    with open(os.path.join(OUT_DIR, "srk_fwd.inc"), "w") as f:
        f.write(fwd)
    with open(os.path.join(OUT_DIR, "srk_bwd.inc"), "w") as f:
        f.write(bwd)
    tmp = tempfile.mktemp(suffix=".so", dir=OUT_DIR)
    cmd = ["g++", *CXXFLAGS, "-I", HERE, "-I", os.path.join(HERE, "ref_shim"),
           os.path.join(HERE, "ref_driver.cpp"), "-o", tmp]
    subprocess.check_call(cmd, cwd=HERE)
    os.replace(tmp, LIB)
    return LIB


C2F = os.path.join(REF_ROOT, "jrender/renderer/dr/softras/cuda/soft_rasterize_coarse_to_fine.py")
C2F_LIB = os.path.join(OUT_DIR, "libsoftras_c2f_ref.so")


def _extract_c2f():
    """`cuda_header` of the reference's coarse-to-fine forward (C2F:21-763) and the bin margin literal it interpolates
    into the launch (`blur_radius`, C2F:15 -> the third argument of TriangleBoundingBoxKernel in `cuda_src`, C2F:802-805)."""
    import re
    captured = []
    stub = types.ModuleType("jittor")

    def code(shapes, dtypes, inputs, **kw):
        captured.append(kw)
        return [_Var(s, d) for s, d in zip(shapes, dtypes)]

    stub.code = code
    saved = sys.modules.get("jittor")
    sys.modules["jittor"] = stub
    try:
        spec = importlib.util.spec_from_file_location("_ref_c2f", C2F)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        B, NF, T, IS, K = 1, 1, 1, 32, 16
        fv, tex = _Var((B, NF, 9)), _Var((B, NF, T, 3))
        info, aggr = _Var((B, NF, 27)), _Var((B, 2, IS, IS))
        rgba, ids = _Var((B, 4, IS, IS)), _Var((B, K, IS, IS), "int32")
        mod.forward_soft_rasterize_coarse_to_fine(fv, tex, info, aggr, rgba, ids, IS, 1, 100, 1e-3, 1e-5, 2, 9.21, 1e-4, 1, 2, 0, 1, 16, 64)
    finally:
        if saved is None:
            del sys.modules["jittor"]
        else:
            sys.modules["jittor"] = saved
    header, src = captured[0]["cuda_header"], captured[0]["cuda_src"]
    m = re.search(r"TriangleBoundingBoxKernel<<<[^>]*>>>\(\s*faces_p,\s*batch_size \* num_faces,\s*([0-9.eE+-]+),", src)
    if not m:
        raise RuntimeError("reference coarse-to-fine launch changed; re-check the blur_radius extraction")
    return header, m.group(1)


def build_c2f(force=False):
    """Build (or reuse) oracle/_ref/libsoftras_c2f_ref.so: the reference's coarse-to-fine forward for the host, its coarse kernel
    launched as one thread so that the bin lists are ascending (oracle/ref_c2f_driver.cpp).  None when the reference tree is not mounted."""
    if not os.path.exists(C2F):
        return C2F_LIB if os.path.exists(C2F_LIB) else None
    deps = [C2F, os.path.join(HERE, "ref_c2f_driver.cpp"), os.path.join(HERE, "ref_shim/cuda_runtime.h"), os.path.abspath(__file__)]
    if not force and os.path.exists(C2F_LIB) and os.path.getmtime(C2F_LIB) >= max(os.path.getmtime(d) for d in deps):
        return C2F_LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    header, blur = _extract_c2f()
    with open(os.path.join(OUT_DIR, "c2f_fwd.inc"), "w") as f:
        f.write(header)
    with open(os.path.join(OUT_DIR, "c2f_params.h"), "w") as f:
        f.write("#define C2F_BLUR_RADIUS %sf\n" % blur if "." in blur or "e" in blur.lower() else "#define C2F_BLUR_RADIUS %s.f\n" % blur)
    tmp = tempfile.mktemp(suffix=".so", dir=OUT_DIR)
    subprocess.check_call(["g++", *CXXFLAGS, "-I", HERE, "-I", os.path.join(HERE, "ref_shim"),
                           os.path.join(HERE, "ref_c2f_driver.cpp"), "-o", tmp], cwd=HERE)
    os.replace(tmp, C2F_LIB)
    return C2F_LIB


FMA_LIB = os.path.join(OUT_DIR, "libsoftras_ref_fma.so")


def build_fma(force=False):
    """The SAME reference sources compiled with floating-point contraction ON (-ffp-contract=fast -mfma), which is what
    nvcc does by default (--fmad=true): oracle/_ref/libsoftras_ref_fma.so.  Not an oracle - a measuring stick: how far
    does the reference move from ITSELF under a legitimate recompile (bench.py `parity.reference_platform_envelope`,
    tools/ref_platform_envelope.py).  Needs the extracted kernel strings of build()."""
    if build(force=force) is None or not os.path.exists(os.path.join(OUT_DIR, "srk_fwd.inc")):
        return FMA_LIB if os.path.exists(FMA_LIB) else None
    deps = [LIB, os.path.join(HERE, "ref_driver.cpp"), os.path.abspath(__file__)]
    if not force and os.path.exists(FMA_LIB) and os.path.getmtime(FMA_LIB) >= max(os.path.getmtime(d) for d in deps):
        return FMA_LIB
    tmp = tempfile.mktemp(suffix=".so", dir=OUT_DIR)
    flags = [f for f in CXXFLAGS if f != "-ffp-contract=off"] + ["-ffp-contract=fast", "-mfma"]
    subprocess.check_call(["g++", *flags, "-I", HERE, "-I", os.path.join(HERE, "ref_shim"),
                           os.path.join(HERE, "ref_driver.cpp"), "-o", tmp], cwd=HERE)
    os.replace(tmp, FMA_LIB)
    return FMA_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(build_fma(force="--force" in sys.argv))
    print(build_n3mr(force="--force" in sys.argv))
    print(build_textures(force="--force" in sys.argv))
    print(build_c2f(force="--force" in sys.argv))
