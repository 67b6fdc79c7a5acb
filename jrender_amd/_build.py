"""Builds jrender_amd/csrc/libjrender_hip.so in-tree with hipcc for gfx950.

``python -m jrender_amd._build [--force]``.  The flags matter for parity:
``-ffp-contract=off`` (no FMA contraction) and no fast-math, because the per-pixel
face-index buffer must match the reference bit for bit.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libjrender_hip.so")
SOURCES = ["jr_api.cpp", "jr_comm.cpp", "binning.hip", "softras_forward.hip", "softras_forward_precise.hip", "softras_backward.hip", "aux_kernels.hip", "loss_kernels.hip", "optim_kernels.hip", "n3mr_kernels.hip"]
HEADERS = ["jr_kernels.h", "softras_device.h", "jr_tuning.h", "../../include/jrender_hip.h"]
INCLUDES = {"softras_forward_precise.hip": ["softras_forward.hip"]}     # sources that #include other sources
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
         "-fPIC", "-Wall", "-Wno-unused-function"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, variant=None, defines=()):
    """variant/defines: an ablation build (tools/ablate): objects and the library get a suffix, the
    product library is untouched.  Tuning switches are the JR_TUNE_* macros of csrc/jr_tuning.h."""
    suffix = "" if not variant else "_" + variant
    lib = os.path.join(CSRC, "libjrender_hip%s.so" % suffix)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, os.path.splitext(s)[0] + suffix + ".o")
        objs.append(obj)
        also = [os.path.join(CSRC, d) for d in INCLUDES.get(s, [])]
        if force or variant or _stale(obj, [src] + also + hdrs):
            jobs.append([HIPCC, *FLAGS, *["-D" + d for d in defines], "-x", "hip", "-c", src, "-o", obj])
    if jobs:
        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd, cwd=CSRC)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    if jobs or _stale(lib, objs):
        tmp = lib + ".tmp%d" % os.getpid()
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", tmp], cwd=CSRC)
        os.replace(tmp, lib)
    if variant:
        for o in objs:
            os.unlink(o)
    return lib


if __name__ == "__main__":
    # python -m jrender_amd._build [--force] [--variant NAME -DJR_TUNE_X=1 ...]
    args = sys.argv[1:]
    variant = args[args.index("--variant") + 1] if "--variant" in args else None
    defines = [a[2:] for a in args if a.startswith("-D")]
    print(build(force="--force" in args, verbose=True, variant=variant, defines=defines))
