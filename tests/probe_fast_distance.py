#!/usr/bin/env python3
"""CPU experiment for VERDICT r5 "next" #2 (a guard-band FAST distance in the forward): what would it cost in parity?

    python tests/probe_fast_distance.py [--faces 39000] [--image-size 1024] [--sigma 1e-5] [--guard 1e-3]        (~ 5 minutes on 8 cores)

The proposal: for OUTSIDE pairs compute the squared distance on a contracted / reciprocal fast path, take the cull decision from it
unless it lies in a guard band around the threshold (then fall back to the exact tree), and let the coverage D of the survivors come
from the fast value.  The GPU access of round 6 closed before any of it could be built; this measures the PARITY side on the CPU, with
the oracle: oracle/softras_oracle.c compiled a second time with -DORC_EXPERIMENT_FAST_D -mfma, where the outside distance is evaluated
once more with FMA contraction and a reciprocal multiply (euclid_outside_fast), the decision stays exact, and D of surviving outside
pairs is taken from the fast value.  Reported against the unmodified oracle on the same view:

  * index buffer identical (by construction: decisions are exact) - checked;
  * RGBA error by the tests' tolerance formula and by SURVEY 8(d)'s;
  * gradient error of the reference backward run on the two forwards' saved tensors (element-wise with the 1e-3 floor, and max-norm);
  * how many outside pairs fall into the guard band (the fallback rate a kernel would pay), how many decisions the fast value alone
    would flip, the largest relative error of the fast distance among survivors.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import Oracle                              # noqa: E402
from jrender_amd import synthetic as syn              # noqa: E402
from tests.util import RGBA_ATOL, bits_equal, grad_err, grad_err_elementwise, rel_err      # noqa: E402


def build(guard, variant="fma+rcp"):
    out = os.path.join(tempfile.gettempdir(), "libsoftras_oracle_fastd_%s_%s.so" % (str(guard).replace(".", "_").replace("-", "m"), variant.replace("+", "_")))
    extra = {"fma+rcp": [], "fma": ["-DORC_EXPERIMENT_NO_RCP"], "rcp": ["-DORC_EXPERIMENT_NO_FMA"]}[variant]
    src = os.path.join(ROOT, "oracle", "softras_oracle.c")
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-mfma", "-fPIC", "-shared",
                           "-DORC_EXPERIMENT_FAST_D", "-DORC_EXPERIMENT_GUARD=%sf" % repr(float(guard)), *extra, src, "-o", out, "-lm"])
    dis = subprocess.run(["objdump", "-d", "--no-show-raw-insn", out], capture_output=True, text=True).stdout
    per_fn, cur = {}, None
    for line in dis.splitlines():
        if line.endswith(">:"):
            cur = line.split("<")[1][:-2]
        elif cur and "vfm" in line:
            per_fn[cur] = per_fn.get(cur, 0) + 1
    return out, per_fn


def survey_rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / (np.abs(b) + 1e-6 * np.abs(b).max())))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--faces", type=int, default=39000)
    ap.add_argument("--image-size", type=int, default=1024)
    ap.add_argument("--sigma", type=float, default=1e-5)
    ap.add_argument("--guard", type=float, default=1e-3)
    ap.add_argument("--variant", default="fma+rcp", choices=["fma+rcp", "fma", "rcp"], help="what makes the distance 'fast': contraction, reciprocal multiply, or both")
    a = ap.parse_args()
    lib, fma = build(a.guard, a.variant)
    assert (fma.get("euclid_outside_fast", 0) > 0) == ("fma" in a.variant), "contraction is not what was asked for: %r" % fma
    fv, tex = syn.sphere_views(a.faces, 1)
    kw = dict(image_size=a.image_size, sigma_val=a.sigma)
    exact = Oracle("port", nthreads=0)
    fast = Oracle("port", nthreads=0)
    fast.lib = C.CDLL(lib)
    fast.lib.orc_ub_events.restype = C.c_long
    t0 = time.time()
    se = exact.forward(fv, tex, **kw)
    t1 = time.time()
    fast.lib.orc_exp_reset()
    sf = fast.forward(fv, tex, **kw)
    cnt = (C.c_double * 5)()
    fast.lib.orc_exp_counters(cnt)
    pairs, outside, guard, flips, max_rel = [float(x) for x in cnt]
    g = np.random.default_rng(0).uniform(-1, 1, se["soft_colors"].shape).astype(np.float32)
    ge, gf = exact.backward(se, g, nthreads=0)[0], exact.backward(sf, g, nthreads=0)[0]
    ge2 = exact.backward(se, g, nthreads=0)[0]                     # the reference backward against ITSELF: float-atomic order noise
    out = {
        "workload": "%d-face sphere, one view, %dx%d, sigma %g, Renderer defaults" % (a.faces, a.image_size, a.image_size, a.sigma), "fast_path": a.variant,
        "fma_instructions": fma, "seconds_per_forward": round(t1 - t0, 1),
        "ids_identical": bits_equal(se["faces_id_buffer"], sf["faces_id_buffer"]),
        "rgba_err_tests_formula (<= 1 passes)": rel_err(sf["soft_colors"], se["soft_colors"], RGBA_ATOL),
        "rgba_rel_floor_1e-6": survey_rel(sf["soft_colors"], se["soft_colors"]),
        "rgba_max_abs": float(np.abs(sf["soft_colors"] - se["soft_colors"]).max()),
        "covered_pixels": int((se["faces_id_buffer"][:, 0] >= 0).sum()),
        "pixels_with_rgba_abs_diff_over": {t: int((np.abs(sf["soft_colors"] - se["soft_colors"]).max(1) > float(t)).sum()) for t in ("1e-6", "1e-5", "1e-4", "1e-3")},
        "aggrs_rel_floor_1e-6": survey_rel(sf["aggrs_info"], se["aggrs_info"]),
        "grad_faces_elementwise_floor_1e-3": grad_err_elementwise(gf, ge), "grad_faces_max_norm": grad_err(gf, ge),
        "grad_faces_order_noise_elementwise": grad_err_elementwise(ge2, ge),
        "euclid_pairs": pairs, "outside_pairs": outside, "guard": a.guard, "outside_pairs_in_guard_band": guard,
        "guard_band_frac_of_outside": guard / max(outside, 1), "decisions_the_fast_value_would_flip": flips,
        "max_rel_error_of_fast_distance_among_survivors": max_rel,
    }
    print(json.dumps(out, indent=1))
    return out


if __name__ == "__main__":
    main()
