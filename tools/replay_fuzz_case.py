#!/usr/bin/env python3
"""Re-run one case saved by tests/fuzz_parity.py (gpurun_out/fuzz_fail_<seed>_<i>.npz) and print every
parity figure — with JRENDER_LIB=<variant .so> to see which build a difference belongs to."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import Oracle                                                          # noqa: E402
from jrender_amd import _ffi                                                       # noqa: E402
from jrender_amd.renderer.dr.softras import SoftRasterizeFunction                  # noqa: E402
from tests.util import RGBA_ATOL, bits_equal, rel_err                              # noqa: E402

z = np.load(sys.argv[1])
kw = eval(str(z["kw"]))
fv, tex, g = z["fv"], z["tex"], z["g"]
port = Oracle("port", nthreads=0)
ref = port.forward(fv, tex, **kw)
rgf, rgt = port.backward(ref, g)
fn = SoftRasterizeFunction(ctx=_ffi.Context.default(), **kw)
fn(fv, tex)
_, _, rgba, info, aggr, ids = [x.numpy() for x in fn.save_vars]
gf, gt = fn.grad(g)
gf, gt = gf.numpy().reshape(rgf.shape), gt.numpy()
print("lib", _ffi.LIB_PATH)
print("kw", kw, "NF", fv.shape[1], "B", fv.shape[0])
print("faces_info exact", bits_equal(info, ref["faces_info"]), "| ids exact", bits_equal(ids, ref["faces_id_buffer"]),
      "| rgba x tol %.3g" % rel_err(rgba, ref["soft_colors"], RGBA_ATOL), "| aggr x tol %.3g" % rel_err(aggr, ref["aggrs_info"], RGBA_ATOL))
for a, b, name in ((gf, rgf, "grad_faces"), (gt, rgt, "grad_textures")):
    fa, fb = np.isfinite(a), np.isfinite(b)
    both = fa & fb & (np.abs(a) < 8e37) & (np.abs(b) < 8e37)
    e = float(np.abs(a[both].astype(np.float64) - b[both]).max() / max(np.abs(b[both]).max(), 1e-30)) if both.any() else 0.0
    i = np.unravel_index(np.argmax(np.where(both, np.abs(a.astype(np.float64) - b), 0)), a.shape)
    print("%s: non-finite ours %d ref %d, pattern equal %s, max |b| %.3g, err/max %.3g at %s (ours %.6g ref %.6g)"
          % (name, (~fa).sum(), (~fb).sum(), np.array_equal(fa, fb), np.abs(b[both]).max() if both.any() else 0, e, i, a[i], b[i]))
