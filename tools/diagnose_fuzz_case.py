#!/usr/bin/env python3
"""GPU box: where does the backward of one saved fuzz case (tests/fuzz_parity.py -> fuzz_fail_<seed>_<i>.npz) differ from the
reference's?  Both backwards run on the HIP forward's saved tensors (so only the backward arithmetic differs); the worst face is
then attributed pixel by pixel (upstream gradient masked to one pixel at a time) and the forward quantities of that pair are printed."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import Oracle                                                          # noqa: E402
from jrender_amd import _ffi                                                       # noqa: E402
from jrender_amd.renderer.dr.softras import SoftRasterizeFunction                  # noqa: E402

z = np.load(sys.argv[1])
kw = eval(str(z["kw"]))
fv, tex, g = z["fv"], z["tex"], z["g"]
port = Oracle("port", nthreads=0)
ref = port.forward(fv, tex, **kw)
ctx = _ffi.Context.default()
fn = SoftRasterizeFunction(ctx=ctx, **kw)
fn(fv, tex)
_, _, rgba, info, aggr, ids = [x.numpy() for x in fn.save_vars]
mine = dict(ref)
mine["soft_colors"], mine["aggrs_info"] = rgba, aggr


def both(gg):
    a = fn.grad(ctx.array(gg))[0].numpy().reshape(fv.shape)
    b = port.backward(mine, gg)[0].reshape(fv.shape)
    return a.astype(np.float64), b.astype(np.float64)


a, b = both(g)
scale = np.abs(b).max()
err = np.abs(a - b) / scale
print("kw", kw)
print("max |ref| %.4g, max err / max %.3g" % (scale, err.max()))
per_face = err.reshape(fv.shape[0], fv.shape[1], 9).max(-1)
order = np.dstack(np.unravel_index(np.argsort(-per_face, axis=None), per_face.shape))[0][:5]
for bi, fi in order:
    print("view %d face %3d: err/max %.3g | ours %s | ref %s" % (bi, fi, per_face[bi, fi], np.array2string(a[bi, fi].reshape(-1), precision=5), np.array2string(b[bi, fi].reshape(-1), precision=5)))
bi, fi = order[0]
IS = kw["image_size"]
K = ids.shape[1]
holders = [(y, x) for y in range(IS) for x in range(IS) if fi in ids[bi, :, y, x]]
print("face %d of view %d is buffered by %d pixels (K = %d)" % (fi, bi, len(holders), K))
rows = []
for (y, x) in holders:
    gm = np.zeros_like(g)
    gm[bi, :, y, x] = g[bi, :, y, x]
    am, bm = both(gm)
    rows.append((np.abs(am[bi, fi] - bm[bi, fi]).max() / scale, y, x, am[bi, fi].reshape(-1), bm[bi, fi].reshape(-1)))
rows.sort(key=lambda r: -r[0])
for e, y, x, am, bm in rows[:4]:
    print("  pixel (row %d, col %d): err/max %.3g\n     ours %s\n     ref  %s" % (y, x, e, np.array2string(am, precision=6), np.array2string(bm, precision=6)))
    print("     rgba %s aggr %s ids %s g %s" % (rgba[bi, :, y, x], aggr[bi, :, y, x], ids[bi, :, y, x], g[bi, :, y, x]))
e, y, x, _, _ = rows[0]
# the pair's forward quantities from the reference formulas (SRK:20-70, 331-358) in float64
f = fv[bi, fi].astype(np.float64)
xp = (2 * x + 1 - IS) / IS
yp = (2 * (IS - 1 - y) + 1 - IS) / IS
inv = info[bi, fi, :9].astype(np.float64).reshape(3, 3)
w = inv @ np.array([xp, yp, 1.0])
print("  pair (pixel row %d col %d, face %d): w = %s (sum %.9f), face xy %s" % (y, x, fi, w, w.sum(), np.array2string(f[:, :2].reshape(-1), precision=5)))
if kw["dist_func"] == "barycentric":
    dis = float(np.min(w)); sign = 1.0 if (w > 0).all() else -1.0
    print("  min w %.9g -> barycentric distance term; sigma %.1e: x/sigma = %.6g" % (dis, kw["sigma_val"], dis * abs(dis) / kw["sigma_val"]))
