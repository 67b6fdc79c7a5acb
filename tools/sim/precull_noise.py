#!/usr/bin/env python3
"""How far can the reference's float distance fall short of the geometric one?  (CPU; needs the oracle.)

The forward kernel's pre-cull (softras_forward.hip) rejects a (pixel, face) pair when the pixel lies more
than rad + margin beyond the line of an edge.  That is only valid if the REFERENCE would cull the pair, and
the reference's distance is a float computation with its own noise.  This tool takes pairs the reference
KEEPS (they are in its K-buffer: oracle.forward_subset) and evaluates the kernel's test on them in float32,
mirroring the kernel's formulas:
  * violations  = kept pairs the pre-cull would reject (must be 0),
  * worst ratio = (geometric distance beyond rad) / (E1 + E2) over kept pairs — how much of the 2.5x safety
                  factor the worst pair uses up.
usage: precull_noise.py IS NFACES sphere|soup [views] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from jrender_amd import synthetic as syn      # noqa: E402
from oracle import Oracle                     # noqa: E402

F = np.float32
IS, NF, scene = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 0
K = 16
fv, tex = syn.sphere_views(NF, B, azimuth0=7.0 * seed) if scene == "sphere" else syn.triangle_soup(NF, B, seed=seed)
port = Oracle("port", nthreads=0)
rng = np.random.default_rng(seed)
pix = rng.choice(B * IS * IS, min(B * IS * IS, 600000), replace=False)
sub = port.forward_subset(fv, tex, pix, image_size=IS, max_faces_per_pixel_for_grad=K)
ids, info = sub["ids"], sub["faces_info"]
EPS = F(2.0 ** -24)
thr = F(F(np.log(1 / 1e-4 - 1)) * F(1e-5))
rad = np.sqrt(thr, dtype=F)
b, r = np.divmod(pix, IS * IS)
row, col = np.divmod(r, IS)
xp = ((2 * col + 1 - IS).astype(F) / F(IS)).astype(F)
yp = ((2 * (IS - 1 - row) + 1 - IS).astype(F) / F(IS)).astype(F)
viol = pairs = 0
worst = 0.0
elig = 0
for k in range(K):
    fid = ids[:, k]
    m = fid >= 0
    inv = info[b[m], fid[m], :9]
    f = fv[b[m], fid[m]]
    x, y = f[:, :, 0], f[:, :, 1]
    xlo, xhi, ylo, yhi = x.min(1) - rad, x.max(1) + rad, y.min(1) - rad, y.max(1) + rad
    X, Y = np.maximum(np.abs(xlo), np.abs(xhi)), np.maximum(np.abs(ylo), np.abs(yhi))
    pos = np.sqrt(X * X + Y * Y)
    ext = (xhi - xlo) + (yhi - ylo)
    g = np.stack([np.sqrt(inv[:, 3 * q] ** 2 + inv[:, 3 * q + 1] ** 2) for q in range(3)], 1)
    S = np.stack([np.abs(inv[:, 3 * q]) * X + np.abs(inv[:, 3 * q + 1]) * Y + np.abs(inv[:, 3 * q + 2]) for q in range(3)], 1)
    sv = (S * np.sqrt(x * x + y * y)).sum(1)
    ca = np.abs((inv[:, 0] + inv[:, 3]) + inv[:, 6])
    cb = np.abs((inv[:, 1] + inv[:, 4]) + inv[:, 7])
    cd = np.abs(((inv[:, 2] + inv[:, 5]) + inv[:, 8]) - F(1))
    e1 = ((ca * X + cb * Y + cd) + F(4) * EPS * S.sum(1)) * pos
    e2 = F(3) * EPS * sv + F(4) * EPS * (g.max(1) * ext + F(1)) * pos
    margin = F(2.5) * (e1 + e2) + F(1.0001) * rad
    ok = margin <= F(1.5) * rad
    w = np.stack([inv[:, 3 * q] * xp[m] + inv[:, 3 * q + 1] * yp[m] + inv[:, 3 * q + 2] for q in range(3)], 1)
    rej = ((w + margin[:, None] * g) < 0).any(1) & ok
    viol += int(rej.sum())
    pairs += int(m.sum())
    elig += int(ok.sum())
    excess = (-(w.astype(np.float64) / g).min(1) - float(rad))
    sel = excess > 0
    if sel.any():
        worst = max(worst, float((excess[sel] / (e1 + e2)[sel].astype(np.float64)).max()))
print("%s NF=%d IS=%d views=%d seed=%d: kept pairs %d, eligible %.3f, violations %d, worst excess/(E1+E2) %.3f"
      % (scene, NF, IS, B, seed, pairs, elig / max(pairs, 1), viol, worst))
