#!/usr/bin/env python3
"""Which edge is nearest to an INSIDE pixel?  (CPU; needs the oracle.)

The reference projects an inside pixel onto all three edge LINES and keeps the nearest (SRK:68-105, strict '<',
first edge wins ties).  Geometrically the distance to the line of edge e (vertices e, e+1) is w_k * h_k for the
opposite vertex k = e+2: its barycentric weight times its altitude h_k = 1 / |grad w_k|.  The kernels use that to
SELECT the edge worth projecting exactly (softras_device.h: inside_edge_select): with a per-face band B that bounds
how far the reference's float result can be from the geometric distance, every edge with
        w_k h_k <= min_j (w_j h_j) + 2 B
is a candidate; one candidate -> only that edge is projected (it IS the reference's argmin), several -> all three
as before.  The record stores s_k = h_k / (2 B), so the test is  w_k s_k <= min_j (w_j s_j) + 1.

B (float32, the formula of csrc/softras_device.h: build_face_geo) = 32 EPS / sqrt(min |Dn|) + 3 (E1 + E2):
  EPS / sqrt(Dn_e) = scale of the foot point's displacement along the edge (sym differences carry ~4 EPS absolute
                     error, the quotient divides by |edge|^2),
  E1               = the weights do not sum to 1 (one rounded det): pixel displaced by (sum w - 1) * position,
  E2               = rounding of the weights times the vertex positions, of the offset products.
This tool replays the reference's float32 arithmetic (association order of SRK:73-92) on sampled inside pairs:
  wrong    = pairs whose reference argmin edge is NOT a candidate            (must be 0),
  noise/B  = max |sqrt(dd_e) - w_k h_k| / B                                  (safety factor = 1 / that),
  share of inside pairs with 1 / 2 / 3 candidates (the second and third cost a full three-edge evaluation of
  the wavefront trip they are in).
usage: inside_edge.py IS NFACES sphere|soup|fuzz [views] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from jrender_amd import synthetic as syn      # noqa: E402
from oracle import Oracle                     # noqa: E402

F = np.float32
IS, NF, scene = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
B = int(sys.argv[4]) if len(sys.argv) > 4 else 2
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 0
K = 16
rng = np.random.default_rng(seed)
if scene == "sphere":
    fv, tex = syn.sphere_views(NF, B, azimuth0=7.0 * seed)
elif scene == "soup":
    fv, tex = syn.triangle_soup(NF, B, seed=seed)
else:   # fuzz: triangles of wildly different sizes and aspect ratios, slivers included
    c = rng.uniform(-1.1, 1.1, (B, NF, 1, 2))
    size = 10.0 ** rng.uniform(-3.0, -0.3, (B, NF, 1, 1))
    off = rng.uniform(-1, 1, (B, NF, 3, 2)) * size
    squash = 10.0 ** rng.uniform(-3, 0, (B, NF, 1, 1))
    off[..., 1:2] *= squash
    ang = rng.uniform(0, np.pi, (B, NF, 1, 1))
    rot = np.concatenate([off[..., 0:1] * np.cos(ang) - off[..., 1:2] * np.sin(ang),
                          off[..., 0:1] * np.sin(ang) + off[..., 1:2] * np.cos(ang)], -1)
    z = rng.uniform(2, 4, (B, NF, 3, 1))
    fv = np.concatenate([c + rot, z], -1).astype(F)
    tex = rng.uniform(0, 1, (B, NF, 1, 3)).astype(F)
port = Oracle("port", nthreads=0)
pix = rng.choice(B * IS * IS, min(B * IS * IS, 800000), replace=False)
sub = port.forward_subset(fv, tex, pix, image_size=IS, max_faces_per_pixel_for_grad=K)
ids, info = sub["ids"], sub["faces_info"]
EPS = F(2.0 ** -24)
b, r = np.divmod(pix, IS * IS)
row, col = np.divmod(r, IS)
xp = ((2 * col + 1 - IS).astype(F) / F(IS)).astype(F)
yp = ((2 * (IS - 1 - row) + 1 - IS).astype(F) / F(IS)).astype(F)


def select_scales(inv, x, y, Dn):
    """s_k = h_k / (2 B) in float32, mirroring build_face_geo"""
    X = np.abs(x).max(1)
    Y = np.abs(y).max(1)
    pos = np.sqrt(X * X + Y * Y)
    ext = (x.max(1) - x.min(1)) + (y.max(1) - y.min(1))
    g = np.stack([np.sqrt(inv[:, 3 * q] * inv[:, 3 * q] + inv[:, 3 * q + 1] * inv[:, 3 * q + 1]) for q in range(3)], 1).astype(F)
    S = np.stack([np.abs(inv[:, 3 * q]) * X + np.abs(inv[:, 3 * q + 1]) * Y + np.abs(inv[:, 3 * q + 2]) for q in range(3)], 1).astype(F)
    sv = (S * np.sqrt(x * x + y * y)).sum(1)
    ca = np.abs((inv[:, 0] + inv[:, 3]) + inv[:, 6])
    cb = np.abs((inv[:, 1] + inv[:, 4]) + inv[:, 7])
    cd = np.abs(((inv[:, 2] + inv[:, 5]) + inv[:, 8]) - F(1))
    e1 = ((ca * X + cb * Y + cd) + F(4) * EPS * S.sum(1)) * pos
    e2 = F(3) * EPS * sv + F(4) * EPS * (g.max(1) * ext + F(1)) * pos
    dmin = np.abs(Dn).min(1)
    with np.errstate(all="ignore"):
        band = (F(32) * EPS / np.sqrt(dmin) + F(3) * (e1 + e2)).astype(F)
        h = (F(1) / g).astype(F)
        s = (h / (F(2) * band)[:, None]).astype(F)
    return s, h, band


tot = wrong = 0
noise_rel = 0.0
need = np.zeros(4, np.int64)
for k in range(K):
    fid = ids[:, k]
    m = fid >= 0
    inv = info[b[m], fid[m], :9]
    sym = info[b[m], fid[m], 9:18].reshape(-1, 3, 3)
    f = fv[b[m], fid[m]]
    x, y = f[:, :, 0], f[:, :, 1]
    w = np.stack([(inv[:, 3 * q] * xp[m] + inv[:, 3 * q + 1] * yp[m]) + inv[:, 3 * q + 2] for q in range(3)], 1).astype(F)
    ins = ((w > 0) & (w < 1)).all(1)
    if not ins.any():
        continue
    inv, sym, x, y, w = inv[ins], sym[ins], x[ins], y[ins], w[ins]
    dd = np.zeros((w.shape[0], 3), F)
    Dn = np.zeros((w.shape[0], 3), F)
    with np.errstate(all="ignore"):
        for e in range(3):                                          # SRK:73-92
            e1_ = (e + 1) % 3
            a = (sym[:, e] - sym[:, e1_]).astype(F)
            num = ((w[:, 0] * a[:, 0] + w[:, 1] * a[:, 1]) + w[:, 2] * a[:, 2]) - a[:, e1_]
            Dn[:, e] = a[:, e] - a[:, e1_]
            tv = (num / Dn[:, e]).astype(F)
            t = np.zeros_like(w)
            t[:, e] = tv
            t[:, e1_] = F(1) - tv
            u = (t - w).astype(F)
            ex = (u[:, 0] * x[:, 0] + u[:, 1] * x[:, 1]) + u[:, 2] * x[:, 2]
            ey = (u[:, 0] * y[:, 0] + u[:, 1] * y[:, 1]) + u[:, 2] * y[:, 2]
            dd[:, e] = ex * ex + ey * ey
    # reference: best starts at 1e8, strict '<' in edge order; NaN never wins
    ddc = np.where(dd < F(1e8), dd, np.inf)
    ref_e = np.argmin(ddc, 1)
    has_ref = np.isfinite(ddc).any(1)
    s, h, band = select_scales(inv, x, y, Dn)
    with np.errstate(all="ignore"):
        q = (w * s).astype(F)                                       # by vertex
        lim = q.min(1) + F(1)
        cand_v = q <= lim[:, None]
    cand_e = np.stack([cand_v[:, 2], cand_v[:, 0], cand_v[:, 1]], 1)   # edge e is opposite vertex e+2
    nb = cand_e.sum(1)
    single = nb == 1
    # single-candidate pairs must have picked the reference's edge (and the reference must have one)
    sel_e = np.argmax(cand_e, 1)
    bad = single & has_ref & (sel_e != ref_e)
    wrong += int(bad.sum())
    tot += len(ref_e)
    for c in (0, 1, 2, 3):
        need[c] += int((nb == c).sum())
    with np.errstate(all="ignore"):
        qe = np.stack([w[:, 2] * h[:, 2], w[:, 0] * h[:, 0], w[:, 1] * h[:, 1]], 1).astype(np.float64)
        d = np.abs(np.sqrt(dd.astype(np.float64)) - qe) / band.astype(np.float64)[:, None]
    d = d[np.isfinite(d)]
    if d.size:
        noise_rel = max(noise_rel, float(d.max()))
print("%s NF=%d IS=%d views=%d seed=%d: inside pairs %d, wrong %d, noise/B %.3f, candidates 0/1/2/3: %.4f %.4f %.4f %.4f"
      % (scene, NF, IS, B, seed, tot, wrong, noise_rel, need[0] / max(tot, 1), need[1] / max(tot, 1), need[2] / max(tot, 1),
         need[3] / max(tot, 1)))
