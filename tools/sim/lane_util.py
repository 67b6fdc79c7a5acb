#!/usr/bin/env python3
"""CPU model of the forward raster loop's lane utilisation on the headline scene (no GPU needed).

For every 8x8 tile of one 1024^2 view of the 39k-face sphere: the ascending list of faces whose grown
box touches the tile, each pixel's private subset (per-pixel box test, as the kernel's ballots do), and
per (pixel, face) whether the pair survives the distance cull (float64 geometry, not bit-exact: this is
a statistics tool).  Then trip counts of alternative schedules:
  batch64   : what the kernel does — batches of 64 survivors, trips = sum over batches of max_i n_i
  batch128  : same with 128-face batches
  stream    : lanes never wait at batch boundaries, trips = max_i sum n_i   (ring buffer upper bound)
  ideal     : ceil(total pairs / 64)
and the same after removing pairs a conservative pre-cull would reject.
"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from jrender_amd import synthetic as syn

IS = int(os.environ.get("IS", 1024)); NF = int(os.environ.get("NF", 39000))
fv, _tex = syn.sphere_views(NF, 1)
f = fv[0].astype(np.float64)                       # [NF,3,3]
# the kernel's pre-cull margin needs the reference's float face_inv (its det rounding is the dominant term)
from oracle import Oracle
_info = Oracle("port").forward_subset(fv, _tex, np.zeros(1, np.int64), image_size=IS)["faces_info"][0]
def kernel_margin(ids, rad):
    F = np.float32; EPS = F(2.0 ** -24); rad = F(rad)
    inv = _info[ids, :9]; xx = fv[0][ids][:, :, 0]; yy = fv[0][ids][:, :, 1]
    xlo_, xhi_, ylo_, yhi_ = xx.min(1) - rad, xx.max(1) + rad, yy.min(1) - rad, yy.max(1) + rad
    X = np.maximum(np.abs(xlo_), np.abs(xhi_)); Y = np.maximum(np.abs(ylo_), np.abs(yhi_))
    pos = np.sqrt(X * X + Y * Y); ext = (xhi_ - xlo_) + (yhi_ - ylo_)
    g = np.stack([np.sqrt(inv[:, 3 * q] ** 2 + inv[:, 3 * q + 1] ** 2) for q in range(3)], 1)
    S = np.stack([np.abs(inv[:, 3 * q]) * X + np.abs(inv[:, 3 * q + 1]) * Y + np.abs(inv[:, 3 * q + 2]) for q in range(3)], 1)
    sv = (S * np.sqrt(xx * xx + yy * yy)).sum(1)
    ca = np.abs((inv[:, 0] + inv[:, 3]) + inv[:, 6]); cb = np.abs((inv[:, 1] + inv[:, 4]) + inv[:, 7])
    cd = np.abs(((inv[:, 2] + inv[:, 5]) + inv[:, 8]) - F(1))
    e1 = ((ca * X + cb * Y + cd) + F(4) * EPS * S.sum(1)) * pos
    e2 = F(3) * EPS * sv + F(4) * EPS * (g.max(1) * ext + F(1)) * pos
    m = F(2.5) * (e1 + e2) + F(1.0001) * rad
    return np.where(m <= F(1.5) * rad, m, np.inf).astype(np.float64)
sigma, dist_eps = 1e-5, np.log(1 / 1e-4 - 1)
thr = dist_eps * sigma; rad = np.sqrt(thr)
x = f[:, :, 0]; y = f[:, :, 1]
xlo, xhi = x.min(1) - rad, x.max(1) + rad
ylo, yhi = y.min(1) - rad, y.max(1) + rad
centre = (2 * np.arange(IS) + 1 - IS) / IS        # pixel centres; row r has y = centre[IS-1-r]
# pixel ranges (inclusive) of each face box
cx0 = np.searchsorted(centre, xlo, "left"); cx1 = np.searchsorted(centre, xhi, "right") - 1
cy0 = np.searchsorted(centre, ylo, "left"); cy1 = np.searchsorted(centre, yhi, "right") - 1   # in yi space
ok = (cx0 <= cx1) & (cy0 <= cy1)
T = 8
nt = IS // T
tile_faces = [[] for _ in range(nt * nt)]
for i in np.nonzero(ok)[0]:
    for ty in range(cy0[i] // T, cy1[i] // T + 1):
        for tx in range(cx0[i] // T, cx1[i] // T + 1):
            tile_faces[ty * nt + tx].append(i)

def seg_dist2(px, py, ax, ay, bx, by):
    dx, dy = bx - ax, by - ay
    t = ((px - ax) * dx + (py - ay) * dy) / np.maximum(dx * dx + dy * dy, 1e-300)
    t = np.clip(t, 0, 1)
    ex, ey = ax + t * dx - px, ay + t * dy - py
    return ex * ex + ey * ey

rng = np.random.default_rng(0)
tiles = [t for t in range(nt * nt) if tile_faces[t]]
sample = rng.choice(tiles, size=min(len(tiles), int(os.environ.get("TILES", 1500))), replace=False)
acc = {}
def add(k, v): acc[k] = acc.get(k, 0) + v
for t in sample:
    ids = np.asarray(tile_faces[t]); ty, tx = divmod(t, nt)
    px = centre[tx * T:(tx + 1) * T]; py = centre[ty * T:(ty + 1) * T]        # yi space
    PX, PY = np.meshgrid(px, py)                                             # [8,8]
    PX = PX.reshape(1, 64); PY = PY.reshape(1, 64)
    inbox = (PX >= xlo[ids, None]) & (PX <= xhi[ids, None]) & (PY >= ylo[ids, None]) & (PY <= yhi[ids, None])   # [n,64]
    X = x[ids]; Y = y[ids]
    # inside test + distance
    d2 = np.minimum.reduce([seg_dist2(PX, PY, X[:, k, None], Y[:, k, None], X[:, (k + 1) % 3, None], Y[:, (k + 1) % 3, None]) for k in range(3)])
    def edge(k):
        a, b = k, (k + 1) % 3
        return (X[:, b, None] - X[:, a, None]) * (PY - Y[:, a, None]) - (Y[:, b, None] - Y[:, a, None]) * (PX - X[:, a, None])
    e0, e1, e2 = edge(0), edge(1), edge(2)
    inside = ((e0 >= 0) & (e1 >= 0) & (e2 >= 0)) | ((e0 <= 0) & (e1 <= 0) & (e2 <= 0))
    survive = inbox & (inside | (d2 < thr))
    # conservative pre-culls (float64 here; the kernel would add a safety margin)
    area2 = np.abs((X[:, 1] - X[:, 0]) * (Y[:, 2] - Y[:, 0]) - (X[:, 2] - X[:, 0]) * (Y[:, 1] - Y[:, 0]))
    sgn = np.sign((X[:, 1] - X[:, 0]) * (Y[:, 2] - Y[:, 0]) - (X[:, 2] - X[:, 0]) * (Y[:, 1] - Y[:, 0]))[:, None]
    L = [np.hypot(X[:, (k + 1) % 3] - X[:, k], Y[:, (k + 1) % 3] - Y[:, k])[:, None] for k in range(3)]
    m = kernel_margin(ids, rad)[:, None]
    add("margin_over_rad", float(np.mean(np.minimum(m, 10 * rad)) / rad))
    halfplane = inbox & ~((sgn * e0 < -m * L[0]) | (sgn * e1 < -m * L[1]) | (sgn * e2 < -m * L[2]))
    cxm = X.mean(1)[:, None]; cym = Y.mean(1)[:, None]
    Rc = np.sqrt(np.max((X - cxm) ** 2 + (Y - cym) ** 2, axis=1))[:, None]
    circle = (PX - cxm) ** 2 + (PY - cym) ** 2 < (Rc + np.minimum(m, 1e9)) ** 2
    both = halfplane & circle
    assert not (survive & ~both).any()
    for name, M in (("box", inbox), ("halfplane", halfplane), ("hp+circle", both), ("exact", survive)):
        keep = M.any(1)                       # faces nobody needs are dropped from the batch by the tile cull
        Mk = M[keep]
        n = Mk.shape[0]
        add(name + ".pairs", int(Mk.sum()))
        add(name + ".faces", n)
        for bs in (64, 128):
            trips = sum(int(Mk[s:s + bs].sum(0).max()) for s in range(0, n, bs))
            add("%s.trips%d" % (name, bs), trips)
        # what the kernel does: batches are formed from the BOX survivors, the refined masks only thin them out
        kb = inbox.any(1)
        Mb = M[kb]
        add(name + ".trips64boxbatch", sum(int(Mb[s:s + 64].sum(0).max()) for s in range(0, Mb.shape[0], 64)))
        add(name + ".stream", int(Mk.sum(0).max()) if n else 0)
        add(name + ".ideal", -(-int(Mk.sum()) // 64))
    add("tiles", 1)
    add("survive_pairs", int(survive.sum())); add("inside_pairs", int((inbox & inside).sum()))
print("tiles sampled", acc["tiles"], " box pairs/tile %.0f  faces/tile %.1f" % (acc["box.pairs"] / acc["tiles"], acc["box.faces"] / acc["tiles"]))
print("mean margin/rad %.3f" % (acc["margin_over_rad"] / acc["tiles"]))
print("survive %.3f of box pairs, inside %.3f" % (acc["survive_pairs"] / acc["box.pairs"], acc["inside_pairs"] / acc["box.pairs"]))
for name in ("box", "halfplane", "hp+circle", "exact"):
    p = acc[name + ".pairs"]
    print("%-10s box-batched trips64 %d (%.3f of box)" % (name, acc[name + ".trips64boxbatch"], acc[name + ".trips64boxbatch"] / acc["box.trips64boxbatch"]))
    print("%-10s pairs %.3f | trips64 %7d (util %.2f) trips128 %7d (%.2f) stream %7d (%.2f) ideal %7d | trips64 vs box %.3f"
          % (name, p / acc["box.pairs"], acc[name + ".trips64"], p / 64 / acc[name + ".trips64"],
             acc[name + ".trips128"], p / 64 / acc[name + ".trips128"], acc[name + ".stream"], p / 64 / acc[name + ".stream"],
             acc[name + ".ideal"], acc[name + ".trips64"] / acc["box.trips64"]))
