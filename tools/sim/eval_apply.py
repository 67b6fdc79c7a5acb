#!/usr/bin/env python3
"""CPU model of the forward's EVALUATE / APPLY split on the headline scene (no GPU needed).

The shipped kernel (round 2) runs the whole per-(pixel, face) arithmetic with lane = pixel: a trip of the
raster loop costs the full pair function and its trip count per batch is the MAXIMUM over the 64 pixels of the
faces a pixel needs.  Round 3 splits the pair function:

  evaluate  lane = any pair of a compact pair list (pixel-major, ascending face inside a pixel), 64 pairs per
            trip, stateless arithmetic only (barycentrics -> distance -> cull -> coverage -> clip -> depth -> zn),
            one 16-byte result cell per pair in LDS
  apply     lane = pixel walks ITS contiguous cells: alpha, online softmax, K-buffer insert

A batch of staged faces is processed in ROUNDS: a round is a range of face slots whose pair count fits the
cell buffer (CAP cells); a range that does not fit is halved.  This script counts, for a sample of tiles of one
1024^2 view of the 39k-face sphere: evaluate trips, apply trips, list-building trips and rounds for several
(BATCH, CAP), and prices them with instruction counts per trip (E evaluate, A apply, L list building, R per
round, B per batch) next to the shipped schedule (trips x F).  Geometry in float64 (statistics, not bit-exact).
"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from jrender_amd import synthetic as syn

IS = int(os.environ.get("IS", 1024)); NF = int(os.environ.get("NF", 39000))
fv, _tex = syn.sphere_views(NF, 1)
f = fv[0].astype(np.float64)
sigma, dist_eps = 1e-5, np.log(1 / 1e-4 - 1)
thr = dist_eps * sigma; rad = np.sqrt(thr)
x = f[:, :, 0]; y = f[:, :, 1]
xlo, xhi = x.min(1) - rad, x.max(1) + rad
ylo, yhi = y.min(1) - rad, y.max(1) + rad
centre = (2 * np.arange(IS) + 1 - IS) / IS
cx0 = np.searchsorted(centre, xlo, "left"); cx1 = np.searchsorted(centre, xhi, "right") - 1
cy0 = np.searchsorted(centre, ylo, "left"); cy1 = np.searchsorted(centre, yhi, "right") - 1
ok = (cx0 <= cx1) & (cy0 <= cy1)
T = 8
nt = IS // T
tile_faces = [[] for _ in range(nt * nt)]
for i in np.nonzero(ok)[0]:
    for ty in range(cy0[i] // T, cy1[i] // T + 1):
        for tx in range(cx0[i] // T, cx1[i] // T + 1):
            tile_faces[ty * nt + tx].append(i)


def tile_masks(t, margin=1.25):
    """[n,64] masks of tile t: box, half-plane pre-cull (margin x radius), exact survivors; faces ascending"""
    ids = np.asarray(tile_faces[t]); ty, tx = divmod(t, nt)
    px = centre[tx * T:(tx + 1) * T]; py = centre[ty * T:(ty + 1) * T]
    PX, PY = np.meshgrid(px, py)
    PX = PX.reshape(1, 64); PY = PY.reshape(1, 64)
    inbox = (PX >= xlo[ids, None]) & (PX <= xhi[ids, None]) & (PY >= ylo[ids, None]) & (PY <= yhi[ids, None])
    X = x[ids]; Y = y[ids]

    def seg(ax, ay, bx, by):
        dx, dy = bx - ax, by - ay
        tt = np.clip(((PX - ax) * dx + (PY - ay) * dy) / np.maximum(dx * dx + dy * dy, 1e-300), 0, 1)
        ex, ey = ax + tt * dx - PX, ay + tt * dy - PY
        return ex * ex + ey * ey
    d2 = np.minimum.reduce([seg(X[:, k, None], Y[:, k, None], X[:, (k + 1) % 3, None], Y[:, (k + 1) % 3, None]) for k in range(3)])

    def edge(k):
        a, b = k, (k + 1) % 3
        return (X[:, b, None] - X[:, a, None]) * (PY - Y[:, a, None]) - (Y[:, b, None] - Y[:, a, None]) * (PX - X[:, a, None])
    e = [edge(0), edge(1), edge(2)]
    inside = ((e[0] >= 0) & (e[1] >= 0) & (e[2] >= 0)) | ((e[0] <= 0) & (e[1] <= 0) & (e[2] <= 0))
    survive = inbox & (inside | (d2 < thr))
    sgn = np.sign((X[:, 1] - X[:, 0]) * (Y[:, 2] - Y[:, 0]) - (X[:, 2] - X[:, 0]) * (Y[:, 1] - Y[:, 0]))[:, None]
    L = [np.hypot(X[:, (k + 1) % 3] - X[:, k], Y[:, (k + 1) % 3] - Y[:, k])[:, None] for k in range(3)]
    m = margin * rad
    hp = inbox & ~((sgn * e[0] < -m * L[0]) | (sgn * e[1] < -m * L[1]) | (sgn * e[2] < -m * L[2]))
    kb = inbox.any(1)                                  # the tile cull keeps faces whose box reaches a pixel
    return inbox[kb], hp[kb], survive[kb], (inbox & inside)[kb]


def rounds_of(M, cap):
    """face-slot ranges of one batch (rows of M) so that every range has <= cap pairs: halve until it fits"""
    out, j0, n = [], 0, M.shape[0]
    while j0 < n:
        j1 = n
        while M[j0:j1].sum() > cap and j1 - j0 > 1:
            j1 = j0 + (j1 - j0 + 1) // 2
        out.append((j0, j1)); j0 = j1
    return out


def main():
    rng = np.random.default_rng(0)
    tiles = [t for t in range(nt * nt) if tile_faces[t]]
    sample = rng.choice(tiles, size=min(len(tiles), int(os.environ.get("TILES", 1500))), replace=False)
    configs = [(56, 10 ** 9), (56, 1024), (56, 768), (56, 512), (40, 768), (40, 512), (32, 512), (24, 512), (32, 384), (24, 384), (16, 256)]
    acc = {c: dict(ev=0, ap=0, lb=0, rounds=0, batches=0, apcells=0) for c in configs}
    base = dict(trips=0, pairs=0, hp=0, surv=0, inside_trips=0, tiles=0, heavy_trips=0, heavy_ev=0)
    heavy = (0, None)
    for t in sample:
        inbox, hp, surv, ins = tile_masks(t)
        base["tiles"] += 1; base["pairs"] += int(inbox.sum()); base["hp"] += int(hp.sum()); base["surv"] += int(surv.sum())
        n = hp.shape[0]
        tr = sum(int(hp[s:s + 56].sum(0).max()) for s in range(0, n, 56))
        base["trips"] += tr
        if hp.sum() > heavy[0]:
            heavy = (int(hp.sum()), t, tr, n)
        for (bs, cap) in configs:
            a = acc[(bs, cap)]
            for s in range(0, n, bs):
                Mb = hp[s:s + bs]
                a["batches"] += 1
                for (j0, j1) in rounds_of(Mb, cap):
                    Mr = Mb[j0:j1]
                    pairs = int(Mr.sum())
                    if pairs == 0:
                        continue
                    a["rounds"] += 1
                    a["ev"] += -(-pairs // 64)
                    a["ap"] += int(Mr.sum(0).max())
                    a["lb"] += int(Mr.sum(0).max())
                    a["apcells"] += pairs
    F, E, A, Lc, R, Bc = 230, 165, 75, 7, 60, 150      # instructions per trip / round / batch
    print("tiles %d | box pairs/tile %.0f, after pre-cull %.0f (%.3f), exact survivors %.3f of box" % (
        base["tiles"], base["pairs"] / base["tiles"], base["hp"] / base["tiles"], base["hp"] / base["pairs"], base["surv"] / base["pairs"]))
    cur = base["trips"] * F + 0
    print("shipped schedule: %d trips (lane use %.2f) x %d instr = %.0f per tile" % (
        base["trips"], base["hp"] / 64 / base["trips"], F, cur / base["tiles"]))
    print("heaviest sampled tile: %d pairs, %d faces, %d trips shipped" % (heavy[0], heavy[3], heavy[2]))
    print("%-12s %8s %8s %8s %7s %7s | %9s %6s" % ("batch,cap", "ev trips", "ap trips", "ap util", "rounds", "batches", "instr/tile", "vs now"))
    for c in configs:
        a = acc[c]
        cost = a["ev"] * E + a["ap"] * A + a["lb"] * Lc + a["rounds"] * R + a["batches"] * Bc
        print("%-12s %8d %8d %8.2f %7d %7d | %9.0f %6.3f" % (
            "%d,%s" % (c[0], "inf" if c[1] > 10 ** 8 else c[1]), a["ev"], a["ap"], a["apcells"] / 64 / a["ap"], a["rounds"], a["batches"],
            cost / base["tiles"], cost / (cur + base["tiles"] * 0 + (acc[(56, 10 ** 9)]["batches"]) * Bc)))


if __name__ == "__main__":
    main()
