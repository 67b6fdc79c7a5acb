#!/usr/bin/env python3
"""Which edge is nearest to an INSIDE pixel?  (CPU; needs the oracle.)

The reference projects an inside pixel onto all three edge LINES and keeps the nearest (SRK:68-105, strict '<',
first edge wins ties).  Round-2 EXPERIMENT (tools/ablate/patches/inside_edge_select_fe5fc2b.patch, not in the
product: exact but slower, profiles/r02_ab_inside_select.log): SELECT the edge worth projecting from a geometric
model of what the reference's float arithmetic computes, and fall back to all three projections when the model
cannot separate the edges:

  * the reference measures from pixel' = sum_k w_k P_k (SRK:87-92: u = t - w, offset = sum u_k P_k), and its weights
    do not sum to one (face_inv is star / det with ONE rounded det): with sm1 = ((w0 + w1) + w2) - 1 the weight of
    pixel' is   wr_k = w_k - sm1 * inv[3k+2]   (w_k is affine: w_k(sum w_j P_j) = w_k + c_k (1 - sum w_j)),
    so the distance of pixel' to the line opposite vertex k is  q_k = |wr_k| h_k,  h_k = 1 / |grad w_k|;
  * the foot point is displaced ALONG the edge by the error of the projection parameter (differences of
    face_sym = x x' + y y' + 1 carry ~4 EPS absolute error, the quotient divides by |edge|^2): the reference's
    distance is sqrt(q_k^2 + d^2) with  d <= dl_k = 32 EPS (2 + sqrt(Dmax / Dn_e)) / sqrt(Dn_e)  (second order!);
  * everything else (rounding of w, of the offsets) stays below  eta = 32 EPS (max_k S_k h_k + pos + max_k |c_k| h_k).
Edge k is a CANDIDATE iff  q_k - eta <= min_j sqrt(q_j^2 + dl_j^2) + eta.  The record stored s_k = h_k / (2 eta) and
(dl_k / (2 eta))^2, so the test is  q~_k <= min_j sqrt(q~_j^2 + dl~_j^2) + 1  (3 sqrt per pair, no division).

This tool replays the reference's float32 arithmetic (association order of SRK:73-92) on sampled inside pairs:
  wrong    = single-candidate pairs whose candidate is NOT the reference's argmin          (must be 0),
  up/down  = max of (reference - model upper bound) / band, (model lower bound - reference) / band  (must be < 1;
             1 / that = safety factor),
  share of inside pairs with 1 / 2 / 3 candidates (2 and 3 cost a full three-edge evaluation of their trip).
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
    """s_k = h_k / (2 eta), dl2_k = (dl_k / (2 eta))^2 by VERTEX k (edge k+1 is opposite), float32 like build_face_geo"""
    X = np.abs(x).max(1)
    Y = np.abs(y).max(1)
    pos = np.sqrt(X * X + Y * Y)
    with np.errstate(all="ignore"):
        g = np.stack([np.sqrt(inv[:, 3 * q] * inv[:, 3 * q] + inv[:, 3 * q + 1] * inv[:, 3 * q + 1]) for q in range(3)], 1).astype(F)
        h = (F(1) / g).astype(F)
        S = np.stack([np.abs(inv[:, 3 * q]) * X + np.abs(inv[:, 3 * q + 1]) * Y + np.abs(inv[:, 3 * q + 2]) for q in range(3)], 1).astype(F)
        aD = np.abs(Dn)
        slack = F(16) * EPS                                         # absolute error allowance of a sym difference
        ehi = np.sqrt(aD + slack)                                   # upper bound of the edge length
        L = np.sqrt(aD.max(1) + slack) + F(2.0 ** -10) * pos        # how far pixel' can be from a vertex (|sm1| <= 2^-10)
        tvb = (ehi * L[:, None] + slack) / aD + F(1)                # bound of |tv| of the reference's quotient
        dl_e = (F(float(os.environ.get("C1", 12))) * EPS * (F(1) + tvb) * ehi / aD).astype(F)   # by edge
        eta = (F(32) * EPS * ((S * h).max(1) + pos + (np.abs(inv[:, [2, 5, 8]]) * h).max(1)) + F(8) * EPS * tvb.max(1) * pos).astype(F)
        dl_v = np.stack([dl_e[:, 1], dl_e[:, 2], dl_e[:, 0]], 1)                               # vertex k <- edge k+1
        r2e = (F(1) / (F(2) * eta)).astype(F)
        s_ = (h * r2e[:, None]).astype(F)
        dl2 = ((dl_v * r2e[:, None]) ** 2).astype(F)
    return s_, dl2, h, eta, dl_v


tot = wrong = 0
noise = [0.0, 0.0]
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
    s, dl2, h, eta, dl_v = select_scales(inv, x, y, Dn)
    with np.errstate(all="ignore"):
        sm1 = (((w[:, 0] + w[:, 1]) + w[:, 2]) - F(1)).astype(F)
        wr = np.stack([w[:, q] - sm1 * inv[:, 3 * q + 2] for q in range(3)], 1).astype(F)
        q = (np.abs(wr) * s).astype(F)                              # by vertex, in units of 2 eta
        rr = np.sqrt((q * q + dl2).astype(F)).astype(F)
        lim = rr.min(1) + F(1)
        cand_v = q <= lim[:, None]
        cond = np.abs(sm1) <= F(2.0 ** -10)
    cand_e = np.stack([cand_v[:, 2], cand_v[:, 0], cand_v[:, 1]], 1)   # edge e is opposite vertex e+2
    nb = cand_e.sum(1)
    single = (nb == 1) & cond
    sel_e = np.argmax(cand_e, 1)
    bad = single & (~has_ref | (sel_e != ref_e))
    wrong += int(bad.sum())
    tot += len(ref_e)
    nb = np.where(cond, nb, 3)                                  # ill-conditioned pair: all three edges
    for c in (0, 1, 2, 3):
        need[c] += int((nb == c).sum())
    with np.errstate(all="ignore"):
        qd = (np.abs(wr) * h).astype(np.float64)                    # by vertex, NDC
        qe = np.stack([qd[:, 2], qd[:, 0], qd[:, 1]], 1)
        dle = np.stack([dl_v[:, 2], dl_v[:, 0], dl_v[:, 1]], 1).astype(np.float64)
        ref = np.sqrt(dd.astype(np.float64))
        e64 = eta.astype(np.float64)[:, None]
        up = np.where(cond[:, None], (ref - np.sqrt(qe * qe + dle * dle)) / e64, 0.0)   # must stay <= 1
        down = np.where(cond[:, None], (qe - ref) / e64, 0.0)
    for arr, idx in ((up, 0), (down, 1)):
        arr = arr[np.isfinite(arr)]
        if arr.size:
            noise[idx] = max(noise[idx], float(arr.max()))
print("%s NF=%d IS=%d views=%d seed=%d: inside pairs %d, wrong %d, up %.3f down %.3f (of eta), candidates 0/1/2/3: %.4f %.4f %.4f %.4f"
      % (scene, NF, IS, B, seed, tot, wrong, noise[0], noise[1], need[0] / max(tot, 1), need[1] / max(tot, 1), need[2] / max(tot, 1),
         need[3] / max(tot, 1)))
