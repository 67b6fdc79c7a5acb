#!/usr/bin/env python3
"""CPU model of the counter bumps of the list-building kernels (binning.hip: k_face_setup's bin counts, k_bin_fill's cursor
bumps): how many device-scope atomics does a launch issue for a given number of ballot-matching rounds?

    python tools/sim/bin_atomics.py [--faces 39000] [--batch 8] [--image-size 1024] [--bin 32] [--scene sphere|soup]

The kernels give one lane to a face and walk the bins of its pixel rectangle, one bin per lane and loop trip; in a trip the lanes
that target the same bin are grouped by `wave_bin_match` (match against the first pending lane, one group per round) and each
group issues ONE atomic; lanes still unmatched when the rounds are used up issue one each.  This replays exactly that on the
synthetic scene of the bench (jrender_amd/synthetic.py) and counts.  It explains the measured times of profiles/
r06_c16_match_rounds.txt: the kernels' time follows the number of atomics, not the bytes they store.
The rectangle here is the face's box widened by the cull radius sqrt(dist_eps * sigma) and clipped to pixel centres - what
binning.hip's pixel_range computes exactly (the model does not need its last-ulp walk)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from jrender_amd import synthetic as syn          # noqa: E402


def rectangles(fv, IS, sigma=1e-5, dist_eps=1e-4):
    """-> px0, px1, row0, row1 per face (inclusive; empty when px0 > px1), flattened over [B * NF]"""
    rad = np.sqrt(np.log(1.0 / dist_eps - 1.0) * sigma)
    x, y = fv[..., 0].reshape(-1, 3).astype(np.float64), fv[..., 1].reshape(-1, 3).astype(np.float64)
    xlo, xhi, ylo, yhi = x.min(1) - rad, x.max(1) + rad, y.min(1) - rad, y.max(1) + rad
    # pixel centre c(i) = (2 i + 1 - IS) / IS;  lo <= c(i) <= hi
    lo = lambda v: np.clip(np.ceil((v * IS + IS - 1) / 2), 0, IS - 1).astype(np.int64)
    hi = lambda v: np.clip(np.floor((v * IS + IS - 1) / 2), -1, IS - 1).astype(np.int64)
    px0, px1, yi0, yi1 = lo(xlo), hi(xhi), lo(ylo), hi(yhi)
    empty = (px0 > px1) | (yi0 > yi1) | (xhi < -1) | (xlo > 1) | (yhi < -1) | (ylo > 1)
    row0, row1 = IS - 1 - yi1, IS - 1 - yi0
    px0 = np.where(empty, 1, px0); px1 = np.where(empty, 0, px1)
    return px0, px1, row0, row1


def count(fv, IS, bin_px, rounds_list):
    B, NF = fv.shape[:2]
    px0, px1, row0, row1 = rectangles(fv, IS)
    lg = int(np.log2(bin_px))
    bins_x = (IS + bin_px - 1) >> lg
    bx0, by0 = px0 >> lg, row0 >> lg
    nbx = np.where(px0 <= px1, (px1 >> lg) - bx0 + 1, 0)
    nb = nbx * np.where(px0 <= px1, (row1 >> lg) - by0 + 1, 0)
    view = np.arange(B * NF) // NF
    base = view * bins_x * bins_x
    total = B * NF
    out = {r: 0 for r in rounds_list}
    pairs = int(nb.sum())
    trips = distinct = 0
    for w0 in range(0, total, 64):
        sl = slice(w0, min(w0 + 64, total))
        n, nx, x0, y0, bb = nb[sl], np.maximum(nbx[sl], 1), bx0[sl], by0[sl], base[sl]
        for it in range(int(n.max()) if n.size else 0):
            act = it < n
            tb = np.where(act, bb + (y0 + it // nx) * bins_x + x0 + it % nx, -1)
            keys = tb[act]
            trips += 1
            distinct += len(set(keys.tolist()))
            for r in rounds_list:
                out[r] += atomics_of_a_trip(keys, r)
    return dict(pairs=pairs, trips=trips, wavefronts=(total + 63) // 64, mean_distinct_bins_per_trip=distinct / max(trips, 1), atomics=out)


def atomics_of_a_trip(keys, rounds):
    """wave_bin_match: lanes in lane order; a round takes the first pending lane's bin and retires every lane with that bin.
    rounds = ('adaptive', n): up to n rounds, stop once three rounds have found groups of one (and at least 4 were run)."""
    adaptive = isinstance(rounds, tuple)
    limit = rounds[1] if adaptive else rounds
    pending = list(keys.tolist())
    atomics = singles = 0
    for rnd in range(limit):
        if not pending:
            break
        k = pending[0]
        grp = sum(1 for q in pending if q == k)
        singles += grp == 1
        if adaptive and rnd >= 3 and singles >= 3:
            break                                   # (the kernel leaves THIS round's group unmatched as well)
        pending = [q for q in pending if q != k]
        atomics += 1
    return atomics + len(pending)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--faces", type=int, default=39000)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--image-size", type=int, default=1024)
    ap.add_argument("--bin", type=int, default=32)
    ap.add_argument("--scene", default="sphere")
    a = ap.parse_args()
    fv = (syn.sphere_views(a.faces, a.batch) if a.scene == "sphere" else syn.triangle_soup(a.faces, a.batch, seed=0))[0]
    r = count(fv, a.image_size, a.bin, [4, 8, 16, ("adaptive", 16), 64])
    print("%s %d faces x %d views at %d^2, %d-pixel bins: %d (face, bin) pairs, %d wavefronts, %d trips, %.1f distinct bins per trip"
          % (a.scene, a.faces, a.batch, a.image_size, a.bin, r["pairs"], r["wavefronts"], r["trips"], r["mean_distinct_bins_per_trip"]))
    for k, v in r["atomics"].items():
        print("  rounds %-16s atomics per launch %8d   (%.2f per pair)" % (k, v, v / max(r["pairs"], 1)))
    return r


if __name__ == "__main__":
    main()
