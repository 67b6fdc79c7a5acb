#!/usr/bin/env python3
"""Golden vectors for the texel samplers, generated from the reference's own kernels compiled for the host
(oracle/_ref/libtextures_ref.so — needs /root/reference).  Small synthetic image + texture coordinates that
exercise wrapping (negative, > 1), both triangle halves of the softras lattice, bilinear and nearest.
Coordinates that are exact integers are avoided: the reference's n3mr kernel oscillates on them (see
jrender_amd/io/obj.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import TexturesOracle      # noqa: E402

rng = np.random.default_rng(2024)
o = TexturesOracle()
image = rng.uniform(0, 1, (23, 17, 3)).astype(np.float32)
NF = 40
out = {"image": image}
tc_in = rng.uniform(0.02, 0.98, (NF, 3, 2)).astype(np.float32)
upd = (rng.uniform(size=NF) < 0.8).astype(np.int32)
for R in (1, 3, 5):
    base = rng.uniform(0, 1, (NF, R * R, 3)).astype(np.float32)
    out["softras_R%d_in" % R] = base
    out["softras_R%d_out" % R] = o.softras(image, tc_in, base, upd)
tc_wild = rng.uniform(-1.7, 2.7, (NF, 3, 2)).astype(np.float32)
for ts in (2, 4):
    base = rng.uniform(0, 1, (NF, ts, ts, ts, 3)).astype(np.float32)
    out["n3mr_ts%d_in" % ts] = base
    for w in range(4):
        for b in range(2):
            tc = tc_in if w == 3 else tc_wild
            out["n3mr_ts%d_w%d_b%d_out" % (ts, w, b)] = o.n3mr(image, tc, base, upd, w, b)
out.update(tc_in=tc_in, tc_wild=tc_wild, is_update=upd)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "textures_golden.npz"), **out)
print("written", sum(v.nbytes for v in out.values()), "bytes")
