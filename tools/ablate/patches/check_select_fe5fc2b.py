#!/usr/bin/env python3
"""GPU box: does inside_edge_select ever change a result?  Needs the instrumented build
(python tools/ablate/build.py check_select): the backward projects all three edges of every INSIDE (pixel, face)
pair as well and counts the pairs whose nearest point / barycentric offsets differ in any bit from what the
selected single projection gave.  Scenes: the headline sphere, the random-triangle soup, sliver-rich random
meshes at several sizes.  Prints one line per scene; the last line is the total (mismatches must be 0)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("JRENDER_LIB", os.path.join(ROOT, "jrender_amd", "csrc", "libjrender_hip_check_select.so"))
sys.path.insert(0, ROOT)
from jrender_amd import _ffi, synthetic as syn                                     # noqa: E402
from jrender_amd.renderer.dr.softras.soft_rasterize import SoftRasterizeFunction   # noqa: E402


def fuzz_mesh(rng, B, NF):
    c = rng.uniform(-1.1, 1.1, (B, NF, 1, 2))
    size = 10.0 ** rng.uniform(-3.0, -0.3, (B, NF, 1, 1))
    off = rng.uniform(-1, 1, (B, NF, 3, 2)) * size
    off[..., 1:2] *= 10.0 ** rng.uniform(-3, 0, (B, NF, 1, 1))
    ang = rng.uniform(0, np.pi, (B, NF, 1, 1))
    rot = np.concatenate([off[..., 0:1] * np.cos(ang) - off[..., 1:2] * np.sin(ang),
                          off[..., 0:1] * np.sin(ang) + off[..., 1:2] * np.cos(ang)], -1)
    fv = np.concatenate([c + rot, rng.uniform(2, 4, (B, NF, 3, 1))], -1).astype(np.float32)
    return fv, rng.uniform(0, 1, (B, NF, 1, 3)).astype(np.float32)


ctx = _ffi.Context(0)
rng = np.random.default_rng(11)
scenes = [("sphere 39000 x 8 @1024", syn.sphere_views(39000, 8), 1024),
          ("soup 39000 x 4 @1024", syn.triangle_soup(39000, 4, seed=3), 1024),
          ("sphere 280 x 4 @256", syn.sphere_views(280, 4), 256),
          ("sphere 3300 x 4 @512", syn.sphere_views(3300, 4), 512)]
for i, (nf, IS) in enumerate(((2000, 256), (20000, 512), (60000, 1024), (500, 128), (8000, 300))):
    scenes.append(("slivers %d x 2 @%d" % (nf, IS), fuzz_mesh(rng, 2, nf), IS))
tot = np.zeros(2, np.int64)
ctx.section_clocks()
for name, (fv, tex), IS in scenes:
    fn = SoftRasterizeFunction(image_size=IS, ctx=ctx)
    fn.execute(ctx.array(fv), ctx.array(tex))
    g = ctx.array(np.random.default_rng(7).uniform(-1, 1, (fv.shape[0], 4, IS, IS)).astype(np.float32))
    fn.grad(g)
    c = ctx.section_clocks()
    print("%-26s inside pairs %12d   mismatches %d   decided by the weights %.4f   inside lanes in a trip with an undecided one %.4f"
          % (name, c[18], c[19], c[16] / max(c[18], 1), c[17] / max(c[18], 1)), flush=True)
    tot += (c[18], c[19])
print("TOTAL inside pairs %d mismatches %d" % (tot[0], tot[1]))
sys.exit(1 if tot[1] or not tot[0] else 0)
