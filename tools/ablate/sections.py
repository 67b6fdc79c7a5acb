#!/usr/bin/env python3
"""GPU box: where do the raster kernels spend their (wall-clock, per-wavefront) time?  Needs the instrumented
build (python tools/ablate/build.py sections): s_memtime laps per section, summed over wavefronts.  The laps
perturb the kernels (each is an s_memtime + s_waitcnt): read the SHARES, not the totals.
    python tools/ablate/sections.py [--heavy] [--shape faces,views,image_size[,sigma[,rgb]]]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HEAVY = "--heavy" in sys.argv          # wavefront 0 of the heaviest bin's 16 tiles in the four-wavefront path, ONE view
SETUP = "--setup" in sys.argv          # k_face_setup / k_bin_fill instead of the raster kernels (variant sections_setup)
os.environ.setdefault("JRENDER_LIB", os.path.join(ROOT, "jrender_amd", "csrc", "libjrender_hip_sections%s.so" % ("_heavy" if HEAVY else ("_setup" if SETUP else ""))))
sys.path.insert(0, ROOT)
from jrender_amd import _ffi, synthetic as syn                                     # noqa: E402
from jrender_amd.renderer.dr.softras.soft_rasterize import SoftRasterizeFunction   # noqa: E402

ctx = _ffi.Context(0)
NB, NF, IS, KW = (1 if HEAVY else 8), 39000, 1024, {}
if "--shape" in sys.argv:          # --shape faces,views,image_size[,sigma[,rgb]]   e.g. config 4's operator call: --heavy --shape 3300,64,64,1e-4,hard
    a = sys.argv[sys.argv.index("--shape") + 1].split(",")
    NF, NB, IS = int(a[0]), int(a[1]), int(a[2])
    if len(a) > 3:
        KW["sigma_val"] = float(a[3])
    if len(a) > 4:
        KW["aggr_func_rgb"] = a[4]
fv, tex = syn.sphere_views(NF, NB)
if "--spot" in sys.argv:            # --spot IMAGE_SIZE: the spot cow (5 856 faces, 25 texels per face) of BASELINE configs[0..1], one view
    z = np.load(os.path.join(ROOT, "tests", "golden", "g1_spot.npz"))
    fv, tex, NB, IS = z["fv"], z["tex"], 1, int(sys.argv[sys.argv.index("--spot") + 1])
fv, tex = ctx.array(fv), ctx.array(tex)
g = ctx.array(np.random.default_rng(7).uniform(-1, 1, (NB, 4, IS, IS)).astype(np.float32))
fn = SoftRasterizeFunction(image_size=IS, ctx=ctx, **KW)
for _ in range(2):
    fn.execute(fv, tex); fn.grad(g)
ctx.section_clocks()
for _ in range(5):
    fn.execute(fv, tex); fn.grad(g)
c = np.asarray(ctx.section_clocks(), np.float64)
if SETUP:
    nw_setup = -(-fv.shape[0] * fv.shape[1] // 64)
    for name, lo, labels in (("k_face_setup", 0, ["face load + face_setup", "faces_info store", "record + store", "pixel ranges", "bin counts"]),
                             ("k_bin_fill", 8, ["rectangle load", "append loop"])):
        tot = c[lo:lo + 8].sum()
        print(name, "clocks per wavefront and launch %.0f (%.2f us at 2.4 GHz)" % (tot / nw_setup / 5, tot / nw_setup / 5 / 2400))
        for i, lab in enumerate(labels):
            print("   %-24s %5.1f %%   %8.0f clocks" % (lab, 100 * c[lo + i] / max(tot, 1), c[lo + i] / nw_setup / 5))
    sys.exit(0)
if HEAVY:
    tot = c[0:8].sum()
    nt = (ctx.bin_size() // 8) ** 2      # tiles of the heaviest bin (JR_BIN_SIZE=8: the ONE tile that lists the most faces)
    info = ctx.last_launch()
    print("heaviest bin (%d faces listed, bin size %d: %d tile(s)), wavefront %s of a %d-wavefront workgroup, %d launches: %.3g clocks per tile and launch"
          % (ctx.last_stats()["max_faces_in_bin"], ctx.bin_size(), nt, os.environ.get("JR_SECTIONS_WAVE_LABEL", "?"), info["wavefronts_per_workgroup"], 5, tot / nt / 5))
    for i, lab in enumerate(["barrier wait / stage", "masks", "pair list", "claimed tasks / evaluate", "-", "inside pairs", "apply", "stores"]):
        print("   %-24s %5.1f %%   %9.0f clocks per tile" % (lab, 100 * c[i] / max(tot, 1), c[i] / nt / 5))
    sys.exit(0)
for name, lo, labels in (("forward", 0, ["set-up", "cull + stage", "ballots + pre-cull", "raster loop", "stores"]),
                         ("backward", 8, ["tile state + sort", "extraction", "staging + items", "gather", "pair arithmetic", "reduce + atomics"])):
    tot = c[lo:lo + 8].sum()
    print(name, "total clocks %.3g" % tot)
    for i, lab in enumerate(labels):
        print("   %-20s %5.1f %%" % (lab, 100 * c[lo + i] / max(tot, 1)))
