#!/usr/bin/env python3
"""GPU box: forward (incl. set-up) / backward time of small and medium configurations under the run-time launch policy
(jr_softras_set_launch_policy: heavy threshold x workgroup size), synchronised per step.  The table of
profiles/r04_experiments.md call 28 comes from this script."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jrender_amd import _ffi, synthetic as syn
from jrender_amd.renderer.dr.softras.soft_rasterize import SoftRasterizeFunction
ctx = _ffi.Context(0)
def timed(fv, tex, IS, steps=30, warm=5, **kw):
    fv, tex = ctx.array(fv), ctx.array(tex)
    B = fv.shape[0]
    g = ctx.array(np.random.default_rng(1).uniform(-1, 1, (B, 4, IS, IS)).astype(np.float32))
    fn = SoftRasterizeFunction(image_size=IS, ctx=ctx, **kw)
    ev = [ctx.event() for _ in range(3)]
    tf, tb = [], []
    for i in range(steps + warm):
        ctx.record(ev[0]); fn.execute(fv, tex); ctx.record(ev[1]); fn.grad(g); ctx.record(ev[2])
        ctx.synchronize()
        if i >= warm:
            tf.append(ctx.elapsed_ms(ev[0], ev[1])); tb.append(ctx.elapsed_ms(ev[1], ev[2]))
    info = ctx.last_launch(); st = ctx.last_stats()
    return float(np.median(tf)), float(np.median(tb)), info["heavy_bins"], info["wavefronts_per_workgroup"], st["max_faces_in_bin"]
spot = np.load(os.path.join(ROOT, "tests", "golden", "g1_spot.npz"))
v, f = syn.uv_sphere(52, 27)
rows = [("C1 spot 256^2 B=1", spot["fv"], spot["tex"], 256, {}),
        ("sphere 3300 1024^2 B=1", *syn.sphere_views(3300, 1), 1024, {}),
        ("C4-like 3300f 64^2 B=64", *syn.sphere_views(3300, 64), 64, dict(sigma_val=1e-4, aggr_func_rgb="hard")),
        ("C4-like 3300f 64^2 B=8", *syn.sphere_views(3300, 8), 64, dict(sigma_val=1e-4, aggr_func_rgb="hard")),
        ("sphere 39000 256^2 B=8", *syn.sphere_views(39000, 8), 256, {}),
        ("sphere 39000 1024^2 B=1", *syn.sphere_views(39000, 1), 1024, {})]
for hm, w in ((-1, 0), (1024, 0), (2048, 0), (0, 0), (-1, 4), (-1, 8), (256, 0)):
    ctx.set_launch_policy(hm, w)
    for name, fv, tex, IS, kw in rows:
        tf, tb, hb, wv, mx = timed(fv, tex, IS, **kw)
        print("heavy_min %5d waves %d | %-26s fwd %.3f bwd %.3f ms  heavy_bins %5d wpw %d max_in_bin %d" % (hm, w, name, tf, tb, hb, wv, mx), flush=True)
