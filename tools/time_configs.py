#!/usr/bin/env python3
"""GPU box: fwd+bwd latency of the BASELINE.json configurations other than the headline one (which bench.py
times), device-resident inputs, median of 30 steps after 5 warm-ups.  One line per configuration."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jrender_amd import _ffi, synthetic as syn                                     # noqa: E402
from jrender_amd.renderer.dr.softras.soft_rasterize import SoftRasterizeFunction   # noqa: E402

ctx = _ffi.Context(0)


def timed(fv, tex, IS, steps=30, warm=5, **kw):
    fv, tex = ctx.array(fv), ctx.array(tex)
    B = fv.shape[0]
    g = ctx.array(np.random.default_rng(1).uniform(-1, 1, (B, 4, IS, IS)).astype(np.float32))
    fn = SoftRasterizeFunction(image_size=IS, ctx=ctx, **kw)
    ev = [ctx.event() for _ in range(steps + 1)]
    for _ in range(warm):
        fn.execute(fv, tex); fn.grad(g)
    ctx.record(ev[0])
    for i in range(steps):
        fn.execute(fv, tex); fn.grad(g)
        ctx.record(ev[i + 1])
    ctx.synchronize()
    return float(np.median([ctx.elapsed_ms(ev[i], ev[i + 1]) for i in range(steps)]))


spot = np.load(os.path.join(ROOT, "tests", "golden", "g1_spot.npz"))
rows = [("C1 spot cow 5 856 faces T=25, 256^2, B=1 (silhouette config)", spot["fv"], spot["tex"], 256, {}),
        ("C2 spot cow 5 856 faces T=25, 1024^2, B=1", spot["fv"], spot["tex"], 1024, {}),
        ("sphere 280 faces, 256^2, B=1", *syn.sphere_views(280, 1), 256, {}),
        ("sphere 3 300 faces, 1024^2, B=1", *syn.sphere_views(3300, 1), 1024, {}),
        ("sphere 39 000 faces, 1024^2, B=1", *syn.sphere_views(39000, 1), 1024, {}),
        ("C3 sphere 39 000 faces, 1024^2, B=8, K=32", *syn.sphere_views(39000, 8), 1024, dict(max_faces_per_pixel_for_grad=32)),
        ("C3 sphere 39 000 faces, 1024^2, B=8, K=64", *syn.sphere_views(39000, 8), 1024, dict(max_faces_per_pixel_for_grad=64)),
        ("C4-like sphere 3 300 faces, 64^2, B=64, sigma 1e-4, hard rgb", *syn.sphere_views(3300, 64), 64, dict(sigma_val=1e-4, aggr_func_rgb="hard"))]
for name, fv, tex, IS, kw in rows:
    print("%-66s %8.3f ms fwd+bwd" % (name, timed(fv, tex, IS, **kw)), flush=True)
import re
import subprocess
# C4 (BASELINE configs[3]): the optimisation LOOP of demo2 (the script's own clock around its iterations; process start,
# context creation and the synthetic targets are outside), with the chain around the rasteriser on the device and on the host
for fe in ("device", "host"):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "demo2_deform.py"), "--iters", "200", "--front-end", fe],
                         capture_output=True, text=True)
    m = re.search(r"(\d+) iterations, (\d+) views on (\d+) rank\(s\): ([0-9.]+) s", out.stdout)
    if not m:
        print("demo2_deform.py --front-end %s FAILED\n%s" % (fe, (out.stdout + out.stderr)[-1500:]))
        continue
    print("demo2_deform.py --front-end %-6s: %s iterations, %s views at 64^2, 1 rank: %.2f ms per iteration"
          % (fe, m.group(1), m.group(2), float(m.group(4)) / int(m.group(1)) * 1e3), flush=True)
