import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
from jrender_amd import _ffi, synthetic as syn
from jrender_amd.renderer.dr.softras.soft_rasterize import SoftRasterizeFunction
ctx = _ffi.Context(0)
spot = np.load("/root/repo/tests/golden/g1_spot.npz")
for name, fv, tex, IS in (("C1 spot 256", spot["fv"], spot["tex"], 256), ("C2 spot 1024", spot["fv"], spot["tex"], 1024), ("39k B=1 1024", *syn.sphere_views(39000, 1), 1024), ("280 256", *syn.sphere_views(280, 1), 256)):
    fvd, texd = ctx.array(fv), ctx.array(tex)
    g = ctx.array(np.random.default_rng(1).uniform(-1, 1, (1, 4, IS, IS)).astype(np.float32))
    fn = SoftRasterizeFunction(image_size=IS, ctx=ctx)
    for _ in range(5): fn.execute(fvd, texd); fn.grad(g)
    ctx.profile_enable(True); ctx.profile_collect()
    import time
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn.execute(fvd, texd); fn.grad(g)
    ctx.synchronize(); wall = (time.perf_counter() - t0) / 20 * 1e3
    ph = ctx.profile_collect(); ctx.profile_enable(False)
    print(name, "wall %.3f ms/step |" % wall, {k: round(v[0] / 20, 4) for k, v in ph.items()}, ctx.last_stats())
