"""K-buffer sharing statistics (GPU box): how many pixels of a tile hold the same face."""
import sys; sys.path.insert(0, '.')
import numpy as np
from jrender_amd import _ffi, synthetic as syn
from jrender_amd.renderer.dr.softras import SoftRasterizeFunction
ctx = _ffi.Context(0)
fv, tex = syn.sphere_views(39000, 1)
fn = SoftRasterizeFunction(image_size=1024, ctx=ctx); fn(fv, tex)
ids = fn.save_vars[5].numpy()[0]          # [16,1024,1024]
print("pairs", int((ids >= 0).sum()), "touched px", int((ids[0] >= 0).sum()))
for T in (8, 4):
    n = 1024 // T
    a = ids.reshape(16, n, T, n, T).transpose(1, 3, 0, 2, 4).reshape(n * n, -1)   # per tile: 16*T*T ids
    a = np.sort(a, axis=1)
    valid = a >= 0
    newf = valid & np.concatenate([np.ones((a.shape[0], 1), bool), a[:, 1:] != a[:, :-1]], 1)
    nfaces = newf.sum(1); npairs = valid.sum(1)
    ne = nfaces > 0
    print("tile %dx%d: non-empty tiles %d, (tile,face) pairs %d, pairs/(tile,face) %.2f, faces/tile mean %.1f max %d, steps/tile(=max K per px) n/a"
          % (T, T, ne.sum(), nfaces.sum(), npairs.sum() / nfaces.sum(), nfaces[ne].mean(), nfaces.max()))
