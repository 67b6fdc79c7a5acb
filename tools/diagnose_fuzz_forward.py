#!/usr/bin/env python3
"""GPU box: where does the FORWARD of one saved fuzz case differ from the reference's (RGBA / aggrs_info beyond 1e-4)?"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import Oracle                                                          # noqa: E402
from jrender_amd import _ffi                                                       # noqa: E402
from jrender_amd.renderer.dr.softras import SoftRasterizeFunction                  # noqa: E402
from tests.util import RGBA_ATOL, RGBA_RTOL, bits_equal                            # noqa: E402

z = np.load(sys.argv[1])
kw = eval(str(z["kw"]))
fv, tex = z["fv"], z["tex"]
port = Oracle("port", nthreads=0)
ref = port.forward(fv, tex, **kw)
fn = SoftRasterizeFunction(ctx=_ffi.Context.default(), **kw)
fn(fv, tex)
_, _, rgba, info, aggr, ids = [x.numpy() for x in fn.save_vars]
print("kw", kw, "NF", fv.shape[1])
print("faces_info exact", bits_equal(info, ref["faces_info"]), "ids exact", bits_equal(ids, ref["faces_id_buffer"]))
for name, a, b in (("rgba", rgba, ref["soft_colors"]), ("aggr", aggr, ref["aggrs_info"])):
    ratio = np.abs(a.astype(np.float64) - b) / (RGBA_RTOL * np.abs(b) + RGBA_ATOL)
    print("%s: worst ratio %.3g; elements above 1: %d of %d" % (name, ratio.max(), (ratio > 1).sum(), ratio.size))
    for idx in np.dstack(np.unravel_index(np.argsort(-ratio, axis=None)[:6], ratio.shape))[0]:
        bi, c, y, x = idx
        print("   view %d ch %d row %d col %d: ours %.9g ref %.9g (ratio %.3g) | rgba ours %s ref %s | aggr ours %s ref %s | ids %s"
              % (bi, c, y, x, a[bi, c, y, x], b[bi, c, y, x], ratio[bi, c, y, x], rgba[bi, :, y, x], ref["soft_colors"][bi, :, y, x],
                 aggr[bi, :, y, x], ref["aggrs_info"][bi, :, y, x], ids[bi, :, y, x]))
