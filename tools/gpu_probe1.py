import sys, time
sys.path.insert(0, '.')
import numpy as np
from oracle import Oracle
from jrender_amd import synthetic as syn, _ffi
from jrender_amd.renderer.dr.softras.soft_rasterize import SoftRasterizeFunction
from tests.util import *
port = Oracle("port", nthreads=0)
ctx = _ffi.Context.default()
print("devices", _ffi.device_count())
cases = [
 ("sphere280/64", syn.sphere_views(280, 2), dict(image_size=64)),
 ("soup500/96", syn.triangle_soup(500, 1, seed=1), dict(image_size=96)),
 ("sphere3300-T4/128", syn.sphere_views(3300, 1, texels=4), dict(image_size=128)),
 ("sphere3300/256 hardrgb", syn.sphere_views(3300, 1), dict(image_size=256, aggr_func_rgb='hard')),
 ("soup2000/200 bary", syn.triangle_soup(2000, 2, seed=2), dict(image_size=200, dist_func='barycentric', sigma_val=1e-4)),
 ("sphere280/vertex", syn.sphere_views(280, 1, texels=3), dict(image_size=100, texture_type='vertex')),
]
for name,(fv,tex),kw in cases:
    a = port.forward(fv, tex, **kw)
    f = SoftRasterizeFunction(**kw)
    t=time.time(); out = f(fv, tex); ctx.synchronize(); dt=time.time()-t
    fvd, texd, rgba, info, aggr, ids = [x.numpy() for x in f.save_vars]
    print(name, "stats", ctx.last_stats(), "t=%.1fms"%(dt*1e3), "ub", port.ub_events())
    print("   info bits", bits_equal(info, a["faces_info"]), " ids bits", bits_equal(ids, a["faces_id_buffer"]), "mismatch px", int((ids!=a["faces_id_buffer"]).any(1).sum()))
    print("   rgba err ratio", rel_err(rgba, a["soft_colors"], RGBA_ATOL), " aggr", rel_err(aggr, a["aggrs_info"], RGBA_ATOL))
    g = np.random.default_rng(0).uniform(-1,1,rgba.shape).astype(np.float32)
    gf_o, gt_o = port.backward(a, g)
    gf, gt = f.grad(g); gf, gt = gf.numpy().reshape(gf_o.shape), gt.numpy()
    print("   grad_faces err %.3g elementwise %.3g | grad_tex err %.3g ew %.3g"%(grad_err(gf,gf_o), grad_err_elementwise(gf,gf_o), grad_err(gt,gt_o), grad_err_elementwise(gt,gt_o)))
