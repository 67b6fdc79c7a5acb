# replay a saved case through check_against with the oracle-based predicate
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from oracle import Oracle
from jrender_amd import _ffi
from jrender_amd.renderer.dr.softras import SoftRasterizeFunction
from tests import fuzz_parity
for f in sys.argv[1:]:
    z = np.load(f); kw = eval(str(z["kw"])); fv, tex, g = z["fv"], z["tex"], z["g"]
    port = Oracle("port", nthreads=0)
    ref = port.forward(fv, tex, **kw)
    fn = SoftRasterizeFunction(ctx=_ffi.Context.default(), **kw); fn(fv, tex)
    try:
        print(f, "->", fuzz_parity.check_against(ref, fn, g, port.backward(ref, g), oracle=port))
    except AssertionError as e:
        print(f, "FAIL", e)
