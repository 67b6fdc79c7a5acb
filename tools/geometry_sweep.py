#!/usr/bin/env python3
"""GPU box: bin size x heavy threshold x workgroup size on the critical-path configurations.

For every configuration the reference point is the 32-pixel / default-policy launch; every other (bin size, threshold,
wavefronts) combination must reproduce its index buffer, colours, aggregates and faces_info BIT FOR BIT (the bin geometry
only decides which kernel organisation computes a tile, never what it computes) and its gradients to 1e-4 of the largest
component (float atomics).  Timing: median over `--steps` steps of forward (incl. set-up) + backward, the step as the
operator's caller sees it (events on the context stream, no host synchronisation inside a step), plus the per-phase
brackets of the library (set-up / lists / forward raster / backward raster).

    python tools/geometry_sweep.py [--quick] [--configs c1,c2,...] > gpurun_out/geometry_sweep.txt
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jrender_amd import _ffi, synthetic as syn                                     # noqa: E402
from jrender_amd.renderer.dr.softras.soft_rasterize import SoftRasterizeFunction   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--quick", action="store_true")
ap.add_argument("--steps", type=int, default=25)
ap.add_argument("--configs", default="")
ap.add_argument("--bins", default="32,16,8")
ap.add_argument("--one", default="", help="bin,heavy_min,waves: time this ONE policy per configuration (e.g. under rocprofv3) instead of the sweep")
args = ap.parse_args()
ctx = _ffi.Context(0)


def configs():
    spot = np.load(os.path.join(ROOT, "tests", "golden", "g1_spot.npz"))
    c4 = dict(sigma_val=1e-4, aggr_func_rgb="hard")
    rows = [("c1_spot256", spot["fv"], spot["tex"], 256, {}),
            ("c2_spot1024", spot["fv"], spot["tex"], 1024, {}),
            ("c4_3300f_64px_b64", *syn.sphere_views(3300, 64), 64, c4),
            ("c4_3300f_64px_b8", *syn.sphere_views(3300, 8), 64, c4),
            ("b1_39k_1024", *syn.sphere_views(39000, 1), 1024, {}),
            ("b2_39k_1024", *syn.sphere_views(39000, 2), 1024, {}),
            ("b4_39k_1024", *syn.sphere_views(39000, 4), 1024, {}),
            ("b8_39k_1024", *syn.sphere_views(39000, 8), 1024, {}),
            ("s3300_1024", *syn.sphere_views(3300, 1), 1024, {}),
            ("s280_256", *syn.sphere_views(280, 1), 256, {}),
            ("s39k_256_b8", *syn.sphere_views(39000, 8), 256, {})]
    want = [c for c in args.configs.split(",") if c]
    return [r for r in rows if (r[0] in want if want else r[0] != "b8_39k_1024")]      # (the headline batch only when asked for)


THRESH = {32: (512, 384, 256, 768, 0), 16: (192, 128, 96, 64, 256, 384, 0), 8: (96, 64, 48, 32, 128, 192, 0)}
if args.quick:
    THRESH = {32: (512, 0), 16: (192, 96, 0), 8: (96, 48, 0)}


def run(fv, tex, g, IS, kw, steps, warm=4):
    fn = SoftRasterizeFunction(image_size=IS, ctx=ctx, **kw)
    ev = [ctx.event() for _ in range(steps + 1)]
    for _ in range(warm):
        fn.execute(fv, tex); fn.grad(g)
    ctx.synchronize()
    ctx.profile_enable(True); ctx.profile_collect()
    ctx.record(ev[0])
    for i in range(steps):
        fn.execute(fv, tex); gf, gt = fn.grad(g)
        ctx.record(ev[i + 1])
    ctx.synchronize()
    ph = ctx.profile_collect(); ctx.profile_enable(False)
    t = float(np.median([ctx.elapsed_ms(ev[i], ev[i + 1]) for i in range(steps)]))
    out = dict(ids=fn.save_vars[5].numpy(), rgba=fn.save_vars[2].numpy(), aggrs=fn.save_vars[4].numpy(),
               info=fn.save_vars[3].numpy(), gf=gf.numpy(), gt=gt.numpy())
    return t, {k: v[0] / max(v[1], 1) for k, v in ph.items()}, out, ctx.last_launch(), ctx.last_stats()


bins = [int(b) for b in args.bins.split(",")]
print("# config | bin heavy_min waves | step ms | setup lists fwd bwd (ms, event brackets) | heavy bins, wpw, max list, pairs | parity")
for name, fv, tex, IS, kw in configs():
    fv_d, tex_d = ctx.array(fv), ctx.array(tex)
    g_d = ctx.array(np.random.default_rng(1).uniform(-1, 1, (fv.shape[0], 4, IS, IS)).astype(np.float32))
    ctx.set_bin_size(32); ctx.set_launch_policy(-1, 0)
    t0, ph0, ref, li0, st0 = run(fv_d, tex_d, g_d, IS, kw, args.steps)
    gmax = max(np.abs(ref["gf"]).max(), 1e-30)
    best = (t0, 32, -1, 0)
    ctx.set_bin_size(0); ctx.set_launch_policy(-1, 0)        # what the library chooses on its own for this shape
    ta, pha, outa, lia, sta = run(fv_d, tex_d, g_d, IS, kw, args.steps)
    bada = [k for k in ("ids", "rgba", "aggrs", "info") if not np.array_equal(outa[k].view(np.int32), ref[k].view(np.int32))]
    print("%-18s | auto: bin %d heavy_min %d waves %d | %.4f | %.4f %.4f %.4f %.4f | %5d %d %5d %8d | %s" % (
        name, ctx.bin_size(), lia["heavy_min_faces"], lia["wavefronts_per_workgroup"], ta, pha["bin_count"], pha["bin_fill_sort"],
        pha["fwd_raster"], pha["bwd_raster"], lia["heavy_bins"], lia["wavefronts_per_workgroup"], sta["max_faces_in_bin"],
        sta["bin_face_pairs"], "bit-exact" if not bada else "MISMATCH %s" % bada), flush=True)
    one = [int(x) for x in args.one.split(",")] if args.one else None
    for b in ([one[0]] if one else bins):
        for hm in ([one[1]] if one else THRESH[b]):
            for w in ([one[2]] if one else ((0,) if hm == 0 else (0, 4, 8))):
                ctx.set_bin_size(b); ctx.set_launch_policy(hm, w)
                t, ph, out, li, st = run(fv_d, tex_d, g_d, IS, kw, args.steps)
                bad = [k for k in ("ids", "rgba", "aggrs", "info") if not np.array_equal(out[k].view(np.int32), ref[k].view(np.int32))]
                gerr = float(np.abs(out["gf"] - ref["gf"]).max() / gmax)
                ok = "bit-exact, grad %.1e" % gerr if not bad and gerr <= 1e-4 else "MISMATCH %s grad %.1e" % (bad, gerr)
                print("%-18s | %2d %4d %d | %.4f | %.4f %.4f %.4f %.4f | %5d %d %5d %8d | %s" % (
                    name, b, hm, w, t, ph["bin_count"], ph["bin_fill_sort"], ph["fwd_raster"], ph["bwd_raster"],
                    li["heavy_bins"], li["wavefronts_per_workgroup"], st["max_faces_in_bin"], st["bin_face_pairs"], ok), flush=True)
                if not bad and t < best[0]:
                    best = (t, b, hm, w)
    print("## %-18s reference (bin 32, round-4 policy) %.4f ms | automatic policy %.4f ms | best of the sweep %.4f ms at bin %d heavy_min %d waves %d" % (
        (name, t0, ta) + best), flush=True)
ctx.set_bin_size(0); ctx.set_launch_policy(-1, 0)
