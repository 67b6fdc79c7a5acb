#!/usr/bin/env python3
"""Gradient parity of builds of libjrender_hip.so against the reference, as NUMBERS.

    python tools/grad_parity.py [--size 1024] [--faces 39000] [name=path/to/libjrender_hip_X.so ...]

GPU box.  The parent renders ONE view with the reference's own kernels compiled for the host (oracle/_ref) and computes

  ref_f32        the reference's float backward (all cores: float atomics in whatever order the threads run)
  ref_serial     the same on one thread
  exact_sum      the same float per-pair terms summed in DOUBLE (ref_softras_backward_exactsum): the reference's gradient
                 without the order noise of its float atomics
  f64            the reference's backward kernel instantiated for double (ref_softras_backward_f64)

then every library (the product first) runs in a child process (JRENDER_LIB) and reports, element-wise with the three
normalisations of bench.py's err_metrics, its gradients against exact_sum / ref_f32 / f64 - once from ITS OWN saved
tensors (forward + backward differences) and once from the ORACLE's saved tensors (backward arithmetic alone).
"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SAMPLE = os.path.join(os.environ.get("TMPDIR", "/tmp"), "jr_grad_sample.npz")


def metrics(a, b):
    from bench import err_metrics
    m = err_metrics(a, b)
    return {"max_norm": m["max_norm"], "floor_1e-3": m["rel_floor_1e-3"], "floor_1e-6": m["rel_floor_1e-6"]}


def fmt(m):
    return "max-norm %.2e | 1e-3 floor %.2e | 1e-6 floor %.2e" % (m["max_norm"], m["floor_1e-3"], m["floor_1e-6"])


def make_sample(NF, IS, K):
    from oracle import Oracle
    from jrender_amd import synthetic as syn
    orc = Oracle("reference", nthreads=0)
    fv, tex = syn.sphere_views(NF, 1)
    g = np.random.default_rng(11).uniform(-1, 1, (1, 4, IS, IS)).astype(np.float32)
    t0 = time.time()
    a = orc.forward(fv, tex, image_size=IS, max_faces_per_pixel_for_grad=K)
    gfN, gtN = orc.backward(a, g, nthreads=orc.num_procs())
    gf1, gt1 = orc.backward(a, g, nthreads=1)
    gfS, gtS = orc.backward_exactsum(a, g)
    gf64, gt64 = orc.backward_f64(a, g)
    print("oracle view %dx%d, %d faces: %.1f s on %d threads" % (IS, IS, NF, time.time() - t0, orc.num_procs()), flush=True)
    np.savez(SAMPLE, fv=fv, tex=tex, g=g, IS=IS, K=K, soft_colors=a["soft_colors"], faces_info=a["faces_info"],
             aggrs_info=a["aggrs_info"], ids=a["faces_id_buffer"], gf_ref=gfN, gt_ref=gtN, gf_serial=gf1, gt_serial=gt1,
             gf_sum=gfS, gt_sum=gtS, gf_f64=gf64, gt_f64=gt64)
    print("reference against itself (order noise of its float atomics):")
    print("  grad_faces     all cores vs exact sum : " + fmt(metrics(gfN, gfS)))
    print("  grad_faces     serial    vs exact sum : " + fmt(metrics(gf1, gfS)))
    print("  grad_faces     all cores vs serial    : " + fmt(metrics(gfN, gf1)))
    print("  grad_textures  all cores vs exact sum : " + fmt(metrics(gtN, gtS)))
    print("reference float arithmetic against its double instantiation (conditioning, not noise):")
    print("  grad_faces     exact sum vs f64       : " + fmt(metrics(gfS, gf64)))
    m = float(np.abs(gf64).max())
    well = np.abs(gfS - gf64).reshape(-1, 9).max(1) <= 1e-4 * m
    print("  faces whose float gradient is within 1e-4 max|g| of the double one: %.1f %%" % (100 * well.mean()), flush=True)


def child():
    from jrender_amd import _ffi
    from jrender_amd.renderer.dr.softras.soft_rasterize import SoftRasterizeFunction
    z = np.load(SAMPLE)
    ctx = _ffi.Context.default()
    IS, K = int(z["IS"]), int(z["K"])
    fn = SoftRasterizeFunction(image_size=IS, max_faces_per_pixel_for_grad=K, ctx=ctx)
    img = fn.execute(ctx.array(z["fv"]), ctx.array(z["tex"]))
    g = ctx.array(z["g"])
    gf, gt = [x.numpy() for x in fn.grad(g)]
    ids_ok = bool((fn.save_vars[5].numpy() == z["ids"]).all())
    from bench import err_metrics
    rgba = err_metrics(img.numpy(), z["soft_colors"])
    # the backward alone: the oracle's saved tensors instead of this build's own forward outputs
    own = fn.save_vars
    fn.save_vars = (own[0], own[1], ctx.array(z["soft_colors"]), ctx.array(z["faces_info"]), ctx.array(z["aggrs_info"]),
                    ctx.array(z["ids"]))
    gf2, gt2 = [x.numpy() for x in fn.grad(g)]
    # timing of the two kernels on this view (one view: the multi-wavefront kernels)
    fn.save_vars = own
    ev = [ctx.event() for _ in range(3)]
    tf, tb = [], []
    for _ in range(12):
        ctx.record(ev[0]); fn.execute(ctx.array(z["fv"]), ctx.array(z["tex"])); ctx.record(ev[1]); fn.grad(g); ctx.record(ev[2])
        tf.append(ctx.elapsed_ms(ev[0], ev[1])); tb.append(ctx.elapsed_ms(ev[1], ev[2]))
    out = {"ids_bit_exact": ids_ok, "rgba_floor_1e-6": rgba["rel_floor_1e-6"], "fwd_ms_b1": float(np.median(tf)), "bwd_ms_b1": float(np.median(tb))}
    for tag, a, t in (("own_forward", gf, gt), ("oracle_forward", gf2, gt2)):
        out[tag] = {"faces_vs_exact_sum": metrics(a, z["gf_sum"]), "faces_vs_ref_f32": metrics(a, z["gf_ref"]),
                    "faces_vs_f64": metrics(a, z["gf_f64"]), "textures_vs_exact_sum": metrics(t, z["gt_sum"])}
    print("RESULT " + json.dumps(out), flush=True)


def main():
    args = sys.argv[1:]
    if args and args[0] == "--child":
        return child()
    IS = int(args[args.index("--size") + 1]) if "--size" in args else 1024
    NF = int(args[args.index("--faces") + 1]) if "--faces" in args else 39000
    libs = [("product", os.path.join(ROOT, "jrender_amd", "csrc", "libjrender_hip.so"))]
    libs += [tuple(a.split("=", 1)) for a in args if "=" in a and not a.startswith("--")]
    make_sample(NF, IS, 16)
    for name, path in libs:
        if not os.path.exists(path):
            print("%-12s missing (%s)" % (name, path))
            continue
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, JRENDER_LIB=path),
                           capture_output=True, text=True, cwd=ROOT)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print("%-12s FAILED\n%s" % (name, (p.stdout + p.stderr)[-2000:]))
            continue
        r = json.loads(line[0][7:])
        print("== %s: ids bit-exact %s, rgba %.2e (1e-6 floor), one view fwd %.3f ms bwd %.3f ms" % (name, r["ids_bit_exact"], r["rgba_floor_1e-6"], r["fwd_ms_b1"], r["bwd_ms_b1"]))
        for tag in ("own_forward", "oracle_forward"):
            for k, m in r[tag].items():
                print("   %-15s %-22s %s" % (tag, k, fmt(m)))
        print(json.dumps({"variant": name, **r}), flush=True)


if __name__ == "__main__":
    main()
