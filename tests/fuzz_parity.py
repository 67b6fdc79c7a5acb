#!/usr/bin/env python3
"""Randomised parity sweep (not collected by pytest; run on a GPU box):

    python tests/fuzz_parity.py --cases 400 --seed 1

Every case draws a scene (soup / sphere views / degenerate-heavy soup), the operator modes, K, image size,
sigma, gamma, near/far, texture layout and batch size at random - and, since round 5, how the launch is organised (bin size 8 / 16 / 32 /
automatic, heavy-bin threshold, 4 / 8 wavefronts per workgroup, precise colour path) -, renders with the HIP path, and applies the same
checks as tests/test_gpu_parity.py against the oracle: faces_info and the face-index buffer bit-exact, RGBA and
aggrs_info within 1e-4, gradients within 1e-4 of the largest component.  Cases that hit the reference's
undefined-behaviour corner (oracle counter) are skipped and counted."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import Oracle                                                          # noqa: E402
from jrender_amd import _ffi, synthetic as syn                                     # noqa: E402
from jrender_amd.renderer.dr.softras import SoftRasterizeFunction                  # noqa: E402
from tests.util import RGBA_ATOL, bits_equal, grad_err, rel_err                   # noqa: E402


def check_against(ref, fn, g, ref_grads, oracle=None):
    """The bars of north_star: index buffer (and faces_info) bit-exact, RGBA / aggrs_info 1e-4, gradients 1e-4 of
    the largest component with the same non-finite pattern.  (tests/test_gpu_parity.py adds an element-wise sanity
    bound for its curated scenes; random operator settings such as gamma=1e-4 with a 1-unit depth range amplify
    cancellation noise 1e4-fold in single entries, in the reference's own float arithmetic as much as here.)"""
    fv, tex, rgba, info, aggr, ids = [x.numpy() for x in fn.save_vars]
    assert bits_equal(info, ref["faces_info"]), "faces_info not bit-exact"
    assert bits_equal(ids, ref["faces_id_buffer"]), "face-index buffer differs"
    assert rel_err(rgba, ref["soft_colors"], RGBA_ATOL) <= 1.0, "rgba"
    assert rel_err(aggr, ref["aggrs_info"], RGBA_ATOL) <= 1.0, "aggrs_info"
    gf, gt = fn.grad(g)
    status = "ok"
    for a, b, name in ((gf.numpy().reshape(ref_grads[0].shape), ref_grads[0], "grad_faces"), (gt.numpy(), ref_grads[1], "grad_textures")):
        if np.isfinite(b).all() and np.abs(b).max() < 1e30:
            e = grad_err(a, b)
            # 1e-4 is the bar.  Above it the case is "ill-conditioned" if — and only if — the error is explained by
            # the forward's in-tolerance differences: a pixel covered by ONE face has colour o == texel k up to
            # rounding, its gradient carries (k - o) / D / gamma, i.e. the forward's last-bit noise amplified up to
            # 1e4-fold (traced on such cases: entries come out with the reference's magnitude and the OPPOSITE sign),
            # which differs between any two float implementations of the forward, the reference's CPU and CUDA builds
            # included.  The predicate: the reference's backward run on OUR saved tensors (our soft_colors / aggrs_info;
            # ids and faces_info are bit-identical anyway) must agree with our backward to the 1e-4 bar.  Without an
            # oracle handle the old empirical cap of 1e-2 applies.
            if e > 1e-4 and oracle is not None:
                mine = dict(ref)
                mine["soft_colors"], mine["aggrs_info"] = rgba, aggr
                again = oracle.backward(mine, g)[0 if name == "grad_faces" else 1]
                e2 = grad_err(a, again)
                assert e2 <= 1e-4, (name, e, "not explained by the forward's rounding: %.3g against the reference backward on our saved tensors" % e2)
                assert e <= 1e-1, (name, e, "explained by the forward's rounding, but beyond the absolute ceiling of 1e-1")
            else:
                assert e <= 1e-2, (name, e)
            if e > 1e-4:
                status = "illcond"
        else:
            # The reference's own gradient overflowed or is about to (softmax weights of faces the forward skipped,
            # SRK:1308): where both are finite the remainder is a sum of ~1e37 terms that cancel -> held to 5e-2;
            # the non-finite pattern must agree except for entries within a factor 4 of FLT_MAX.
            fa, fb = np.isfinite(a), np.isfinite(b)
            near_max = (np.abs(np.where(fa, a, 0)) > 8e37) | (np.abs(np.where(fb, b, 0)) > 8e37)
            assert np.array_equal(fa | near_max, fb | near_max), (name, "non-finite pattern")
            both = fa & fb & ~near_max
            if both.any():
                e = float(np.abs(a[both].astype(np.float64) - b[both]).max() / max(np.abs(b[both]).max(), 1e-30))
                # 5e-2: seed 21 case 144 (barycentric / sum / vertex colours, textures gradient at 1.2e37) has ONE entry
                # of 4e34 off by 1.4 % of the largest finite entry, identically on every build since round 1
                # (tools/ablate/cases/fuzz_fail_21_144.npz): cancellation among ~1e37 terms, 3.5e-4 of their size
                assert e <= 5e-2, (name, e, "reference gradient non-finite")
            if status == "ok":
                status = "overflow"
    return status


def draw_case(rng, big=False):
    kind = rng.choice(["soup", "sphere", "big_faces", "far_soup"])
    B = int(rng.choice([1, 1, 2, 3]))
    texture_type = str(rng.choice(["surface", "surface", "vertex"]))
    texels = 3 if texture_type == "vertex" else int(rng.choice([1, 1, 4, 9]))
    if kind == "sphere":
        nf = int(rng.choice([280, 280, 3300]))
        fv, tex = syn.sphere_views(nf, B, texels=texels, seed=int(rng.integers(1 << 30)), azimuth0=float(rng.uniform(0, 360)),
                                   elevation=float(rng.uniform(-60, 60)))
    else:
        nf = int(rng.integers(1, 12000 if big else 2500))
        scale = {"soup": float(rng.uniform(0.8, 4.0)), "big_faces": float(rng.uniform(8, 30)), "far_soup": float(rng.uniform(1, 3))}[kind]
        fv, tex = syn.triangle_soup(nf, B, seed=int(rng.integers(1 << 30)), texels=texels, scale=scale)
        if kind == "far_soup":
            fv[..., :2] *= float(rng.uniform(1.2, 2.5))            # many faces partly / fully off-screen
    kw = dict(image_size=int(rng.integers(9, 330 if big else 140)),
              dist_func=str(rng.choice(["euclidean", "euclidean", "barycentric", "hard"])),
              aggr_func_rgb=str(rng.choice(["softmax", "softmax", "hard"])),
              aggr_func_alpha=str(rng.choice(["prod", "sum", "hard"])),
              texture_type=texture_type,
              max_faces_per_pixel_for_grad=int(rng.choice([1, 2, 5, 16, 16, 16, 17, 33, 64])),
              sigma_val=float(rng.choice([1e-4, 1e-5, 1e-5, 1e-6])),
              gamma_val=float(rng.choice([1e-4, 1e-4, 1e-3, 1e-2])),
              fill_back=bool(rng.integers(2)),
              near=float(rng.choice([1.0, 1.0, 2.5])), far=float(rng.choice([100.0, 100.0, 3.5])))
    return kind, fv, tex, kw


def draw_launch(rng):
    """How the library organises the launch - never what it computes: the operator's bin_size, the heavy-bin threshold and the
    workgroup size of the multi-wavefront kernel (jr_softras_set_launch_policy), the colour-path arithmetic (round 5)."""
    return dict(bin_size=int(rng.choice([0, 0, 8, 16, 32])), heavy_min=int(rng.choice([-1, -1, 4, 24, 100])),
                waves=int(rng.choice([0, 0, 4, 8])), precise=bool(rng.integers(8) == 0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-failures", type=int, default=1)
    ap.add_argument("--big", action="store_true", help="up to 12 000 faces and 330^2 pixels (seconds per case in the oracle)")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    ctx = _ffi.Context.default()
    port = Oracle("port", nthreads=0)
    done = skipped = failed = overflowed = illcond = 0
    t0 = time.time()
    for i in range(args.cases):
        kind, fv, tex, kw = draw_case(rng, args.big)
        ref = port.forward(fv, tex, **kw)
        if port.ub_events():
            skipped += 1
            continue
        launch = draw_launch(rng)
        ctx.set_launch_policy(launch["heavy_min"], launch["waves"])
        fn = SoftRasterizeFunction(ctx=ctx, bin_size=launch["bin_size"], precise_colour=launch["precise"], **kw)
        fn(fv, tex)
        g = rng.uniform(-1, 1, ref["soft_colors"].shape).astype(np.float32)
        try:
            st = check_against(ref, fn, g, port.backward(ref, g), oracle=port)
            overflowed += st == "overflow"
            illcond += st == "illcond"
        except AssertionError as e:
            print("FAIL case %d (%s, NF=%d, B=%d, %r, launch %r): %s" % (i, kind, fv.shape[1], fv.shape[0], kw, launch, e), flush=True)
            np.savez("gpurun_out/fuzz_fail_%d_%d.npz" % (args.seed, i), fv=fv, tex=tex, kw=repr(kw), g=g)
            failed += 1
            if failed >= args.max_failures:
                raise SystemExit(1)
            continue
        done += 1
    ctx.set_launch_policy(-1, 0)
    if illcond > max(2, 0.02 * max(done, 1)):
        print("WARNING: %d of %d cases are ill-conditioned (> 2 %%): a forward regression inside the 1e-4 RGBA bar would look like this" % (illcond, done))
        failed = max(failed, 1)
    print("fuzz: %d cases passed (%d with an overflowing reference gradient, %d ill-conditioned: gradient error > 1e-4 explained by the forward's rounding), "
          "%d failed, %d skipped (reference UB corner), seed %d, %.1f s" % (done, overflowed, illcond, failed, skipped, args.seed, time.time() - t0))
    if failed:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
