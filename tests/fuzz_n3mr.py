#!/usr/bin/env python3
"""Randomised parity sweep of the NMR path against oracle/n3mr_oracle.c (the plain-C restatement, pinned
bit-identical to the reference's own kernels compiled for the host by tests/test_oracle.py).  Not collected by pytest; run on a GPU box:  python tests/fuzz_n3mr.py --cases 200
Checks: face_index / weight / depth / face_inv / sampling maps bit-exact, rgb 1e-6, gradients 1e-4 of max."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import N3mrOracle                                                      # noqa: E402
import jrender_amd as jr                                                           # noqa: E402
from jrender_amd.renderer.dr.n3mr import RasterizeFunction                         # noqa: E402
from tests.util import bits_equal, grad_err                                        # noqa: E402


def draw(rng):
    B = int(rng.choice([1, 2, 3]))
    ts = int(rng.choice([2, 2, 3, 4, 5]))                             # ts = 1 reads out of bounds in the reference (rejected here)
    IS = int(rng.integers(5, 200))
    kind = rng.choice(["sphere", "soup", "big"])
    if kind == "sphere":
        nf = int(rng.choice([280, 3300]))
        v, f = jr.synthetic.sphere_mesh(nf)
        eyes = np.stack([np.asarray(jr.get_points_from_angles(2.732, float(rng.uniform(-50, 50)), float(rng.uniform(0, 360))), np.float32)
                         for _ in range(B)])
        ndc = jr.perspective(jr.look_at(np.broadcast_to(v[None], (B,) + v.shape), eyes), 30.)
        ff = np.concatenate([f, f[:, ::-1]])
        faces = np.ascontiguousarray(ndc[:, ff])
    else:
        nf = int(rng.integers(1, 1500))
        scale = float(rng.uniform(1, 4)) if kind == "soup" else float(rng.uniform(8, 40))
        fv, _ = jr.synthetic.triangle_soup(nf, B, seed=int(rng.integers(1 << 30)), scale=scale)
        fv[..., :2] *= float(rng.uniform(0.8, 1.6))
        faces = np.concatenate([fv, fv[:, :, ::-1]], 1)
    tex = rng.uniform(0, 1, (B, faces.shape[1], ts, ts, ts, 3)).astype(np.float32)
    flags = [(True, True, True), (True, True, True), (False, True, False), (True, False, False), (False, True, True)][int(rng.integers(5))]
    kw = dict(image_size=IS, near=float(rng.choice([0.1, 0.1, 2.2])), far=float(rng.choice([100.0, 3.6])),
              eps=float(rng.choice([1e-3, 1e-2])), background_color=tuple(float(x) for x in rng.uniform(0, 1, 3)))
    return kind, faces.astype(np.float32), tex, kw, flags


def run(cases=200, seed=0):
    """-> number of failed cases (stops at the first); tests/test_gpu_fuzz_slice.py runs a fixed-seed slice."""
    import types
    args = types.SimpleNamespace(cases=cases, seed=seed)
    o = N3mrOracle("port")        # the C restatement (bit-identical to the reference build, any image size)
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    for i in range(args.cases):
        kind, faces, tex, kw, (rrgb, ra, rd) = draw(rng)
        ref = o.forward(faces, tex if rrgb else None, return_rgb=rrgb, return_alpha=ra, return_depth=rd, **kw)
        fn = RasterizeFunction(kw["image_size"], kw["near"], kw["far"], kw["eps"], kw["background_color"], rrgb, ra, rd)
        fn(faces, tex if rrgb else None)
        f, t, fim, wm, dm, rgb, alpha, fivm, sidx, swt = fn.save_vars
        try:
            assert bits_equal(fim.numpy(), ref["face_index_map"]), "face_index_map"
            assert bits_equal(dm.numpy(), ref["depth_map"]), "depth_map"
            assert bits_equal(wm.numpy(), ref["weight_map"]), "weight_map"
            if rd:
                assert bits_equal(fivm.numpy().reshape(ref["face_inv_map"].shape), ref["face_inv_map"]), "face_inv_map"
            if rrgb:
                assert bits_equal(sidx.numpy(), ref["sampling_index_map"]) and bits_equal(swt.numpy(), ref["sampling_weight_map"]), "sampling"
                assert np.allclose(rgb.numpy(), ref["rgb_map"], rtol=1e-6, atol=1e-7), "rgb"
            if ra:
                assert bits_equal(alpha.numpy(), ref["alpha_map"]), "alpha"
            shape = ref["face_index_map"].shape
            g_rgb = rng.uniform(-1, 1, shape + (3,)).astype(np.float32) if rrgb else None
            g_a = rng.uniform(-1, 1, shape).astype(np.float32) if ra else None
            g_d = rng.uniform(-1, 1, shape).astype(np.float32) if rd else None
            gfo, gto = o.backward(ref, g_rgb, g_a, g_d)
            gf, gt = fn.grad(g_rgb, g_a, g_d)
            e = grad_err(gf.numpy().reshape(gfo.shape), gfo)
            assert e <= 1e-4, ("grad_faces", e)
            if rrgb:
                e = grad_err(gt.numpy(), gto)
                assert e <= 1e-4, ("grad_textures", e)
        except AssertionError as ex:
            print("FAIL case %d (%s NF=%d B=%d ts=%d flags=%r %r): %s" % (i, kind, faces.shape[1], faces.shape[0], tex.shape[2], (rrgb, ra, rd), kw, ex), flush=True)
            os.makedirs("gpurun_out", exist_ok=True)
            np.savez("gpurun_out/fuzz_n3mr_fail_%d_%d.npz" % (args.seed, i), faces=faces, tex=tex, kw=repr(kw), flags=repr((rrgb, ra, rd)))
            return 1
    print("fuzz_n3mr: %d cases passed, seed %d, %.1f s" % (args.cases, args.seed, time.time() - t0))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    raise SystemExit(run(args.cases, args.seed))


if __name__ == "__main__":
    main()
