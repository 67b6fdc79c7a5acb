#!/usr/bin/env python3
"""Texture optimisation through the NMR renderer — the workload of the reference's demo4-optim_textures.py on the HIP NMR path
(`dr_type='n3mr'`), with the backward chain written out (no autograd framework):

    textures [1,NF,4,4,4,3] --tanh (demo4-optim_textures.py:43)--> cube textures --ambient lighting (intensity 1, no directional light,
      :38)--> NMR rasteriser (HIP: z-buffer, texture sampling; look_at camera, orthographic, :38) --> image
      loss = sum((image - image_ref)^2) (:44);   Adam(lr 0.03, betas (0.5, 0.999)) (:68), a random azimuth per iteration (:41)

    python examples/demo4_optim_textures.py [--iters 200] [--image-size 256] [--obj mesh.obj --ref image.npy]

Without --obj / --ref (the reference's data files are not shipped here) the mesh is a UV sphere and the targets are renders of the
SAME sphere under a procedural "true" texture from 12 fixed azimuths; every iteration draws one of them (the reference compares
every random view with ONE photograph, which only constrains what that photograph shows).

The rasteriser forward / backward run on the GPU (jr_n3mr_forward / jr_n3mr_backward through Renderer.render_mesh /
Renderer.grad_textures); tanh, its derivative and the 1.1 M-element Adam step are NumPy or - with --device-adam, the default -
one jr_adam_step launch on the device-resident parameter (jrender_amd/optim.py).
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import jrender_amd as jr                                                     # noqa: E402


def true_textures(v, f, ts):
    """A smooth procedural colour per face (function of the face centre's direction), constant over the face's texture cube."""
    c = v[f].mean(1)
    c = c / np.maximum(np.linalg.norm(c, axis=1, keepdims=True), 1e-9)
    rgb = 0.5 + 0.45 * np.stack([np.sin(3.0 * c[:, 0] + 1.0), np.cos(4.0 * c[:, 1]), np.sin(5.0 * c[:, 2] - 0.5)], 1)
    return np.broadcast_to(rgb[None, :, None, None, None, :], (1, f.shape[0], ts, ts, ts, 3)).astype(np.float32).copy()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--obj', default=None, help="template mesh (.obj); default: a 1 352-vertex UV sphere")
    ap.add_argument('--ref', default=None, help=".npy reference image [3,S,S] in [0,1] compared with EVERY view (the reference's protocol)")
    ap.add_argument('--iters', type=int, default=200)
    ap.add_argument('--image-size', type=int, default=256)
    ap.add_argument('--texture-size', type=int, default=4)
    ap.add_argument('--views', type=int, default=12, help="synthetic targets: azimuths 0, 360/n, ...")
    ap.add_argument('--host-adam', action='store_true', help="NumPy Adam on a host parameter instead of jr_adam_step on the device")
    ap.add_argument('--history-out', default=None)
    ap.add_argument('--quiet', action='store_true')
    ap.add_argument('--seed', type=int, default=1)
    args = ap.parse_args(argv)
    rng = np.random.default_rng(args.seed)                                   # demo4-optim_textures.py:17 seeds NumPy's global generator with 1

    if args.obj:
        v, f = jr.load_obj(args.obj)
    else:
        v, f = jr.synthetic.uv_sphere(52, 27)
    v = (np.asarray(v, np.float32) * 0.6).astype(np.float32)                  # :26
    f = np.asarray(f, np.int32)
    ts = args.texture_size
    ctx = jr.Context.default()
    renderer = jr.Renderer(image_size=args.image_size, camera_mode='look_at', perspective=False, light_intensity_directionals=0.0,
                           light_intensity_ambient=1.0, dr_type='n3mr')       # :38

    def render(tex, azimuth):
        renderer.transform.set_eyes_from_angles(2.732, 0, azimuth)            # :42
        return renderer.render_mesh(jr.Mesh(v, f, textures=tex, dr_type='n3mr'), mode='rgb')

    if args.ref:
        ref = np.load(args.ref).astype(np.float32)[None]
        azimuths, targets = None, None
    else:
        azimuths = np.arange(args.views, dtype=np.float32) * 360.0 / args.views
        t_true = true_textures(v / 0.6, f, ts)
        targets = [render(t_true.copy(), a).numpy() for a in azimuths]

    device_adam = not args.host_adam
    params = np.ones((1, f.shape[0], ts, ts, ts, 3), np.float32)             # :30
    p_dev = ctx.array(params) if device_adam else None
    opt = jr.Adam([p_dev if device_adam else params], lr=0.03, betas=(0.5, 0.999))
    hist = []
    t0 = time.perf_counter()
    for it in range(args.iters):
        if device_adam:
            params = p_dev.numpy()
        th = np.tanh(params)
        if targets is None:
            az, target = float(rng.uniform(0, 360)), ref
        else:
            k = int(rng.integers(0, len(azimuths)))
            az, target = float(azimuths[k]), targets[k]
        img = render(th.copy(), az)                                           # (the lighting step rewrites mesh.textures: hand it a copy)
        diff = img.numpy() - target
        loss = float((diff.astype(np.float64) ** 2).sum())
        g_tex = renderer.grad_textures(2.0 * diff)                            # d loss / d tanh(textures): rasteriser backward + lighting VJP
        g_par = (g_tex * (1.0 - th * th)).astype(np.float32)                  # tanh'
        opt.step([ctx.array(g_par) if device_adam else g_par])
        hist.append(loss)
        if not args.quiet and (it % 20 == 0 or it == args.iters - 1):
            print("iter %4d  loss %.3f" % (it, loss), flush=True)
    ctx.synchronize()
    main.loop_seconds = time.perf_counter() - t0
    if not args.quiet:
        print("%d iterations, %.2f ms per iteration; loss %.3f -> %.3f" % (args.iters, main.loop_seconds / max(args.iters, 1) * 1e3, hist[0], hist[-1]))
    if args.history_out:
        np.save(args.history_out, np.asarray(hist, np.float32))
    main.textures = np.tanh(p_dev.numpy() if device_adam else params)
    return hist


if __name__ == '__main__':
    main()
