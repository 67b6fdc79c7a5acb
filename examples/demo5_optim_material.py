#!/usr/bin/env python3
"""Material optimisation through the Cook-Torrance lighting step — the workloads of the reference's demo5-optim_metallic_textures.py
and demo6-optim_roughness_textures.py on the HIP SoftRas path, with the backward chain written out (no autograd framework):

    metallic / roughness [1,NF,T,1] --mean over the texels--> per-face Cook-Torrance shading of the textures (GGX, Smith, Schlick:
      renderer/lighting/directional_lighting.py:86-130; lit = clip(textures * diffuse + specular, 0, 1), lighting.py:203-204)
      --look_at + perspective--> SoftRas (HIP) --> image;   loss = sum((image - image_ref)^2)
    backward: jr_softras_backward -> grad of the lit textures -> Lighting.backward_material (the VJP of the lighting step)

    python examples/demo5_optim_material.py --param metallic     # demo5: metallic from 0, roughness fixed at 0.5, directional light 1.0, ambient 0, Adam(0.1), 20 steps
    python examples/demo5_optim_material.py --param roughness    # demo6: roughness from 1, metallic fixed at 0.4, Renderer defaults, Adam(0.1), 15 steps

Without --obj / --ref (the reference's data files are not shipped here) the mesh is a UV sphere with a grey texture and the target is
a render of the same sphere with the "true" material (a smooth per-face map), from the demos' camera (2.732, 30, 140).
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import jrender_amd as jr                                                     # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--param', choices=['metallic', 'roughness'], default='metallic')
    ap.add_argument('--obj', default=None, help="template mesh (.obj, with its texture); default: a grey UV sphere")
    ap.add_argument('--ref', default=None, help=".npy reference image [3,S,S] in [0,1]")
    ap.add_argument('--iters', type=int, default=None, help="default: 20 (metallic) / 15 (roughness), as the reference's demos")
    ap.add_argument('--image-size', type=int, default=256)
    ap.add_argument('--texture-res', type=int, default=4)
    ap.add_argument('--history-out', default=None)
    ap.add_argument('--quiet', action='store_true')
    args = ap.parse_args(argv)
    metallic_run = args.param == 'metallic'
    iters = args.iters if args.iters is not None else (20 if metallic_run else 15)

    T = args.texture_res * args.texture_res
    if args.obj:
        mesh0 = jr.Mesh.from_obj(args.obj, texture_res=args.texture_res, load_texture=True, dr_type='softras')
        v, f, tex = mesh0.vertices, mesh0.faces, np.array(mesh0.textures, np.float32)
    else:
        v, f = jr.synthetic.uv_sphere(40, 21)
        v, f = np.asarray(v, np.float32)[None], np.asarray(f, np.int32)[None]
        tex = np.full((1, f.shape[1], T, 3), 0.55, np.float32)
    nf = f.shape[1]
    if metallic_run:                                                          # demo5-optim_metallic_textures.py:28-33
        renderer = jr.Renderer(image_size=args.image_size, dr_type='softras', light_intensity_directionals=1.0, light_intensity_ambient=0.0)
        metallic = np.zeros((1, nf, T, 1), np.float32)
        roughness = np.full((1, nf, T, 1), 0.5, np.float32)
    else:                                                                     # demo6-optim_roughness_textures.py:28-33
        renderer = jr.Renderer(image_size=args.image_size, dr_type='softras')
        metallic = np.full((1, nf, T, 1), 0.4, np.float32)
        roughness = np.ones((1, nf, T, 1), np.float32)

    def render(m, r):
        renderer.transform.set_eyes_from_angles(2.732, 30, 140)
        return renderer(v, f, tex.copy(), metallic_textures=m, roughness_textures=r)   # (the lighting step rewrites the textures it is given)

    if args.ref:
        ref = np.load(args.ref).astype(np.float32)[None]
    else:                                      # the "true" material: a smooth map over the sphere
        c = np.asarray(v[0])[np.asarray(f[0])].mean(1)
        wave = (0.5 + 0.5 * np.sin(4.0 * c[:, 0] + 2.0 * c[:, 1]))[None, :, None, None].astype(np.float32)
        if metallic_run:
            ref = render(np.broadcast_to(0.9 * wave, metallic.shape).astype(np.float32), roughness).numpy()
        else:
            ref = render(metallic, np.broadcast_to(0.25 + 0.5 * wave, roughness.shape).astype(np.float32)).numpy()

    param = metallic if metallic_run else roughness
    opt = jr.Adam([param], lr=0.1, betas=(0.5, 0.999))
    hist = []
    t0 = time.perf_counter()
    for it in range(iters):
        img = render(metallic, roughness)
        diff = img.numpy() - ref
        loss = float((diff.astype(np.float64) ** 2).sum())
        gm, gr = renderer.grad_material(2.0 * diff)
        opt.step([gm if metallic_run else gr])
        hist.append(loss)
        if not args.quiet:
            print("iter %3d  loss %.4f  %s in [%.3f, %.3f]" % (it, loss, args.param, float(param.min()), float(param.max())), flush=True)
    main.loop_seconds = time.perf_counter() - t0
    if args.history_out:
        np.save(args.history_out, np.asarray(hist, np.float32))
    main.param = param
    return hist


if __name__ == '__main__':
    main()
