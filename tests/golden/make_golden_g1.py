#!/usr/bin/env python3
"""Fixtures for BASELINE.json's configs on their NAMED inputs — needs /root/reference/data.

  g1_spot.npz   C1/C2: data/obj/spot/spot_triangulated.obj (5 856 faces, texture_res=5) through the host
                front-end of demo1-render.py (Mesh.from_obj -> Lighting -> look_at/perspective at distance 2.732,
                elevation 30, azimuth 0), i.e. the face_vertices / textures the rasteriser receives, plus
                goldens from the reference's own kernels compiled for the host (oracle/_ref): the 256x256
                silhouette (C1) and, at 1024x1024 (C2), ids / RGBA at 16 384 sampled pixels and the gradients of
                a seeded upstream gradient that is non-zero on those pixels.
  g1_demo2.npz  C4: data/obj/sphere/sphere_1352.obj, data/camera.npy, the alpha channel of data/source.npy
                (uint8) and data/results/output_deform/deform_00000.png (the reference's own first frame).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import jrender_amd as jr                      # noqa: E402
from oracle import Oracle                     # noqa: E402

DATA = "/root/reference/data"
HERE = os.path.dirname(os.path.abspath(__file__))


def spot():
    mesh = jr.Mesh.from_obj(os.path.join(DATA, "obj/spot/spot_triangulated.obj"), load_texture=True, texture_res=5,
                            texture_type='surface', dr_type='softras')
    r = jr.Renderer(dr_type='softras')
    r.transform.set_eyes_from_angles(2.732, 30, 0)
    mesh = r.lighting(mesh, r.transform.eyes)
    mesh = r.transform(mesh)
    fv = np.ascontiguousarray(mesh.face_vertices, np.float32)
    tex = np.ascontiguousarray(mesh.face_textures, np.float32)
    ref = Oracle("reference", nthreads=0)
    c1 = ref.forward(fv, tex, image_size=256)["soft_colors"][0, 3]
    s = ref.forward(fv, tex, image_size=1024)
    rng = np.random.default_rng(2)
    touched = np.flatnonzero(s["faces_id_buffer"][0, 0].reshape(-1) >= 0)
    pix = np.unique(np.concatenate([rng.choice(touched, 12288, replace=False), rng.choice(1024 * 1024, 4096, replace=False)]))
    G = np.zeros((1, 4, 1024 * 1024), np.float32)
    G[0][:, pix] = rng.uniform(-1, 1, (4, pix.size)).astype(np.float32)
    G = G.reshape(1, 4, 1024, 1024)
    gf, gt = ref.backward(s, G)
    np.savez_compressed(os.path.join(HERE, "g1_spot.npz"), fv=fv, tex=tex, c1_alpha=c1, pix=pix,
                        ids=s["faces_id_buffer"].reshape(16, -1)[:, pix].T, rgba=s["soft_colors"].reshape(4, -1)[:, pix].T,
                        g=G.reshape(4, -1)[:, pix].T, grad_faces=gf, grad_textures=gt)


def demo2():
    from PIL import Image
    v, f = jr.load_obj(os.path.join(DATA, "obj/sphere/sphere_1352.obj"))
    src = np.load(os.path.join(DATA, "source.npy"))
    cam = np.load(os.path.join(DATA, "camera.npy")).astype(np.float32)
    alpha = np.ascontiguousarray(src[:, 3])                       # anti-aliased 8-bit silhouettes
    frame0 = np.asarray(Image.open(os.path.join(DATA, "results/output_deform/deform_00000.png")))
    np.savez_compressed(os.path.join(HERE, "g1_demo2.npz"), vertices=v, faces=f, cameras=cam,
                        alpha=alpha, frame0=frame0)


if __name__ == "__main__":
    if "--demo2-only" not in sys.argv:
        spot()
    demo2()
    for n in ("g1_spot.npz", "g1_demo2.npz"):
        print(n, os.path.getsize(os.path.join(HERE, n)))
