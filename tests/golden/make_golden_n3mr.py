#!/usr/bin/env python3
"""Generates tests/golden/n3mr_*.npz from the REFERENCE's own NMR kernels (oracle/_ref/libn3mr_ref.so,
i.e. /root/reference/jrender/renderer/dr/n3mr/cuda/rasterize.py compiled for the host, run serially).
Run in the build container:  python tests/golden/make_golden_n3mr.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import N3mrOracle                  # noqa: E402
import jrender_amd as jr                       # noqa: E402


def scene(nf, batch, ts, seed, az0=20.0):
    v, f = jr.synthetic.sphere_mesh(nf)
    eyes = np.stack([np.asarray(jr.get_points_from_angles(2.732, 30., az0 + 90.0 * b), np.float32) for b in range(batch)])
    ndc = jr.perspective(jr.look_at(np.broadcast_to(v[None], (batch,) + v.shape), eyes), 30.)
    ff = np.concatenate([f, f[:, ::-1]])                                   # fill_back duplication (N3R:63-64)
    faces = np.ascontiguousarray(ndc[:, ff])
    tex = np.random.default_rng(seed).uniform(0, 1, (batch, ff.shape[0], ts, ts, ts, 3)).astype(np.float32)
    return faces, tex


CASES = {
    "n3mr_sphere280_ts2_32": (lambda: scene(280, 2, 2, 1), dict(image_size=32, background_color=[0.1, 0.2, 0.3])),
    "n3mr_sphere280_ts4_48": (lambda: scene(280, 1, 4, 2, az0=77.0), dict(image_size=48, eps=1e-3)),
}


def main():
    o = N3mrOracle()
    for name, (gen, kw) in CASES.items():
        faces, tex = gen()
        s = o.forward(faces, tex, **kw)
        rng = np.random.default_rng(sum(map(ord, name)))
        g_rgb = rng.uniform(-1, 1, s["rgb_map"].shape).astype(np.float32)
        g_a = rng.uniform(-1, 1, s["alpha_map"].shape).astype(np.float32)
        g_d = rng.uniform(-1, 1, s["depth_map"].shape).astype(np.float32)
        gf, gt = o.backward(s, g_rgb, g_a, g_d)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), faces=faces, textures=tex, params=json.dumps(kw),
                            face_index_map=s["face_index_map"], weight_map=s["weight_map"], depth_map=s["depth_map"],
                            face_inv_map=s["face_inv_map"], rgb_map=s["rgb_map"], alpha_map=s["alpha_map"],
                            sampling_index_map=s["sampling_index_map"], sampling_weight_map=s["sampling_weight_map"],
                            grad_rgb=g_rgb, grad_alpha=g_a, grad_depth=g_d, grad_faces=gf, grad_textures=gt)
        print(name, "covered %.2f" % (s["face_index_map"] >= 0).mean(), os.path.getsize(os.path.join(HERE, name + ".npz")))


if __name__ == "__main__":
    main()
