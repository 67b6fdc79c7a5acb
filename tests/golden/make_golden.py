#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE's own kernels (oracle/_ref, i.e.
/root/reference/jrender/renderer/dr/softras/cuda/soft_rasterize.py compiled for the host by
oracle/build_ref.py).  Run in the build container (needs /root/reference):

    python tests/golden/make_golden.py

Each file holds the seeded inputs, the parameters and every output of the forward and (serial,
hence deterministic) backward op.  The reference ships no tests or golden vectors of its own
(SURVEY.md §4), so these pin the oracle and the HIP path to the reference arithmetic.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import Oracle                      # noqa: E402
from jrender_amd import synthetic as syn       # noqa: E402

CASES = {
    # name: (generator, kwargs for the op)
    "sphere280_default_48": (lambda: syn.sphere_views(280, 2), dict(image_size=48)),
    "soup300_default_40": (lambda: syn.triangle_soup(300, 1, seed=11, scale=2.0), dict(image_size=40)),
    "soup300_T9_hardrgb_37": (lambda: syn.triangle_soup(300, 1, seed=12, texels=9, scale=2.0),
                              dict(image_size=37, aggr_func_rgb="hard", sigma_val=1e-4)),
    "sphere280_vertex_bary_sum_33": (lambda: syn.sphere_views(280, 1, texels=3),
                                     dict(image_size=33, texture_type="vertex", dist_func="barycentric",
                                          aggr_func_alpha="sum", sigma_val=1e-4)),
    "soup300_hard_hard_K3_32": (lambda: syn.triangle_soup(300, 1, seed=13, scale=3.0),
                                dict(image_size=32, dist_func="hard", aggr_func_alpha="hard",
                                     max_faces_per_pixel_for_grad=3, fill_back=False)),
    "sphere280_nearcull_K20_36": (lambda: syn.sphere_views(280, 1),
                                  dict(image_size=36, near=2.2, far=3.0, max_faces_per_pixel_for_grad=20,
                                       sigma_val=3e-5, gamma_val=1e-2)),
}


def main():
    ref = Oracle("reference", nthreads=0)
    for name, (gen, kw) in CASES.items():
        fv, tex = gen()
        out = ref.forward(fv, tex, **kw)
        g = np.random.default_rng(sum(map(ord, name))).uniform(-1, 1, out["soft_colors"].shape).astype(np.float32)
        gf, gt = ref.backward(out, g)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"), face_vertices=out["face_vertices"], textures=out["textures"],
            params=json.dumps(kw), faces_info=out["faces_info"], aggrs_info=out["aggrs_info"],
            soft_colors=out["soft_colors"], faces_id_buffer=out["faces_id_buffer"], grad_soft_colors=g,
            grad_faces=gf, grad_textures=gt)
        print(name, "touched %.2f" % (out["faces_id_buffer"][:, 0] >= 0).mean(),
              "bytes", os.path.getsize(os.path.join(HERE, name + ".npz")))


if __name__ == "__main__":
    main()
