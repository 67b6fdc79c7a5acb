#!/usr/bin/env python3
"""Generates tests/golden/c2f_*.npz from the reference's own COARSE-TO-FINE forward kernels
(/root/reference/jrender/renderer/dr/softras/cuda/soft_rasterize_coarse_to_fine.py compiled for the host by
oracle/build_ref.py: build_c2f, its coarse kernel launched as one thread = ascending bin lists, oracle/ref_c2f_driver.cpp)
and the reference's backward (the same operator for both forward paths, SRW:105-133).  Run in the build container:

    python tests/golden/make_golden_c2f.py

Same keys as make_golden.py's files, so the same tests read them (tests/test_oracle.py: the C restatement; tests/test_gpu_parity.py:
the HIP path, which receives the file's `bin_size` exactly as the reference operator would): both are pinned to what the reference's
BINNED per-pixel kernel writes, not only to its bin_size = 0 kernel.  Surface textures only: for vertex colours the binned kernel's
sampler is not perspective-correct (C2F:439-441 against SRK:168-171), one of the deviations tests/test_oracle_c2f.py lists.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import C2fOracle, Oracle           # noqa: E402
from jrender_amd import synthetic as syn       # noqa: E402


def crowded(nf, texels, seed, batch=1):
    fv, tex = syn.triangle_soup(nf, batch, seed=seed, texels=texels, scale=5.0)
    fv[..., :2] *= 0.55
    return fv, tex


CASES = {
    # name: (generator, kwargs of the reference operator incl. bin_size / max_elems_per_bin)
    "c2f_sphere280_bin16_72": (lambda: syn.sphere_views(280, 2), dict(image_size=72, bin_size=16, max_elems_per_bin=280)),
    "c2f_soup300_T4_hardrgb_sum_bin16_64": (lambda: crowded(300, 4, seed=21),
                                            dict(image_size=64, bin_size=16, max_elems_per_bin=300, aggr_func_rgb="hard",
                                                 aggr_func_alpha="sum", sigma_val=1e-4)),
    "c2f_soup500_K4_bin8_40": (lambda: crowded(500, 1, seed=22),
                               dict(image_size=40, bin_size=8, max_elems_per_bin=500, sigma_val=3e-5, max_faces_per_pixel_for_grad=4)),
}


def main():
    c2f, ref = C2fOracle(), Oracle("reference", nthreads=0)
    for name, (gen, kw) in CASES.items():
        fv, tex = gen()
        out = c2f.forward(fv, tex, **kw)
        assert int(out["elems_per_bin"].max()) <= out["max_elems_per_bin"], "a bin overflowed: the reference would truncate its list"
        out["params"] = dict(out["params"], bin_size=kw["bin_size"], max_elems_per_bin=kw["max_elems_per_bin"])
        g = np.random.default_rng(sum(map(ord, name))).uniform(-1, 1, out["soft_colors"].shape).astype(np.float32)
        gf, gt = ref.backward(out, g)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"), face_vertices=out["face_vertices"], textures=out["textures"],
            params=json.dumps(kw), faces_info=out["faces_info"], aggrs_info=out["aggrs_info"],
            soft_colors=out["soft_colors"], faces_id_buffer=out["faces_id_buffer"], grad_soft_colors=g,
            grad_faces=gf, grad_textures=gt)
        print(name, "touched %.2f" % (out["faces_id_buffer"][:, 0] >= 0).mean(), "longest list", int(out["elems_per_bin"].max()),
              "bytes", os.path.getsize(os.path.join(HERE, name + ".npz")))


if __name__ == "__main__":
    main()
