"""CPU tests of the oracle itself (no GPU): the C restatement against the golden vectors
generated from the reference's own kernels, and — where oracle/_ref is available — against the
reference build directly on fresh seeded inputs in every mode."""
import glob
import itertools
import json
import os

import numpy as np
import pytest

from oracle import Oracle, have_ref
from jrender_amd import synthetic as syn
from tests.util import bits_equal

GOLDEN = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
                if not os.path.basename(p).startswith(("n3mr_", "regress_", "textures_", "g1_", "host_", "pin_")))


@pytest.fixture(scope="module")
def port():
    return Oracle("port", nthreads=0)


def test_golden_files_present():
    assert len(GOLDEN) >= 6


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_port_matches_reference_golden(port, path):
    z = np.load(path)
    kw = json.loads(str(z["params"]))
    out = port.forward(z["face_vertices"], z["textures"], **kw)
    assert port.ub_events() == 0
    for k in ("faces_info", "aggrs_info", "soft_colors", "faces_id_buffer"):
        assert bits_equal(out[k], z[k]), k
    gf, gt = port.backward(out, z["grad_soft_colors"])
    assert bits_equal(gf, z["grad_faces"])
    assert bits_equal(gt, z["grad_textures"])


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_port_matches_reference_build_all_modes(port):
    ref = Oracle("reference", nthreads=0)
    fv, tex = syn.triangle_soup(200, 1, seed=3, texels=4, scale=2.0)
    fvv, texv = syn.sphere_views(280, 1, texels=3)
    g = None
    for dist, rgb, alpha, tt, fb in itertools.product(
            ["hard", "barycentric", "euclidean"], ["hard", "softmax"], ["hard", "sum", "prod"],
            ["surface", "vertex"], [True, False]):
        f, t = (fv, tex) if tt == "surface" else (fvv, texv)
        kw = dict(image_size=32, dist_func=dist, aggr_func_rgb=rgb, aggr_func_alpha=alpha,
                  texture_type=tt, fill_back=fb, sigma_val=1e-4, max_faces_per_pixel_for_grad=5)
        a, b = ref.forward(f, t, **kw), port.forward(f, t, **kw)
        assert port.ub_events() == 0
        for k in ("faces_info", "aggrs_info", "soft_colors", "faces_id_buffer"):
            assert bits_equal(a[k], b[k]), (kw, k)
        if g is None:
            g = np.random.default_rng(5).uniform(-1, 1, a["soft_colors"].shape).astype(np.float32)
        for x, y in zip(ref.backward(a, g), port.backward(b, g)):
            assert bits_equal(x, y), kw


def test_subset_equals_full(port):
    fv, tex = syn.sphere_views(280, 2)
    kw = dict(image_size=40)
    full = port.forward(fv, tex, **kw)
    pix = np.sort(np.random.default_rng(0).choice(2 * 40 * 40, 300, replace=False))
    sub = port.forward_subset(fv, tex, pix, **kw)
    b, r = np.divmod(pix, 40 * 40)
    assert bits_equal(sub["ids"], full["faces_id_buffer"].reshape(2, 16, -1)[b, :, r])
    assert bits_equal(sub["rgba"], full["soft_colors"].reshape(2, 4, -1)[b, :, r])
    assert bits_equal(sub["aggr"], full["aggrs_info"].reshape(2, 2, -1)[b, :, r])
    g = np.zeros_like(full["soft_colors"])
    g.reshape(2, 4, -1)[b, :, r] = np.random.default_rng(1).uniform(-1, 1, (300, 4))
    gf, gt = port.backward(full, g)
    gf2, gt2 = port.backward_subset(full, g, pix)
    assert bits_equal(gf, gf2) and bits_equal(gt, gt2)


def test_empty_and_offscreen(port):
    # all faces outside the view volume -> empty image, every slot -1, zero gradients
    fv, tex = syn.triangle_soup(50, 1, seed=1)
    fv = fv + np.array([5.0, 5.0, 0.0], np.float32)
    out = port.forward(fv, tex, image_size=16)
    assert (out["faces_id_buffer"] == -1).all()
    assert (out["soft_colors"] == 0).all()
    gf, gt = port.backward(out, np.ones_like(out["soft_colors"]))
    assert (gf == 0).all() and (gt == 0).all()


def test_k_buffer_semantics(port):
    # 5 stacked full-screen triangles at depths 5,4,3,2,1.5 with K=2:
    # slots fill with faces 0,1 then the farthest slot is replaced while a nearer face arrives
    tri = np.array([[-3, -3, 0], [3, -3, 0], [0, 3, 0]], np.float32)
    fv = np.stack([tri + np.array([0, 0, z], np.float32) for z in (5, 4, 3, 2, 1.5)])[None]
    tex = np.ones((1, 5, 1, 3), np.float32)
    out = port.forward(fv, tex, image_size=4, max_faces_per_pixel_for_grad=2)
    ids = out["faces_id_buffer"][0, :, 2, 2]
    # arrival: [0,1] (max=slot0:5) ; 2 (3<5) -> [2,1] max slot1:4 ; 3 -> [2,3] max slot0:3 ; 4 -> [4,3]
    assert ids.tolist() == [4, 3]


# ---- NMR: plain-C restatement (oracle/n3mr_oracle.c) vs golden vectors and vs the reference build ----------
N3MR_GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "n3mr_*.npz")))


@pytest.mark.parametrize("path", N3MR_GOLDEN, ids=[os.path.basename(p)[:-4] for p in N3MR_GOLDEN])
def test_n3mr_port_reproduces_golden_vectors(path):
    """The golden files were generated from the reference's own NMR kernels; the C restatement must reproduce
    every map and both gradients bit for bit."""
    import json
    from oracle import N3mrOracle
    z = np.load(path)
    kw = json.loads(str(z["params"]))
    o = N3mrOracle("port")
    s = o.forward(z["faces"], z["textures"], **kw)
    for k in ("face_index_map", "weight_map", "depth_map", "face_inv_map", "rgb_map", "alpha_map",
              "sampling_index_map", "sampling_weight_map"):
        assert bits_equal(s[k], z[k].reshape(s[k].shape)), k
    gf, gt = o.backward(s, z["grad_rgb"], z["grad_alpha"], z["grad_depth"])
    assert bits_equal(gf, z["grad_faces"].reshape(gf.shape)) and bits_equal(gt, z["grad_textures"])


def test_n3mr_port_bit_identical_to_reference_build():
    import jrender_amd as jr
    from oracle import N3mrOracle
    try:
        ref = N3mrOracle("reference")
    except FileNotFoundError:
        pytest.skip("/root/reference not mounted and oracle/_ref/libn3mr_ref.so not shipped")
    port = N3mrOracle("port")
    rng = np.random.default_rng(0)
    for i in range(40):
        B = int(rng.choice([1, 2]))
        if rng.integers(2):
            v, f = jr.synthetic.sphere_mesh(280)
            eyes = np.stack([np.asarray(jr.get_points_from_angles(2.732, float(rng.uniform(-50, 50)),
                                                                  float(rng.uniform(0, 360))), np.float32) for _ in range(B)])
            ndc = jr.perspective(jr.look_at(np.broadcast_to(v[None], (B,) + v.shape), eyes), 30.)
            faces = np.ascontiguousarray(ndc[:, np.concatenate([f, f[:, ::-1]])]).astype(np.float32)
        else:
            fv, _ = jr.synthetic.triangle_soup(int(rng.integers(1, 500)), B, seed=int(rng.integers(1 << 30)),
                                               scale=float(rng.uniform(1, 30)))
            fv[..., :2] *= float(rng.uniform(0.8, 1.6))
            faces = np.concatenate([fv, fv[:, :, ::-1]], 1).astype(np.float32)
        ts, IS = int(rng.choice([2, 3, 4])), int(rng.choice([8, 16, 32, 48, 64]))
        tex = rng.uniform(0, 1, (B, faces.shape[1], ts, ts, ts, 3)).astype(np.float32)
        rrgb, ra, rd = [(True, True, True), (False, True, False), (True, False, False), (False, True, True)][int(rng.integers(4))]
        kw = dict(image_size=IS, near=float(rng.choice([0.1, 2.2])), far=float(rng.choice([100., 3.6])),
                  eps=float(rng.choice([1e-3, 1e-2])), background_color=(0.1, 0.2, 0.3),
                  return_rgb=rrgb, return_alpha=ra, return_depth=rd)
        a = ref.forward(faces, tex if rrgb else None, **kw)
        b = port.forward(faces, tex if rrgb else None, **kw)
        for k in ("face_index_map", "weight_map", "depth_map", "face_inv_map", "rgb_map", "alpha_map",
                  "sampling_index_map", "sampling_weight_map", "faces_inv"):
            assert bits_equal(a[k], b[k]), (i, k)
        shape = a["face_index_map"].shape
        g = [rng.uniform(-1, 1, shape + (3,)).astype(np.float32) if rrgb else None,
             rng.uniform(-1, 1, shape).astype(np.float32) if ra else None,
             rng.uniform(-1, 1, shape).astype(np.float32) if rd else None]
        for x, y in zip(ref.backward(a, *g), port.backward(b, *g)):
            assert bits_equal(x, y), i
