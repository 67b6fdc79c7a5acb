"""The reference's COARSE-TO-FINE forward against its own bin_size = 0 forward (CPU; VERDICT r5 "missing" #4).

`SoftRasterizeFunction(bin_size > 0)` of the reference runs other kernels (soft_rasterize_coarse_to_fine.py, C2F): a bounding-box
kernel, a block-cooperative binning kernel and a per-pixel kernel that walks its bin's list instead of all faces.  This package
honours `bin_size` as a bin geometry and promises the bin_size = 0 RESULTS for every value.  What that promise
means in the reference's own terms is established here, from the reference's kernels compiled for the host
(oracle/build_ref.py: build_c2f; the binning kernel launched as ONE thread - a legal launch of its stride loops that makes
every list ascending, oracle/ref_c2f_driver.cpp):

  * with ascending, untruncated lists the binned forward writes THE SAME BITS as the bin_size = 0 forward - every output, all
    36 distance x colour x alpha x culling modes with surface textures;
  * where it does not, the reference deviates from itself, in five documented ways that this package deliberately does not
    reproduce (each shown below on a scene that triggers it and nothing else);
  * tests/golden/c2f_*.npz are written by the binned kernels; tests/test_oracle.py holds the C restatement to them and
    tests/test_gpu_parity.py the HIP path (which gets `bin_size` like the reference operator would).
"""
import itertools

import numpy as np
import pytest

from oracle import Oracle, have_ref
from jrender_amd import synthetic as syn
from tests.util import bits_equal

KEYS = ("soft_colors", "faces_info", "aggrs_info", "faces_id_buffer")


def _c2f():
    try:
        from oracle import C2fOracle
        return C2fOracle(nthreads=0)
    except (FileNotFoundError, OSError):
        return None


pytestmark = pytest.mark.skipif(not have_ref() or _c2f() is None, reason="oracle/_ref not built (needs /root/reference)")


@pytest.fixture(scope="module")
def ref():
    return Oracle("reference", nthreads=0)


@pytest.fixture(scope="module")
def c2f():
    return _c2f()


def crowded(nf, texels, seed, batch=1):
    fv, tex = syn.triangle_soup(nf, batch, seed=seed, texels=texels, scale=5.0)
    fv[..., :2] *= 0.55
    return fv, tex


def same(a, b):
    return {k: bits_equal(a[k], b[k]) for k in KEYS}


def lists_ascending_and_complete(out):
    e, n, M = out["bin_elems"], out["elems_per_bin"], out["max_elems_per_bin"]
    assert int(n.max()) <= M, "a bin overflowed"
    valid = e >= 0
    assert (valid.sum(-1) == n).all()                                   # every counted face is listed
    assert (valid[..., 1:] <= valid[..., :-1]).all()                    # ... contiguously from slot 0
    pairs = valid[..., 1:] & valid[..., :-1]
    assert (np.diff(e, axis=-1)[pairs] > 0).all()                       # ... in ascending order (global ids: batch * NF + face)


def test_binned_forward_with_ascending_lists_writes_the_bits_of_the_unbinned_forward(ref, c2f):
    """36 modes (surface textures), two views, an image size that is not a multiple of the bin size, K = 5 so that the K-nearest
    buffer replaces as well as appends.  'hard' colour: aggrs_info[1] is the winning face - the binned kernel stores the GLOBAL id
    (C2F:701 `face_index_min = fn`, fn = view * NF + face), the unbinned one the face's own; everything else is the same bits."""
    fv, tex = crowded(300, 4, seed=3, batch=2)
    NF = fv.shape[1]
    for dist, rgb, alpha, fb in itertools.product(["hard", "barycentric", "euclidean"], ["hard", "softmax"], ["hard", "sum", "prod"], [True, False]):
        kw = dict(image_size=56, dist_func=dist, aggr_func_rgb=rgb, aggr_func_alpha=alpha, fill_back=fb, sigma_val=1e-4,
                  max_faces_per_pixel_for_grad=5)
        a = ref.forward(fv, tex, **kw)
        b = c2f.forward(fv, tex, bin_size=16, max_elems_per_bin=NF, **kw)
        lists_ascending_and_complete(b)
        if rgb == "hard":
            winner = a["aggrs_info"][:, 1].copy()
            for view in range(1, winner.shape[0]):
                winner[view][winner[view] >= 0] += view * NF
            assert bits_equal(winner, b["aggrs_info"][:, 1]) and bits_equal(a["aggrs_info"][:, 0], b["aggrs_info"][:, 0]), kw
            assert all(bits_equal(a[k], b[k]) for k in KEYS if k != "aggrs_info"), (kw, same(a, b))
        else:
            assert all(same(a, b).values()), (kw, same(a, b))


@pytest.mark.parametrize("image_size,bin_size", [(64, 16), (100, 16), (40, 8), (256, 32), (27 * 4, 4)])
def test_every_bin_geometry_the_reference_accepts(ref, c2f, image_size, bin_size):
    fv, tex = syn.sphere_views(280, 2)
    a = ref.forward(fv, tex, image_size=image_size)
    b = c2f.forward(fv, tex, bin_size=bin_size, max_elems_per_bin=280, image_size=image_size)
    lists_ascending_and_complete(b)
    assert all(same(a, b).values()), same(a, b)


def test_more_than_27_bins_per_edge_is_refused(c2f):
    fv, tex = syn.sphere_views(280, 1)
    with pytest.raises(ValueError):                                     # C2F:16-18
        c2f.forward(fv, tex, bin_size=4, max_elems_per_bin=280, image_size=27 * 4 + 1)


# ---- where the reference's binned path differs from its own bin_size = 0 path (none of it reproduced by this package) ------------

def test_deviation_1_overflowing_lists_are_truncated(ref, c2f):
    """C2F:236-261: a chunk of 512 faces whose entries do not fit the bin's remaining room is dropped WHOLE (the counter still
    advances); the default room is num_faces / 5 (SRW:85-90).  Pixels of such a bin lose faces."""
    fv, tex = crowded(1300, 1, seed=4)
    kw = dict(image_size=64, sigma_val=1e-4)
    a = ref.forward(fv, tex, **kw)
    b = c2f.forward(fv, tex, bin_size=16, max_elems_per_bin=0, **kw)                  # the reference's default: 260
    assert b["max_elems_per_bin"] == 260 and int(b["elems_per_bin"].max()) > 260
    listed = (b["bin_elems"] >= 0).sum(-1)
    assert int(listed.max()) < int(b["elems_per_bin"].max())                         # counted but not listed
    assert not bits_equal(a["faces_id_buffer"], b["faces_id_buffer"]) and not bits_equal(a["soft_colors"], b["soft_colors"])
    full = c2f.forward(fv, tex, bin_size=16, max_elems_per_bin=1300, **kw)
    assert all(same(a, full).values())


def test_deviation_2_faces_reaching_behind_the_camera_are_dropped(ref, c2f):
    """C2F:119-120: a face with any vertex at z < 1e-8 is skipped by the binning.  The bin_size = 0 kernel keeps it: its alpha
    contribution is accumulated before the depth cull (SRK:350-365) and it enters the K-nearest buffer where its depth is in range."""
    fv, tex = crowded(300, 1, seed=3)
    fv[0, 5, 0, 2] = -0.5
    kw = dict(image_size=64, sigma_val=1e-4)
    a = ref.forward(fv, tex, **kw)
    b = c2f.forward(fv, tex, bin_size=16, max_elems_per_bin=300, **kw)
    assert not (b["bin_elems"] == 5).any() and (a["faces_id_buffer"] == 5).any()
    assert (a["soft_colors"][:, 3] != b["soft_colors"][:, 3]).any()
    fv[0, 5, 0, 2] = 2.5
    assert all(same(ref.forward(fv, tex, **kw), c2f.forward(fv, tex, bin_size=16, max_elems_per_bin=300, **kw)).values())


def test_deviation_3_the_bin_margin_ignores_sigma(ref, c2f):
    """C2F:15, :104: boxes are widened by sqrt(0.01) = 0.1 NDC whatever sigma is, while a face reaches sqrt(dist_eps * sigma) far
    (SRK:316).  Up to sigma ~ 1.08e-3 (radius 0.1) nothing is lost; beyond, faces near a bin border vanish from its pixels."""
    fv, tex = crowded(300, 1, seed=3)
    for sigma, expect_same in ((1e-3, True), (3e-3, False)):
        kw = dict(image_size=64, sigma_val=sigma)
        a = ref.forward(fv, tex, **kw)
        b = c2f.forward(fv, tex, bin_size=16, max_elems_per_bin=300, **kw)
        assert all(same(a, b).values()) == expect_same, (sigma, same(a, b))


def test_deviation_4_vertex_colours_are_not_perspective_correct(ref, c2f):
    """C2F:439-441 interpolates vertex colours with the clipped barycentrics as they are; the bin_size = 0 forward divides by depth
    (SRK:168-171).  Coverage, alpha, aggregates and the index buffer agree; the colours differ by per cents."""
    fv, tex = syn.sphere_views(280, 2, texels=3)
    kw = dict(image_size=56, texture_type="vertex", sigma_val=1e-4)
    a = ref.forward(fv, tex, **kw)
    b = c2f.forward(fv, tex, bin_size=16, max_elems_per_bin=280, **kw)
    s = same(a, b)
    assert s["faces_info"] and s["faces_id_buffer"] and s["aggrs_info"] and not s["soft_colors"]
    assert bits_equal(a["soft_colors"][:, 3], b["soft_colors"][:, 3])
    assert 1e-3 < float(np.abs(a["soft_colors"][:, :3] - b["soft_colors"][:, :3]).max()) < 0.2


def test_deviation_5_list_order_decides_the_bits(ref, c2f):
    """The one deviation that is not in the kernels' text but in their launch: <<<64, 512>>> blocks race for list segments
    (C2F:236), so a bin's faces arrive chunk-permuted from run to run.  What another arrival order does to a pixel, shown with the
    reference's bin_size = 0 kernel on the face-reversed scene (ids mapped back): the K nearest faces are the same SET, but they
    sit in other slots of the index buffer, and alpha product / online softmax round differently - the outputs are no longer the
    same bits, only close.  A bit-exact index buffer therefore needs one defined order: ascending."""
    fv, tex = crowded(300, 1, seed=3)
    kw = dict(image_size=64, sigma_val=1e-4, max_faces_per_pixel_for_grad=4)
    a = ref.forward(fv, tex, **kw)
    perm = np.arange(fv.shape[1])[::-1]
    r = ref.forward(fv[:, perm], tex[:, perm], **kw)
    ids = r["faces_id_buffer"].copy()
    ids[ids >= 0] = perm[ids[ids >= 0]]                                   # back to the original numbering
    assert (np.sort(ids, 1) == np.sort(a["faces_id_buffer"], 1)).all()    # same K nearest faces per pixel ...
    assert (ids != a["faces_id_buffer"]).any()                            # ... in other slots
    assert not bits_equal(a["soft_colors"], r["soft_colors"])
    assert np.allclose(a["soft_colors"], r["soft_colors"], rtol=1e-4, atol=1e-6)


def test_the_reference_s_own_first_demo2_frame(ref, c2f):
    """The one output file the reference ships for this path: data/results/output_deform/deform_00000.png, written by
    demo2-deform.py:96-99 at iteration 0 from its CUDA coarse-to-fine kernels (bin_size=16, max_elems_per_bin=2700).  The same
    kernels compiled for the host, fed by this package's host front end (DeformModel + look_at + perspective(15 deg) as NumPy),
    reproduce it to the 8 bits it holds - 99 % of the pixels exactly, all within 3 grey levels (nvcc's contraction and exp against
    -ffp-contract=off) - and, once more, write the bits of the bin_size = 0 forward.  tests/golden/g1_demo2.npz carries the
    reference's mesh, cameras and that frame (make_golden_g1.py)."""
    import os
    import jrender_amd as jr
    from jrender_amd.renderer import transform as T
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "g1_demo2.npz"))
    v, f, cam = z["vertices"], z["faces"], z["cameras"]
    template = jr.DeformModel(v, f).forward().reshape(1, -1, 3)                     # iteration 0: zero displacement
    eye = T.get_points_from_angles(cam[0:1, 0], cam[0:1, 1], cam[0:1, 2])
    screen = T.perspective(T.look_at(template.astype(np.float32), eye), angle=15.)
    fv = screen[0][f.astype(np.int64)][None].astype(np.float32)
    tex = np.ones((1, f.shape[0], 1, 3), np.float32)
    kw = dict(image_size=64, sigma_val=1e-4, aggr_func_rgb="hard")
    b = c2f.forward(fv, tex, bin_size=16, max_elems_per_bin=2700, **kw)
    lists_ascending_and_complete(b)
    assert f.shape[0] // 5 < int(b["elems_per_bin"].max()) <= 2700                  # (why the demo passes max_elems_per_bin: the operator's default room, NF / 5, would truncate)
    frame = (255 * b["soft_colors"][0, 3]).astype(np.uint8).astype(np.int32)        # demo2-deform.py:98
    d = np.abs(frame - z["frame0"].astype(np.int32))
    assert (d == 0).mean() >= 0.985 and (d <= 1).mean() >= 0.995 and d.max() <= 3, ((d == 0).mean(), (d <= 1).mean(), d.max())
    assert all(same(ref.forward(fv, tex, **kw), b).values())
