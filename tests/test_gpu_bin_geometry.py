"""Screen-bin geometry (round 5): bins of 8, 16 or 32 pixels per side, chosen per launch.

The reference exposes the choice to its caller - `bin_size` / `max_elems_per_bin` of
jrender/renderer/dr/softras/soft_rasterize.py:85-99 -> cuda/soft_rasterize_coarse_to_fine.py:16-18, :129-280 (its own
binning), and demo2-deform.py:65 passes bin_size=16 for 64^2 images.  Here the bin size only decides how finely the face
lists, the launch order and the heavy-tile classification follow the image: every pixel still sees its faces in ascending
index order, so the index buffer, faces_info, colours and aggregates must be THE SAME BITS for every geometry, through
every kernel organisation (one wavefront per tile, 4- / 8-wavefront pipeline, split backward tiles), and equal to the
oracle's.  jr_softras_bin_size / jr_softras_last_launch prove which geometry and which kernels really ran.
"""
import itertools

import numpy as np
import pytest

from oracle import Oracle
from jrender_amd import _ffi, synthetic as syn
from jrender_amd.renderer.dr.softras import SoftRasterizeFunction
from tests.test_gpu_parity import ELEMENTWISE_TOL, ELEMENTWISE_TOL_BARYCENTRIC, check_against, grad_err
from tests.test_gpu_heavy_modes import crowded_soup
from tests.util import bits_equal

pytestmark = pytest.mark.gpu

# faces a bin of that size must list to be "heavy" in these tests: the crowded scenes put hundreds of faces into a
# 32-pixel bin, so nearly every tile takes the multi-wavefront path under each geometry
LOW = {32: 96, 16: 48, 8: 24}


@pytest.fixture(scope="module")
def gctx():
    ctx = _ffi.Context(0)                      # own context: bin size and policy must not leak into the other modules
    yield ctx
    ctx.set_bin_size(0)
    ctx.set_launch_policy(-1, 0)
    ctx.close()


@pytest.fixture(scope="module")
def port():
    return Oracle("port", nthreads=0)


def outputs(fn, g):
    outs = [x.numpy() for x in fn.save_vars[2:]]          # soft_colors, faces_info, aggrs_info, faces_id_buffer
    gf, gt = fn.grad(g)
    return outs, gf.numpy(), gt.numpy()


SCENES = {
    # name: (face_vertices, textures, image size, operator kwargs) - image sizes that are / are not multiples of 8, 16, 32
    "soup64": lambda: (*crowded_soup(1000, 4, seed=59, batch=2), 64, dict(sigma_val=1e-4)),
    "soup100_hard": lambda: (*crowded_soup(700, 1, seed=60), 100, dict(sigma_val=3e-5, aggr_func_rgb="hard", aggr_func_alpha="sum")),
    "sphere280_at_72": lambda: (*syn.sphere_views(280, 3), 72, dict()),
    "sphere3300_at_160_k33": lambda: (*syn.sphere_views(3300, 1), 160, dict(max_faces_per_pixel_for_grad=33, sigma_val=1e-4)),
    "soup_tiny_13": lambda: (*crowded_soup(300, 1, seed=61), 13, dict(sigma_val=1e-4, dist_func="barycentric")),
}


@pytest.mark.parametrize("scene", sorted(SCENES))
def test_every_geometry_gives_the_same_bits(gctx, port, scene):
    fv, tex, IS, kw = SCENES[scene]()
    kw = dict(image_size=IS, **kw)
    g = np.random.default_rng(5).uniform(-1, 1, (fv.shape[0], 4, IS, IS)).astype(np.float32)
    # the oracle once, against the reference geometry of rounds 1-4 (32-pixel bins, one wavefront per tile)
    gctx.set_bin_size(32); gctx.set_launch_policy(0, 0)
    ref = port.forward(fv, tex, **kw)
    if port.ub_events():
        pytest.skip("input hits the reference's undefined-behaviour corner (SRK:107-121)")
    fn = SoftRasterizeFunction(ctx=gctx, **kw)
    fn(fv, tex)
    check_against(ref, fn, g, port.backward(ref, g),
                  ELEMENTWISE_TOL_BARYCENTRIC if kw.get("dist_func") == "barycentric" else ELEMENTWISE_TOL)
    base, base_gf, base_gt = outputs(fn, g)
    seen_heavy = 0
    for b, (thr, waves) in itertools.product((8, 16, 32), (("low", 4), ("low", 8), (-1, 0), (0, 0))):
        gctx.set_bin_size(b)
        gctx.set_launch_policy(LOW[b] if thr == "low" else thr, waves)
        fn = SoftRasterizeFunction(ctx=gctx, **kw)
        fn(fv, tex)
        assert gctx.bin_size() == b and gctx.last_stats()["bins_per_image"] == ((IS + b - 1) // b) ** 2
        info = gctx.last_launch()
        seen_heavy += info["heavy_bins"] > 0
        outs, gf, gt = outputs(fn, g)
        for a, r, name in zip(outs, base, ("soft_colors", "faces_info", "aggrs_info", "faces_id_buffer")):
            assert bits_equal(a, r), (scene, b, thr, waves, name, info)
        assert grad_err(gf, base_gf) <= 2e-6 and grad_err(gt, base_gt) <= 2e-6      # float atomics: only the order of the sums differs
    if "soup" in scene and IS >= 64:
        assert seen_heavy >= 6, "the crowded scenes must reach the multi-wavefront kernels under every bin size"


@pytest.mark.parametrize("bin_size", [8, 16])
@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("dist,rgb,alpha", [("euclidean", "softmax", "prod"), ("euclidean", "hard", "hard"),
                                            ("barycentric", "softmax", "sum"), ("hard", "hard", "prod")])
def test_pipeline_modes_under_small_bins(gctx, port, bin_size, waves, dist, rgb, alpha):
    """The reference's per-mode branches (SRK:331-358, :390-419) through the 4- / 8-wavefront pipeline when the bins are
    one tile or 2x2 tiles: checked against the oracle, and the launch record proves the path."""
    fv, tex = crowded_soup(900, 4, seed=41)
    kw = dict(image_size=64, dist_func=dist, aggr_func_rgb=rgb, aggr_func_alpha=alpha, sigma_val=1e-4)
    gctx.set_bin_size(bin_size); gctx.set_launch_policy(LOW[bin_size], waves)
    ref = port.forward(fv, tex, **kw)
    if port.ub_events():
        pytest.skip("input hits the reference's undefined-behaviour corner (SRK:107-121)")
    fn = SoftRasterizeFunction(ctx=gctx, **kw)
    fn(fv, tex)
    info = gctx.last_launch()
    assert info["four_wavefront_kernel"] and info["heavy_bins"] >= 2 and info["wavefronts_per_workgroup"] == waves, info
    assert gctx.bin_size() == bin_size
    g = np.random.default_rng(3).uniform(-1, 1, ref["soft_colors"].shape).astype(np.float32)
    check_against(ref, fn, g, port.backward(ref, g), ELEMENTWISE_TOL_BARYCENTRIC if dist == "barycentric" else ELEMENTWISE_TOL)


def test_bin_size_kwarg_is_honoured_and_rounded(gctx):
    """SoftRasterizeFunction(bin_size=...) - the reference operator's argument (SRW:85-99) - selects the geometry of ITS
    launches only: rounded up to 8 / 16 / 32, the context's own setting is back afterwards, and the backward reuses the
    forward's records (same token) under the same geometry."""
    fv, tex = syn.sphere_views(3300, 2)
    g = np.random.default_rng(0).uniform(-1, 1, (2, 4, 64, 64)).astype(np.float32)
    gctx.set_bin_size(0); gctx.set_launch_policy(-1, 0)
    auto = gctx.bin_size(64)
    assert auto in (8, 16, 32)
    base = None
    for asked, expect in ((0, auto), (5, 8), (8, 8), (10, 16), (16, 16), (17, 32), (32, 32), (64, 32)):
        fn = SoftRasterizeFunction(image_size=64, sigma_val=1e-4, aggr_func_rgb='hard', bin_size=asked,
                                   max_elems_per_bin=2700, ctx=gctx)
        fn(fv, tex)
        assert gctx.bin_size() == expect, (asked, gctx.bin_size())
        assert gctx.bin_size(64) == auto                      # the context's own choice is untouched
        outs, gf, gt = outputs(fn, g)
        assert gctx.bin_size() == expect                      # the backward ran under the operator's geometry
        if base is None:
            base = (outs, gf)
        for a, r in zip(outs, base[0]):
            assert bits_equal(a, r)
        assert grad_err(gf, base[1]) <= 2e-6


def test_backward_rebuilds_when_the_geometry_changed_in_between(gctx):
    """A backward that presents the forward's token under ANOTHER bin size must not walk the forward's launch order with
    its own tile mapping: the records are rebuilt (jrender_hip.h: token reuse needs the same resolved geometry)."""
    fv, tex = crowded_soup(800, 1, seed=7, batch=2)
    g = np.random.default_rng(1).uniform(-1, 1, (2, 4, 64, 64)).astype(np.float32)
    grads = []
    gctx.set_launch_policy(48, 4)              # (one explicit threshold for all: set_launch_policy itself invalidates tokens)
    for fwd_bin, bwd_bin in ((32, 32), (32, 8), (8, 32), (16, 8)):
        gctx.set_bin_size(fwd_bin)
        fn = SoftRasterizeFunction(image_size=64, sigma_val=1e-4, ctx=gctx)
        fn(fv, tex)
        assert gctx.last_launch()["heavy_bins"] > 0
        gctx.set_bin_size(bwd_bin)             # no generation bump: only the geometry check stands between the backward and stale records
        grads.append(fn.grad(g)[0].numpy())
        assert gctx.bin_size() == bwd_bin
    for gf in grads[1:]:
        assert grad_err(gf, grads[0]) <= 2e-6


def test_automatic_bin_size_follows_image_batch_and_mesh_density(gctx):
    """The default geometry (jr_softras_bin_size, jrender_hip.h): 8-pixel bins up to 128^2, 16 up to 512^2 unless the mesh is dense
    for the image (more than 100 faces per 16-pixel bin on average), above 512^2 16 while the launch stays under 4 Mpixels; and the
    launch really runs under what the query reports."""
    gctx.set_bin_size(0); gctx.set_launch_policy(-1, 0)
    for IS, B, NF, expect in ((64, 64, 3300, 8), (128, 1, 39000, 8), (256, 1, 5856, 16), (256, 8, 39000, 32), (512, 1, 39000, 16),
                              (512, 1, 110000, 32), (1024, 1, 39000, 16), (1024, 4, 39000, 16), (1024, 8, 39000, 32), (2048, 1, 5856, 16),
                              (2048, 2, 5856, 32), (256, 1, 0, 16)):
        assert gctx.bin_size(IS, B, NF) == expect, (IS, B, NF, gctx.bin_size(IS, B, NF))
    for nf, IS, expect in ((3300, 256, 16), (39000, 256, 32)):
        fv, tex = syn.sphere_views(nf, 1)
        SoftRasterizeFunction(image_size=IS, ctx=gctx)(fv, tex)
        assert gctx.bin_size() == expect == gctx.bin_size(IS, 1, fv.shape[1])
