"""The multi-wavefront kernels in EVERY mode (VERDICT r3 next #1a).

The product gives a whole workgroup (4 or 8 wavefronts, evaluate / apply pipeline: softras_forward.hip tile_heavy_pipe)
to the tiles of bins that list more than 512 faces, and splits such tiles over four wavefronts in small backwards
(softras_backward.hip, tune::bwd_split).  With the default threshold only the limb of a 39k-face sphere gets there, i.e.
euclidean / softmax / prod at K = 16.  jr_softras_set_launch_policy lowers the threshold at RUN TIME, so that here nearly
every tile of a small scene takes the pipeline, with 4 and with 8 wavefronts, through the reference's per-mode branches:
  distance / alpha   cuda/soft_rasterize.py:331-358
  hard / softmax rgb cuda/soft_rasterize.py:390-419
  K-buffer, K != 16  cuda/soft_rasterize.py:369-385
Bars as everywhere: face-index buffer and faces_info bit-exact, RGBA / aggregates 1e-4, gradients 1e-4 of the largest
gradient.  jr_softras_last_launch proves that the pipeline really ran (heavy bins > 0, the forced workgroup size).
"""
import itertools

import numpy as np
import pytest

from oracle import Oracle
from jrender_amd import _ffi, synthetic as syn
from jrender_amd.renderer.dr.softras import SoftRasterizeFunction
from tests.test_gpu_parity import ELEMENTWISE_TOL, ELEMENTWISE_TOL_BARYCENTRIC, check_against
from tests.util import bits_equal

pytestmark = pytest.mark.gpu

HEAVY_MIN = 96          # faces listed in a 32x32 bin; the scenes below put 150-700 into every bin they touch


@pytest.fixture(scope="module")
def hctx():
    ctx = _ffi.Context(0)                      # own context: the policy must not leak into the other test modules
    ctx.set_bin_size(32)                       # this module's scenes and threshold are sized for 32-pixel bins (the other geometries: test_gpu_bin_geometry.py)
    yield ctx
    ctx.set_launch_policy(-1, 0)
    ctx.close()


@pytest.fixture(scope="module")
def port():
    return Oracle("port", nthreads=0)


def crowded_soup(nf, texels, seed, batch=1):
    """Random triangles over a quarter of the screen (several hundred per 32x32 bin, tens per pixel), depths spread so
    that the K-buffer replaces as well as appends."""
    fv, tex = syn.triangle_soup(nf, batch, seed=seed, texels=texels, scale=5.0)
    fv[..., :2] *= 0.55
    return fv, tex


def run_forced(ctx, port, fv, tex, waves, expect_heavy=True, seed=0, **kw):
    ctx.set_launch_policy(HEAVY_MIN, waves)
    ref = port.forward(fv, tex, **kw)
    if port.ub_events():
        pytest.skip("input hits the reference's undefined-behaviour corner (SRK:107-121)")
    fn = SoftRasterizeFunction(ctx=ctx, **kw)
    fn(fv, tex)
    info = ctx.last_launch()
    if expect_heavy:
        assert info["four_wavefront_kernel"] and info["heavy_bins"] >= 2, info
        assert info["wavefronts_per_workgroup"] == waves and info["heavy_min_faces"] == HEAVY_MIN, info
        assert ctx.last_stats()["max_faces_in_bin"] > 2 * HEAVY_MIN
    else:
        assert not info["four_wavefront_kernel"] and info["wavefronts_per_workgroup"] == 1, info
    g = np.random.default_rng(seed).uniform(-1, 1, ref["soft_colors"].shape).astype(np.float32)
    check_against(ref, fn, g, port.backward(ref, g),
                  ELEMENTWISE_TOL_BARYCENTRIC if kw.get("dist_func") == "barycentric" else ELEMENTWISE_TOL)
    return ref, fn


@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("dist,rgb,alpha", list(itertools.product(
    ["hard", "barycentric", "euclidean"], ["hard", "softmax"], ["hard", "sum", "prod"])))
def test_pipeline_all_modes(hctx, port, dist, rgb, alpha, waves):
    fv, tex = crowded_soup(900, 4, seed=41)
    run_forced(hctx, port, fv, tex, waves, image_size=64, dist_func=dist, aggr_func_rgb=rgb, aggr_func_alpha=alpha,
               sigma_val=1e-4)


@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("K", [1, 16, 17, 33, 64])
def test_pipeline_k_values(hctx, port, K, waves):
    fv, tex = crowded_soup(1100, 1, seed=43)
    ref, fn = run_forced(hctx, port, fv, tex, waves, image_size=64, max_faces_per_pixel_for_grad=K, sigma_val=1e-4)
    ids = ref["faces_id_buffer"]
    if K <= 17:       # the replace path of the K-buffer is exercised: pixels with a full buffer exist
        assert (ids[:, K - 1] >= 0).mean() > 0.05


@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("texels", [1, 4, 9])
@pytest.mark.parametrize("rgb", ["hard", "softmax"])
def test_pipeline_texture_resolutions(hctx, port, texels, rgb, waves):
    fv, tex = crowded_soup(800, texels, seed=47, batch=2)
    run_forced(hctx, port, fv, tex, waves, image_size=56, aggr_func_rgb=rgb, sigma_val=3e-5)


@pytest.mark.parametrize("rgb", ["hard", "softmax"])
def test_vertex_colours_keep_the_single_wavefront_kernel(hctx, port, rgb):
    # the cell of the pipeline has no room for three vertex colours: texture_type='vertex' renders with one wavefront
    # per tile whatever the policy says (and the backward split still runs)
    fv, tex = crowded_soup(800, 3, seed=53)
    run_forced(hctx, port, fv, tex, 8, expect_heavy=False, image_size=56, texture_type="vertex", aggr_func_rgb=rgb,
               sigma_val=1e-4)


def test_workgroup_sizes_and_threshold_give_the_same_bits(hctx, port):
    """4 / 8 wavefronts per heavy tile, the default threshold (no heavy tile in this scene) and 'never': the same
    index buffer, colours and aggregates bit for bit (the same device functions on the same operands in the same
    per-pixel order), gradients equal up to the order of the float atomics."""
    fv, tex = crowded_soup(1000, 4, seed=59, batch=2)
    kw = dict(image_size=64, sigma_val=1e-4, max_faces_per_pixel_for_grad=16)
    g = np.random.default_rng(5).uniform(-1, 1, (2, 4, 64, 64)).astype(np.float32)
    outs = []
    for heavy_min, waves in ((HEAVY_MIN, 4), (HEAVY_MIN, 8), (-1, 0), (0, 0)):
        hctx.set_launch_policy(heavy_min, waves)
        fn = SoftRasterizeFunction(ctx=hctx, **kw)
        fn(fv, tex)
        info = hctx.last_launch()
        assert (info["heavy_bins"] > 0) == (heavy_min == HEAVY_MIN), info
        outs.append([x.numpy() for x in fn.save_vars[2:]] + [fn.grad(g)[0].numpy()])
    for o in outs[1:]:
        for a, b in zip(o[:-1], outs[0][:-1]):
            assert bits_equal(a, b)
        scale = np.abs(outs[0][-1]).max()
        assert np.abs(o[-1] - outs[0][-1]).max() <= 2e-6 * scale


def test_first_forward_of_a_shape_takes_the_same_path_as_the_second(hctx):
    """The workgroup size depends on the number of heavy tiles, which the device counts: a shape the context has not seen
    waits for this forward's own count, a shape it has seen launches speculatively with the remembered one - the same
    choice (it used to be 4 wavefronts on the first call, 8 from the second on)."""
    hctx.set_launch_policy(HEAVY_MIN, 0)        # automatic workgroup size; also forgets the remembered shapes
    fv, tex = crowded_soup(700, 1, seed=61)
    fv2, tex2 = crowded_soup(1300, 1, seed=62, batch=2)
    seen = {}
    for rep in range(3):                         # alternating two shapes: both stay remembered
        for name, (a, b, size) in (("one", (fv, tex, 64)), ("two", (fv2, tex2, 96))):
            fn = SoftRasterizeFunction(image_size=size, sigma_val=1e-4, ctx=hctx)
            fn(a, b)
            info = hctx.last_launch()
            assert info["four_wavefront_kernel"] and info["heavy_bins"] > 0
            seen.setdefault(name, []).append((info["wavefronts_per_workgroup"], info["heavy_bins"]))
    for name, v in seen.items():
        assert len(set(v)) == 1, (name, v)
