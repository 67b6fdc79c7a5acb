"""BASELINE.json's configurations on their NAMED inputs, fixtures from
tests/golden/make_golden_g1.py (goldens come from the reference's own kernels compiled for the host):

  C1  spot cow, SoftRas silhouette 256x256                      -> whole image vs golden
  C2  spot cow, SoftRas RGBA 1024x1024 + gradients              -> ids bit-exact / RGBA 1e-4 at 16k golden pixels,
                                                                   whole id buffer vs the C port, gradients 1e-4
  C4  demo2-deform: sphere_1352 + camera.npy + source.npy       -> frame 0 against the reference's own
                                                                   deform_00000.png (8-bit), loss falls on the real data
  C5  NMR, 78 000 faces at 1024x1024                            -> every map bit-exact vs the C restatement
(C3 = 39k faces x 8 views at 1024^2 is tests/test_gpu_fullsize.py.)"""
import os

import numpy as np
import pytest

import jrender_amd as jr
from jrender_amd import _ffi
from jrender_amd.renderer.dr.softras import SoftRasterizeFunction
from oracle import N3mrOracle, Oracle
from tests.util import RGBA_ATOL, bits_equal, grad_err, rel_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def spot():
    return np.load(os.path.join(GOLD, "g1_spot.npz"))


def test_c1_spot_silhouette_256(spot):
    fn = SoftRasterizeFunction(image_size=256)
    rgba = fn(spot["fv"], spot["tex"]).numpy()
    assert rel_err(rgba[0, 3], spot["c1_alpha"], RGBA_ATOL) <= 1.0
    assert 0.1 < (rgba[0, 3] > 0.5).mean() < 0.6                          # a cow-sized silhouette


def test_c2_spot_rgba_1024_and_gradients(spot):
    fv, tex, pix = spot["fv"], spot["tex"], spot["pix"]
    fn = SoftRasterizeFunction(image_size=1024)
    rgba = fn(fv, tex).numpy()
    ids = fn.save_vars[5].numpy()
    assert bits_equal(ids.reshape(16, -1)[:, pix].T, spot["ids"])          # golden = the reference's kernels
    assert rel_err(rgba.reshape(4, -1)[:, pix].T, spot["rgba"], RGBA_ATOL) <= 1.0
    port = Oracle("port", nthreads=0)                                     # whole image vs the C restatement
    ref = port.forward(fv, tex, image_size=1024)
    assert bits_equal(ids, ref["faces_id_buffer"]) and bits_equal(fn.save_vars[3].numpy(), ref["faces_info"])
    assert rel_err(rgba, ref["soft_colors"], RGBA_ATOL) <= 1.0
    G = np.zeros((1, 4, 1024 * 1024), np.float32)
    G[0][:, pix] = spot["g"].T
    gf, gt = fn.grad(G.reshape(1, 4, 1024, 1024))
    assert grad_err(gf.numpy().reshape(spot["grad_faces"].shape), spot["grad_faces"]) <= 1e-4
    assert grad_err(gt.numpy(), spot["grad_textures"]) <= 1e-4


def test_c2_vertex_gradients_through_the_renderer():
    """Renderer.grad_vertices (rasteriser backward -> scatter -> camera VJP) on a spot-sized textured mesh
    against the oracle's face gradients pushed through the same host chain."""
    from jrender_amd.structures.mesh import face_vertices_backward
    v, f = jr.synthetic.sphere_mesh(3300)
    tex = np.random.default_rng(0).uniform(0, 1, (f.shape[0], 25, 3)).astype(np.float32)
    r = jr.Renderer(image_size=256, dr_type='softras')
    r.transform.set_eyes_from_angles(2.732, 30, 0)
    img = r.render_mesh(jr.Mesh(v, f, textures=tex), mode='rgb')
    G = np.random.default_rng(1).uniform(-1, 1, img.shape).astype(np.float32)
    gv = r.grad_vertices(grad_rgb=G)
    fn = r.rasterizer._fn
    fv_in, tex_in = fn.save_vars[0].numpy(), fn.save_vars[1].numpy()
    port = Oracle("port", nthreads=0)
    ref = port.forward(fv_in, tex_in, image_size=256)
    g4 = np.concatenate([G, np.zeros((1, 1) + G.shape[2:], np.float32)], 1)
    rgf, _ = port.backward(ref, g4)
    assert rgf.shape[1] == f.shape[0]                # softras' fill_back is the kernel's double_side flag: no appended faces
    want = r.transform.transformer.backward(face_vertices_backward(rgf, f[None], v.shape[0]), v[None])
    assert grad_err(gv, want) <= 1e-4


def _demo2_model_and_renderer(z, views):
    import importlib.util
    spec = importlib.util.spec_from_file_location("demo2", os.path.join(os.path.dirname(GOLD), "..", "examples", "demo2_deform.py"))
    demo2 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo2)
    model = demo2.Model(z["vertices"], z["faces"])
    r = jr.Renderer(image_size=64, sigma_val=1e-4, aggr_func_rgb='hard', camera_mode='look_at', viewing_angle=15,
                    dr_type='softras', bin_size=16, max_elems_per_bin=2700, max_faces_per_pixel_for_grad=16)
    cam = z["cameras"][:views]
    r.transform.set_eyes_from_angles(cam[:, 0], cam[:, 1], cam[:, 2])
    return demo2, model, r


def test_c4_demo2_first_frame_matches_reference_png():
    """demo2-deform.py:96-99 saves (255 * images_pred[0]).astype(uint8) at iteration 0 = the untouched template
    sphere seen from camera 0.  The reference's frame came from its CUDA coarse-to-fine path (bin_size=16); 8-bit
    agreement is all this file can show."""
    z = np.load(os.path.join(GOLD, "g1_demo2.npz"))
    demo2, model, r = _demo2_model_and_renderer(z, 4)
    v = model.forward()
    mesh = jr.Mesh(np.repeat(v, 4, 0), np.repeat(model.faces, 4, 0))
    sil = r.render_mesh(mesh, mode='silhouettes').numpy().reshape(4, 64, 64)
    mine = (255 * sil[0]).astype(np.uint8).astype(np.int32)
    ref = z["frame0"].astype(np.int32)
    d = np.abs(mine - ref)
    assert (d <= 1).mean() >= 0.995 and d.max() <= 3, ((d <= 1).mean(), d.max())


def test_c4_demo2_loss_falls_on_the_real_data():
    z = np.load(os.path.join(GOLD, "g1_demo2.npz"))
    demo2, _, _ = _demo2_model_and_renderer(z, 1)
    import tempfile
    d = tempfile.mkdtemp()
    np.save(os.path.join(d, "source.npy"), np.repeat(z["alpha"][:, None], 4, 1))      # [N,4,64,64]: alpha in channel 3
    np.save(os.path.join(d, "camera.npy"), z["cameras"])
    hist = demo2.main(["-i", os.path.join(d, "source.npy"), "-c", os.path.join(d, "camera.npy"), "-b", "32",
                       "--iters", "40", "--quiet", "--template-vertices", os.path.join(GOLD, "g1_demo2.npz")])
    assert hist[-1] < hist[0] - 0.08, (hist[0], hist[-1])          # 0.81 -> 0.69 in 40 iterations


def test_c5_nmr_78k_faces_at_1024():
    from jrender_amd.renderer.dr.n3mr import RasterizeFunction
    v, f = jr.synthetic.sphere_mesh(39000)
    eye = np.asarray(jr.get_points_from_angles(2.732, 30., 20.), np.float32)
    ndc = jr.perspective(jr.look_at(v[None], eye), 30.)
    ff = np.concatenate([f, f[:, ::-1]])
    faces = np.ascontiguousarray(ndc[:, ff])
    tex = np.random.default_rng(0).uniform(0, 1, (1, ff.shape[0], 2, 2, 2, 3)).astype(np.float32)
    o = N3mrOracle("port")                                                # any image size; bit-identical to the reference build
    ref = o.forward(faces, tex, image_size=1024)
    fn = RasterizeFunction(1024, 0.1, 100, 1e-3, (0, 0, 0), True, True, True)
    fn(faces, tex)
    _, _, fim, wm, dm, rgb, alpha, fivm, sidx, swt = fn.save_vars
    assert bits_equal(fim.numpy(), ref["face_index_map"]) and bits_equal(dm.numpy(), ref["depth_map"])
    assert bits_equal(wm.numpy(), ref["weight_map"]) and bits_equal(sidx.numpy(), ref["sampling_index_map"])
    assert np.allclose(rgb.numpy(), ref["rgb_map"], rtol=1e-6, atol=1e-7)
    rng = np.random.default_rng(1)
    g = [rng.uniform(-1, 1, ref[k].shape).astype(np.float32) for k in ("rgb_map", "alpha_map", "depth_map")]
    gf, gt = fn.grad(*g)
    rgf, rgt = o.backward(ref, *g)
    assert grad_err(gf.numpy().reshape(rgf.shape), rgf) <= 1e-4 and grad_err(gt.numpy(), rgt) <= 1e-4
