"""GPU parity of the NMR ("n3mr") path: HIP kernels through the C ABI vs the reference's own
kernels compiled for the host (oracle/_ref/libn3mr_ref.so) and the golden vectors generated from
them.  Bars: face_index_map, weight_map, depth_map, face_inv_map, sampling indices/weights and
rgb bit-exact (the forward has no transcendental and no order dependence except the z-test, whose
ties resolve to the lowest face id on both sides); gradients within 1e-4 of max |grad| (per-face
scan-line sums are accumulated per lane, float atomics in the depth / texture passes)."""
import glob
import json
import os

import numpy as np
import pytest

import jrender_amd as jr
from jrender_amd.renderer.dr.n3mr import RasterizeFunction
from tests.util import bits_equal, grad_err

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "n3mr_*.npz")))


def _scene(nf, batch, ts, seed, az0=20.0):
    v, f = jr.synthetic.sphere_mesh(nf)
    eyes = np.stack([np.asarray(jr.get_points_from_angles(2.732, 30., az0 + 90.0 * b), np.float32) for b in range(batch)])
    ndc = jr.perspective(jr.look_at(np.broadcast_to(v[None], (batch,) + v.shape), eyes), 30.)
    ff = np.concatenate([f, f[:, ::-1]])
    tex = np.random.default_rng(seed).uniform(0, 1, (batch, ff.shape[0], ts, ts, ts, 3)).astype(np.float32)
    return np.ascontiguousarray(ndc[:, ff]), tex


def _check(ref, fn, g_rgb, g_a, g_d, ref_grads):
    f, tex, fim, wm, dm, rgb, alpha, fivm, sidx, swt = fn.save_vars
    assert bits_equal(fim.numpy(), ref["face_index_map"]), "face_index_map"
    assert bits_equal(dm.numpy(), ref["depth_map"]), "depth_map"
    assert bits_equal(wm.numpy(), ref["weight_map"]), "weight_map"
    assert bits_equal(fivm.numpy().reshape(ref["face_inv_map"].shape), ref["face_inv_map"]), "face_inv_map"
    assert bits_equal(sidx.numpy(), ref["sampling_index_map"]) and bits_equal(swt.numpy(), ref["sampling_weight_map"])
    assert np.allclose(rgb.numpy(), ref["rgb_map"], rtol=1e-6, atol=1e-7)
    assert bits_equal(alpha.numpy(), ref["alpha_map"])
    gf, gt = fn.grad(g_rgb, g_a, g_d)
    assert grad_err(gf.numpy().reshape(ref_grads[0].shape), ref_grads[0]) <= 1e-4
    assert grad_err(gt.numpy(), ref_grads[1]) <= 1e-4


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_n3mr_golden(path):
    z = np.load(path)
    kw = json.loads(str(z["params"]))
    fn = RasterizeFunction(kw["image_size"], 0.1, 100, kw.get("eps", 1e-3), kw.get("background_color", (0, 0, 0)),
                           True, True, True)
    fn(z["faces"], z["textures"])
    _check(z, fn, z["grad_rgb"], z["grad_alpha"], z["grad_depth"],
           (z["grad_faces"].reshape(z["faces"].shape[0], -1, 3, 3), z["grad_textures"]))


@pytest.mark.parametrize("nf,IS,ts", [(280, 64, 2), (3300, 128, 3), (3300, 256, 2),
                                       (280, 48, 9),     # ts=9: texel gradients beyond the LDS accumulators
                                       (39000, 512, 2)]) # the bench mesh (78 000 faces with fill_back), walks of 500 px
def test_n3mr_vs_reference_build(nf, IS, ts):
    from oracle import N3mrOracle
    try:
        o = N3mrOracle()
    except FileNotFoundError:
        pytest.skip("oracle/_ref/libn3mr_ref.so not available")
    faces, tex = _scene(nf, 2, ts, 3)
    ref = o.forward(faces, tex, image_size=IS, background_color=(0.3, 0.2, 0.1))
    rng = np.random.default_rng(4)
    g_rgb = rng.uniform(-1, 1, ref["rgb_map"].shape).astype(np.float32)
    g_a = rng.uniform(-1, 1, ref["alpha_map"].shape).astype(np.float32)
    g_d = rng.uniform(-1, 1, ref["depth_map"].shape).astype(np.float32)
    fn = RasterizeFunction(IS, 0.1, 100, 1e-3, (0.3, 0.2, 0.1), True, True, True)
    fn(faces, tex)
    _check(ref, fn, g_rgb, g_a, g_d, o.backward(ref, g_rgb, g_a, g_d))


def test_n3mr_silhouette_only_and_renderer_api():
    from oracle import N3mrOracle
    faces, _ = _scene(280, 1, 2, 5)
    fn = RasterizeFunction(64, 0.1, 100, 1e-3, None, False, True, False)
    rgb, alpha, depth = fn(faces)
    assert rgb is None and depth is None and 0.2 < alpha.numpy().mean() < 0.6
    ga = np.random.default_rng(0).uniform(-1, 1, (1, 64, 64)).astype(np.float32)
    gf, gt = fn.grad(None, ga, None)
    assert gt is None and np.isfinite(gf.numpy()).all() and np.abs(gf.numpy()).max() > 0
    try:
        o = N3mrOracle()
        s = o.forward(faces, None, image_size=64, return_rgb=False, return_depth=False)
        gfo, _ = o.backward(s, None, ga, None)
        assert grad_err(gf.numpy().reshape(gfo.shape), gfo) <= 1e-4
    except FileNotFoundError:
        pass
    # Renderer(dr_type='n3mr'): flip + permute + anti-aliasing pool on the host
    v, f = jr.synthetic.sphere_mesh(280)
    mesh = jr.Mesh(v, f, textures=np.random.default_rng(1).uniform(0, 1, (f.shape[0], 2, 2, 2, 3)).astype(np.float32),
                   dr_type='n3mr')
    r = jr.Renderer(image_size=32, dr_type='n3mr', anti_aliasing=True)
    rgb = r.render_mesh(mesh, mode='rgb')
    assert rgb.shape == (1, 3, 32, 32) and rgb.numpy().max() <= 1.0 + 1e-6
    mesh.reset_()
    sil = r.render_mesh(mesh, mode='silhouettes')
    assert sil.shape == (1, 1, 32, 32) or sil.shape == (1, 32, 32)


def test_n3mr_validation():
    faces, tex = _scene(280, 1, 2, 5)
    with pytest.raises(RuntimeError, match="near"):
        RasterizeFunction(32, -1.0, 100, 1e-3, (0, 0, 0), True, True, True)(faces, tex)
    with pytest.raises(ValueError):
        RasterizeFunction(32, 0.1, 100, 1e-3, (0, 0, 0), True, False, False)(faces, None)


def _degenerate_scene():
    faces, tex = _scene(280, 1, 2, 9)
    faces = faces.copy()
    n = faces.shape[1]
    faces[0, 5] = faces[0, 5, 0]                 # all three vertices coincide: zero area, w_sum = 0 -> zp = NaN
    faces[0, 40, 1] = faces[0, 40, 0]            # two coincide: zero area, collinear
    faces[0, n // 2 + 7, 2, 0] = np.nan          # NaN x in one vertex
    faces[0, 90, :, 2] = np.nan                  # NaN depths
    # a coincident-vertex face in front of everything, over an otherwise EMPTY corner of the image
    faces[0, 11] = np.asarray([-0.9, -0.9, 1.5], np.float32)
    return faces, tex


def test_n3mr_degenerate_and_nan_faces():
    """a zero-area face (zp = NaN) must not be drawn — the reference's `zp < depth_map` is false
    for NaN (N3K:147).  Maps bit-exact vs the reference build incl. coincident-vertex and NaN faces."""
    from oracle import N3mrOracle
    o = N3mrOracle()
    faces, tex = _degenerate_scene()
    for IS in (64, 128):
        try:
            ref = o.forward(faces, tex, image_size=IS)
        except RuntimeError:
            continue
        fn = RasterizeFunction(IS, 0.1, 100, 1e-3, (0, 0, 0), True, True, True)
        fn(faces, tex)
        f, t, fim, wm, dm, rgb, alpha, fivm, sidx, swt = fn.save_vars
        assert bits_equal(fim.numpy(), ref["face_index_map"]), "face_index_map"
        assert not np.isin(fim.numpy(), [5, 11, 40]).any()          # the zero-area faces are never drawn
        d, dr = dm.numpy(), ref["depth_map"]
        assert np.array_equal(np.isnan(d), np.isnan(dr)) and bits_equal(np.nan_to_num(d), np.nan_to_num(dr))
        assert bits_equal(alpha.numpy(), ref["alpha_map"])
        w, wr = wm.numpy(), ref["weight_map"]
        assert np.array_equal(np.isnan(w), np.isnan(wr)) and bits_equal(np.nan_to_num(w), np.nan_to_num(wr))


def test_n3mr_functional_api_on_device_and_its_backward():
    """N3F:240-256 (permute, vertical flip, 2x2 mean pool) as HIP kernels: values against NumPy on the raw maps,
    the backward of the whole call against the oracle's gradients pushed through the same transform."""
    from oracle import N3mrOracle
    from jrender_amd.renderer.dr.n3mr import RasterizeRGBAD
    from jrender_amd import _ffi
    o = N3mrOracle()
    faces, tex = _scene(280, 2, 2, 7)
    rng = np.random.default_rng(8)
    for aa in (False, True):
        IS = 32
        S = IS * (2 if aa else 1)
        op = RasterizeRGBAD(IS, aa, 0.1, 100, 1e-3, (0.2, 0.3, 0.4), True, True, True)
        out = op(faces, tex)
        assert all(isinstance(out[k], _ffi.DeviceArray) for k in ("rgb", "alpha", "depth"))
        raw_rgb, raw_a, raw_d = op.fn.save_vars[5].numpy(), op.fn.save_vars[6].numpy(), op.fn.save_vars[4].numpy()

        def tr(x):                                        # [B,S,S,C] bottom-up -> [B,C,IS,IS]
            x = x.transpose(0, 3, 1, 2)[:, :, ::-1, :]
            return x.reshape(x.shape[0], x.shape[1], IS, S // IS, IS, S // IS).mean((3, 5)) if aa else x
        assert np.allclose(out["rgb"].numpy(), tr(raw_rgb), atol=1e-6) and out["rgb"].shape == (2, 3, IS, IS)
        a = out["alpha"].numpy()
        assert a.shape == ((2, 1, IS, IS) if aa else (2, IS, IS))
        assert np.allclose(a.reshape(2, 1, IS, IS), tr(raw_a[..., None]), atol=1e-6)
        assert np.allclose(out["depth"].numpy().reshape(2, 1, IS, IS), tr(raw_d[..., None]), atol=1e-5)
        g_rgb = rng.uniform(-1, 1, (2, 3, IS, IS)).astype(np.float32)
        g_a = rng.uniform(-1, 1, a.shape).astype(np.float32)
        g_d = rng.uniform(-1, 1, a.shape).astype(np.float32)
        gf, gt = op.backward(g_rgb, g_a, g_d)

        def trb(g):                                       # adjoint of tr: [B,C,IS,IS] -> [B,S,S,C] bottom-up
            g = g.reshape(2, -1, IS, IS)
            if aa:
                g = np.repeat(np.repeat(g, 2, axis=2), 2, axis=3) / 4
            return np.ascontiguousarray(g[:, :, ::-1, :].transpose(0, 2, 3, 1))
        ref = o.forward(faces, tex, image_size=S, background_color=(0.2, 0.3, 0.4))
        rgf, rgt = o.backward(ref, trb(g_rgb), trb(g_a)[..., 0], trb(g_d)[..., 0])
        assert grad_err(gf.numpy().reshape(rgf.shape), rgf) <= 1e-4 and grad_err(gt.numpy(), rgt) <= 1e-4


def test_n3mr_renderer_gradients_textures_and_vertices():
    """Renderer(dr_type='n3mr') end to end on the device with gradients (demo4-optim_textures' chain):
    d/d textures is exact (the image is linear in the texels) -> finite differences; d/d vertices is NMR's
    approximate gradient -> the oracle's gradients pushed through the same host chain."""
    from oracle import N3mrOracle
    from jrender_amd.structures.mesh import face_vertices_backward
    v, f = jr.synthetic.sphere_mesh(280)
    v = (v * 0.6).astype(np.float32)
    rng = np.random.default_rng(3)
    tex = rng.uniform(0.2, 0.8, (1, f.shape[0], 2, 2, 2, 3)).astype(np.float32)
    r = jr.Renderer(image_size=24, camera_mode='look_at', light_intensity_directionals=0.3, light_intensity_ambient=0.6,
                    dr_type='n3mr', anti_aliasing=True)
    r.transform.set_eyes_from_angles(2.732, 20.0, 40.0)

    def render(t):
        return r.render_mesh(jr.Mesh(v, f, textures=t.copy(), dr_type='n3mr'), mode='rgb')
    img = render(tex)
    assert img.shape == (1, 3, 24, 24)
    G = rng.uniform(-1, 1, img.shape).astype(np.float32)
    gt = r.grad_textures(G)
    assert gt.shape == tex.shape
    base = float((img.numpy().astype(np.float64) * G).sum())
    big = np.argsort(-np.abs(gt).reshape(-1))[:4]
    for i in big:
        t2 = tex.copy().reshape(-1)
        t2[i] += 0.05
        fd = (float((render(t2.reshape(tex.shape)).numpy().astype(np.float64) * G).sum()) - base) / 0.05
        assert np.isclose(gt.reshape(-1)[i], fd, rtol=2e-3, atol=1e-5), (i, gt.reshape(-1)[i], fd)
    # vertices: same forward, then the oracle's backward on the rasteriser's own inputs
    render(tex)
    gv = r.grad_vertices(grad_rgb=G)
    op = r.rasterizer._op
    fv_in, tex_in = op.fn.save_vars[0].numpy(), op.fn.save_vars[1].numpy()
    o = N3mrOracle()
    S = 48
    ref = o.forward(fv_in, tex_in, image_size=S, near=r.rasterizer.near, far=r.rasterizer.far, eps=r.rasterizer.rasterizer_eps,
                    background_color=r.rasterizer.background_color, return_rgb=True, return_alpha=False, return_depth=False)
    g = np.repeat(np.repeat(G, 2, axis=2), 2, axis=3) / 4
    rgf, _ = o.backward(ref, np.ascontiguousarray(g[:, :, ::-1, :].transpose(0, 2, 3, 1)), None, None)
    nf = f.shape[0]
    rgf = rgf[:, :nf] + rgf[:, nf:, ::-1]
    want = r.transform.transformer.backward(face_vertices_backward(rgf, f[None], v.shape[0]), v[None])
    assert grad_err(gv, want) <= 1e-4


def test_n3mr_silhouette_and_depth_modes_use_the_module_defaults():
    """the reference's render_silhouettes / render_depth call rasterize_silhouettes / rasterize_depth
    with (faces, image_size, anti_aliasing) only (N3R:61-80), i.e. near=0.1, far=100, eps=1e-4 whatever the
    rasterizer was built with.  A rasterizer with OTHER near / far must give the oracle's default-parameter maps,
    and its silhouette gradient must be the oracle's at eps=1e-4 (eps enters the backward as dist +/- eps)."""
    from oracle import N3mrOracle
    from jrender_amd.renderer.dr.n3mr.rasterizer import N3mrRasterizer
    from jrender_amd.structures.mesh import face_vertices as v2f
    v, f = jr.synthetic.sphere_mesh(280)
    eye = np.asarray(jr.get_points_from_angles(2.732, 25., 30.), np.float32)
    ndc = jr.perspective(jr.look_at((v * 0.7).astype(np.float32)[None], eye), 30.)
    S = 32
    ras = N3mrRasterizer(image_size=S, anti_aliasing=False, near=2.4, far=2.9, fill_back=True)   # would clip half the sphere
    alpha = ras.render_silhouettes(ndc, f[None])
    ff = np.concatenate([f, f[:, ::-1]])
    faces = np.ascontiguousarray(v2f(ndc, ff[None]))
    o = N3mrOracle()
    ref = o.forward(faces, None, image_size=S, near=0.1, far=100, eps=1e-4, return_rgb=False, return_alpha=True, return_depth=False)
    a = alpha.numpy() if hasattr(alpha, "numpy") else np.asarray(alpha)
    assert np.array_equal(a.reshape(S, S), ref["alpha_map"].reshape(S, S)[::-1])       # functional API flips rows (N3F:243-247)
    G = np.random.default_rng(5).uniform(-1, 1, a.shape).astype(np.float32)
    gf, _ = ras.backward(grad_silhouettes=G)
    ga = np.ascontiguousarray(G.reshape(1, S, S)[:, ::-1])
    rgf, _ = o.backward(ref, None, ga, None)
    assert grad_err(gf.numpy().reshape(rgf.shape), rgf) <= 1e-4
    # eps = 1e-3 (the rasterizer's own) gives a measurably different gradient: the test can tell the two apart
    ref3 = o.forward(faces, None, image_size=S, near=0.1, far=100, eps=1e-3, return_rgb=False, return_alpha=True, return_depth=False)
    rgf3, _ = o.backward(ref3, None, ga, None)
    assert grad_err(rgf3, rgf) > 1e-3
    depth = ras.render_depth(ndc, f[None])
    d = depth.numpy() if hasattr(depth, "numpy") else np.asarray(depth)
    refd = o.forward(faces, None, image_size=S, near=0.1, far=100, eps=1e-4, return_rgb=False, return_alpha=False, return_depth=True)
    assert np.array_equal(d.reshape(S, S), refd["depth_map"].reshape(S, S)[::-1])


def test_n3mr_zbuffer_keys_are_reusable_across_forwards_of_one_context():
    """k_n3mr_resolve clears the z-buffer keys it read, so that a context's next forward needs no memset launch: smaller, equal and
    larger images in turn on ONE context (the larger one regrows the key array) must each match the oracle bit for bit."""
    from oracle import N3mrOracle
    o = N3mrOracle("port")
    for nf, batch, IS, seed in ((280, 2, 96, 1), (280, 1, 48, 2), (3300, 2, 96, 3), (280, 2, 160, 4), (280, 1, 96, 5), (3300, 2, 160, 6)):
        faces, tex = _scene(nf, batch, 2, seed, az0=10.0 * seed)
        ref = o.forward(faces, tex, image_size=IS)
        fn = RasterizeFunction(IS, 0.1, 100, 1e-3, (0, 0, 0), True, True, True)
        fn(faces, tex)
        f, t, fim, wm, dm, rgb, alpha, fivm, sidx, swt = fn.save_vars
        assert bits_equal(fim.numpy(), ref["face_index_map"]), (nf, batch, IS)
        assert bits_equal(dm.numpy(), ref["depth_map"]) and bits_equal(alpha.numpy(), ref["alpha_map"])
