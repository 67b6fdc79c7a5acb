"""End-to-end through the preserved API surface on the GPU: jr.Renderer(dr_type='softras')
.render_mesh(mesh) == oracle on the same lit / transformed mesh; config-2-like case (T=25 surface
textures, 1024x1024, batch 1) checked at random pixels; a few steps of the demo2-style silhouette
fitting loop (IoU loss, hand-written backward chain) must reduce the loss."""
import numpy as np
import pytest

import jrender_amd as jr
from oracle import Oracle
from tests.util import RGBA_ATOL, bits_equal, grad_err, rel_err

pytestmark = pytest.mark.gpu


def _mesh(nf=280, batch=1, texels=1, seed=0):
    v, f = jr.synthetic.sphere_mesh(nf)
    tex = jr.synthetic.face_colors(nf, texels, seed)
    return jr.Mesh(np.broadcast_to(v[None], (batch,) + v.shape).copy(), f, textures=tex[None].repeat(batch, 0))


def test_renderer_matches_oracle_on_lit_mesh():
    port = Oracle("port", nthreads=0)
    r = jr.Renderer(image_size=96, dr_type='softras')
    r.transform.set_eyes_from_angles(2.732, 30., 20.)
    m = _mesh(280, 1, texels=4)
    rgb = r.render_mesh(m, mode='rgb').numpy()
    # render_mesh mutated the mesh (lighting, transform) exactly like the reference: m now holds NDC
    ref = port.forward(m.face_vertices, m.face_textures, image_size=96, fill_back=True)
    assert rgb.shape == (1, 3, 96, 96)
    assert rel_err(rgb, ref["soft_colors"][:, :3], RGBA_ATOL) <= 1.0
    m.reset_()
    sil = r.render_mesh(m, mode='silhouettes').numpy()
    assert rel_err(sil, ref["soft_colors"][:, 3], RGBA_ATOL) <= 1.0
    assert 0.2 < sil.mean() < 0.6


def test_config2_like_textured_1024_random_pixels():
    port = Oracle("port", nthreads=0)
    IS, K = 1024, 16
    fv, tex = jr.synthetic.sphere_views(3300, 1, texels=25)
    fn = jr.SoftRasterizeFunction(image_size=IS)
    rgba = fn(fv, tex).numpy()
    ids = fn.save_vars[5].numpy()
    rng = np.random.default_rng(0)
    touched = np.flatnonzero(ids[0, 0].reshape(-1) >= 0)
    pix = np.unique(np.concatenate([rng.choice(touched, 3000, replace=False), rng.choice(IS * IS, 1000)]))
    sub = port.forward_subset(fv, tex, pix, image_size=IS)
    assert bits_equal(ids.reshape(1, K, -1)[0][:, pix].T, sub["ids"])
    assert rel_err(rgba.reshape(4, -1)[:, pix].T, sub["rgba"], RGBA_ATOL) <= 1.0
    g = np.zeros((1, 4, IS, IS), np.float32)
    g.reshape(4, -1)[:, pix] = rng.uniform(-1, 1, (4, len(pix)))
    gf, gt = fn.grad(g)
    s = dict(face_vertices=fv.reshape(1, -1, 9), textures=tex, soft_colors=rgba, faces_info=fn.save_vars[3].numpy(),
             aggrs_info=fn.save_vars[4].numpy(), faces_id_buffer=ids, params=dict(image_size=IS))
    gfo, gto = port.backward_subset(s, g, pix)
    assert grad_err(gf.numpy().reshape(gfo.shape), gfo) <= 1e-4
    assert grad_err(gt.numpy(), gto) <= 1e-4


def test_demo2_style_silhouette_fitting_reduces_loss():
    # target: silhouettes of a sphere of radius 0.8 from 4 cameras; start: radius 1.0; optimise a
    # per-vertex radial scale with plain gradient descent through rasteriser -> transform -> gather
    v, f = jr.synthetic.uv_sphere(24, 16)
    B, IS = 4, 64
    eyes = jr.get_points_from_angles(np.full(B, 2.732, np.float32), np.full(B, 30., np.float32),
                                     np.arange(B, dtype=np.float32) * 90)
    la = jr.LookAt(True, 30, 1.0, eye=eyes)
    ras = jr.SoftRasterizer(image_size=IS, sigma_val=1e-4, aggr_func_rgb='hard', fill_back=True)

    def render(scale):
        verts = np.broadcast_to((v * scale[:, None])[None], (B,) + v.shape).astype(np.float32)
        m = jr.Mesh(la(verts), f)
        return verts, m, ras(m, 'silhouettes').numpy()
    _, _, target = render(np.full(v.shape[0], 0.8, np.float32))
    scale = np.ones(v.shape[0], np.float32)
    losses = []
    for it in range(12):
        verts, m, sil = render(scale)
        losses.append(float(jr.neg_iou_loss(sil, target)))
        gsil = jr.neg_iou_loss_backward(sil, target)
        gfv, _ = ras.backward(grad_silhouettes=gsil)
        from jrender_amd.structures.mesh import face_vertices_backward
        gndc = face_vertices_backward(gfv.numpy().reshape(B, -1, 3, 3), m.faces, v.shape[0])
        gworld = la.backward(gndc, verts)
        gscale = (gworld * v[None]).sum((0, 2))
        scale = scale - 0.5 * gscale / (np.abs(gscale).max() + 1e-12) * 0.05
    assert losses[-1] < losses[0] * 0.7, losses


def test_demo2_deform_example_reduces_loss():
    """examples/demo2_deform.py (reference demo2-deform.py: sigmoid/tanh vertex parametrisation, neg-IoU +
    Laplacian + flatten, Adam) on a synthetic target: the loss must fall and the mesh stay finite."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "demo2_deform.py")
    spec = importlib.util.spec_from_file_location("demo2_deform", path)
    demo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo)
    hist = demo.main(["--iters", "60", "--batch-size", "8", "--quiet"])
    assert np.isfinite(hist).all()
    assert hist[-1] < 0.8 * hist[0], (hist[0], hist[-1])


def test_gbuffer_rasterizer_matches_oracle_modes():
    """render2's caller of the operator: vertex attributes, barycentric distance, hard aggregation (+MSAA)."""
    from oracle import Oracle
    port = Oracle("port", nthreads=0)
    fv, _ = jr.synthetic.sphere_views(280, 1)
    attr = np.random.default_rng(5).uniform(0, 1, fv.shape[1:]).astype(np.float32)      # [NF,3,3] per-vertex attribute
    gb = jr.GBufferRasterizer(image_size=48, near=1, far=100, fill_back=True)
    img = gb.Rasterize(fv[0], attr)
    ref = port.forward(fv, attr[None], image_size=48, near=1, far=100, texture_type="vertex",
                       dist_func="barycentric", aggr_func_rgb="hard")
    assert img.shape == (48, 48, 3)
    assert np.allclose(img, ref["soft_colors"][0, :3].transpose(1, 2, 0), rtol=1e-4, atol=1e-6)
    img2 = gb.Rasterize(fv[0], attr, MSAA=True)
    ref2 = port.forward(fv, attr[None], image_size=96, near=1, far=100, texture_type="vertex",
                        dist_func="barycentric", aggr_func_rgb="hard")["soft_colors"][0, :3]
    pooled = ref2.reshape(3, 48, 2, 48, 2).mean((2, 4)).transpose(1, 2, 0)
    assert np.allclose(img2, pooled, rtol=1e-4, atol=1e-6)
    depth = gb.Rasterize_depth(fv[0])
    refd = port.forward(fv, np.ones_like(fv[0])[None], image_size=48, near=1, far=100, texture_type="vertex",
                        dist_func="hard", aggr_func_rgb="hard")["aggrs_info"][0, 0]
    assert np.allclose(depth, refd, rtol=1e-6)


def test_bench_line_contract():
    """bench.py prints ONE JSON line with the driver's contract fields, the roofline and the CPU baseline."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--faces", "280",
                          "--image-size", "64", "--batch", "2"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "images/s" and c["sample"]
    assert abs(d["value"] - 2 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
