"""SURVEY §8(f) widened: the reference's other optimisation loops on this path -

  demo4-optim_textures.py:24-76            NMR renderer, cube textures through tanh, Adam        -> examples/demo4_optim_textures.py
  demo5-optim_metallic_textures.py:21-44   SoftRas renderer, Cook-Torrance lighting, metallic    -> examples/demo5_optim_material.py --param metallic
  demo6-optim_roughness_textures.py        the same with roughness                               -> ... --param roughness

each with: the loss falls on a self-contained synthetic target, and the gradient the loop uses (rasteriser backward on the GPU ->
lighting VJP, renderer/lighting/directional_lighting.py:86-130) against central differences of the rendered loss.  The lighting
forward itself is pinned to the reference's files by tests/test_host_reference.py (fixture out_ct_diffuse / out_ct_specular made from
the reference's directional_lighting.py through oracle/jittor_shim); the float64 VJP checks are tests/test_host.py."""
import importlib.util
import os

import numpy as np
import pytest

import jrender_amd as jr

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_example(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "examples", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("host_adam", [False, True])
def test_demo4_texture_optimisation_loss_falls(host_adam):
    demo4 = load_example("demo4_optim_textures")
    hist = demo4.main(["--iters", "60", "--image-size", "64", "--views", "6", "--quiet"] + (["--host-adam"] if host_adam else []))
    assert len(hist) == 60 and np.isfinite(hist).all()
    first, last = float(np.mean(hist[:6])), float(np.mean(hist[-6:]))       # (every iteration draws another view: compare windows)
    assert last < 0.6 * first, (first, last)
    t = demo4.main.textures
    assert t.shape[2:] == (4, 4, 4, 3) and np.abs(t).max() <= 1.0 and t.std() > 0.02     # tanh range; the faces took different colours


def test_demo4_device_and_host_adam_follow_the_same_curve():
    demo4 = load_example("demo4_optim_textures")
    a = demo4.main(["--iters", "12", "--image-size", "48", "--views", "4", "--quiet"])
    b = demo4.main(["--iters", "12", "--image-size", "48", "--views", "4", "--quiet", "--host-adam"])
    assert np.allclose(a, b, rtol=2e-3), (a, b)


@pytest.mark.parametrize("param,drop", [("metallic", 0.6), ("roughness", 0.85)])
def test_demo5_demo6_material_optimisation_loss_falls(param, drop):
    demo5 = load_example("demo5_optim_material")
    hist = demo5.main(["--param", param, "--image-size", "96", "--quiet"])
    assert len(hist) == (20 if param == "metallic" else 15) and np.isfinite(hist).all()
    assert hist[-1] < drop * hist[0], hist
    p = demo5.main.param
    assert p.shape[2:] == (16, 1) and p.std() > 1e-3                          # the map left its constant start


@pytest.mark.parametrize("param", ["metallic", "roughness"])
def test_material_gradient_against_central_differences_through_the_renderer(param):
    """Renderer.grad_material (jr_softras_backward -> fold -> Lighting.backward_material) against central differences of the
    rendered loss sum((image - ref)^2), for the faces with the largest gradient.  The rasteriser is linear in the lit textures and
    the lighting step smooth in the material away from the clip: a 0.02 step resolves the derivative to a few per cent in float32."""
    v, f = jr.synthetic.uv_sphere(16, 9)
    v, f = np.asarray(v, np.float32)[None], np.asarray(f, np.int32)[None]
    nf, T = f.shape[1], 4
    rng = np.random.default_rng(5)
    tex = rng.uniform(0.2, 0.6, (1, nf, T, 3)).astype(np.float32)
    M = rng.uniform(0.2, 0.7, (1, nf, 1, 1)).astype(np.float32) * np.ones((1, 1, T, 1), np.float32)
    R = rng.uniform(0.35, 0.8, (1, nf, 1, 1)).astype(np.float32) * np.ones((1, 1, T, 1), np.float32)
    r = jr.Renderer(image_size=64, dr_type='softras', light_intensity_directionals=0.9, light_intensity_ambient=0.2,
                    light_directions=[0.3, 1.0, -0.6])
    ref = rng.uniform(0, 1, (1, 3, 64, 64)).astype(np.float32)           # Renderer.__call__ renders mode='rgb': three channels

    def loss(m, rr):
        r.transform.set_eyes_from_angles(2.732, 30, 140)
        img = r(v, f, tex.copy(), metallic_textures=m, roughness_textures=rr).numpy()
        return float(((img.astype(np.float64) - ref) ** 2).sum()), img
    _, img = loss(M, R)
    gm, gr = r.grad_material(2.0 * (img - ref))
    g = gm if param == "metallic" else gr
    assert g.shape == M.shape and np.isfinite(g).all()
    per_face = g.sum(axis=(2, 3))[0]
    faces = np.argsort(-np.abs(per_face))[:4]
    assert np.abs(per_face[faces]).min() > 0
    h = 0.02
    for fi in faces:
        a, b = (M if param == "metallic" else R).copy(), (M if param == "metallic" else R).copy()
        a[0, fi] += h; b[0, fi] -= h
        fd = (loss(a, R)[0] - loss(b, R)[0]) / (2 * h) if param == "metallic" else (loss(M, a)[0] - loss(M, b)[0]) / (2 * h)
        assert np.isclose(per_face[fi], fd, rtol=0.05, atol=1e-3 * np.abs(per_face).max()), (param, fi, per_face[fi], fd)
