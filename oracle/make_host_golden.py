#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — executes the reference's OWN host-side Python (transforms, lighting, losses) from
/root/reference under the NumPy-backed `jittor` shim (oracle/jittor_shim) and writes inputs + outputs to
tests/golden/host_ref.npz.  Nothing is copied: the files are loaded by path where they lie.

    python oracle/make_host_golden.py            # regenerate the fixture (needs /root/reference)

`tests/test_host_reference.py` compares the jrender_amd mirrors with the fixture (runs anywhere), and — where
/root/reference is mounted — re-executes this script's cases and checks that the committed fixture is what the
reference produces today.

Reference files executed:
  renderer/transform/look_at.py, perspective.py, look.py, orthogonal.py, projection.py
  renderer/utils/get_points_from_angles.py
  renderer/lighting/ambient_lighting.py, directional_lighting.py, lighting.py (Lighting / AmbientLighting /
  DirectionalLighting / the legacy `lighting()`; its SSS branch needs the rasteriser and is not called)
  loss/iou_loss.py, laplacian_loss.py, flatten_loss.py
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("JRENDER_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "..", "tests", "golden", "host_ref.npz")
F32 = np.float32


def have_reference():
    return os.path.exists(os.path.join(REF_ROOT, "jrender", "renderer", "transform", "look_at.py"))


class _Loaded:
    pass


def load_reference_host():
    """-> namespace of the reference's host functions / classes, executing under the shim.  sys.modules is
    restored afterwards (the shim must not leak into the test process as `jittor`)."""
    saved = {k: v for k, v in sys.modules.items() if k == "jittor" or k.startswith(("jittor.", "jrender", "skimage"))}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, os.path.join(HERE, "jittor_shim"))
    ns = _Loaded()
    try:
        jt = importlib.import_module("jittor")
        ns.jt = jt

        def load(name, rel):
            spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, "jrender", rel))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[name] = mod
            spec.loader.exec_module(mod)
            return mod

        def package(name):
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
            return m

        def stub(name, **attrs):
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
            return m

        for rel in ("look_at", "perspective", "look", "orthogonal", "projection"):
            setattr(ns, rel, getattr(load("_ref_host_" + rel, "renderer/transform/%s.py" % rel), rel))
        ns.get_points_from_angles = load("_ref_host_gpfa", "renderer/utils/get_points_from_angles.py").get_points_from_angles
        ns.neg_iou_loss = load("_ref_host_iou", "loss/iou_loss.py").neg_iou_loss
        ns.LaplacianLoss = load("_ref_host_lap", "loss/laplacian_loss.py").LaplacianLoss
        ns.FlattenLoss = load("_ref_host_flat", "loss/flatten_loss.py").FlattenLoss
        # lighting.py imports its siblings with `from . import *` and a few modules its SSS branch needs: give it the
        # package it expects, with the two sibling files executed for real and the rest stubbed (never called here)
        for p in ("jrender", "jrender.renderer", "jrender.renderer.lighting", "jrender.renderer.dr",
                  "jrender.renderer.dr.softras", "jrender.io", "jrender.io.utils", "jrender.renderer.utils", "skimage"):
            package(p)
        stub("jrender.renderer.dr.softras.soft_rasterize", SoftRasterizeFunction=None)
        stub("jrender.io.utils.load_textures", _load_textures_for_softras=None)
        stub("jrender.renderer.utils.gaussian_blur", gaussian_blur=None)
        stub("jrender.renderer.utils.ToStretchMap", computeStretchMap=None)
        stub("skimage.io", imsave=None)
        pkg = sys.modules["jrender.renderer.lighting"]
        amb = load("jrender.renderer.lighting.ambient_lighting", "renderer/lighting/ambient_lighting.py")
        dire = load("jrender.renderer.lighting.directional_lighting", "renderer/lighting/directional_lighting.py")
        for m in (amb, dire):
            for k, v in m.__dict__.items():
                if not k.startswith("_"):
                    setattr(pkg, k, v)
        lig = load("jrender.renderer.lighting.lighting", "renderer/lighting/lighting.py")
        ns.ambient_lighting, ns.directional_lighting = amb.ambient_lighting, dire.directional_lighting
        ns.Lighting, ns.lighting_legacy = lig.Lighting, lig.lighting
    finally:
        sys.path.remove(os.path.join(HERE, "jittor_shim"))
        for k in [k for k in sys.modules if k == "jittor" or k.startswith(("jittor.", "jrender.", "skimage", "_ref_host_")) or k == "jrender"]:
            del sys.modules[k]
        sys.modules.update(saved)
    return ns


def _sphere(seg=14, rings=11):
    """the 280-face UV sphere of jrender_amd.synthetic, restated so that this script has no product imports"""
    lat = np.pi * (np.arange(1, rings) / rings)
    lon = 2 * np.pi * (np.arange(seg) / seg)
    ring = np.stack([np.outer(np.sin(lat), np.cos(lon)), np.repeat(np.cos(lat)[:, None], seg, 1),
                     np.outer(np.sin(lat), np.sin(lon))], axis=-1).reshape(-1, 3)
    verts = np.concatenate([[[0, 1, 0]], ring, [[0, -1, 0]]])
    idx = lambda r, s: 1 + r * seg + (s % seg)
    faces = [(0, idx(0, s + 1), idx(0, s)) for s in range(seg)]
    for r in range(rings - 2):
        for s in range(seg):
            a, b, c, d = idx(r, s), idx(r, s + 1), idx(r + 1, s), idx(r + 1, s + 1)
            faces += [(a, b, d), (a, d, c)]
    faces += [(verts.shape[0] - 1, idx(rings - 2, s), idx(rings - 2, s + 1)) for s in range(seg)]
    return verts.astype(F32), np.asarray(faces, np.int32)


def _unit(v):
    return (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(F32)


def cases(ref):
    """name -> array: inputs (`in_*`) and the reference's outputs (`out_*`)."""
    jt = ref.jt
    V = lambda a: jt.array(np.array(a))
    npy = lambda v: np.array(v.data if hasattr(v, "data") and not isinstance(v, np.ndarray) else v)
    rng = np.random.default_rng(2024)
    d = {}
    # ---- transforms ----
    v = (rng.normal(size=(2, 40, 3)) * 0.4).astype(F32)
    eyes = np.array([[0.3, 1.2, -2.5], [-1.0, 0.4, -2.0]], F32)
    d["in_vertices"], d["in_eyes"] = v, eyes
    d["out_look_at_batch"] = npy(ref.look_at(V(v), V(eyes)))
    d["out_look_at_tuple"] = npy(ref.look_at(V(v), (0.0, 1.0, -2.732)))
    d["out_look_at_at_up"] = npy(ref.look_at(V(v), V(eyes), at=[0.1, -0.2, 0.05], up=[0.0, 0.0, 1.0]))
    cam = ref.look_at(V(v), V(eyes))
    d["out_perspective_30"] = npy(ref.perspective(cam, 30.))
    d["out_perspective_47"] = npy(ref.perspective(cam, 47.5))
    d["out_look_right"] = npy(ref.look(V(v), V(eyes[0]), direction=[0.2, -0.1, 1.0], up=[0, 1, 0]))
    d["out_look_left"] = npy(ref.look(V(v), V(eyes[0]), direction=[0.2, -0.1, 1.0], up=[0, 1, 0], coordinate="left"))
    d["out_orthogonal"] = npy(ref.orthogonal(cam, 0.7))
    K = np.array([[[300., 0, 128.], [0, 310., 120.], [0, 0, 1.]]], F32)
    th = 0.3
    R = np.array([[[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]]], F32)
    t = np.array([[[0.1, -0.05, 3.0]]], F32)
    dist = np.array([[0.05, -0.02, 0.001, 0.002, 0.01]], F32)
    d["in_K"], d["in_R"], d["in_t"], d["in_dist"] = K, R, t, dist
    d["out_projection"] = npy(ref.projection(V(v), V(K), V(R), V(t), V(dist), 256))
    d["out_points_scalar"] = np.asarray(ref.get_points_from_angles(2.732, 30., 55.), np.float64)
    dist3, el3, az3 = np.array([2.0, 2.732, 3.5], F32), np.array([10., 30., -45.], F32), np.array([0., 55., 200.], F32)
    d["in_angles"] = np.stack([dist3, el3, az3])
    d["out_points_array"] = npy(ref.get_points_from_angles(V(dist3), V(el3), V(az3)))
    # ---- losses ----
    pred, targ = rng.uniform(0, 1, (4, 32, 32)).astype(F32), (rng.uniform(0, 1, (4, 32, 32)) > 0.5).astype(F32)
    d["in_pred"], d["in_target"] = pred, targ
    d["out_neg_iou"] = npy(ref.neg_iou_loss(V(pred), V(targ)))
    sv, sf = _sphere()
    xs = (sv[None] * np.array([[[1.0]], [[0.8]]], F32) + rng.normal(size=(2,) + sv.shape).astype(F32) * 0.03).astype(F32)
    d["in_sphere_v"], d["in_sphere_f"], d["in_loss_x"] = sv, sf, xs
    lap = ref.LaplacianLoss(V(sv), V(sf), average=False)
    d["out_laplacian"] = npy(lap(V(xs)))
    d["out_laplacian_avg"] = npy(ref.LaplacianLoss(V(sv), V(sf), average=True)(V(xs)))
    d["out_laplacian_matrix"] = npy(lap.laplacian)
    fl = ref.FlattenLoss(V(sf), average=False)
    d["out_flatten"] = npy(fl(V(xs)))
    d["out_flatten_avg"] = npy(ref.FlattenLoss(V(sf), average=True)(V(xs)))
    d["out_flatten_quads"] = np.stack([npy(fl.v0s), npy(fl.v1s), npy(fl.v2s), npy(fl.v3s)])
    # ---- lighting ----
    B, N = 2, 60
    normals = _unit(rng.normal(size=(B, N, 3)))
    positions = (rng.normal(size=(B, N, 3)) * 0.5).astype(F32)
    eye = np.array([[0.0, 0.5, -2.7], [1.0, 0.2, -2.2]], F32)
    metallic = rng.uniform(0.1, 0.9, (B, N, 4, 1)).astype(F32)
    rough = rng.uniform(0.2, 0.9, (B, N, 4, 1)).astype(F32)
    d["in_normals"], d["in_positions"], d["in_eye"], d["in_metallic"], d["in_roughness"] = normals, positions, eye, metallic, rough
    d["out_ambient"] = npy(ref.ambient_lighting(jt.zeros((B, N, 3)), 0.4, (1.0, 0.9, 0.8)))
    dl, sl = ref.directional_lighting(jt.zeros((B, N, 3)), jt.zeros((B, N, 3)), V(normals), 0.6, (1.0, 0.8, 0.7), (0.3, 1.0, -0.4),
                                      V(positions), eye, False, None, None)
    d["out_lambert_diffuse"], d["out_lambert_specular"] = npy(dl), npy(sl)
    dl, sl = ref.directional_lighting(jt.zeros((B, N, 3)), jt.zeros((B, N, 3)), V(normals), 0.6, (1.0, 0.8, 0.7), (0.3, 1.0, -0.4),
                                      V(positions), eye, True, V(metallic), V(rough))
    d["out_ct_diffuse"], d["out_ct_specular"] = npy(dl), npy(sl)

    # Lighting module on a mesh-like object (the fields Lighting.execute reads, lighting.py:177-223)
    fv = (sv[sf][None] * np.array([[[[1.0]]], [[[0.9]]]], F32)).astype(F32)                  # [2,NF,3,3]
    v10, v12 = fv[:, :, 0] - fv[:, :, 1], fv[:, :, 2] - fv[:, :, 1]
    snorm = _unit(np.cross(v12, v10))                                                       # any unit normals will do:
    vnorm = _unit(np.broadcast_to(sv[None], (2,) + sv.shape) + 0.1)                          # the lighting is what is pinned
    tex = rng.uniform(0.05, 1.4, (2, sf.shape[0], 4, 3)).astype(F32)                         # > 1 exercises the clamp
    met = rng.uniform(0.1, 0.9, (2, sf.shape[0], 4, 1)).astype(F32)
    rou = rng.uniform(0.2, 0.9, (2, sf.shape[0], 4, 1)).astype(F32)
    d["in_mesh_fv"], d["in_mesh_snorm"], d["in_mesh_vnorm"], d["in_mesh_tex"] = fv, snorm, vnorm, tex
    d["in_mesh_metallic"], d["in_mesh_roughness"] = met, rou
    verts2 = np.broadcast_to(sv[None], (2,) + sv.shape).astype(F32)
    vtex = rng.uniform(0.05, 1.2, (2, sv.shape[0], 4, 3)).astype(F32)                        # 4-D "vertex" textures (lighting.py:213)
    d["in_mesh_vtex"] = vtex

    def mesh(textures, specular):
        m = types.SimpleNamespace()
        m.textures, m.normal_textures, m.with_SSS = V(textures), None, False
        m.faces = V(np.broadcast_to(sf[None], (2,) + sf.shape).astype(np.int32))
        m.face_vertices, m.surface_normals = V(fv), V(snorm)
        m.vertices, m.vertex_normals = V(verts2), V(vnorm)
        m.with_specular = specular
        m.metallic_textures, m.roughness_textures = (V(met), V(rou)) if specular else (None, None)
        return m
    L = ref.Lighting('surface', 0.45, [1, 0.95, 0.9], 0.55, [0.9, 1, 1], [0.2, 1.0, -0.3])
    d["out_lighting_surface"] = npy(L(mesh(tex, False), eye).textures)
    d["out_lighting_surface_specular"] = npy(L(mesh(tex, True), eye).textures)
    Lv = ref.Lighting('vertex', 0.45, [1, 0.95, 0.9], 0.55, [0.9, 1, 1], [0.2, 1.0, -0.3])
    d["out_lighting_vertex"] = npy(Lv(mesh(vtex, False), eye).textures)
    # the legacy functional lighting (lighting.py:14-54): faces [B,NF,3,3], cube textures [B,NF,t,t,t,3]
    ctex = rng.uniform(0, 1, (2, sf.shape[0], 2, 2, 2, 3)).astype(F32)
    d["in_cube_tex"] = ctex
    d["out_lighting_legacy"] = npy(ref.lighting_legacy(V(fv), V(ctex.copy()), 0.5, 0.5, (1, 1, 1), (1, 0.9, 0.8), (0, 1, 0)))
    return {k: np.asarray(a) for k, a in d.items()}


def main():
    if not have_reference():
        raise SystemExit("reference not mounted at %s" % REF_ROOT)
    d = cases(load_reference_host())
    np.savez_compressed(OUT, **d)
    print("wrote %s: %d arrays, %.1f KB" % (os.path.normpath(OUT), len(d), os.path.getsize(OUT) / 1024))


if __name__ == "__main__":
    main()
