"""Seeded synthetic SoftRas workloads — NumPy only, no file deps.

The reference's README quotes fwd+bwd times on 280 / 3.3k / 39k-face meshes that
do not ship with it; these generators are the stand-ins used by the tests and by
bench.py:

* ``uv_sphere(seg, rings)``  lat-long sphere with exactly ``2*seg*(rings-1)`` faces:
  280 = (14, 11), 3 300 = (50, 34), 39 000 = (150, 131).
* ``sphere_views(...)``      that sphere seen from B cameras on a turntable,
  already transformed to NDC with ``look_at`` + ``perspective`` (defaults of
  jrender's Renderer: distance 2.732, elevation 30 deg, viewing angle 30 deg).
* ``triangle_soup(...)``     random small triangles, rejection of slivers.
"""
import numpy as np

from .renderer.transform import get_points_from_angles, look_at, perspective

F32 = np.float32

SPHERE_SHAPES = {280: (14, 11), 3300: (50, 34), 39000: (150, 131)}


def uv_sphere(seg, rings, radius=1.0):
    """Vertices [nv,3] f32 and faces [nf,3] i32 of a lat-long sphere.

    ``rings`` latitude bands; the two polar bands are triangle fans, the others
    quads split in two => nf = 2*seg*(rings-1), nv = seg*(rings-1) + 2.
    """
    lat = np.pi * (np.arange(1, rings) / rings)                     # rings-1 interior circles
    lon = 2 * np.pi * (np.arange(seg) / seg)
    ring = np.stack([np.outer(np.sin(lat), np.cos(lon)),
                     np.repeat(np.cos(lat)[:, None], seg, 1),
                     np.outer(np.sin(lat), np.sin(lon))], axis=-1).reshape(-1, 3)
    verts = np.concatenate([[[0, 1, 0]], ring, [[0, -1, 0]]]) * radius
    top, bottom = 0, verts.shape[0] - 1
    idx = lambda r, s: 1 + r * seg + (s % seg)
    faces = []
    for s in range(seg):
        faces.append((top, idx(0, s + 1), idx(0, s)))
    for r in range(rings - 2):
        for s in range(seg):
            a, b, c, d = idx(r, s), idx(r, s + 1), idx(r + 1, s), idx(r + 1, s + 1)
            faces.append((a, b, d))
            faces.append((a, d, c))
    for s in range(seg):
        faces.append((bottom, idx(rings - 2, s), idx(rings - 2, s + 1)))
    return verts.astype(F32), np.asarray(faces, np.int32)


def sphere_mesh(num_faces, radius=1.0):
    if num_faces not in SPHERE_SHAPES:
        raise ValueError("num_faces must be one of %s" % sorted(SPHERE_SHAPES))
    return uv_sphere(*SPHERE_SHAPES[num_faces], radius=radius)


def face_colors(num_faces, texels=1, seed=0):
    rng = np.random.default_rng(1000 + seed)
    return rng.uniform(0, 1, (num_faces, texels, 3)).astype(F32)


def sphere_views(num_faces=39000, batch=1, distance=2.732, elevation=30.0, radius=1.0,
                 viewing_angle=30.0, texels=1, seed=0, azimuth0=0.0):
    """-> face_vertices [B,NF,3,3] (NDC x,y; camera z), textures [B,NF,T,3]."""
    verts, faces = sphere_mesh(num_faces, radius)
    az = azimuth0 + 360.0 * np.arange(batch) / batch
    eyes = np.stack([np.asarray(get_points_from_angles(float(distance), float(elevation), float(a)), F32)
                     for a in az])
    v = np.broadcast_to(verts[None], (batch,) + verts.shape)
    ndc = perspective(look_at(v, eyes), angle=viewing_angle)
    fv = ndc[:, faces]                                               # [B,NF,3,3]
    tex = np.broadcast_to(face_colors(faces.shape[0], texels, seed)[None],
                          (batch, faces.shape[0], texels, 3))
    return np.ascontiguousarray(fv, F32), np.ascontiguousarray(tex, F32)


def triangle_soup(num_faces, batch=1, seed=0, texels=1, scale=1.2, zrange=(2.0, 4.0)):
    """Random triangles in NDC: centre U(-0.9,0.9)^2, offsets U(-r,r)^2 with
    r = scale*sqrt(2/NF), per-vertex z U(zrange); slivers (|det| < 1e-4 r^2,
    ill-conditioned for any fp32 rasteriser) are re-drawn."""
    rng = np.random.default_rng(seed)
    r = scale * np.sqrt(2.0 / num_faces)
    out = np.empty((batch, num_faces, 3, 3), F32)
    for b in range(batch):
        todo = np.arange(num_faces)
        while todo.size:
            c = rng.uniform(-0.9, 0.9, (todo.size, 1, 2))
            xy = c + rng.uniform(-r, r, (todo.size, 3, 2))
            z = rng.uniform(zrange[0], zrange[1], (todo.size, 3, 1))
            tri = np.concatenate([xy, z], -1).astype(F32)
            e1, e2 = tri[:, 1, :2] - tri[:, 0, :2], tri[:, 2, :2] - tri[:, 0, :2]
            det = e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]
            ok = np.abs(det) >= 1e-4 * r * r
            out[b, todo[ok]] = tri[ok]
            todo = todo[~ok]
    tex = rng.uniform(0, 1, (batch, num_faces, texels, 3)).astype(F32)
    return out, tex
