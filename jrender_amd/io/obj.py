"""OBJ / MTL / texture I/O — NumPy + Pillow mirror of jrender/io (both renderer variants).

load_obj             jrender/io/load_obj.py:9-22 -> utils/_load_obj_for_softras.py:142-207,
                     utils/_load_obj_for_n3mr.py:113-160 (dr_type='n3mr')
load_textures        utils/_load_obj_for_softras.py:41-140 (Kd colours, map_Kd images)
load_textures_n3mr   utils/_load_obj_for_n3mr.py:30-110 (cube textures [NF, ts, ts, ts, 3])
sample_textures      utils/load_textures.py:11-69: the reference's CUDA sampler restated in NumPy
                     (per-face R x R texels at barycentric sample points, bilinear fetch)
sample_textures_n3mr utils/load_textures.py:103-219: the NMR sampler (ts^3 barycentric lattice, four wrapping
                     modes, bilinear / nearest)
save_obj             geometry only (texture atlas export is out of scope)

Both samplers reproduce the reference's float / double promotions operation by operation and are
BIT-EXACT against the reference kernels compiled for the host (tests/test_textures.py, oracle/_ref).
Two quirks of the reference are pinned, not reproduced: (1) its sampler outputs are fresh buffers, so
texels of faces that no material image updates are UNINITIALISED there — here they keep their Kd colour;
(2) the NMR kernel rewrites the face's texture coordinates in place from every one of its ts^3 threads
(load_textures.py:151-172), which is idempotent except for a coordinate that is an exact integer under
REPEAT (1.0 -> 0.0 -> 1.0 ...: the result depends on the thread interleaving) — here the wrapping is
applied once.

Load-time only, not on the hot path.  Deliberate differences: faces are returned as int32 (the
reference returns float32 indices, :174); texels of faces that no material image updates keep
their Kd colour (the reference leaves them uninitialised because its kernel output is a fresh
buffer, load_textures.py:4, :41).
"""
import os

import numpy as np

__all__ = ["load_obj", "load_mtl", "load_textures", "load_textures_n3mr", "sample_textures", "sample_textures_n3mr",
           "save_obj", "TEXTURE_WRAPPING"]

TEXTURE_WRAPPING = {'REPEAT': 0, 'MIRRORED_REPEAT': 1, 'CLAMP_TO_EDGE': 2, 'CLAMP_TO_BORDER': 3}   # _load_obj_for_n3mr.py:7-8

F32 = np.float32


def _imread(path):
    from PIL import Image
    return np.asarray(Image.open(path)).astype(F32) / 255.


def load_mtl(filename_mtl):
    texture_filenames, colors, material_name, normal_filename = {}, {}, '', ""
    with open(filename_mtl) as f:
        for line in f.readlines():
            s = line.split()
            if not s:
                continue
            if s[0] == 'newmtl':
                material_name = s[1]
            if s[0] == 'map_Kd':
                texture_filenames[material_name] = s[1]
            if s[0] == 'Kd':
                colors[material_name] = np.array(list(map(float, s[1:4])))
            if s[0] == 'map_normal':
                normal_filename = s[2]
    return colors, texture_filenames, normal_filename


def sample_textures(image, face_texcoords, textures, is_update):
    """load_textures.py:11-69.  image [H,W,3] (already flipped), face_texcoords [NF,3,2],
    textures [NF,R*R,3] (updated copy is returned), is_update [NF] bool/int."""
    image = np.asarray(image, F32)
    faces = np.asarray(face_texcoords, F32)
    out = np.array(textures, F32, copy=True)
    NF, RR = out.shape[:2]
    R = int(np.sqrt(RR))
    H, W = image.shape[:2]
    i = np.arange(RR)
    w_y, w_x = i // R, i % R
    lower = (w_x + w_y) < R
    w0 = np.where(lower, (w_x + 1. / 3.) / R, ((R - 1. - w_x) + 2. / 3.) / R).astype(F32)
    w1 = np.where(lower, (w_y + 1. / 3.) / R, ((R - 1. - w_y) + 2. / 3.) / R).astype(F32)
    w2 = ((1. - w0.astype(np.float64)) - w1.astype(np.float64)).astype(F32)   # "1. - w0 - w1" is double arithmetic
    sel = np.flatnonzero(np.asarray(is_update) != 0)
    if sel.size == 0:
        return out
    f = faces[sel]                                                    # [n,3,2]
    pos_x = ((f[:, 0, 0, None] * w0 + f[:, 1, 0, None] * w1 + f[:, 2, 0, None] * w2) * F32(W - 1)).astype(F32)
    pos_y = ((f[:, 0, 1, None] * w0 + f[:, 1, 1, None] * w1 + f[:, 2, 1, None] * w2) * F32(H - 1)).astype(F32)
    x0 = pos_x.astype(np.int64)
    y0 = pos_y.astype(np.int64)
    wx1 = (pos_x - x0.astype(F32)).astype(F32)          # float - (int) -> float (int64 operands would promote to double)
    wy1 = (pos_y - y0.astype(F32)).astype(F32)
    wx0, wy0 = (F32(1) - wx1).astype(F32), (F32(1) - wy1).astype(F32)
    flat = image.reshape(-1, 3)
    n = flat.shape[0]

    def px(yy, xx):
        return flat[np.clip(yy * W + xx, 0, n - 1)]
    y1 = (pos_y + F32(1)).astype(F32).astype(np.int64)                      # (int)(pos_y + 1)
    c = px(y0, x0) * (wx0 * wy0)[..., None] + px(y1, x0) * (wx0 * wy1)[..., None] + \
        px(y0, x0 + 1) * (wx1 * wy0)[..., None] + px(y1, x0 + 1) * (wx1 * wy1)[..., None]
    out[sel] = c.astype(F32)
    return out


def _wrap_mod(x, y):
    """load_textures.py:109-117: x > 0 ? fmod(x, y) : y + fmod(x, y) (float)."""
    x = np.asarray(x, F32)
    y = F32(y)
    return np.where(x > 0, np.fmod(x, y), (y + np.fmod(x, y)).astype(F32)).astype(F32)


def sample_textures_n3mr(image, face_texcoords, textures, is_update, texture_wrapping=0, use_bilinear=True):
    """load_textures.py:103-219.  image [H,W,3] (already flipped), face_texcoords [NF,3,2], textures
    [NF,ts,ts,ts,3] (updated copy is returned), is_update [NF]; texture_wrapping is the integer code of
    TEXTURE_WRAPPING."""
    image = np.asarray(image, F32)
    out = np.array(textures, F32, copy=True)
    NF, ts = out.shape[:2]
    H, W = image.shape[:2]
    sel = np.flatnonzero(np.asarray(is_update) != 0)
    if sel.size == 0:
        return out
    if texture_wrapping == 3:                                          # CLAMP_TO_BORDER: texture_[k] = 0
        out[sel] = 0
        return out
    i = np.arange(ts ** 3)
    d = [(((i // (ts * ts)) % ts) / (ts - 1.)).astype(F32), (((i // ts) % ts) / (ts - 1.)).astype(F32),
         ((i % ts) / (ts - 1.)).astype(F32)]                           # int / double -> float
    ssum = ((d[0] + d[1]) + d[2]).astype(F32)
    pos = ssum > 0
    with np.errstate(invalid="ignore", divide="ignore"):
        d = [np.where(pos, (dk / ssum).astype(F32), dk) for dk in d]
    f = np.array(np.asarray(face_texcoords, F32)[sel], copy=True)      # [n,3,2]
    if texture_wrapping == 0:
        f = _wrap_mod(f, 1.)
    elif texture_wrapping == 1:
        f = np.where(_wrap_mod(f, 2.) < 1, _wrap_mod(f, 1.), (F32(1) - _wrap_mod(f, 1.)).astype(F32)).astype(F32)
    elif texture_wrapping == 2:
        f = np.maximum(np.minimum(f, F32(1)), F32(0))
    pos_x = (((f[:, 0, 0, None] * d[0] + f[:, 1, 0, None] * d[1]) + f[:, 2, 0, None] * d[2]) * F32(W - 1)).astype(F32)
    pos_y = (((f[:, 0, 1, None] * d[0] + f[:, 1, 1, None] * d[1]) + f[:, 2, 1, None] * d[2]) * F32(H - 1)).astype(F32)
    flat = image.reshape(-1)
    n = flat.shape[0]

    def px(yy, xx):
        idx = (yy * W * 3 + xx * 3)[..., None] + np.arange(3)
        return flat[np.clip(idx, 0, n - 1)]
    if use_bilinear:
        x0, y0 = pos_x.astype(np.int64), pos_y.astype(np.int64)
        wx1 = (pos_x - x0.astype(F32)).astype(F32)
        wy1 = (pos_y - y0.astype(F32)).astype(F32)
        wx0, wy0 = (F32(1) - wx1).astype(F32), (F32(1) - wy1).astype(F32)
        y1 = np.minimum((pos_y + F32(1)).astype(F32).astype(np.int64), H - 1)
        x1 = np.minimum(x0 + 1, W - 1)
        c = px(y0, x0) * (wx0 * wy0)[..., None]
        c = c + px(y1, x0) * (wx0 * wy1)[..., None]
        c = c + px(y0, x1) * (wx1 * wy0)[..., None]
        c = c + px(y1, x1) * (wx1 * wy1)[..., None]
    else:
        rnd = lambda v: np.where(v >= 0, np.floor(v.astype(np.float64) + 0.5), np.ceil(v.astype(np.float64) - 0.5)).astype(np.int64)   # noqa: E731
        c = px(rnd(pos_y), rnd(pos_x))
    out[sel] = c.astype(F32).reshape((sel.size,) + out.shape[1:])
    return out


def _parse_texture_faces(lines):
    """texture-coordinate indices per (fan-triangulated) face and its material (shared by both loaders)."""
    faces, material_names, material_name = [], [], ''

    def tidx(tok):
        return int(tok.split('/')[1]) if '/' in tok and '//' not in tok else 0
    for line in lines:
        s = line.split()
        if not s:
            continue
        if s[0] == 'f':
            vs = s[1:]
            v0 = tidx(vs[0])
            for i in range(len(vs) - 2):
                faces.append((v0, tidx(vs[i + 1]), tidx(vs[i + 2])))
                material_names.append(material_name)
        if s[0] == 'usemtl':
            material_name = s[1]
    return np.vstack(faces).astype(np.int32) - 1, material_names


def _material_image(filename_obj, fn):
    image = _imread(os.path.join(os.path.dirname(filename_obj), fn))
    if image.ndim == 2:
        image = np.stack((image,) * 3, -1)
    if image.shape[2] == 4:
        image = image[:, :, :3]
    return np.ascontiguousarray(image[::-1, :, :])


def load_textures_n3mr(filename_obj, filename_mtl, texture_res, texture_wrapping='REPEAT', use_bilinear=True):
    """_load_obj_for_n3mr.py:30-110 -> textures [NF, ts, ts, ts, 3] (default colour 0.5, Kd, then map_Kd)."""
    with open(filename_obj) as f:
        lines = f.readlines()
    vt = np.vstack([[float(v) for v in l.split()[1:3]] for l in lines if l.split() and l.split()[0] == 'vt']).astype(F32)
    faces, material_names = _parse_texture_faces(lines)
    texcoords = vt[faces]
    colors, texture_filenames, _ = load_mtl(filename_mtl)
    textures = np.full((faces.shape[0], 3), 0.5, F32)
    names = np.array(material_names)
    for material, color in colors.items():
        textures[names == material] = color
    ts = int(texture_res)
    textures = np.ascontiguousarray(np.broadcast_to(textures[:, None, None, None, :], (faces.shape[0], ts, ts, ts, 3)), F32)
    for material, fn in texture_filenames.items():
        textures = sample_textures_n3mr(_material_image(filename_obj, fn), texcoords, textures, names == material,
                                        TEXTURE_WRAPPING[texture_wrapping], bool(use_bilinear))
    return textures


def load_textures(filename_obj, filename_mtl, texture_res):
    """-> textures [NF, R*R, 3], face_texcoords [NF,3,2]."""
    with open(filename_obj) as f:
        lines = f.readlines()
    vt = [[float(v) for v in l.split()[1:3]] for l in lines if l.split() and l.split()[0] == 'vt']
    faces, material_names, material_name = [], [], ''

    def tidx(tok):
        return int(tok.split('/')[1]) if '/' in tok and '//' not in tok else 0
    for line in lines:
        s = line.split()
        if not s:
            continue
        if s[0] == 'f':
            vs = s[1:]
            v0 = tidx(vs[0])
            for i in range(len(vs) - 2):
                faces.append((v0, tidx(vs[i + 1]), tidx(vs[i + 2])))
                material_names.append(material_name)
        if s[0] == 'usemtl':
            material_name = s[1]
    faces = np.vstack(faces).astype(np.int32) - 1
    texcoords = np.vstack(vt).astype(F32)[faces] if vt else np.zeros((faces.shape[0], 3, 2), F32)
    colors, texture_filenames, _ = load_mtl(filename_mtl)
    textures = np.ones((faces.shape[0], 3), F32)
    names = np.array(material_names)
    for material, color in colors.items():
        textures[names == material] = color
    textures = np.repeat(textures[:, None, :], texture_res ** 2, axis=1)
    for material, fn in texture_filenames.items():
        image = _imread(os.path.join(os.path.dirname(filename_obj), fn))
        if image.ndim == 2:
            image = np.stack((image,) * 3, -1)
        if image.shape[2] == 4:
            image = image[:, :, :3]
        image = image[::-1, :, :]
        textures = sample_textures(image, texcoords, textures, names == material)
    return textures, texcoords


def load_obj(filename_obj, normalization=False, load_texture=False, dr_type='softras', texture_res=4,
             texture_type='surface', texture_wrapping='REPEAT', use_bilinear=True):
    assert dr_type in ['softras', 'n3mr']
    assert texture_type in ['surface', 'vertex']
    with open(filename_obj) as f:
        lines = f.readlines()
    vertices = np.vstack([[float(v) for v in l.split()[1:4]] for l in lines
                          if l.split() and l.split()[0] == 'v']).astype(F32)
    faces = []
    for line in lines:
        s = line.split()
        if s and s[0] == 'f':
            vs = s[1:]
            v0 = int(vs[0].split('/')[0])
            for i in range(len(vs) - 2):
                faces.append((v0, int(vs[i + 1].split('/')[0]), int(vs[i + 2].split('/')[0])))
    faces = np.vstack(faces).astype(np.int32) - 1
    textures = face_texcoords = None
    if dr_type == 'n3mr':                                              # _load_obj_for_n3mr.py:113-160
        if load_texture:
            for line in lines:
                if line.startswith('mtllib'):
                    filename_mtl = os.path.join(os.path.dirname(filename_obj), line.split()[1])
                    textures = load_textures_n3mr(filename_obj, filename_mtl, texture_res, texture_wrapping, use_bilinear)
            if textures is None:
                raise Exception('Failed to load textures.')
        if normalization:
            vertices = vertices - vertices.min(0)
            vertices = vertices / np.abs(vertices).max()
            vertices = vertices * 2
            vertices = vertices - vertices.max(0) / 2
        return (vertices, faces, textures) if load_texture else (vertices, faces)
    if load_texture and texture_type == 'surface':
        for line in lines:
            if line.startswith('mtllib'):
                filename_mtl = os.path.join(os.path.dirname(filename_obj), line.split()[1])
                textures, face_texcoords = load_textures(filename_obj, filename_mtl, texture_res)
        if textures is None:
            raise Exception('Failed to load textures.')
    elif load_texture and texture_type == 'vertex':
        textures = np.vstack([[float(v) for v in l.split()[4:7]] for l in lines
                              if l.split() and l.split()[0] == 'v']).astype(F32)
    if normalization:
        vertices = vertices - vertices.min(0)
        vertices = vertices / np.abs(vertices).max()
        vertices = vertices * 2
        vertices = vertices - vertices.max(0) / 2
    if load_texture:
        return vertices, faces, textures, None, None, face_texcoords
    return vertices, faces


def save_obj(filename, vertices, faces, textures=None):
    if textures is not None:
        raise NotImplementedError("texture atlas export is outside the accelerated path")
    vertices = np.asarray(vertices)
    faces = np.asarray(faces).astype(np.int64)
    with open(filename, 'w') as f:
        f.write('# %s\n' % os.path.basename(filename))
        for v in vertices:
            f.write('v %.8f %.8f %.8f\n' % (v[0], v[1], v[2]))
        f.write('\n')
        for fc in faces:
            f.write('f %d %d %d\n' % (fc[0] + 1, fc[1] + 1, fc[2] + 1))
