"""OBJ / MTL / texture I/O — NumPy + Pillow mirror of jrender/io (softras variant).

load_obj           jrender/io/load_obj.py:9-22 -> utils/_load_obj_for_softras.py:142-207
load_textures      utils/_load_obj_for_softras.py:41-140 (Kd colours, map_Kd images)
sample_textures    utils/load_textures.py:11-69: the reference's CUDA sampler restated in NumPy
                   (per-face R x R texels at barycentric sample points, bilinear fetch)
save_obj           geometry only (texture atlas export is out of scope)

Load-time only, not on the hot path.  Deliberate differences: faces are returned as int32 (the
reference returns float32 indices, :174); texels of faces that no material image updates keep
their Kd colour (the reference leaves them uninitialised because its kernel output is a fresh
buffer, load_textures.py:4, :41).
"""
import os

import numpy as np

__all__ = ["load_obj", "load_mtl", "load_textures", "sample_textures", "save_obj"]

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
    w2 = (1. - w0 - w1).astype(F32)
    sel = np.flatnonzero(np.asarray(is_update) != 0)
    if sel.size == 0:
        return out
    f = faces[sel]                                                    # [n,3,2]
    pos_x = (f[:, 0, 0, None] * w0 + f[:, 1, 0, None] * w1 + f[:, 2, 0, None] * w2) * (W - 1)
    pos_y = (f[:, 0, 1, None] * w0 + f[:, 1, 1, None] * w1 + f[:, 2, 1, None] * w2) * (H - 1)
    x0 = pos_x.astype(np.int64)
    y0 = pos_y.astype(np.int64)
    wx1 = pos_x - x0
    wy1 = pos_y - y0
    wx0, wy0 = 1 - wx1, 1 - wy1
    flat = image.reshape(-1, 3)
    n = flat.shape[0]

    def px(yy, xx):
        return flat[np.clip(yy * W + xx, 0, n - 1)]
    c = px(y0, x0) * (wx0 * wy0)[..., None] + px(y0 + 1, x0) * (wx0 * wy1)[..., None] + \
        px(y0, x0 + 1) * (wx1 * wy0)[..., None] + px(y0 + 1, x0 + 1) * (wx1 * wy1)[..., None]
    out[sel] = c.astype(F32)
    return out


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
    if dr_type == 'n3mr':
        raise NotImplementedError("the n3mr cube-texture loader is not part of this path")
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
