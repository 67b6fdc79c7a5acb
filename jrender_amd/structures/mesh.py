"""Mesh container — NumPy host-side mirror of jrender/structures/mesh.py (MESH:8-375) and
jrender/structures/utils/faces_vertices.py:4-19.

Holds vertices [B,NV,3] f32, faces [B,NF,3] i32 and textures, derives the rasteriser's inputs
``face_vertices`` [B,NF,3,3] and ``face_textures``, surface / vertex normals for lighting, and keeps
the originals for ``reset_()`` (Transform and Lighting mutate the mesh, like the reference).
Float face indices (the reference's softras OBJ loader returns them, _load_obj_for_softras.py:174)
are accepted and cast.
"""
from typing import List

import numpy as np

from .. import _ffi

__all__ = ["Mesh", "face_vertices", "join_meshes_as_scene"]

F32 = np.float32

_device_cache = {}          # (context id, kind, shape, content) -> DeviceArray: face arrays / default textures


def _cached_device(ctx, kind, shape, content, make):
    key = (id(ctx), kind, tuple(shape), content)     # the content itself (bytes of a face array): equal keys ARE equal arrays
    hit = _device_cache.get(key)
    if hit is None or hit.ctx is not ctx or hit.ptr is None:
        if len(_device_cache) >= 16:
            _device_cache.pop(next(iter(_device_cache)))
        hit = _device_cache[key] = make()
    return hit


def _shared_faces(faces):
    """faces [B,NF,3] -> the [NF,3] all views share, or None when the views have different faces."""
    if faces.shape[0] == 1 or faces.strides[0] == 0:
        return faces[0]
    return faces[0] if all(np.array_equal(faces[0], faces[b]) for b in range(1, faces.shape[0])) else None


def device_faces(ctx, faces):
    """[NF,3] int32 on the device, uploaded once per distinct face array."""
    f = np.ascontiguousarray(faces, np.int32)
    return _cached_device(ctx, "faces", f.shape, f.tobytes(), lambda: ctx.array(f))


def face_vertices(vertices, faces):
    """faces_vertices.py:4-19: vertices [B,NV,C] x faces [B,NF,3] -> [B,NF,3,C].  Device vertices [B,NV,3]: the HIP gather
    (jr_face_vertices_forward) when the views share one face array, one launch per view otherwise."""
    if isinstance(vertices, _ffi.DeviceArray):
        faces = np.asarray(faces)
        assert vertices.ndim == 3 and faces.ndim == 3 and vertices.shape[2] == 3
        assert faces.shape[0] in (1, vertices.shape[0])
        assert faces.shape[2] == 3
        ctx, lib = vertices.ctx, _ffi.load()
        B, NV = vertices.shape[:2]
        NF = faces.shape[1]
        out = ctx.empty((B, NF, 3, 3), F32)
        shared = _shared_faces(faces)
        if shared is not None:
            _ffi._check(lib.jr_face_vertices_forward(ctx.handle, vertices.ptr, device_faces(ctx, shared).ptr, out.ptr, B, NV, NF))
        else:
            for b in range(B):
                _ffi._check(lib.jr_face_vertices_forward(ctx.handle, vertices.view(b, b + 1).ptr, device_faces(ctx, faces[b]).ptr,
                                                         out.view(b, b + 1).ptr, 1, NV, NF))
        return out
    vertices = np.asarray(vertices)
    faces = np.asarray(faces)
    assert vertices.ndim == 3 and faces.ndim == 3
    assert vertices.shape[0] == faces.shape[0]
    assert faces.shape[2] == 3
    idx = faces.astype(np.int64)
    return vertices[np.arange(idx.shape[0])[:, None, None], idx]           # [B,NF,3,C]


def face_vertices_backward(grad_fv, faces, num_vertices):
    """Scatter-add VJP of face_vertices (was Jittor autograd): [B,NF,3,C] -> [B,NV,C].  Device gradients [B,NF,3,3]: the HIP
    scatter (jr_face_vertices_backward; float atomics, i.e. summation order varies in the last bits)."""
    if isinstance(grad_fv, _ffi.DeviceArray):
        faces = np.asarray(faces)
        ctx, lib = grad_fv.ctx, _ffi.load()
        B, NF = grad_fv.shape[:2]
        assert grad_fv.size == B * NF * 9 and faces.shape[-2] == NF
        out = ctx.empty((B, int(num_vertices), 3), F32)
        shared = _shared_faces(faces.reshape((-1,) + faces.shape[-2:]))
        if shared is not None:
            _ffi._check(lib.jr_face_vertices_backward(ctx.handle, grad_fv.ptr, device_faces(ctx, shared).ptr, out.ptr, B,
                                                      int(num_vertices), NF))
        else:
            for b in range(B):
                _ffi._check(lib.jr_face_vertices_backward(ctx.handle, grad_fv.view(b, b + 1).ptr, device_faces(ctx, faces[b]).ptr,
                                                          out.view(b, b + 1).ptr, 1, int(num_vertices), NF))
        return out
    grad_fv = np.asarray(grad_fv, F32)
    faces = np.asarray(faces).astype(np.int64)
    B, NF = faces.shape[:2]
    C = grad_fv.shape[-1]
    flat = (faces + (np.arange(B, dtype=np.int64) * num_vertices)[:, None, None]).reshape(-1)
    g = grad_fv.reshape(B * NF * 3, C)
    out = np.stack([np.bincount(flat, weights=g[:, c], minlength=B * num_vertices) for c in range(C)], -1)
    return out.reshape(B, num_vertices, C).astype(F32)


def _normalize(v, eps, axis):
    n = np.sqrt(np.sum(v * v, axis=axis, keepdims=True))
    return v / np.maximum(n, eps)


class Mesh(object):
    def __init__(self, vertices, faces, textures=None, texture_res=1, texture_type='surface',
                 dr_type='softras', metallic_textures=None, roughness_textures=None,
                 normal_textures=None, TBN=None, with_SSS=False, face_texcoords=None):
        # vertices may live on the device (DeviceArray [B,NV,3], float32): the camera transform, the face gather and
        # the rasteriser then run without a host round trip; normals / lighting still take the host arrays
        self._vertices = vertices if isinstance(vertices, _ffi.DeviceArray) else np.asarray(vertices, F32)
        self._faces = np.asarray(faces).astype(np.int32)
        if self._vertices.ndim == 2:
            self._vertices = self._vertices.reshape((1,) + tuple(self._vertices.shape))
        if self._faces.ndim == 2:
            self._faces = self._faces[None]
        self.texture_type = texture_type
        self.batch_size = self._vertices.shape[0]
        self.num_vertices = self._vertices.shape[1]
        self.num_faces = self._faces.shape[1]
        if self._faces.shape[0] == 1 and self.batch_size > 1:
            self._faces = np.broadcast_to(self._faces, (self.batch_size,) + self._faces.shape[1:])
        self._face_vertices = None
        self._face_vertices_update = True
        self._surface_normals = None
        self._surface_normals_update = True
        self._vertex_normals = None
        self._vertex_normals_update = True
        self._with_specular = True
        self._face_texcoords = None if face_texcoords is None else np.asarray(face_texcoords, F32)[None]
        self._with_SSS = with_SSS
        self._fill_back = False
        self.dr_type = dr_type
        if normal_textures is not None or TBN is not None or with_SSS:
            raise NotImplementedError("normal maps / SSS are outside the accelerated SoftRas path")
        self._normal_textures = None
        self._TBN = None

        B, NF, NV = self.batch_size, self.num_faces, self.num_vertices
        if texture_type == 'surface':
            tshape = (B, NF, texture_res ** 2) if dr_type == 'softras' else (B, NF, texture_res, texture_res, texture_res)
        elif texture_type == 'vertex':
            tshape = (B, NV)
        else:
            raise ValueError('texture type not applicable')
        self._metallic_textures = np.zeros(tshape + (1,), F32) if metallic_textures is None else np.asarray(metallic_textures, F32)
        self._roughness_textures = np.ones(tshape + (1,), F32) if roughness_textures is None else np.asarray(roughness_textures, F32)

        self._default_textures = textures is None
        if textures is None:
            self._textures = np.ones(tshape + (3,), F32)
            self.texture_res = texture_res if texture_type == 'surface' else 1
        else:
            textures = np.asarray(textures, F32)
            if textures.ndim == 3 and texture_type == 'surface':
                textures = textures[None]
            if textures.ndim == 2 and texture_type == 'vertex':
                textures = textures[None]
            if textures.ndim == 5:
                textures = textures[None]
            if textures.shape[0] == 1 and B > 1:
                textures = np.broadcast_to(textures, (B,) + textures.shape[1:])
            self._textures = textures
            if dr_type == 'softras':
                self.texture_res = int(np.sqrt(textures.shape[2])) if texture_type == 'surface' else 1
            else:
                self.texture_res = textures.shape[2]
        self._origin_vertices = self._vertices
        self._origin_faces = self._faces
        self._origin_textures = self._textures
        self._origin_default_textures = self._default_textures

    # ---- properties (MESH:143-211) ----
    @property
    def with_specular(self):
        return self._with_specular

    @with_specular.setter
    def with_specular(self, v):
        self._with_specular = v

    @property
    def with_SSS(self):
        return self._with_SSS

    @property
    def faces(self):
        return self._faces

    @faces.setter
    def faces(self, faces):
        self._faces = np.asarray(faces).astype(np.int32)
        self.num_faces = self._faces.shape[1]
        self._face_vertices_update = self._surface_normals_update = self._vertex_normals_update = True

    @property
    def vertices(self):
        return self._vertices

    @vertices.setter
    def vertices(self, vertices):
        self._vertices = vertices if isinstance(vertices, _ffi.DeviceArray) else np.asarray(vertices, F32)
        self.num_vertices = self._vertices.shape[1]
        self._face_vertices_update = self._surface_normals_update = self._vertex_normals_update = True

    @property
    def textures(self):
        return self._textures

    @textures.setter
    def textures(self, textures):
        self._textures = np.asarray(textures, F32)
        self._default_textures = False

    @property
    def metallic_textures(self):
        return self._metallic_textures

    @metallic_textures.setter
    def metallic_textures(self, v):
        self._metallic_textures = np.asarray(v, F32)

    @property
    def roughness_textures(self):
        return self._roughness_textures

    @roughness_textures.setter
    def roughness_textures(self, v):
        self._roughness_textures = np.asarray(v, F32)

    @property
    def normal_textures(self):
        return self._normal_textures

    @property
    def face_texcoords(self):
        return self._face_texcoords

    @property
    def face_vertices(self):
        if self._face_vertices_update:
            self._face_vertices = face_vertices(self.vertices, self.faces)
            self._face_vertices_update = False
        return self._face_vertices

    @property
    def surface_normals(self):
        """MESH:213-229: cross(v2-v1, v0-v1) normalised, evaluated in float64 like the reference."""
        if self._surface_normals_update:
            fv = np.asarray(self.face_vertices).astype(np.float64)      # device vertices: normals are host work
            v10 = fv[:, :, 0] - fv[:, :, 1]
            v12 = fv[:, :, 2] - fv[:, :, 1]
            self._surface_normals = _normalize(np.cross(v12, v10), 1e-12, 2).astype(F32)
            self._surface_normals_update = False
        return self._surface_normals

    @property
    def vertex_normals(self):
        """MESH:231-248: area-weighted sum of the incident corner normals, normalised (eps 1e-6)."""
        if self._vertex_normals_update:
            fv = np.asarray(self.face_vertices)
            n1 = np.cross(fv[:, :, 2] - fv[:, :, 1], fv[:, :, 0] - fv[:, :, 1])
            n2 = np.cross(fv[:, :, 0] - fv[:, :, 2], fv[:, :, 1] - fv[:, :, 2])
            n0 = np.cross(fv[:, :, 1] - fv[:, :, 0], fv[:, :, 2] - fv[:, :, 0])
            out = np.zeros((self.batch_size, self.num_vertices, 3), F32)
            for b in range(self.batch_size):
                f = self.faces[b].astype(np.int64)
                np.add.at(out[b], f[:, 1], n1[b])
                np.add.at(out[b], f[:, 2], n2[b])
                np.add.at(out[b], f[:, 0], n0[b])
            self._vertex_normals = _normalize(out, 1e-6, 2).astype(F32)
            self._vertex_normals_update = False
        return self._vertex_normals

    @property
    def face_textures(self):
        device = isinstance(self._vertices, _ffi.DeviceArray)
        if device and self.texture_type == 'surface' and self._default_textures:
            ctx, shape = self._vertices.ctx, (self._vertices.shape[0],) + tuple(self._textures.shape[1:])
            return _cached_device(ctx, "ones", shape, 1, lambda: ctx.array(np.ones(shape, F32)))
        if self.texture_type in ['surface']:
            t = self.textures
        elif self.texture_type in ['vertex']:
            t = face_vertices(self.textures, self.faces[:self.textures.shape[0]])
        else:
            raise ValueError('texture type not applicable')
        if not device:
            return t
        # device path: one texture block per VIEW (the camera step may have broadcast one vertex set over B eyes)
        B = self._vertices.shape[0]
        if t.shape[0] != B:
            if t.shape[0] != 1:
                raise ValueError("textures batch %d does not match %d views" % (t.shape[0], B))
            t = np.broadcast_to(t, (B,) + t.shape[1:])
        return self._vertices.ctx.array(t)

    def fill_back_(self):
        if not self._fill_back:
            self.faces = np.concatenate((self.faces, self.faces[:, :, [2, 1, 0]]), axis=1)
            self.textures = np.concatenate((self.textures, self.textures), axis=1)
            self._fill_back = True

    def reset_(self):
        self.vertices = self._origin_vertices
        self.faces = self._origin_faces
        self.textures = self._origin_textures
        self._default_textures = self._origin_default_textures
        self._fill_back = False

    @classmethod
    def from_obj(cls, filename_obj, normalization=False, load_texture=False, dr_type='softras',
                 texture_res=1, texture_type='surface', texture_wrapping='REPEAT', use_bilinear=True,
                 with_SSS=False):
        from ..io import load_obj
        textures = None
        face_texcoords = None
        if load_texture and dr_type == 'n3mr':
            vertices, faces, textures = load_obj(
                filename_obj, normalization=normalization, texture_res=texture_res, load_texture=True,
                dr_type=dr_type, texture_type=texture_type, texture_wrapping=texture_wrapping,
                use_bilinear=use_bilinear)
        elif load_texture:
            vertices, faces, textures, _, _, face_texcoords = load_obj(
                filename_obj, normalization=normalization, texture_res=texture_res, load_texture=True,
                dr_type=dr_type, texture_type=texture_type)
        else:
            vertices, faces = load_obj(filename_obj, normalization=normalization, texture_res=texture_res,
                                       load_texture=False, dr_type=dr_type)
        return cls(vertices, faces, textures, texture_res, texture_type, dr_type=dr_type,
                   with_SSS=with_SSS, face_texcoords=face_texcoords)

    def save_obj(self, filename_obj, save_texture=False, texture_res_out=16):
        from ..io import save_obj
        if self.batch_size != 1:
            raise ValueError('Could not save when batch size >= 1')
        if save_texture:
            raise NotImplementedError("texture atlas export is outside the accelerated path")
        save_obj(filename_obj, self.vertices[0], self.faces[0])


def join_meshes_as_scene(meshes: List[Mesh], include_texture: bool = True) -> Mesh:
    """MESH:330-375."""
    vert = meshes[0].vertices
    face = meshes[0].faces
    nv = vert.shape[1]
    for mesh in meshes[1:]:
        vert = np.concatenate([vert, mesh.vertices], axis=1)
        face = np.concatenate([face, mesh.faces + nv], axis=1)
        nv += mesh.vertices.shape[1]
    if not include_texture:
        return Mesh(vert, face)
    dr_type, texture_type = meshes[0].dr_type, meshes[0].texture_type
    if not all(dr_type == m.dr_type and texture_type == m.texture_type for m in meshes):
        raise ValueError("Inconsistent textures in join_meshes_as_scene (dr_type or texture_type).")
    tex = np.concatenate([m.textures for m in meshes], axis=1)
    return Mesh(vertices=vert, faces=face, textures=tex, texture_type=texture_type, dr_type=dr_type)
