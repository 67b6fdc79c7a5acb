"""NMR hard rasteriser — host-side mirror of jrender/renderer/dr/n3mr/n3mr.py (N3F:13-346).

``RasterizeFunction`` keeps the reference's constructor, ``execute(faces, textures)`` and
``grad(grad_rgb, grad_alpha, grad_depth)`` protocol and its ten saved tensors (N3F:116); the five
JIT ops + the host compositing between them are ``jr_n3mr_forward`` / ``jr_n3mr_backward``.
The functional API (``rasterize_rgbad`` & co., N3F:189-346) does the NHWC->NCHW permute, the
vertical flip and the optional 2x2 mean pool on the host (NumPy), like the reference's tensor ops.
"""
import ctypes as C

import numpy as np

from .... import _ffi

__all__ = ["RasterizeFunction", "Rasterize", "rasterize_rgbad", "rasterize", "rasterize_silhouettes",
           "rasterize_depth", "DEFAULT_IMAGE_SIZE", "DEFAULT_ANTI_ALIASING", "DEFAULT_NEAR", "DEFAULT_FAR",
           "DEFAULT_EPS", "DEFAULT_BACKGROUND_COLOR"]

DEFAULT_IMAGE_SIZE = 256
DEFAULT_ANTI_ALIASING = True
DEFAULT_NEAR = 0.1
DEFAULT_FAR = 100
DEFAULT_EPS = 1e-4
DEFAULT_BACKGROUND_COLOR = (0, 0, 0)


def _f(v):
    return C.c_float(float(np.float32(v)))


def _p(a):
    return None if a is None else a.ptr


class RasterizeFunction:
    def __init__(self, image_size, near, far, eps, background_color, return_rgb=False, return_alpha=False,
                 return_depth=False, ctx=None):
        self.image_size, self.near, self.far, self.eps = image_size, near, far, eps
        self.background_color = background_color
        self.return_rgb, self.return_alpha, self.return_depth = return_rgb, return_alpha, return_depth
        self.ctx = ctx
        self.save_vars = None

    def execute(self, faces, textures=None):
        ctx = self.ctx or _ffi.Context.default()
        f = faces.clone() if isinstance(faces, _ffi.DeviceArray) else ctx.array(np.asarray(faces, np.float32))
        self.batch_size, self.num_faces = f.shape[:2]
        B, NF, IS = self.batch_size, self.num_faces, int(self.image_size)
        if f.size != B * NF * 9:
            raise ValueError("faces must be [B, NF, 3, 3], got %s" % (f.shape,))
        tex = None
        self.texture_size = None
        if self.return_rgb:
            if textures is None:
                raise ValueError("return_rgb needs textures [B, NF, ts, ts, ts, 3]")
            tex = textures if isinstance(textures, _ffi.DeviceArray) else ctx.array(np.asarray(textures, np.float32))
            self.texture_size = tex.shape[2]
        TS = self.texture_size or 0
        faces_inv = ctx.empty((B, NF, 9))
        face_index_map = ctx.empty((B, IS, IS), np.int32)
        weight_map = ctx.empty((B, IS, IS, 3))
        depth_map = ctx.empty((B, IS, IS))
        rgb_map = ctx.empty((B, IS, IS, 3)) if self.return_rgb else None
        sidx = ctx.empty((B, IS, IS, 8), np.int32) if self.return_rgb else None
        swt = ctx.empty((B, IS, IS, 8)) if self.return_rgb else None
        alpha_map = ctx.empty((B, IS, IS)) if self.return_alpha else None
        face_inv_map = ctx.empty((B, IS, IS, 3, 3)) if self.return_depth else None
        bg = None
        if self.return_rgb and self.background_color is not None:
            bg = (C.c_float * 3)(*[float(np.float32(c)) for c in self.background_color])
        _ffi._check(_ffi.load().jr_n3mr_forward(
            ctx.handle, f.ptr, _p(tex), faces_inv.ptr, face_index_map.ptr, weight_map.ptr, depth_map.ptr,
            _p(face_inv_map), _p(rgb_map), _p(alpha_map), _p(sidx), _p(swt), B, NF, TS, IS, _f(self.near),
            _f(self.far), _f(self.eps), bg, int(self.return_rgb), int(self.return_alpha), int(self.return_depth)))
        self._ctx = ctx
        self.save_vars = (f, tex, face_index_map, weight_map, depth_map, rgb_map, alpha_map, face_inv_map,
                          sidx, swt)                                                        # N3F:116
        return rgb_map, alpha_map, depth_map if self.return_depth else None

    __call__ = execute

    def grad(self, grad_rgb_map=None, grad_alpha_map=None, grad_depth_map=None):
        if self.save_vars is None:
            raise RuntimeError("grad() called before execute()")
        f, tex, fim, wm, dm, rgb, alpha, fivm, sidx, swt = self.save_vars
        ctx = self._ctx
        B, NF, IS = self.batch_size, self.num_faces, int(self.image_size)

        def dev(g, like):
            if like is None:
                return None
            if g is None:
                return ctx.zeros(like.shape)                                                 # N3F:41-55
            return g if isinstance(g, _ffi.DeviceArray) else ctx.array(np.asarray(g, np.float32))
        g_rgb = dev(grad_rgb_map, rgb) if self.return_rgb else None
        g_a = dev(grad_alpha_map, alpha) if self.return_alpha else None
        g_d = dev(grad_depth_map, dm) if self.return_depth else None
        grad_faces = ctx.empty(f.shape)
        grad_textures = ctx.empty(tex.shape) if self.return_rgb else None
        _ffi._check(_ffi.load().jr_n3mr_backward(
            ctx.handle, f.ptr, fim.ptr, wm.ptr, dm.ptr, _p(fivm), _p(rgb), _p(alpha), _p(swt), _p(sidx),
            _p(g_rgb), _p(g_a), _p(g_d), grad_faces.ptr, _p(grad_textures), B, NF, self.texture_size or 0, IS,
            _f(self.eps), int(self.return_rgb), int(self.return_alpha), int(self.return_depth)))
        return grad_faces, grad_textures


class Rasterize:
    """N3F:166-187."""

    def __init__(self, image_size, near, far, eps, background_color, return_rgb=False, return_alpha=False,
                 return_depth=False):
        self.args = (image_size, near, far, eps, background_color, return_rgb, return_alpha, return_depth)

    def __call__(self, faces, textures):
        self.fn = RasterizeFunction(*self.args)
        return self.fn(faces, textures)


def _pool2(x):
    s = x.shape
    return x.reshape(s[:-2] + (s[-2] // 2, 2, s[-1] // 2, 2)).mean((-3, -1)).astype(np.float32)


def rasterize_rgbad(faces, textures=None, image_size=DEFAULT_IMAGE_SIZE, anti_aliasing=DEFAULT_ANTI_ALIASING,
                    near=DEFAULT_NEAR, far=DEFAULT_FAR, eps=DEFAULT_EPS, background_color=DEFAULT_BACKGROUND_COLOR,
                    return_rgb=True, return_alpha=True, return_depth=True):
    """N3F:189-265 -> {'rgb': [B,3,IS,IS], 'alpha': [B,IS,IS], 'depth': [B,IS,IS]} (NumPy)."""
    size = image_size * 2 if anti_aliasing else image_size
    rgb, alpha, depth = Rasterize(size, near, far, eps, background_color, return_rgb, return_alpha,
                                  return_depth)(faces, textures)
    out = {'rgb': None, 'alpha': None, 'depth': None}
    if return_rgb:
        r = rgb.numpy().transpose(0, 3, 1, 2)[:, :, ::-1, :]                              # N3F:240-244
        out['rgb'] = _pool2(r) if anti_aliasing else np.ascontiguousarray(r)
    if return_alpha:
        a = alpha.numpy()[:, ::-1, :]
        out['alpha'] = _pool2(a[:, None]) if anti_aliasing else np.ascontiguousarray(a)
    if return_depth:
        d = depth.numpy()[:, ::-1, :]
        out['depth'] = _pool2(d[:, None]) if anti_aliasing else np.ascontiguousarray(d)
    return out


def rasterize(faces, textures, image_size=DEFAULT_IMAGE_SIZE, anti_aliasing=DEFAULT_ANTI_ALIASING,
              near=DEFAULT_NEAR, far=DEFAULT_FAR, eps=DEFAULT_EPS, background_color=DEFAULT_BACKGROUND_COLOR):
    return rasterize_rgbad(faces, textures, image_size, anti_aliasing, near, far, eps, background_color,
                           True, False, False)['rgb']


def rasterize_silhouettes(faces, image_size=DEFAULT_IMAGE_SIZE, anti_aliasing=DEFAULT_ANTI_ALIASING,
                          near=DEFAULT_NEAR, far=DEFAULT_FAR, eps=DEFAULT_EPS):
    return rasterize_rgbad(faces, None, image_size, anti_aliasing, near, far, eps, None, False, True, False)['alpha']


def rasterize_depth(faces, image_size=DEFAULT_IMAGE_SIZE, anti_aliasing=DEFAULT_ANTI_ALIASING,
                    near=DEFAULT_NEAR, far=DEFAULT_FAR, eps=DEFAULT_EPS):
    return rasterize_rgbad(faces, None, image_size, anti_aliasing, near, far, eps, None, False, False, True)['depth']
