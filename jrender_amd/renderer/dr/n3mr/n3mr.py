"""NMR hard rasteriser — host-side mirror of jrender/renderer/dr/n3mr/n3mr.py (N3F:13-346).

``RasterizeFunction`` keeps the reference's constructor, ``execute(faces, textures)`` and
``grad(grad_rgb, grad_alpha, grad_depth)`` protocol and its ten saved tensors (N3F:116); the five
JIT ops + the host compositing between them are ``jr_n3mr_forward`` / ``jr_n3mr_backward``.
The functional API (``rasterize_rgbad`` & co., N3F:189-346) does the NHWC->NCHW permute, the
vertical flip and the optional 2x2 mean pool on the device (``jr_n3mr_image_forward``) and returns
DeviceArrays; ``RasterizeRGBAD`` is the same call as an object with a ``backward``.
"""
import ctypes as C

import numpy as np

from .... import _ffi

__all__ = ["RasterizeFunction", "Rasterize", "RasterizeRGBAD", "rasterize_rgbad", "rasterize", "rasterize_silhouettes",
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


def _image_forward(x, channels, pool):
    """[B,H,W,C] bottom-up device map -> [B,C,H/pool,W/pool] top-down (jr_n3mr_image_forward)."""
    ctx = x.ctx
    B, H, W = x.shape[:3]
    out = ctx.empty((B, channels, H // pool, W // pool))
    _ffi._check(_ffi.load().jr_n3mr_image_forward(ctx.handle, x.ptr, out.ptr, B, H, W, channels, pool))
    return out


def _image_backward(g, ctx, B, H, W, channels, pool):
    """gradient wrt [B,C,H/pool,W/pool] (host or device) -> gradient wrt the [B,H,W,C] bottom-up map."""
    g = g if isinstance(g, _ffi.DeviceArray) else ctx.array(np.asarray(g, np.float32))
    if g.size != B * channels * (H // pool) * (W // pool):
        raise ValueError("gradient of %d elements for an image [%d,%d,%d,%d]" % (g.size, B, channels, H // pool, W // pool))
    out = ctx.empty((B, H, W, channels) if channels > 1 else (B, H, W))
    _ffi._check(_ffi.load().jr_n3mr_image_backward(ctx.handle, g.ptr, out.ptr, B, H, W, channels, pool))
    return out


class RasterizeRGBAD:
    """rasterize_rgbad (N3F:189-265) as an object that remembers what its backward needs: the NMR op at the
    super-sampled size, then the transpose / vertical flip / 2x2 mean pool ON THE DEVICE
    (jr_n3mr_image_forward; Jittor tensor ops in the reference).  ``backward`` maps image gradients back
    through the same chain to (grad_faces, grad_textures)."""

    def __init__(self, image_size=DEFAULT_IMAGE_SIZE, anti_aliasing=DEFAULT_ANTI_ALIASING, near=DEFAULT_NEAR,
                 far=DEFAULT_FAR, eps=DEFAULT_EPS, background_color=DEFAULT_BACKGROUND_COLOR, return_rgb=True,
                 return_alpha=True, return_depth=True, ctx=None):
        self.pool = 2 if anti_aliasing else 1
        self.size = image_size * self.pool
        self.flags = (return_rgb, return_alpha, return_depth)
        self.fn = RasterizeFunction(self.size, near, far, eps, background_color, return_rgb, return_alpha,
                                    return_depth, ctx=ctx)

    def __call__(self, faces, textures=None):
        rgb, alpha, depth = self.fn(faces, textures)
        rr, ra, rd = self.flags
        out = {'rgb': None, 'alpha': None, 'depth': None}
        if rr:
            out['rgb'] = _image_forward(rgb, 3, self.pool)                                  # [B,3,IS,IS]
        if ra:
            a = _image_forward(alpha, 1, self.pool)
            out['alpha'] = a if self.pool == 2 else a.reshape(a.shape[0], a.shape[2], a.shape[3])   # N3F:252-254
        if rd:
            d = _image_forward(depth, 1, self.pool)
            out['depth'] = d if self.pool == 2 else d.reshape(d.shape[0], d.shape[2], d.shape[3])
        return out

    def backward(self, grad_rgb=None, grad_alpha=None, grad_depth=None):
        ctx = self.fn._ctx
        B, S = self.fn.batch_size, self.size
        rr, ra, rd = self.flags
        g_rgb = _image_backward(grad_rgb, ctx, B, S, S, 3, self.pool) if (rr and grad_rgb is not None) else None
        g_a = _image_backward(grad_alpha, ctx, B, S, S, 1, self.pool) if (ra and grad_alpha is not None) else None
        g_d = _image_backward(grad_depth, ctx, B, S, S, 1, self.pool) if (rd and grad_depth is not None) else None
        return self.fn.grad(g_rgb, g_a, g_d)


def rasterize_rgbad(faces, textures=None, image_size=DEFAULT_IMAGE_SIZE, anti_aliasing=DEFAULT_ANTI_ALIASING,
                    near=DEFAULT_NEAR, far=DEFAULT_FAR, eps=DEFAULT_EPS, background_color=DEFAULT_BACKGROUND_COLOR,
                    return_rgb=True, return_alpha=True, return_depth=True):
    """N3F:189-265 -> {'rgb': [B,3,IS,IS], 'alpha', 'depth': [B,IS,IS] ([B,1,IS,IS] with anti_aliasing, like
    the reference's nn.pool(x.unsqueeze(1)))} as DeviceArrays."""
    return RasterizeRGBAD(image_size, anti_aliasing, near, far, eps, background_color, return_rgb, return_alpha,
                          return_depth)(faces, textures)


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
