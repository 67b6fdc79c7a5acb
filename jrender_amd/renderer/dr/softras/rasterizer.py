"""SoftRasterizer — host-side mirror of jrender/renderer/dr/softras/rasterizer.py (SRR:8-61).

Same constructor arguments, defaults, validation (``ValueError`` on unknown enum strings,
SRR:17-24) and ``execute(mesh, mode)`` behaviour: anti-aliasing renders at 2x and applies a
2x2 mean pool (SRR:45, :54-55, here ``jr_avgpool2x2_forward``), ``mode`` selects
silhouettes (channel 3) / rgb (channels 0-2) / both (SRR:56-61).
``backward(grad_silhouettes=..., grad_rgb=...)`` is the explicit replacement of the Jittor
autograd edge through the pool, the channel selection and the op.
"""
import numpy as np

from .... import _ffi
from .soft_rasterize import SoftRasterizeFunction

__all__ = ["SoftRasterizer"]


def _select_channels(images, c0, c1):
    """images [B,4,H,W] -> [B,c1-c0,H,W] (or [B,H,W] for one channel), device-to-device."""
    B, C, H, W = images.shape
    n = c1 - c0
    out = images.ctx.empty((B, n, H, W) if n > 1 else (B, H, W), np.float32)
    plane = H * W * 4
    # one strided copy: B rows of n planes, C planes apart in the source
    _ffi._check(_ffi.load().jr_memcpy2d_d2d(images.ctx.handle, out.ptr, n * plane, images.ptr + c0 * plane, C * plane,
                                            n * plane, B))
    return out


def _scatter_channels(ctx, shape, grads):
    """inverse of _select_channels: assemble grad_images [B,4,H,W] from per-selection grads."""
    B, C, H, W = shape
    out = ctx.zeros(shape, np.float32)
    plane = H * W * 4
    lib = _ffi.load()
    for (c0, c1), g in grads:
        if g is None:
            continue
        g = g if isinstance(g, _ffi.DeviceArray) else ctx.array(np.asarray(g, np.float32))
        n = c1 - c0
        if g.size != B * n * H * W:
            raise ValueError("gradient has %d elements, expected %d" % (g.size, B * n * H * W))
        _ffi._check(lib.jr_memcpy2d_d2d(ctx.handle, out.ptr + c0 * plane, C * plane, g.ptr, n * plane, n * plane, B))
    return out


class SoftRasterizer:
    def __init__(self, image_size=256, background_color=[0, 0, 0], near=1, far=100,
                 anti_aliasing=False, fill_back=False, eps=1e-3, sigma_val=1e-5,
                 dist_func='euclidean', dist_eps=1e-4, gamma_val=1e-4, aggr_func_rgb='softmax',
                 aggr_func_alpha='prod', texture_type='surface', bin_size=0, max_elems_per_bin=0,
                 max_faces_per_pixel_for_grad=16):
        if dist_func not in ['hard', 'euclidean', 'barycentric']:
            raise ValueError('Distance function only support hard, euclidean and barycentric')
        if aggr_func_rgb not in ['hard', 'softmax']:
            raise ValueError('Aggregate function(rgb) only support hard and softmax')
        if aggr_func_alpha not in ['hard', 'prod', 'sum']:
            raise ValueError('Aggregate function(a) only support hard, prod and sum')
        if texture_type not in ['surface', 'vertex']:
            raise ValueError('Texture type only support surface and vertex')
        self.image_size = image_size
        self.background_color = background_color
        self.near = near
        self.far = far
        self.anti_aliasing = anti_aliasing
        self.eps = eps
        self.fill_back = fill_back
        self.sigma_val = sigma_val
        self.dist_func = dist_func
        self.dist_eps = dist_eps
        self.gamma_val = gamma_val
        self.aggr_func_rgb = aggr_func_rgb
        self.aggr_func_alpha = aggr_func_alpha
        self.texture_type = texture_type
        self.bin_size = bin_size
        self.max_elems_per_bin = max_elems_per_bin
        self.max_faces_per_pixel_for_grad = max_faces_per_pixel_for_grad
        self._fn = None

    def rasterize(self, face_vertices, face_textures):
        """The op on raw tensors: [B,NF,3,3] x [B,NF,T,3] -> images [B,4,IS,IS] (after AA pooling)."""
        image_size = self.image_size * (2 if self.anti_aliasing else 1)
        self._fn = SoftRasterizeFunction(image_size, self.background_color, self.near, self.far,
                                         self.fill_back, self.eps, self.sigma_val, self.dist_func,
                                         self.dist_eps, self.gamma_val, self.aggr_func_rgb,
                                         self.aggr_func_alpha, self.texture_type, self.bin_size,
                                         self.max_elems_per_bin, self.max_faces_per_pixel_for_grad)
        images = self._fn(face_vertices, face_textures)
        if self.anti_aliasing:                                                    # SRR:54-55
            B, C, H, W = images.shape
            pooled = images.ctx.empty((B, C, H // 2, W // 2), np.float32)
            _ffi._check(_ffi.load().jr_avgpool2x2_forward(images.ctx.handle, images.ptr, pooled.ptr,
                                                          B * C, H, W))
            images = pooled
        self._images_shape = images.shape
        return images

    def execute(self, mesh, mode=None):
        images = self.rasterize(mesh.face_vertices, mesh.face_textures)
        self._mode = mode
        if mode == 'silhouettes':
            return _select_channels(images, 3, 4)
        elif mode == 'rgb':
            return _select_channels(images, 0, 3)
        elif mode is None:
            return _select_channels(images, 3, 4), _select_channels(images, 0, 3)

    __call__ = execute

    def backward_images(self, grad_images):
        """grad wrt pooled images [B,4,IS,IS] -> (grad_face_vertices [B,NF,3,3], grad_face_textures)."""
        if self._fn is None:
            raise RuntimeError("backward before execute")
        ctx = self._fn._ctx
        g = grad_images if isinstance(grad_images, _ffi.DeviceArray) else ctx.array(np.asarray(grad_images, np.float32))
        if self.anti_aliasing:
            B, C, H, W = self._images_shape
            up = ctx.empty((B, C, 2 * H, 2 * W), np.float32)
            _ffi._check(_ffi.load().jr_avgpool2x2_backward(ctx.handle, g.ptr, up.ptr, B * C, 2 * H, 2 * W))
            g = up
        return self._fn.grad(g)

    def backward(self, grad_silhouettes=None, grad_rgb=None):
        ctx = self._fn._ctx
        g = _scatter_channels(ctx, self._images_shape, [((3, 4), grad_silhouettes), ((0, 3), grad_rgb)])
        return self.backward_images(g)
