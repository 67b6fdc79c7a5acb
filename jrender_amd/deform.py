"""The deformable-template model of the reference's demo2-deform.py (``Model``, demo2-deform.py:17-47) with the backward
chain written out (the reference gets it from Jittor autograd).

    vertices = parametrisation(template * 0.5, displace, center)            demo2-deform.py:35-41
      base = log(|t| / (1 - |t|));  c = tanh(center);  u = sigmoid(base + displace) * sign(t)
      v = relu(u) * (1 - c) - relu(-u) * (c + 1) + c

``DeformModel(vertices, faces)`` keeps template, parameters and intermediates in NumPy (rounds 1 - 4);
``DeformModel(vertices, faces, ctx=ctx)`` keeps them ON THE GPU: ``forward`` / ``backward`` are one launch each
(``jr_deform_vertices_forward`` / ``_backward``), the parameters are ``DeviceArray``s that ``jr.Adam`` updates in place
with ``jr_adam_step`` - the reference keeps model and optimiser on the GPU too (demo2-deform.py:17-40, :72).
"""
import numpy as np

from . import _ffi
from .loss.losses import FlattenLoss, LaplacianLoss

__all__ = ["DeformModel"]


class DeformModel:
    def __init__(self, vertices, faces, ctx=None):
        self.template = (np.asarray(vertices, np.float32) * 0.5)[None]      # [1,nv,3], |t| < 1          (demo2-deform.py:23)
        self.vertices = self.template                                        # (the name the reference's Model uses)
        self.faces = np.asarray(faces, np.int32)[None]
        self.ctx = ctx
        nv = self.template.shape[1]
        if ctx is None:
            self.displace = np.zeros_like(self.template)                    # demo2-deform.py:28-29
            self.center = np.zeros((1, 1, 3), np.float32)
        else:
            self.template_d = ctx.array(self.template)
            self.displace = ctx.zeros((1, nv, 3))
            self.center = ctx.zeros((1, 1, 3))
        self.laplacian_loss = LaplacianLoss(self.template[0], self.faces[0])   # demo2-deform.py:32-33
        self.flatten_loss = FlattenLoss(self.faces[0])

    def parameters(self):
        return [self.displace, self.center]

    # ---- forward: [1,nv,3] vertices ---------------------------------------------------------------------------------
    def forward(self):
        if self.ctx is not None:
            out = self.ctx.empty(self.template.shape, np.float32)
            _ffi._check(_ffi.load().jr_deform_vertices_forward(self.ctx.handle, self.template_d.ptr, self.displace.ptr,
                                                               self.center.ptr, out.ptr, self.template.shape[1]))
            return out
        a = np.abs(self.template)
        with np.errstate(divide="ignore"):
            base = np.log(a / (1 - a))
        self._c = np.tanh(self.center)
        self._s = 1.0 / (1.0 + np.exp(-(base + self.displace)))
        self._sign = np.sign(self.template)
        u = self._s * self._sign
        self._u = u
        v = np.maximum(u, 0) * (1 - self._c) - np.maximum(-u, 0) * (self._c + 1) + self._c
        return v.astype(np.float32)

    __call__ = forward

    # ---- backward: d(loss)/d(vertices) -> (d/d displace, d/d center) ----------------------------------------------------
    def backward(self, g, *weighted):
        """``g`` [1,nv,3] (+ any number of ``(weight, gradient)`` pairs added to it: the regularisers of
        demo2-deform.py:85-88) -> gradients of the two parameters, in ``parameters()`` order."""
        if self.ctx is not None:
            if len(weighted) > 2:
                raise ValueError("the device path combines up to three gradients per launch")
            terms = [(1.0, g)] + [(float(w), x) for w, x in weighted]
            for _w, x in terms:
                if not isinstance(x, _ffi.DeviceArray) or x.size != self.template.size:
                    raise ValueError("gradients must be DeviceArrays of %d floats" % self.template.size)
            terms += [(0.0, None)] * (3 - len(terms))
            g_disp = self.ctx.empty(self.template.shape, np.float32)
            g_cen = self.ctx.empty((1, 1, 3), np.float32)
            args = []
            for w, x in terms:
                args += [x.ptr if x is not None else None, float(w)]
            _ffi._check(_ffi.load().jr_deform_vertices_backward(self.ctx.handle, self.template_d.ptr, self.displace.ptr,
                                                                self.center.ptr, *args, g_disp.ptr, g_cen.ptr,
                                                                self.template.shape[1]))
            return g_disp, g_cen
        g = np.asarray(g, np.float32)
        for w, x in weighted:
            g = g + w * np.asarray(x, np.float32)
        u, c = self._u, self._c
        g_c = (g * (1 - np.maximum(u, 0) - np.maximum(-u, 0))).sum(1, keepdims=True)
        g_u = g * ((u > 0) * (1 - c) + (u < 0) * (c + 1))
        g_disp = g_u * self._sign * self._s * (1 - self._s)
        return g_disp.astype(np.float32), (g_c * (1 - c * c)).astype(np.float32)
