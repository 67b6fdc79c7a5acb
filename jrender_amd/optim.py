"""Adam — restatement of the optimiser the deformation demo uses (demo2-deform.py:72:
``nn.Adam(model.parameters(), 0.01, betas=(0.5, 0.99))``).  The reference relies on Jittor autograd;
here ``step`` takes the hand-computed gradients, one array per parameter, and updates in place.

Parameters are float32 NumPy arrays (the NumPy path below) or ``DeviceArray``s: those are updated by one
``jr_adam_step`` launch each, moments resident on the GPU (round 5: the reference keeps the optimiser on the GPU
too).  The kernel performs the NumPy path's float operations in the same order (``tests/test_gpu_device_chain.py``)."""
import numpy as np

from . import _ffi

__all__ = ["Adam"]


class Adam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = list(params)
        for p in self.params:
            if not ((isinstance(p, np.ndarray) or isinstance(p, _ffi.DeviceArray)) and p.dtype == np.float32):
                raise TypeError("Adam parameters must be float32 NumPy arrays or DeviceArrays (updated in place)")
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.m = [p.ctx.zeros(p.shape) if isinstance(p, _ffi.DeviceArray) else np.zeros_like(p) for p in self.params]
        self.v = [p.ctx.zeros(p.shape) if isinstance(p, _ffi.DeviceArray) else np.zeros_like(p) for p in self.params]
        self.n_step = 0

    def step(self, grads, iteration=None):
        """One Adam step.  ``iteration`` (a one-element int32 DeviceArray holding the number of steps already taken): the step
        number is read on the DEVICE (jr_adam_step_counted) - what a step recorded into a graph (Context.capture) needs;
        device parameters only, and the caller advances the counter (Context.counter_add)."""
        if len(grads) != len(self.params):
            raise ValueError("need one gradient per parameter (%d != %d)" % (len(grads), len(self.params)))
        if iteration is not None:
            b0, b1 = self.betas
            for p, g, m, v in zip(self.params, grads, self.m, self.v):
                if not (isinstance(p, _ffi.DeviceArray) and isinstance(g, _ffi.DeviceArray)) or g.size != p.size:
                    raise TypeError("a counted step needs DeviceArray parameters and gradients of the same size")
                _ffi._check(_ffi.load().jr_adam_step_counted(p.ctx.handle, p.ptr, g.ptr, m.ptr, v.ptr, p.size, float(self.lr),
                                                             float(b0), float(b1), float(self.eps), float(self.weight_decay),
                                                             iteration.ptr))
            return
        self.n_step += 1
        b0, b1 = self.betas
        c0, c1 = 1.0 - b0 ** self.n_step, 1.0 - b1 ** self.n_step
        for p, g, m, v in zip(self.params, grads, self.m, self.v):
            if isinstance(p, _ffi.DeviceArray):
                if not isinstance(g, _ffi.DeviceArray):
                    g = p.ctx.array(np.asarray(g, np.float32).reshape(p.shape))
                if g.size != p.size:
                    raise ValueError("gradient %s does not match parameter %s" % (g.shape, p.shape))
                _ffi._check(_ffi.load().jr_adam_step(p.ctx.handle, p.ptr, g.ptr, m.ptr, v.ptr, p.size, float(self.lr),
                                                     float(b0), float(b1), float(self.eps), float(self.weight_decay),
                                                     int(self.n_step)))
                continue
            g = np.asarray(g, np.float32).reshape(p.shape)
            if self.weight_decay:
                g = g + self.weight_decay * p
            m *= b0
            m += (1.0 - b0) * g
            v *= b1
            v += (1.0 - b1) * g * g
            p -= (self.lr / c0) * m / (np.sqrt(v / c1) + self.eps)
