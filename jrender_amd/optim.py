"""Adam — NumPy restatement of the optimiser the deformation demo uses (demo2-deform.py:76:
``nn.Adam(model.parameters(), 0.01, betas=(0.5, 0.99))``).  The reference relies on Jittor autograd;
here ``step`` takes the hand-computed gradients, one array per parameter, and updates in place."""
import numpy as np

__all__ = ["Adam"]


class Adam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = list(params)
        for p in self.params:
            if not (isinstance(p, np.ndarray) and p.dtype == np.float32):
                raise TypeError("Adam parameters must be float32 NumPy arrays (updated in place)")
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.m = [np.zeros_like(p) for p in self.params]
        self.v = [np.zeros_like(p) for p in self.params]
        self.n_step = 0

    def step(self, grads):
        if len(grads) != len(self.params):
            raise ValueError("need one gradient per parameter (%d != %d)" % (len(grads), len(self.params)))
        self.n_step += 1
        b0, b1 = self.betas
        c0, c1 = 1.0 - b0 ** self.n_step, 1.0 - b1 ** self.n_step
        for p, g, m, v in zip(self.params, grads, self.m, self.v):
            g = np.asarray(g, np.float32).reshape(p.shape)
            if self.weight_decay:
                g = g + self.weight_decay * p
            m *= b0
            m += (1.0 - b0) * g
            v *= b1
            v += (1.0 - b1) * g * g
            p -= (self.lr / c0) * m / (np.sqrt(v / c1) + self.eps)
