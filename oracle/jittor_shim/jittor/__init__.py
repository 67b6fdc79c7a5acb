"""NumPy-backed stand-in for the part of the `jittor` API that the reference's HOST-side Python touches —
TEST INFRASTRUCTURE ONLY (oracle/): it lets `oracle/make_host_golden.py` EXECUTE the reference's own files
(`jrender/renderer/transform/*.py`, `renderer/lighting/*.py`, `loss/*.py`, `renderer/utils/get_points_from_angles.py`)
from where they lie under /root/reference and write their outputs to `tests/golden/host_ref.npz`, so that the
`jrender_amd` mirrors are compared with the reference itself instead of with a reading of it.

Not a Jittor re-implementation: eager, CPU, no autograd, only what those files call.  Semantics taken from Jittor
(1.3 series; the reference pins no version, requirements.txt:1) and worth stating because they define the fixtures:

* every floating array is float32 (`jt.flags.auto_convert_64_to_32` defaults to 1) and integers are int32;
  Python scalars do not promote (NumPy 2 weak scalars behave the same way);
* `jt.normalize(x, p=2, dim=1, eps=1e-12)` = `x / max(||x||_2 over dim, eps)`;
* `jt.cross` is over the last axis; `Var.broadcast(shape)` broadcasts to `shape`;
* `x += y` updates the Var in place (aliases see it), like `Var.assign`;
* `jt.clamp(x, min_v=None, max_v=None)`; `Var.size(i)`; `Var.numel()`;
* reductions accumulate in float32 (NumPy's pairwise order, not Jittor's: long sums agree to ~1e-6 relative only).
"""
import numpy as np

from . import nn  # noqa: F401  (jittor.nn)

float32 = "float32"
int32 = "int32"


def _np(x):
    return x.data if isinstance(x, Var) else x


def _wrap(a):
    a = np.asarray(a)
    if a.dtype == np.float64:
        a = a.astype(np.float32)
    elif a.dtype == np.int64:
        a = a.astype(np.int32)
    return Var(a)


def _axis(dim):
    if dim is None or dim == ():
        return None
    if isinstance(dim, (list, tuple)):
        return tuple(int(d) for d in dim)
    return int(dim)


class Var:
    __array_priority__ = 100

    def __init__(self, data):
        self.data = np.asarray(data)

    # ---- meta ----
    @property
    def shape(self):
        return list(self.data.shape)

    @property
    def dtype(self):
        return str(self.data.dtype)

    def size(self, i=None):
        return list(self.data.shape) if i is None else int(self.data.shape[i])

    def numel(self):
        return int(self.data.size)

    def numpy(self):
        return np.array(self.data)

    def item(self):
        return self.data.item()

    def __len__(self):
        return self.data.shape[0]

    def __bool__(self):
        return bool(self.data)

    def __float__(self):
        return float(self.data)

    def __repr__(self):
        return "jt.Var(%r)" % (self.data,)

    def stop_grad(self):
        return self

    def detach(self):
        return self

    def float32(self):
        return Var(self.data.astype(np.float32))

    float = float32

    def int32(self):
        return Var(self.data.astype(np.int32))

    int = int32

    # ---- shape ops ----
    def unsqueeze(self, dim):
        return Var(np.expand_dims(self.data, dim))

    def squeeze(self, dim=None):
        return Var(np.squeeze(self.data, dim))

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (list, tuple)):
            shape = shape[0]
        return Var(self.data.reshape(tuple(int(s) for s in shape)))

    view = reshape

    def broadcast(self, shape, dims=()):
        return Var(np.broadcast_to(self.data, tuple(int(s) for s in shape)).copy())

    def transpose(self, *axes):
        if len(axes) == 1 and isinstance(axes[0], (list, tuple)):
            axes = axes[0]
        return Var(np.transpose(self.data, axes if axes else None))

    permute = transpose

    def __getitem__(self, idx):
        if isinstance(idx, tuple):
            idx = tuple(_np(i) for i in idx)
        else:
            idx = _np(idx)
        return Var(self.data[idx])

    def __setitem__(self, idx, value):
        if isinstance(idx, tuple):
            idx = tuple(_np(i) for i in idx)
        else:
            idx = _np(idx)
        self.data[idx] = _np(value)

    # ---- arithmetic (float32 stays float32) ----
    def _bin(self, other, op, rev=False):
        a, b = self.data, _np(other)
        if isinstance(b, np.ndarray) and b.dtype == np.float64:
            b = b.astype(np.float32)
        return _wrap(op(b, a) if rev else op(a, b))

    def __add__(self, o): return self._bin(o, np.add)
    def __radd__(self, o): return self._bin(o, np.add, True)
    def __sub__(self, o): return self._bin(o, np.subtract)
    def __rsub__(self, o): return self._bin(o, np.subtract, True)
    def __mul__(self, o): return self._bin(o, np.multiply)
    def __rmul__(self, o): return self._bin(o, np.multiply, True)
    def __truediv__(self, o): return self._bin(o, np.divide)
    def __rtruediv__(self, o): return self._bin(o, np.divide, True)
    def __neg__(self): return Var(-self.data)
    def __gt__(self, o): return Var(self.data > _np(o))
    def __lt__(self, o): return Var(self.data < _np(o))
    def __ge__(self, o): return Var(self.data >= _np(o))
    def __le__(self, o): return Var(self.data <= _np(o))

    def _inplace(self, res):
        self.data = res.data.astype(self.data.dtype, copy=False)
        return self

    def __iadd__(self, o): return self._inplace(self + o)
    def __isub__(self, o): return self._inplace(self - o)
    def __imul__(self, o): return self._inplace(self * o)
    def __itruediv__(self, o): return self._inplace(self / o)

    def pow(self, e): return _wrap(np.power(self.data, _np(e)))
    __pow__ = pow
    def sqr(self): return Var(self.data * self.data)
    def sqrt(self): return Var(np.sqrt(self.data))
    def abs(self): return Var(np.abs(self.data))
    def exp(self): return Var(np.exp(self.data))

    def sum(self, dim=None, keepdims=False):
        return _wrap(np.sum(self.data, axis=_axis(dim), keepdims=keepdims, dtype=self.data.dtype))

    def max(self, dim=None, keepdims=False):
        return _wrap(np.max(self.data, axis=_axis(dim), keepdims=keepdims))

    def min(self, dim=None, keepdims=False):
        return _wrap(np.min(self.data, axis=_axis(dim), keepdims=keepdims))

    def maximum(self, o): return _wrap(np.maximum(self.data, _np(o)))
    def minimum(self, o): return _wrap(np.minimum(self.data, _np(o)))


nn._Var = Var


def array(data, dtype=None):
    a = np.array(_np(data))
    if dtype is not None:
        a = a.astype(str(dtype))
    return _wrap(a)


def zeros(shape, dtype="float32"):
    if isinstance(shape, int):
        shape = (shape,)
    return Var(np.zeros(tuple(int(s) for s in shape), str(dtype)))


def ones(shape, dtype="float32"):
    if isinstance(shape, int):
        shape = (shape,)
    return Var(np.ones(tuple(int(s) for s in shape), str(dtype)))


def full(shape, val, dtype="float32"):
    return Var(np.full(tuple(int(s) for s in shape), val, str(dtype)))


def zeros_like(x): return Var(np.zeros_like(_np(x)))
def ones_like(x): return Var(np.ones_like(_np(x)))


def sum(x, dim=None, keepdims=False): return x.sum(dim, keepdims)       # noqa: A001
def max(x, dim=None, keepdims=False): return x.max(dim, keepdims)       # noqa: A001
def min(x, dim=None, keepdims=False): return x.min(dim, keepdims)       # noqa: A001
def abs(x): return x.abs()                                              # noqa: A001
def pow(x, e): return x.pow(e)                                          # noqa: A001
def sqrt(x): return x.sqrt()
def tan(x): return Var(np.tan(_np(x)))
def sin(x): return Var(np.sin(_np(x)))
def cos(x): return Var(np.cos(_np(x)))
def exp(x): return Var(np.exp(_np(x)))
def maximum(a, b): return _wrap(np.maximum(_np(a), _np(b)))
def minimum(a, b): return _wrap(np.minimum(_np(a), _np(b)))


def clamp(x, min_v=None, max_v=None):
    a = _np(x)
    if min_v is not None:
        a = np.maximum(a, np.asarray(_np(min_v), a.dtype))
    if max_v is not None:
        a = np.minimum(a, np.asarray(_np(max_v), a.dtype))
    return Var(a)


def normalize(input, p=2, dim=1, eps=1e-12):                             # noqa: A002
    assert p == 2
    a = _np(input)
    n = np.sqrt(np.sum(a * a, axis=dim, keepdims=True, dtype=a.dtype))
    return Var(a / np.maximum(n, np.asarray(eps, a.dtype)))


def cross(a, b, dim=-1):
    return _wrap(np.cross(_np(a), _np(b), axis=dim))


def matmul(a, b):
    return _wrap(np.matmul(_np(a), _np(b)))


def stack(xs, dim=0):
    return _wrap(np.stack([_np(x) for x in xs], axis=dim))


def concat(xs, dim=0):
    return _wrap(np.concatenate([_np(x) for x in xs], axis=dim))


class contrib:
    concat = staticmethod(concat)


def transpose(x, *axes):
    return x.transpose(*axes)


class Function:
    def __call__(self, *a, **k):
        return self.execute(*a, **k)


Module = nn.Module


class _Flags:
    use_cuda = 0
    auto_convert_64_to_32 = 1


flags = _Flags()


def code(*a, **k):
    raise RuntimeError("the NumPy jittor shim executes host-side Python only; the kernels are oracle/_ref's business")
