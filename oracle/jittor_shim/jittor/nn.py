"""`jittor.nn` of the NumPy shim (see jittor/__init__.py): Module / ModuleList / relu only."""
import numpy as np


class Module:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self.execute(*a, **k)


class ModuleList(list):
    def __init__(self, modules=()):
        super().__init__(modules)


_Var = None        # set by jittor/__init__.py (the package is removed from sys.modules after the reference is loaded)


def relu(x):
    return _Var(np.maximum(x.data, np.zeros((), x.data.dtype)))
