"""jrender_amd — MI355X-native SoftRas differentiable rasteriser behind jrender's API surface.

``import jrender_amd as jr`` then ``jr.Renderer(dr_type='softras')``, ``jr.Mesh``,
``jr.soft_rasterize`` ... as with the reference.  The hot path runs in hand-written HIP kernels
(jrender_amd/csrc) through a C ABI (include/jrender_hip.h); the host side is NumPy + ctypes.
"""
from .structures import *          # noqa: F401,F403
from .renderer import *            # noqa: F401,F403
from .loss import *                # noqa: F401,F403
from .io import *                  # noqa: F401,F403
from .optim import Adam            # noqa: F401
from .deform import DeformModel    # noqa: F401
from . import synthetic            # noqa: F401
from ._ffi import Context, DeviceArray   # noqa: F401

__version__ = "0.1"
