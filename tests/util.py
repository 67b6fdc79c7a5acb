"""Shared helpers for the parity tests (tests only)."""
import numpy as np

RGBA_RTOL, RGBA_ATOL = 1e-4, 1e-6       # north_star: 1e-4 relative fp32; floor for alpha's 1 - prod(1-D) cancellation


def bits_equal(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def rel_err(a, b, atol):
    """max over elements of |a-b| / (rtol-scaled) — returns the worst ratio against RGBA_RTOL*|b|+atol."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / (RGBA_RTOL * np.abs(b) + atol)))


def _finite_pair(a, b):
    """The reference itself produces NaN/inf gradients in a few corners (e.g. 0*inf when a clipped
    barycentric sum vanishes); parity then means: same non-finite pattern, finite parts close."""
    a = np.asarray(a, np.float64).reshape(-1)
    b = np.asarray(b, np.float64).reshape(-1)
    fa, fb = np.isfinite(a), np.isfinite(b)
    if not np.array_equal(fa, fb):
        return None, None
    return a[fa], b[fb]


def grad_err(a, b):
    """Gradient error normalised by the largest gradient magnitude (float atomics reorder sums, so
    element-wise relative error is meaningless where contributions cancel)."""
    a, b = _finite_pair(a, b)
    if a is None:
        return float("inf")
    if a.size == 0:
        return 0.0
    scale = max(float(np.max(np.abs(b))), 1e-30)
    return float(np.max(np.abs(a - b))) / scale


def grad_err_elementwise(a, b, floor=1e-3):
    """|a-b| / (|b| + floor*max|b|): the per-element reading of '1e-4 relative' with a magnitude floor."""
    a, b = _finite_pair(a, b)
    if a is None:
        return float("inf")
    if a.size == 0:
        return 0.0
    scale = max(float(np.max(np.abs(b))), 1e-30)
    return float(np.max(np.abs(a - b) / (np.abs(b) + floor * scale)))
