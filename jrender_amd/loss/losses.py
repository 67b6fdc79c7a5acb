"""Losses of the mesh-deformation demo — NumPy mirror of jrender/loss/ with hand-written
backward passes (the reference relied on Jittor autograd):
    neg_iou_loss   iou_loss.py:1-4
    LaplacianLoss  laplacian_loss.py:5-37
    FlattenLoss    flatten_loss.py:5-80
Each ``*_backward`` returns d(loss)/d(input) for an upstream scalar gradient of 1.
"""
import numpy as np

from .. import _ffi

__all__ = ["neg_iou_loss", "neg_iou_loss_backward", "neg_iou_loss_and_grad", "LaplacianLoss", "FlattenLoss"]

F32 = np.float32


def neg_iou_loss_and_grad(predict, target, total_views=None):
    """-> (iou per view [B], d(neg_iou_loss)/d(predict)); the loss is 1 - mean(iou).  Device ``predict`` (DeviceArray
    [B,...]): one HIP launch (jr_neg_iou_loss) computes both and BOTH stay on the device (nothing waits for the GPU:
    read ``iou.numpy()`` when the number is needed); ``total_views`` is the number of views the loss averages over
    when this call only sees a shard of them (default: its own batch)."""
    if isinstance(predict, _ffi.DeviceArray):
        ctx = predict.ctx
        t = target if isinstance(target, _ffi.DeviceArray) else ctx.array(np.asarray(target, F32))
        if t.size != predict.size:
            raise ValueError("target %s does not match predict %s" % (t.shape, predict.shape))
        B = predict.shape[0]
        iou = ctx.empty((B,), F32)
        grad = ctx.empty(predict.shape, F32)
        _ffi._check(_ffi.load().jr_neg_iou_loss(ctx.handle, predict.ptr, t.ptr, iou.ptr, grad.ptr, B, predict.size // B,
                                                float(total_views or B)))
        return iou, grad
    predict, target = np.asarray(predict, F32), np.asarray(target, F32)
    dims = tuple(range(predict.ndim))[1:]
    I = (predict * target).sum(dims)
    U = (predict + target - predict * target).sum(dims) + 1e-6
    g = neg_iou_loss_backward(predict, target)
    if total_views:
        g = g * F32(predict.shape[0] / total_views)
    return (I / U).astype(F32), g


def neg_iou_loss(predict, target):
    if isinstance(predict, _ffi.DeviceArray):
        ctx = predict.ctx
        t = target if isinstance(target, _ffi.DeviceArray) else ctx.array(np.asarray(target, F32))
        if t.size != predict.size:
            raise ValueError("target %s does not match predict %s" % (t.shape, predict.shape))
        B = predict.shape[0]
        iou = ctx.empty((B,), F32)
        _ffi._check(_ffi.load().jr_neg_iou_loss(ctx.handle, predict.ptr, t.ptr, iou.ptr, None, B, predict.size // B, 1.0))
        return F32(1. - iou.numpy().sum() / B)
    predict, target = np.asarray(predict, F32), np.asarray(target, F32)
    dims = tuple(range(predict.ndim))[1:]
    intersect = (predict * target).sum(dims)
    union = (predict + target - predict * target).sum(dims) + 1e-6
    return F32(1. - (intersect / union).sum() / intersect.size)


def neg_iou_loss_backward(predict, target):
    if isinstance(predict, _ffi.DeviceArray):
        return neg_iou_loss_and_grad(predict, target)[1]
    predict, target = np.asarray(predict, F32), np.asarray(target, F32)
    dims = tuple(range(predict.ndim))[1:]
    shape = (-1,) + (1,) * (predict.ndim - 1)
    I = (predict * target).sum(dims).reshape(shape)
    U = ((predict + target - predict * target).sum(dims) + 1e-6).reshape(shape)
    n = predict.shape[0]
    # d(I/U)/dp = (t*U - I*(1-t)) / U^2
    return (-(target * U - I * (1 - target)) / (U * U) / n).astype(F32)


def _csr_arrays(m):
    """Dense [n,n] float matrix -> (rowptr int32 [n+1], col int32 [nnz], val float32 [nnz]), columns ascending per row
    (the order jr_laplacian_loss sums a row in)."""
    m = np.asarray(m)
    rows, cols = np.nonzero(m)                       # row-major: rows ascending, columns ascending within a row
    rowptr = np.zeros(m.shape[0] + 1, np.int32)
    np.cumsum(np.bincount(rows, minlength=m.shape[0]), out=rowptr[1:])
    return rowptr, cols.astype(np.int32), m[rows, cols].astype(F32)


class _DeviceLoss:
    """Per-mesh loss values [B] on the device; ``float(x)`` / ``x.numpy()`` download (and average if the loss was
    built with average=True) — the only points that wait for the GPU."""

    def __init__(self, values, average):
        self.values, self.average = values, average

    def numpy(self):
        v = self.values.numpy()
        return F32(v.sum() / v.shape[0]) if self.average else v

    def __array__(self, dtype=None, copy=None):
        a = np.asarray(self.numpy())
        return a if dtype is None else a.astype(dtype)

    def __float__(self):
        # consistent with numpy(): the batch mean when the loss averages, else the value of the ONE mesh - a per-mesh
        # vector has no float (what float() of the host result says too)
        v = self.numpy()
        if np.ndim(v) and np.size(v) != 1:
            raise TypeError("only a loss built with average=True (or a batch of one mesh) converts to a float; "
                            "use .numpy() for the %d per-mesh values" % np.size(v))
        return float(np.reshape(v, -1)[0]) if np.ndim(v) else float(v)


class LaplacianLoss:
    def __init__(self, vertex, faces, average=False):
        vertex, faces = np.asarray(vertex), np.asarray(faces).astype(np.int64)
        self.nv, self.nf, self.average = vertex.shape[0], faces.shape[0], average
        # laplacian_loss.py:14-29 builds the matrix dense: -1 for every edge (set, not accumulated), the diagonal = the
        # vertex degree, rows divided by it.  ~7 non-zeros per row: built and applied as CSR here (a threaded BLAS needs
        # 30 ms for the dense 1352 x 1352 x 3 product on a 256-thread host, the sparse one 0.1 ms; the headline mesh's
        # 19 502 x 19 502 matrix would be 1.5 GB); `.laplacian` materialises the reference's dense array on demand.
        self._dense = None
        try:
            from scipy.sparse import coo_matrix, csr_matrix, diags
            e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 0]], faces[:, [1, 2]], faces[:, [2, 1]],
                                faces[:, [2, 0]], faces[:, [0, 2]]])
            adj = coo_matrix((np.ones(len(e), F32), (e[:, 0], e[:, 1])), shape=(self.nv, self.nv)).tocsr()
            adj.data[:] = -1.0                                     # duplicates were summed: an edge is -1 however many faces share it
            # lap[r, r] = -(sum of the row) OVERWRITES the diagonal.  A face that repeats a vertex has put a -1 there first
            # (laplacian_loss.py:14-19), and that -1 is part of the sum: the diagonal is then degree + 1 ...
            diag = -np.asarray(adj.sum(1)).reshape(-1).astype(F32)
            off = adj.tolil()
            off.setdiag(0)
            off = off.tocsr()
            off.eliminate_zeros()
            lap = (off + diags(diag, format="csr")).tocsr().astype(F32)
            lap.sort_indices()
            # row / diagonal, element by element like the reference's `laplacian[i, :] /= laplacian[i, i]` (a float division each)
            rows = np.repeat(np.arange(self.nv), np.diff(lap.indptr))
            with np.errstate(divide="ignore", invalid="ignore"):
                lap.data = (lap.data / diag[rows]).astype(F32)
            # ... and a vertex no face refers to has an all-zero row divided by its zero diagonal: 0 / 0 = NaN in EVERY column
            # (the loss of such a mesh is NaN in the reference; a sparse row of explicit NaNs keeps that)
            lonely = np.flatnonzero(diag == 0)
            if lonely.size:
                keep = np.ones(self.nv, bool)
                keep[lonely] = False
                lap = diags(keep.astype(F32), format="csr") @ lap
                nan_rows = coo_matrix((np.full(lonely.size * self.nv, np.nan, F32),
                                       (np.repeat(lonely, self.nv), np.tile(np.arange(self.nv), lonely.size))),
                                      shape=(self.nv, self.nv)).tocsr()
                lap = (lap + nan_rows).tocsr().astype(F32)
                lap.sort_indices()
            self._csr = csr_matrix(lap)
            self._csr_t = csr_matrix(lap.T.tocsr())
            self._csr_t.sort_indices()
        except ImportError:                     # pragma: no cover
            self._csr = self._csr_t = None
            self._dense = self._dense_matrix(faces)

    def _dense_matrix(self, faces):
        lap = np.zeros([self.nv, self.nv], F32)
        lap[faces[:, 0], faces[:, 1]] = -1
        lap[faces[:, 1], faces[:, 0]] = -1
        lap[faces[:, 1], faces[:, 2]] = -1
        lap[faces[:, 2], faces[:, 1]] = -1
        lap[faces[:, 2], faces[:, 0]] = -1
        lap[faces[:, 0], faces[:, 2]] = -1
        r, c = np.diag_indices(self.nv)
        lap[r, c] = -lap.sum(1)
        lap /= lap[r, c][:, None]
        return lap

    @property
    def laplacian(self):
        """The dense [nv, nv] matrix of the reference (laplacian_loss.py:29), materialised on first use."""
        if self._dense is None:
            self._dense = np.asarray(self._csr.todense(), F32)
        return self._dense

    def _apply(self, x, transpose=False):
        m = self._csr_t if transpose else self._csr
        if m is None:
            return np.matmul(self.laplacian.T if transpose else self.laplacian, x)
        x = np.asarray(x, F32)
        if x.ndim == 2:
            return m @ x
        return np.stack([m @ xi for xi in x])

    def __call__(self, x):
        if isinstance(x, _ffi.DeviceArray):
            return self.value_and_grad(x, want_grad=False)[0]
        x = np.asarray(x, F32)
        y = self._apply(x)
        out = (y * y).sum(tuple(range(y.ndim))[1:])
        return out.sum() / x.shape[0] if self.average else out

    def backward(self, x):
        """d(sum over batch of the loss)/dx (divided by the batch size when average=True)."""
        if isinstance(x, _ffi.DeviceArray):
            return self.value_and_grad(x)[1]
        x = np.asarray(x, F32)
        g = 2 * self._apply(self._apply(x), transpose=True)
        return (g / x.shape[0] if self.average else g).astype(F32)

    def _device_matrices(self, ctx):
        """(rowptr, col, val) of L and of its transpose on the device, uploaded once per context."""
        cache = self.__dict__.setdefault("_dev", {})
        if id(ctx) not in cache:
            if self._csr is not None:           # rows ascending, columns ascending within a row (what _csr_arrays gives for the dense matrix)
                arrays = tuple(x for m in (self._csr, self._csr_t)
                               for x in (m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.astype(F32)))
            else:
                arrays = _csr_arrays(self.laplacian) + _csr_arrays(np.ascontiguousarray(self.laplacian.T))
            cache[id(ctx)] = (ctx,) + tuple(ctx.array(a) for a in arrays)
        return cache[id(ctx)][1:]

    def value_and_grad(self, x, want_grad=True):
        """(__call__(x), backward(x)).  Device ``x`` [B,nv,3]: ONE launch (jr_laplacian_loss), both results stay on the
        device — the value is a DeviceArray [B] (its mean when ``average``: [1])."""
        if not isinstance(x, _ffi.DeviceArray):
            return self(x), (self.backward(x) if want_grad else None)
        ctx = x.ctx
        if x.ndim != 3 or x.shape[1] != self.nv or x.shape[2] != 3 or x.dtype != F32:
            raise ValueError("LaplacianLoss: device vertices must be float32 [B, %d, 3], got %s" % (self.nv, x.shape))
        B = x.shape[0]
        m = self._device_matrices(ctx)
        loss = ctx.empty((B,), F32)
        grad = ctx.empty(x.shape, F32) if want_grad else None
        scratch = ctx.empty(x.shape, F32)
        _ffi._check(_ffi.load().jr_laplacian_loss(ctx.handle, *[a.ptr for a in m], x.ptr, scratch.ptr, loss.ptr,
                                                  grad.ptr if want_grad else None, B, self.nv,
                                                  1.0 / B if self.average else 1.0))
        return _DeviceLoss(loss, self.average), grad


class FlattenLoss:
    def __init__(self, faces, average=False):
        faces = np.asarray(faces).astype(np.int64)
        self.nf, self.average = faces.shape[0], average
        edges = sorted(set(tuple(v) for v in np.sort(np.concatenate((faces[:, 0:2], faces[:, 1:3]), axis=0))))
        # the two faces sharing each edge -> opposite vertices v2, v3
        opp = {}
        for f in faces:
            for a in range(3):
                e = tuple(sorted((f[a], f[(a + 1) % 3])))
                opp.setdefault(e, []).append(f[(a + 2) % 3])
        v0s, v1s, v2s, v3s = [], [], [], []
        for e in edges:
            o = opp.get(e, [])
            if len(o) >= 2:
                v0s.append(e[0]); v1s.append(e[1]); v2s.append(o[0]); v3s.append(o[1])
        self.v0s, self.v1s = np.array(v0s, np.int64), np.array(v1s, np.int64)
        self.v2s, self.v3s = np.array(v2s, np.int64), np.array(v3s, np.int64)

    @staticmethod
    def _half(a, b, eps):
        al2 = (a * a).sum(-1)
        bl2 = (b * b).sum(-1)
        al1 = np.sqrt(al2 + eps)
        bl1 = np.sqrt(bl2 + eps)
        ab = (a * b).sum(-1)
        cos = ab / (al1 * bl1 + eps)
        sin = np.sqrt(1 - cos * cos + eps)
        c = a * (ab / (al2 + eps))[..., None]
        cb = b - c
        cbl1 = bl1 * sin
        return cb, cbl1

    def _cos(self, vertices, eps):
        v0, v1 = vertices[:, self.v0s], vertices[:, self.v1s]
        v2, v3 = vertices[:, self.v2s], vertices[:, self.v3s]
        cb1, l1 = self._half(v1 - v0, v2 - v0, eps)
        cb2, l2 = self._half(v1 - v0, v3 - v0, eps)
        return (cb1 * cb2).sum(-1) / (l1 * l2 + eps)

    def __call__(self, vertices, eps=1e-6):
        if isinstance(vertices, _ffi.DeviceArray):
            return self.value_and_grad(vertices, eps, want_grad=False)[0]
        vertices = np.asarray(vertices, np.float64)
        cos = self._cos(vertices, eps)
        loss = ((cos + 1) ** 2).sum(tuple(range(cos.ndim))[1:])
        return (loss.sum() / vertices.shape[0] if self.average else loss).astype(F32)

    @staticmethod
    def _half_vjp(a, b, eps, dcb, dl):
        """VJP of _half: upstream (dcb [.,3], dl [.]) -> (da, db).  Mirrors the forward expression by expression."""
        al2 = (a * a).sum(-1)
        bl2 = (b * b).sum(-1)
        al1 = np.sqrt(al2 + eps)
        bl1 = np.sqrt(bl2 + eps)
        ab = (a * b).sum(-1)
        den = al1 * bl1 + eps
        cos = ab / den
        sin = np.sqrt(1 - cos * cos + eps)
        s = ab / (al2 + eps)
        # l = bl1 * sin
        dbl1 = sin * dl
        dsin = bl1 * dl
        dcos = dsin * (-cos / sin)
        # cos = ab / (al1 * bl1 + eps)
        dab = dcos / den
        dal1 = -ab * bl1 / (den * den) * dcos
        dbl1 = dbl1 + (-ab * al1 / (den * den) * dcos)
        # cb = b - a * s,  s = ab / (al2 + eps)
        db = dcb.copy()
        da = -s[..., None] * dcb
        ds = -(a * dcb).sum(-1)
        dab = dab + ds / (al2 + eps)
        dal2 = -ab / ((al2 + eps) ** 2) * ds
        # norms and the dot product
        dal2 = dal2 + dal1 / (2 * al1)
        dbl2 = dbl1 / (2 * bl1)
        da = da + 2 * a * dal2[..., None] + b * dab[..., None]
        db = db + 2 * b * dbl2[..., None] + a * dab[..., None]
        return da, db

    def backward(self, vertices, eps=1e-6):
        """d(sum over batch of the loss)/d(vertices) (divided by the batch size when average=True): reverse mode
        through the per-edge expression, scattered to the four vertices of every edge pair."""
        return self.value_and_grad(vertices, eps)[1]

    def value_and_grad(self, vertices, eps=1e-6, want_grad=True):
        """(__call__(vertices), backward(vertices)) from ONE evaluation of the per-edge expression.  Device vertices
        [B,nv,3]: one launch (jr_flatten_loss, double arithmetic like this mirror; the gradient is scattered with float
        atomics), both results stay on the device."""
        if isinstance(vertices, _ffi.DeviceArray):
            ctx = vertices.ctx
            if vertices.ndim != 3 or vertices.shape[2] != 3 or vertices.dtype != F32:
                raise ValueError("FlattenLoss: device vertices must be float32 [B, nv, 3], got %s" % (vertices.shape,))
            cache = self.__dict__.setdefault("_dev", {})
            if id(ctx) not in cache:
                pad = lambda a: np.ascontiguousarray(a if len(a) else [0], np.int32)      # noqa: E731  (no zero-byte buffers)
                cache[id(ctx)] = (ctx,) + tuple(ctx.array(pad(a)) for a in (self.v0s, self.v1s, self.v2s, self.v3s))
            idx = cache[id(ctx)][1:]
            B, nv = vertices.shape[:2]
            if len(self.v0s) and int(max(a.max() for a in (self.v0s, self.v1s, self.v2s, self.v3s))) >= nv:
                raise ValueError("FlattenLoss: the mesh has %d vertices, the faces index up to %d" % (nv, int(max(a.max() for a in (self.v0s, self.v1s, self.v2s, self.v3s)))))
            loss = ctx.empty((B,), F32)
            grad = ctx.empty(vertices.shape, F32) if want_grad else None
            _ffi._check(_ffi.load().jr_flatten_loss(ctx.handle, *[a.ptr for a in idx], vertices.ptr, loss.ptr,
                                                    grad.ptr if want_grad else None, B, nv, len(self.v0s), float(eps),
                                                    1.0 / B if self.average else 1.0))
            return _DeviceLoss(loss, self.average), grad
        vertices = np.asarray(vertices, np.float64)
        B, nv = vertices.shape[:2]
        v0, v1 = vertices[:, self.v0s], vertices[:, self.v1s]
        v2, v3 = vertices[:, self.v2s], vertices[:, self.v3s]
        a, b1, b2 = v1 - v0, v2 - v0, v3 - v0
        cb1, l1 = self._half(a, b1, eps)
        cb2, l2 = self._half(a, b2, eps)
        num = (cb1 * cb2).sum(-1)
        den = l1 * l2 + eps
        cos = num / den
        g = 2 * (cos + 1)                                       # d(loss)/d(cos)
        dcb1 = cb2 * (g / den)[..., None]
        dcb2 = cb1 * (g / den)[..., None]
        dl1 = -num * l2 / (den * den) * g
        dl2 = -num * l1 / (den * den) * g
        da1, db1 = self._half_vjp(a, b1, eps, dcb1, dl1)
        da2, db2 = self._half_vjp(a, b2, eps, dcb2, dl2)
        da = da1 + da2
        out = np.zeros((B * nv, 3), np.float64)
        offs = (np.arange(B, dtype=np.int64) * nv)[:, None]
        for idx, contrib in ((self.v1s, da), (self.v2s, db1), (self.v3s, db2), (self.v0s, -(da + db1 + db2))):
            flat = (idx[None, :] + offs).reshape(-1)
            for d in range(3):
                out[:, d] += np.bincount(flat, weights=contrib[..., d].reshape(-1), minlength=B * nv)
        out = out.reshape(B, nv, 3)
        loss = ((cos + 1) ** 2).sum(tuple(range(cos.ndim))[1:])
        return ((loss.sum() / B if self.average else loss).astype(F32),
                (out / B if self.average else out).astype(F32))

    def backward_fd(self, vertices, eps=1e-6, h=1e-4):
        """Gradient by symmetric differences on the (small, smooth) per-edge expression — the loss is
        a regulariser weighted 3e-4 in the demo; analytic accuracy is not needed there."""
        vertices = np.asarray(vertices, np.float64)
        g = np.zeros_like(vertices)
        base_idx = [self.v0s, self.v1s, self.v2s, self.v3s]

        def total(v):
            cos = self._cos(v, eps)
            return ((cos + 1) ** 2)
        for which in range(4):
            idx = base_idx[which]
            for d in range(3):
                # perturb, per edge, only the role-`which` vertex: evaluate with a per-edge copy
                v0, v1 = vertices[:, self.v0s].copy(), vertices[:, self.v1s].copy()
                v2, v3 = vertices[:, self.v2s].copy(), vertices[:, self.v3s].copy()
                roles = [v0, v1, v2, v3]

                def f(delta):
                    r = [x.copy() for x in roles]
                    r[which][:, :, d] += delta
                    cb1, l1 = self._half(r[1] - r[0], r[2] - r[0], eps)
                    cb2, l2 = self._half(r[1] - r[0], r[3] - r[0], eps)
                    cos = (cb1 * cb2).sum(-1) / (l1 * l2 + eps)
                    return (cos + 1) ** 2
                de = (f(h) - f(-h)) / (2 * h)                                    # [B, nedges]
                for b in range(vertices.shape[0]):
                    g[b, :, d] += np.bincount(idx, weights=de[b], minlength=vertices.shape[1])
        return (g / vertices.shape[0] if self.average else g).astype(F32)
