"""Batch sharding of SoftRas renders over the GPUs of one node (one process per GPU).

The reference has no distributed code at all (grep for nccl/mpi/rank -> nothing).
Every view of a batch is independent in the forward and in the backward (all kernels index the
batch element independently, SRK:278, :1216), so the batch dimension shards with no data-path
collective inside the op.  What a caller may want afterwards is an exchange step:

  * ``all_gather``   rendered images (or per-view gradients) of every rank, e.g. to write one
                     image grid or to evaluate a loss that couples views;
  * ``all_reduce``   the gradient of vertices that are SHARED by all views (mesh deformation).

``torch.distributed`` is used purely as the communicator: backend "nccl" (= RCCL over xGMI on
ROCm) for device buffers, "gloo" for host arrays / CPU tests.  Device buffers are handed over
zero-copy through ``__cuda_array_interface__``; nothing else of PyTorch is touched.
"""
import numpy as np

from . import _ffi

__all__ = ["shard_bounds", "BatchShards", "ShardedSoftRasterizer"]


def shard_bounds(batch, world_size):
    """Contiguous, balanced split of ``batch`` items over ``world_size`` ranks -> list of (lo, hi)."""
    base, extra = divmod(int(batch), int(world_size))
    out, lo = [], 0
    for r in range(world_size):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def _dist():
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return None
    return dist


class BatchShards:
    """Collectives over the leading (batch) axis.  Works un-initialised as world_size 1."""

    def __init__(self, group=None):
        d = _dist()
        self.group = group
        self.rank = d.get_rank(group) if d else 0
        self.world = d.get_world_size(group) if d else 1

    def bounds(self, batch):
        return shard_bounds(batch, self.world)

    def local(self, array, batch=None):
        """This rank's slice of a full-batch host array."""
        lo, hi = self.bounds(array.shape[0] if batch is None else batch)[self.rank]
        return array[lo:hi]

    def _as_tensor(self, x):
        import torch
        if isinstance(x, _ffi.DeviceArray):
            x.ctx.synchronize()                       # our stream -> torch's stream hand-off
            return torch.as_tensor(x, device="cuda:%d" % x.ctx.device), True
        return torch.from_numpy(np.ascontiguousarray(x)), False

    def all_gather(self, local, batch):
        """Concatenate every rank's ``local`` [b_r, ...] along axis 0 -> NumPy [batch, ...] on all
        ranks (shards may be uneven; they are padded to the largest for the collective)."""
        if self.world == 1:
            return np.asarray(local)
        import torch
        d = _dist()
        bounds = self.bounds(batch)
        width = max(hi - lo for lo, hi in bounds)
        t, on_gpu = self._as_tensor(local)
        item = tuple(t.shape[1:])
        pad = torch.zeros((width,) + item, dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        out = torch.empty((self.world * width,) + item, dtype=t.dtype, device=t.device)
        d.all_gather_into_tensor(out, pad, group=self.group)
        if on_gpu:
            torch.cuda.synchronize()
        out = out.cpu().numpy().reshape((self.world, width) + item)
        return np.concatenate([out[r, : hi - lo] for r, (lo, hi) in enumerate(bounds)], axis=0)

    def all_reduce_sum(self, x):
        """Sum a (small) host array over ranks — e.g. the gradient of vertices shared by all views."""
        if self.world == 1:
            return np.asarray(x)
        import torch
        d = _dist()
        t = torch.from_numpy(np.ascontiguousarray(x).copy())
        if d.get_backend(self.group) == "nccl":
            t = t.cuda()
        d.all_reduce(t, op=d.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()


class ShardedSoftRasterizer:
    """Renders a global batch [B, NF, 3, 3] with each rank rasterising its contiguous slice.

    ``make_function`` builds the local op (default: the HIP ``SoftRasterizeFunction``); tests plug
    in a CPU function with the same ``execute`` / ``grad`` protocol.
    """

    def __init__(self, make_function=None, group=None, **op_kwargs):
        if make_function is None:
            from .renderer.dr.softras.soft_rasterize import SoftRasterizeFunction
            make_function = lambda **kw: SoftRasterizeFunction(**kw)   # noqa: E731
        self.shards = BatchShards(group)
        self.fn = make_function(**op_kwargs)
        self._batch = None

    def forward_local(self, face_vertices, textures):
        """Full-batch host inputs -> this rank's images [b_r, 4, IS, IS] (device or host array)."""
        self._batch = face_vertices.shape[0]
        fv = self.shards.local(np.asarray(face_vertices, np.float32))
        tex = self.shards.local(np.asarray(textures, np.float32))
        return self.fn.execute(fv, tex)

    def forward(self, face_vertices, textures):
        """-> images of the WHOLE batch on every rank (all-gather of the image shards)."""
        local = self.forward_local(face_vertices, textures)
        return self.shards.all_gather(local, self._batch)

    def backward_local(self, grad_images_full):
        """Full-batch upstream gradient -> this rank's (grad_face_vertices, grad_textures)."""
        g = self.shards.local(np.asarray(grad_images_full, np.float32), self._batch)
        return self.fn.grad(g)

    def backward(self, grad_images_full):
        """-> per-view gradients of the whole batch on every rank (all-gather of the shards)."""
        gf, gt = self.backward_local(grad_images_full)
        return self.shards.all_gather(gf, self._batch), self.shards.all_gather(gt, self._batch)

    def backward_shared_vertices(self, grad_images_full, faces, num_vertices):
        """Views share one vertex set: scatter per-face gradients to vertices locally, sum over the
        rank's views, all-reduce over ranks -> [NV, 3] on every rank."""
        from .structures.mesh import face_vertices_backward
        gf, _ = self.backward_local(grad_images_full)
        gf = np.asarray(gf).reshape(-1, np.asarray(faces).shape[-2], 3, 3)
        if gf.shape[0]:
            fb = np.broadcast_to(np.asarray(faces).reshape(1, -1, 3), (gf.shape[0], gf.shape[1], 3))
            gv = face_vertices_backward(gf, fb, num_vertices).sum(0)
        else:
            gv = np.zeros((num_vertices, 3), np.float32)
        return self.shards.all_reduce_sum(gv)
