"""Batch sharding of SoftRas renders over the GPUs of one node (one process per GPU).

The reference has no distributed code at all (grep for nccl/mpi/rank -> nothing).
Every view of a batch is independent in the forward and in the backward (all kernels index the
batch element independently, SRK:278, :1216), so the batch dimension shards with no data-path
collective inside the op.  What a caller may want afterwards is an exchange step:

  * ``all_gather``   rendered images (or per-view gradients) of every rank, e.g. to write one
                     image grid or to evaluate a loss that couples views;
  * ``all_reduce``   the gradient of vertices that are SHARED by all views (mesh deformation,
                     demo2-deform.py:45 repeats one vertex set over the batch).

The communicator is ``jrender_amd.comm``: RCCL over xGMI through the C ABI for device arrays
(device to device, on the context's stream), a local-socket communicator for host arrays (CPU
tests, several ranks on one GPU).  No PyTorch anywhere.
"""

import numpy as np

from . import _ffi
from . import comm as _comm

__all__ = ["shard_bounds", "BatchShards", "ShardedSoftRasterizer", "launch_ranks"]


def shard_bounds(batch, world_size):
    """Contiguous, balanced split of ``batch`` items over ``world_size`` ranks -> list of (lo, hi)."""
    base, extra = divmod(int(batch), int(world_size))
    out, lo = [], 0
    for r in range(world_size):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


class BatchShards:
    """Collectives over the leading (batch) axis.  Without a communicator: world size 1."""

    def __init__(self, comm=None):
        self.comm = comm if comm is not None else _comm.SingleCommunicator()
        self.rank, self.world = self.comm.rank, self.comm.world

    def bounds(self, batch):
        return shard_bounds(batch, self.world)

    def local(self, array, batch=None):
        """This rank's slice of a full-batch array (host array or DeviceArray)."""
        lo, hi = self.bounds(array.shape[0] if batch is None else batch)[self.rank]
        if isinstance(array, _ffi.DeviceArray):
            return array.view(lo, hi)
        return array[lo:hi]

    def all_gather(self, local, batch):
        """Every rank's ``local`` [b_r, ...] along axis 0 -> [batch, ...] on all ranks.  A
        DeviceArray stays on the device (RCCL), a host array stays on the host."""
        return self.comm.all_gather(local, batch)

    def all_reduce_sum(self, x):
        """Sum over ranks (DeviceArray: in place on the device)."""
        return self.comm.all_reduce_sum(x)


class ShardedSoftRasterizer:
    """Renders a global batch [B, NF, 3, 3] with each rank rasterising its contiguous slice.

    ``make_function`` builds the local op (default: the HIP ``SoftRasterizeFunction``); tests plug
    in a CPU function with the same ``execute`` / ``grad`` protocol.
    """

    def __init__(self, make_function=None, comm=None, **op_kwargs):
        if make_function is None:
            from .renderer.dr.softras.soft_rasterize import SoftRasterizeFunction
            make_function = lambda **kw: SoftRasterizeFunction(**kw)   # noqa: E731
        self.shards = BatchShards(comm)
        self.fn = make_function(**op_kwargs)
        self._batch = None

    def _slice(self, x, batch=None):
        if isinstance(x, _ffi.DeviceArray):
            return self.shards.local(x, batch)
        return self.shards.local(np.asarray(x, np.float32), batch)

    def forward_local(self, face_vertices, textures):
        """Full-batch inputs (host arrays or DeviceArrays) -> this rank's images [b_r, 4, IS, IS]."""
        self._batch = face_vertices.shape[0]
        return self.fn.execute(self._slice(face_vertices), self._slice(textures))

    def forward(self, face_vertices, textures):
        """-> images of the WHOLE batch on every rank (all-gather of the image shards)."""
        local = self.forward_local(face_vertices, textures)
        return self.shards.all_gather(local, self._batch)

    def backward_local(self, grad_images_full):
        """Full-batch upstream gradient -> this rank's (grad_face_vertices, grad_textures)."""
        return self.fn.grad(self._slice(grad_images_full, self._batch))

    def backward(self, grad_images_full):
        """-> per-view gradients of the whole batch on every rank (all-gather of the shards)."""
        gf, gt = self.backward_local(grad_images_full)
        return self.shards.all_gather(gf, self._batch), self.shards.all_gather(gt, self._batch)

    def backward_shared_vertices(self, grad_images_full, faces, num_vertices):
        """Views share one vertex set: scatter per-face gradients to vertices locally, sum over the
        rank's views, all-reduce over ranks -> [NV, 3] on every rank (DeviceArray on the HIP path:
        scatter kernel + ncclAllReduce, nothing leaves the device)."""
        gf, _ = self.backward_local(grad_images_full)
        return shared_vertex_gradient(gf, faces, num_vertices, self.shards.comm)


def shared_vertex_gradient(grad_faces, faces, num_vertices, comm=None):
    """Σ over local views of the face→vertex scatter-add of ``grad_faces`` [b, NF, 3, 3], then the
    sum over ranks.  DeviceArray in -> DeviceArray [NV, 3] out (device scatter + RCCL all-reduce);
    host array in -> host array out."""
    comm = comm if comm is not None else _comm.SingleCommunicator()
    if isinstance(grad_faces, _ffi.DeviceArray):
        ctx = grad_faces.ctx
        faces_d = faces if isinstance(faces, _ffi.DeviceArray) else ctx.array(np.asarray(faces, np.int32).reshape(-1, 3))
        nf = faces_d.size // 3
        gv = ctx.empty((int(num_vertices), 3), np.float32)
        _ffi._check(_ffi.load().jr_face_vertices_backward_shared(
            ctx.handle, grad_faces.ptr, faces_d.ptr, gv.ptr, int(grad_faces.shape[0]), int(num_vertices), int(nf)))
        return comm.all_reduce_sum(gv)
    from .structures.mesh import face_vertices_backward
    faces = np.asarray(faces)
    gf = np.asarray(grad_faces).reshape(-1, faces.shape[-2], 3, 3)
    if gf.shape[0]:
        fb = np.broadcast_to(faces.reshape(1, -1, 3), (gf.shape[0], gf.shape[1], 3))
        gv = face_vertices_backward(gf, fb, num_vertices).sum(0)
    else:
        gv = np.zeros((num_vertices, 3), np.float32)
    return comm.all_reduce_sum(gv.astype(np.float32))


def launch_ranks(script, n, argv, timeout_s=1500.0, name=None, relay_stdout=True):
    """Start ``n`` ranks of ``script`` (one process per GPU, each its own process group) with a private 0700 rendezvous
    directory, relay rank 0's stdout.  ALL children are polled: if any rank exits non-zero (or the whole launch exceeds
    ``timeout_s``) before the others are done, the rest is killed and the launcher returns non-zero after printing the tail
    of the failing rank's stderr (every rank's stderr goes to a file in the rendezvous directory) - a rank that dies before
    ``ncclCommInitRank`` completes must not leave the others waiting for ever.  Used by ``bench.py --gpus N`` and
    ``examples/demo2_deform.py --gpus N``.  -> 0 on success, 1 on failure."""
    import os
    import signal
    import subprocess
    import sys
    import tempfile
    import time
    name = name or os.path.basename(script)
    rdzv = tempfile.mkdtemp(prefix="jrender_%s_" % os.path.splitext(name)[0])
    procs, logs = [], []
    for r in range(n):
        # HSA_ENABLE_IPC_MODE_LEGACY=0: the MI355X host driver of this pool only supports dmabuf IPC; RCCL's
        # cross-process buffer registration fails with `hipIpcGetMemHandle: invalid argument` without it.  An
        # explicit setting in the caller's environment wins.
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   JRENDER_RDZV=os.path.join(rdzv, "rdzv"), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        out = open(os.path.join(rdzv, "rank%d.out" % r), "w+b")
        err = open(os.path.join(rdzv, "rank%d.err" % r), "w+b")
        logs.append((out, err))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env,
                                      stdout=out, stderr=err, start_new_session=True))

    def tail(f, nbytes=3000):
        f.flush(); f.seek(0, 2); size = f.tell(); f.seek(max(0, size - nbytes))
        return f.read().decode(errors="replace")

    def kill_all():
        for q in procs:
            if q.poll() is None:
                try:
                    os.killpg(q.pid, signal.SIGKILL)       # the rank's own process group (start_new_session)
                except OSError:
                    pass
        for q in procs:
            try:
                q.wait(timeout=10)
            except Exception:
                pass

    # rank 0's stdout is relayed WHILE the ranks run (a progress line every 20 iterations of an optimisation loop must not
    # wait for the end of the launch): a second handle on its log file is drained in the poll loop
    relay = open(os.path.join(rdzv, "rank0.out"), "rb") if relay_stdout else None

    def drain():
        if relay is None:
            return
        chunk = relay.read()
        if chunk:
            sys.stdout.write(chunk.decode(errors="replace"))
            sys.stdout.flush()

    failure, t0 = None, time.time()
    while failure is None and any(q.poll() is None for q in procs):
        drain()
        for r, q in enumerate(procs):
            rc = q.poll()
            if rc not in (None, 0):
                failure = "rank %d exited with code %d" % (r, rc)
                break
        else:
            if time.time() - t0 > timeout_s:
                failure = "launch exceeded %.0f s" % timeout_s
            else:
                time.sleep(0.05)
    if failure is None:
        bad = [(r, q.returncode) for r, q in enumerate(procs) if q.returncode]
        if bad:
            failure = "rank %d exited with code %d" % bad[0]
    if failure is not None:
        kill_all()
        drain()
        sys.stderr.write("%s: %s; exit codes %s\n" % (name, failure, [q.returncode for q in procs]))
        for r, (_o, e) in enumerate(logs):
            t = tail(e).strip()
            if t:
                sys.stderr.write("---- rank %d stderr (tail) ----\n%s\n" % (r, t))
        sys.stderr.write("%s: the ranks' complete logs are kept in %s (rank<i>.out / rank<i>.err)\n" % (name, rdzv))
    else:
        for o, _e in logs:
            o.flush()
        drain()
        for r, (_o, e) in enumerate(logs):                 # warnings of healthy ranks stay visible
            t = tail(e).strip()
            if t:
                sys.stderr.write("---- rank %d stderr ----\n%s\n" % (r, t))
    for o, e in logs:
        o.close(); e.close()
    if relay is not None:
        relay.close()
    if failure is not None:                                # the logs stay for the post-mortem (their directory is private: 0700)
        return 1
    try:
        for f in os.listdir(rdzv):
            os.unlink(os.path.join(rdzv, f))
        os.rmdir(rdzv)
    except OSError:
        pass
    return 1 if failure is not None else 0
