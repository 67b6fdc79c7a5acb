"""N>1 path on CPU: world_size-2 `gloo` process group, the sharding/collective logic of
jrender_amd.parallel driven with a CPU stand-in for the local op (the oracle — tests only)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds():
    from jrender_amd.parallel import shard_bounds
    assert shard_bounds(8, 2) == [(0, 4), (4, 8)]
    assert shard_bounds(7, 3) == [(0, 3), (3, 5), (5, 7)]
    assert shard_bounds(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    for B in range(1, 20):
        for W in range(1, 9):
            b = shard_bounds(B, W)
            assert b[0][0] == 0 and b[-1][1] == B and all(x[1] == y[0] for x, y in zip(b, b[1:]))


class _OracleFunction:
    """execute/grad protocol of SoftRasterizeFunction, computed by the CPU oracle."""

    def __init__(self, **kw):
        from oracle import Oracle
        self.kw, self.orc = kw, Oracle("port", nthreads=2)

    def execute(self, fv, tex):
        if fv.shape[0] == 0:
            self.saved = None
            IS = self.kw.get("image_size", 256)
            return np.zeros((0, 4, IS, IS), np.float32)
        self.saved = self.orc.forward(fv, tex, **self.kw)
        return self.saved["soft_colors"]

    def grad(self, g):
        if self.saved is None:
            return np.zeros((0, 1, 3, 3), np.float32), np.zeros((0, 1, 1, 3), np.float32)
        return self.orc.backward(self.saved, g)


def _worker(rank, world, port, B, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jrender_amd import synthetic as syn
    from jrender_amd.parallel import ShardedSoftRasterizer
    fv, tex = syn.sphere_views(280, B)
    kw = dict(image_size=24)
    sh = ShardedSoftRasterizer(make_function=lambda **k: _OracleFunction(**k), **kw)
    images = sh.forward(fv, tex)
    g = np.random.default_rng(0).uniform(-1, 1, images.shape).astype(np.float32)
    gf, gt = sh.backward(g)
    verts, faces = syn.sphere_mesh(280)
    gv = sh.backward_shared_vertices(g, faces, verts.shape[0])
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), images=images, gf=gf, gt=gt, gv=gv)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [4, 3])
def test_gloo_world2_matches_single_process(tmp_path, B):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000) + B
    mp.spawn(_worker, args=(2, port, B, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    from jrender_amd import synthetic as syn
    from jrender_amd.structures.mesh import face_vertices_backward
    fv, tex = syn.sphere_views(280, B)
    one = _OracleFunction(image_size=24)
    ref = one.execute(fv, tex)
    g = np.random.default_rng(0).uniform(-1, 1, ref.shape).astype(np.float32)
    gf, gt = one.grad(g)
    for r in (r0, r1):
        assert np.array_equal(r["images"], ref)          # every rank holds the whole batch
        assert np.array_equal(r["gf"], gf) and np.array_equal(r["gt"], gt)
    verts, faces = syn.sphere_mesh(280)
    gv = face_vertices_backward(gf, np.broadcast_to(faces[None], (B,) + faces.shape), verts.shape[0]).sum(0)
    assert np.allclose(r0["gv"], gv, rtol=1e-5, atol=1e-6) and np.array_equal(r0["gv"], r1["gv"])
