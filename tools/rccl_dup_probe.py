#!/usr/bin/env python3
"""Does RCCL accept two ranks on ONE GPU?  (1-GPU boxes: decides whether a world-2 RCCL test can run there.)"""
import multiprocessing as mp
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rank(r, path, q):
    sys.path.insert(0, ROOT)
    try:
        import numpy as np
        from jrender_amd import _ffi, comm
        ctx = _ffi.Context(0)
        cm = comm.RcclCommunicator(ctx, r, 2, path=path, timeout=60)
        x = ctx.array(np.full((4,), r + 1, np.float32))
        out = cm.all_gather(x, 8).numpy()
        s = cm.all_reduce_sum(x).numpy()
        cm.close()
        q.put((r, "ok", out.tolist(), s.tolist()))
    except Exception as e:
        q.put((r, "error", repr(e)))


if __name__ == "__main__":
    c = mp.get_context("spawn")
    q = c.Queue()
    path = os.path.join(tempfile.mkdtemp(), "rdzv")
    ps = [c.Process(target=rank, args=(r, path, q)) for r in range(2)]
    [p.start() for p in ps]
    for p in ps:
        p.join(120)
        if p.is_alive():
            p.terminate()
            print("rank hung")
    while not q.empty():
        print(q.get())
