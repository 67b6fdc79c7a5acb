"""Worker of tests/test_parallel.py::test_rendezvous_under_the_driver_s_launcher: started by `python -m torch.distributed.run`
exactly as the driver starts bench.py for N > 1, on a box without GPUs.  It runs the REAL rendezvous code of
jrender_amd.comm.RcclCommunicator (rank 0 publishes the unique id, the others poll for it, rank 0 removes the file after the
collective create) around a stand-in for libjrender_hip.so's jr_comm_* entry points, then the host communicator's collectives."""
import ctypes as C
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np                                   # noqa: E402
from jrender_amd import _ffi, comm as jcomm          # noqa: E402


class FakeLib:
    """jr_comm_unique_id: 128 random bytes; jr_comm_create: COLLECTIVE like ncclCommInitRank (returns when every rank has
    called it) - through marker files next to the rendezvous prefix."""

    def __init__(self, prefix):
        self.prefix, self.seen_id, self.world, self.rank = prefix, None, None, None

    def jr_comm_unique_id(self, buf):
        C.memmove(buf, os.urandom(jcomm.ID_BYTES), jcomm.ID_BYTES)
        return 0

    def jr_comm_create(self, ctx_handle, buf, world, rank, out):
        self.seen_id, self.world, self.rank = bytes(buf), int(world), int(rank)
        open("%s.joined.%d" % (self.prefix, rank), "w").close()
        t0 = time.time()
        while len(glob.glob(self.prefix + ".joined.*")) < world:
            if time.time() - t0 > 60:
                return 1
            time.sleep(0.01)
        return 0

    def jr_comm_size(self, h):
        return self.world

    def jr_comm_rank(self, h):
        return self.rank

    def jr_comm_destroy(self, h):
        return 0

    def jr_last_error(self):
        return b"stand-in: the collective create timed out"


class FakeCtx:
    handle, device = None, 0


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    prefix = jcomm.rendezvous_path()
    fake = FakeLib(prefix)
    _ffi.load = lambda: fake
    c = jcomm.RcclCommunicator(FakeCtx(), rank, world)
    rec = {"rank": c.rank, "world": c.world, "local_rank": int(os.environ["LOCAL_RANK"]), "id_sha256": hashlib.sha256(fake.seen_id).hexdigest(),
           "prefix": prefix, "ppid": os.getppid(), "master_port": os.environ.get("MASTER_PORT"), "run_id": os.environ.get("TORCHELASTIC_RUN_ID")}
    # ... and the host communicator (what ranks that share GPUs talk through) under the same launcher
    h = jcomm.init_from_env(None, backend="host")
    s = h.all_reduce_sum_host(np.full(5, rank + 1, np.float32))
    g = h.all_gather(np.full((1, 3), rank, np.float32), world)
    rec.update(sum=s.tolist(), gathered=np.asarray(g).tolist(), max=h.all_reduce_max(float(rank)))
    h.barrier()
    rec["id_file_left"] = os.path.exists(prefix + ".id")
    h.barrier()
    h.close()
    sys.stdout.flush()
    os.write(1, ("WORKER " + json.dumps(rec) + "\n").encode())        # ONE write: three workers share the agent's stdout pipe


if __name__ == "__main__":
    main()
