"""One process per GPU: RCCL communicator over the node's xGMI links, bootstrapped without MPI.

The reference is single-device, so this has no reference behaviour to mirror
except "N-GPU result == 1-GPU result".  Ranks are started by ``python -m torch.distributed.run``
(which only sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment -- torch itself is
never imported here, so only one HIP runtime lives in the process).  Rank 0 creates the RCCL
unique id and publishes it through an atomically renamed file keyed by the launcher's PID and
MASTER_PORT; all ranks of one node share that parent.
"""
from __future__ import annotations

import ctypes as C
import os
import tempfile
import time

import numpy as np

from . import _lib as L


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))),
            int(os.environ.get("WORLD_SIZE", "1")))


def _rendezvous_path():
    port = os.environ.get("MASTER_PORT", "0")
    run_id = os.environ.get("TORCHELASTIC_RUN_ID", "none")
    return os.path.join(tempfile.gettempdir(), f"gcnx_uid_{os.getppid()}_{port}_{run_id}")


def exchange_unique_id(rank, world_size, timeout_s=120.0, path=None):
    """Rank 0 writes the 128-byte id; the others poll for it."""
    lib = L.load()
    path = path or _rendezvous_path()
    if rank == 0:
        buf = C.create_string_buffer(L.UNIQUE_ID_BYTES)
        L.check(lib.gcnx_comm_unique_id(buf))
        tmp = f"{path}.tmp{os.getpid()}"
        with open(tmp, "wb") as fh:
            fh.write(buf.raw)
        os.replace(tmp, path)
        return buf.raw
    t0 = time.time()
    while True:
        try:
            with open(path, "rb") as fh:
                raw = fh.read()
            if len(raw) == L.UNIQUE_ID_BYTES:
                return raw
        except FileNotFoundError:
            pass
        if time.time() - t0 > timeout_s:
            raise TimeoutError(f"rank {rank}: no RCCL unique id at {path} after {timeout_s}s")
        time.sleep(0.01)


class Communicator:
    def __init__(self, ctx, rank, world_size, uid_path=None):
        self.ctx, self.rank, self.world_size = ctx, rank, world_size
        self.h = None
        self._path = uid_path or _rendezvous_path()
        if world_size > 1:
            uid = exchange_unique_id(rank, world_size, path=self._path)
            h = C.c_void_p()
            ctx._ck(ctx.lib.gcnx_comm_init_rank(ctx.h, uid, world_size, rank, C.byref(h)))
            self.h = h
            self._scratch = ctx.zeros(4)
            self.barrier()
            if rank == 0:
                try:
                    os.remove(self._path)
                except OSError:
                    pass

    def allreduce_sum(self, arr, n=None):
        if self.world_size > 1:
            self.ctx._ck(self.ctx.lib.gcnx_allreduce_f32(self.ctx.h, self.h, arr.ptr, n or arr.size, L.RED_SUM))

    def allreduce_host(self, values, op="max"):
        """Small host-side reduction (timings, counters) through the device."""
        v = np.asarray(values, dtype=np.float32).ravel()
        if self.world_size == 1:
            return v.copy()
        assert v.size <= 4
        pad = np.zeros(4, np.float32)
        pad[:v.size] = v
        self._scratch.copy_from_host(pad)
        self.ctx._ck(self.ctx.lib.gcnx_allreduce_f32(self.ctx.h, self.h, self._scratch.ptr, 4,
                                                     L.RED_MAX if op == "max" else L.RED_SUM))
        return self._scratch.numpy()[:v.size]

    def barrier(self):
        if self.world_size > 1:
            self.allreduce_host([0.0], "sum")
        self.ctx.sync()

    def close(self):
        if self.h:
            self.ctx.lib.gcnx_comm_destroy(self.h)
            self.h = None
