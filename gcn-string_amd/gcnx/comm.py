"""One process per GPU: RCCL communicator over the node's xGMI links, bootstrapped without MPI.

The reference is single-device, so this has no reference behaviour to mirror
except "N-GPU result == 1-GPU result".  Ranks are started either by ``bench.py --gpus N`` itself (it spawns one
process per GPU before anything touches the GPU) or by ``python -m torch.distributed.run``; both only set RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment -- torch is never imported here, so only one HIP runtime
lives in the process.  Rank 0 creates the RCCL unique id and publishes it through an atomically renamed file whose
name is keyed by an id the LAUNCHER hands to every rank: ``GCNX_RUN_ID`` (a fresh uuid per launch: bench.py's own
launcher, or any other launcher that exports it), else torchrun's run id + restart count + MASTER_PORT together with
the elastic agent's pid (the one parent torchrun guarantees its workers share), else MASTER_PORT alone (launchers that put
a wrapper shell around every rank share no parent); a file older than the launch is never accepted.  Single node by contract.
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


def _rendezvous_path(env=None):
    env = os.environ if env is None else env
    port = env.get("MASTER_PORT", "0")
    run_id = env.get("GCNX_RUN_ID")
    if run_id:                                       # handed down by the launcher: unique per launch
        key = f"{run_id}_{port}"
    elif "TORCHELASTIC_RUN_ID" in env:               # torchrun: its workers are children of ONE elastic agent
        key = f"{env['TORCHELASTIC_RUN_ID']}_{env.get('TORCHELASTIC_RESTART_COUNT', '0')}_{port}_{os.getppid()}"
    else:                                            # some other launcher that exported neither (mpirun / srun behind per-rank
        key = f"port{port}"                          # wrapper shells: no common parent pid to key on) -- the ranks meet on the port;
    return os.path.join(tempfile.gettempdir(), f"gcnx_uid_{key}")   # a crashed earlier run's file is told apart by its age (below)


_T_IMPORT = time.time()          # this rank's start, to within the interpreter's start-up
_STALE_SLACK_S = 20.0            # ranks of one launch start within this of each other


def _rccl_unique_id():
    lib = L.load()
    buf = C.create_string_buffer(L.UNIQUE_ID_BYTES)
    L.check(lib.gcnx_comm_unique_id(buf))
    return buf.raw


def exchange_unique_id(rank, world_size, timeout_s=120.0, path=None, make_id=_rccl_unique_id):
    """Rank 0 writes the 128-byte id; the others poll for it."""
    path = path or _rendezvous_path()
    if rank == 0:
        raw = make_id()
        tmp = f"{path}.tmp{os.getpid()}"
        with open(tmp, "wb") as fh:
            fh.write(raw)
        os.replace(tmp, path)                        # atomically over whatever an earlier run left under this key
        return raw
    t0 = time.time()
    while True:
        try:
            # a file older than this launch is a crashed earlier run's (rank 0 removes its file after the first barrier; a run
            # that died before that leaves it): never this launch's id, whose file is written after the ranks have started
            if os.path.getmtime(path) >= _T_IMPORT - _STALE_SLACK_S:
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
    capturable = True      # ncclAllReduce on the ctx stream can be recorded into the step's HIP graph

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
