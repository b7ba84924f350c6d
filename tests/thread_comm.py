"""Test infrastructure: a communicator between THREADS of one process that share one GPU.

The box has a single GPU, so RCCL cannot connect two ranks there.  The device-side logic of the multi-GPU step --
sharding, loss normalisation by the global batch, which buffers are all-reduced and when -- does not depend on the
transport, though: ``ThreadCommunicator`` has the interface of ``gcnx.comm.Communicator`` and sums / maxes device
buffers across the ranks of a ``ThreadWorld`` through the host (d2h, barrier, reduce in rank order, h2d).  Every rank
has its own ``gcnx.Context`` (its own HIP streams) on device 0.
"""
import threading

import numpy as np


class ThreadWorld:
    def __init__(self, world_size):
        self.world_size = world_size
        self.barrier = threading.Barrier(world_size)
        self.slots = [None] * world_size
        self.errors = []

    def run(self, fn):
        """fn(rank, comm_factory) in world_size threads; re-raises the first failure."""
        results = [None] * self.world_size

        def work(rank):
            try:
                results[rank] = fn(rank, lambda ctx: ThreadCommunicator(self, ctx, rank))
            except BaseException as e:            # noqa: BLE001 -- reported to the main thread
                self.errors.append(e)
                self.barrier.abort()
        threads = [threading.Thread(target=work, args=(r,)) for r in range(self.world_size)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if self.errors:
            raise self.errors[0]
        return results


class ThreadCommunicator:
    capturable = False     # host-mediated: cannot be recorded into a HIP graph

    def __init__(self, world, ctx, rank):
        self.world, self.ctx, self.rank, self.world_size = world, ctx, rank, world.world_size
        self.calls = 0

    def _reduce(self, host, op):
        w = self.world
        w.slots[self.rank] = host
        w.barrier.wait()
        parts = [np.asarray(s, np.float32) for s in w.slots]
        out = parts[0].copy()
        for p in parts[1:]:                       # rank order: the same sum on every rank
            out = out + p if op == "sum" else np.maximum(out, p)
        w.barrier.wait()                          # everyone has read the slots before they are reused
        return out

    def allreduce_sum(self, arr, n=None):
        self.calls += 1
        n = n or arr.size
        view = arr if n == arr.size else arr.flat(0, n)
        red = self._reduce(view.numpy().ravel(), "sum")
        view.copy_from_host(red.reshape(view.shape))

    def allreduce_host(self, values, op="max"):
        return self._reduce(np.asarray(values, np.float32).ravel(), op)

    def barrier(self):
        self.ctx.sync()
        self.world.barrier.wait()

    def close(self):
        pass
