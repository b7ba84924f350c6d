"""Device-resident dataset and loader.

``DisjointLoader`` (gcnx.loader) assembles every batch on the host -- vstack / block_diag / find, as Spektral does
for gcn.py:316-317 -- and ``DeviceBatch.from_host`` uploads it.  Once the kernels are fast that host work and the
PCIe copy dominate an epoch.  Here the whole dataset is uploaded ONCE as one disjoint union (features, CSR with
gcn_filter already applied per graph -- the filter of a block-diagonal matrix is the block-diagonal of the filters --
labels, node offsets); a batch is then one ``gcnx_collate`` launch that gathers the selected graphs' rows and
re-bases their indices, into buffers sized for the largest possible batch and reused for every batch.  Per batch
only 3(B+1) ints cross PCIe.  Iteration order, shuffling and batch boundaries are those of ``DisjointLoader``.
"""
from __future__ import annotations

import math

import numpy as np

from . import device as D
from .loader import collate_disjoint
from .models import DeviceBatch


class DeviceDataset:
    """All graphs of a ``Dataset`` resident in HBM."""

    def __init__(self, ctx, dataset, normalize=None, weighted=True, symmetric=None):
        self.ctx = ctx
        graphs = [dataset[i] for i in range(len(dataset))]
        (x, a, i), y = collate_disjoint(graphs)
        sizes = np.array([g.n_nodes for g in graphs], np.int64)
        self.n_graphs = len(graphs)
        self.node_ptr_host = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        n = int(self.node_ptr_host[-1])
        seg = D.Segments(ctx, self.node_ptr_host)
        csr = D.DeviceCSR.from_coo(ctx, a.indices, a.values, n, graph_ptr=seg.host, symmetric=symmetric, weighted=weighted)
        if normalize:
            csr = csr.gcn_norm(normalize)
        self.csr, self.symmetric = csr, csr.symmetric    # checked once on the union (symmetric=None); batches inherit it
        rows = np.asarray(a.indices)[:, 0]
        rowptr_host = np.zeros(n + 1, np.int64)
        np.cumsum(np.bincount(rows, minlength=n), out=rowptr_host[1:])
        self.ent_ptr_host = rowptr_host[self.node_ptr_host]          # entry offset of every graph
        self.x = ctx.to_device(x, np.float32)
        y = np.asarray(y, np.float32)
        self.y = ctx.to_device(y.reshape(self.n_graphs, -1), np.float32)
        self.node_ptr = seg.dev
        self.n_features, self.n_labels = self.x.shape[1], self.y.shape[1]
        self.sizes = sizes
        self.nnz_sizes = np.diff(self.ent_ptr_host)

    def __len__(self):
        return self.n_graphs

    def capacity(self, batch_size):
        """Rows / entries of the largest batch of `batch_size` graphs."""
        b = min(batch_size, self.n_graphs)
        return int(np.sort(self.sizes)[-b:].sum()), int(np.sort(self.nnz_sizes)[-b:].sum())


class _BatchBuffers:
    def __init__(self, ctx, ds, batch_size):
        ncap, ecap = ds.capacity(batch_size)
        self.x = ctx.empty((ncap, ds.n_features))
        self.rowptr = ctx.empty(ncap + 1, np.int32)
        self.colidx = ctx.empty(max(ecap, 1), np.int32)
        self.vals = ctx.empty(max(ecap, 1), np.float32) if ds.csr.vals is not None else None
        self.y = ctx.empty((batch_size, ds.n_labels))
        self.gp = ctx.empty(batch_size + 1, np.int32)
        self.ids = ctx.empty(max(ncap, 1), np.int32)     # DisjointLoader's id vector i, written by the collate launch
        self.desc = ctx.empty(3 * (batch_size + 1), np.int32)


def collate_on_device(ds, indices, bufs=None):
    """The DeviceBatch of the graphs `indices` (dataset order positions), assembled by gcnx_collate."""
    ctx = ds.ctx
    sel = np.asarray(indices, np.int64)
    b = len(sel)
    bufs = bufs or _BatchBuffers(ctx, ds, b)
    bn = np.concatenate([[0], np.cumsum(ds.sizes[sel])])
    be = np.concatenate([[0], np.cumsum(ds.nnz_sizes[sel])])
    n, nnz = int(bn[-1]), int(be[-1])
    desc = np.concatenate([sel, [0], bn, be]).astype(np.int32)
    V = D.DeviceArray._view                              # (views without flat()'s checks: eight per batch)
    dview = V(bufs.desc, 0, (int(desc.size),))
    dview.copy_from_host(desc, wait=False)               # queued: the host runs ahead of the GPU across batches
    f, c = ds.n_features, ds.n_labels
    csr = ds.csr
    ctx._ck(ctx.lib.gcnx_collate(ctx.h, dview.ptr, b, ds.node_ptr.ptr, csr.rowptr.ptr, csr.colidx.ptr,
                                 csr.vals.ptr if csr.vals is not None else None, ds.x.ptr, ds.x.ld, f, ds.y.ptr, c,
                                 bufs.rowptr.ptr, bufs.colidx.ptr, bufs.vals.ptr if bufs.vals is not None else None,
                                 bufs.x.ptr, bufs.x.ld, bufs.y.ptr, bufs.gp.ptr, bufs.ids.ptr))
    seg = D.Segments.from_device(ctx, V(bufs.gp, 0, (b + 1,)), bn)
    seg._ids = V(bufs.ids, 0, (max(n, 1),))              # (otherwise built on the host on first use and uploaded)
    a = D.DeviceCSR(ctx, n, nnz, V(bufs.rowptr, 0, (n + 1,)), V(bufs.colidx, 0, (max(nnz, 1),)),
                    V(bufs.vals, 0, (max(nnz, 1),)) if bufs.vals is not None else None, seg.dev, b, ds.symmetric,
                    int(ds.sizes[sel].max()) if b else 0)
    batch = DeviceBatch(ctx, V(bufs.x, 0, (n, f)), a, seg, V(bufs.y, 0, (b, c)))
    batch._bufs = bufs                                   # keeps the capacity buffers alive with the batch
    return batch


class DeviceDisjointLoader:
    """``DisjointLoader`` over a ``DeviceDataset``: same arguments, same order of graphs (same seed -> same
    batches), but yields ``(DeviceBatch, None)`` -- the labels ride in ``batch.y`` -- assembled on the device.
    Every batch reuses one set of capacity-sized buffers: a batch is valid until the next one is drawn."""

    def __init__(self, dataset, batch_size=1, epochs=None, shuffle=True, seed=None):
        assert isinstance(dataset, DeviceDataset)
        self.dataset, self.batch_size, self.epochs, self.shuffle = dataset, int(batch_size), epochs, shuffle
        self._rng = np.random.default_rng(seed) if seed is not None else np.random
        self._bufs = _BatchBuffers(dataset.ctx, dataset, min(self.batch_size, len(dataset)))
        self._gen = self._generator()

    @property
    def steps_per_epoch(self):
        return int(math.ceil(len(self.dataset) / self.batch_size))

    def _generator(self):
        n = len(self.dataset)
        epoch = 0
        while self.epochs is None or epoch < self.epochs:
            order = np.arange(n)
            if self.shuffle:
                self._rng.shuffle(order)
            for s in range(0, n, self.batch_size):
                yield order[s:s + self.batch_size]
            epoch += 1

    def __iter__(self):
        return self

    def __next__(self):
        return collate_on_device(self.dataset, next(self._gen), self._bufs), None

    def load(self):
        return self
