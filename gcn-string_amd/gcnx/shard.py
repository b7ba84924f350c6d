"""Partition a disjoint batch by graph across ranks.

A DisjointLoader batch is block-diagonal (sp.block_diag, 8.A.1): no edge crosses graphs, the
pool and the head are per graph, so the forward pass needs no communication and each rank takes
a contiguous range of graphs.  Boundaries balance the per-graph cost nnz_g*F + n_g*F (SpMM gather
+ dense rows), not the graph count.  O(B) on the host.
"""
from __future__ import annotations

import numpy as np


def partition_graphs(graph_ptr, rowptr, world_size, f=1):
    """Returns bounds[world_size+1]: rank r owns graphs [bounds[r], bounds[r+1])."""
    gp = np.asarray(graph_ptr, dtype=np.int64)
    nodes = np.diff(gp)
    nnz = np.asarray(rowptr, dtype=np.int64)[gp[1:]] - np.asarray(rowptr, dtype=np.int64)[gp[:-1]]
    return partition_by_cost((nnz + nodes).astype(np.float64) * f, world_size)


def partition_by_cost(cost, world_size):
    """Contiguous graph ranges of (nearly) equal summed cost: bounds[world_size+1].  Needs only the per-graph costs,
    so a rank can partition a batch whose graphs it has not built (bench.py: every rank builds its own shard)."""
    cost = np.asarray(cost, dtype=np.float64)
    b = len(cost)
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    total = cum[-1]
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        g = int(np.searchsorted(cum, target, side="left"))
        # pick the nearer boundary, keep bounds monotone and leave at least one graph per rank
        if g > 0 and abs(cum[g - 1] - target) <= abs(cum[min(g, b)] - target):
            g -= 1
        g = max(g, bounds[-1] + (1 if b >= world_size else 0))
        g = min(g, b - (world_size - r) if b >= world_size else b)
        bounds.append(g)
    bounds.append(b)
    return np.asarray(bounds, dtype=np.int64)


def shard_batch(host_batch, rank, world_size):
    """This rank's shard of a synth.HostBatch plus the global number of graphs."""
    bounds = partition_graphs(host_batch.graph_ptr, host_batch.rowptr, world_size, host_batch.f)
    return host_batch.slice_graphs(int(bounds[rank]), int(bounds[rank + 1])), host_batch.n_graphs
