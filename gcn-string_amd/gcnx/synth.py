"""Synthetic workloads of BASELINE.json's configs (no E. coli data exists outside the authors'
cluster: src/scripts/gcn_generator.py:13-27 hard-codes /mnt/mnemo6 paths).

Graph statistics follow the reference's offline builder: residue contact maps at < 10 Angstrom
*including the 0-Angstrom diagonal* (gcn_utills.py:195-227: every node has a self-loop), two
protein chains per graph joined by DCA bridge edges (gcn_utills.py:345-351), symmetric 0/1
adjacency (weights stripped, gcn.py:187-197), one-hot graph labels (gcn.py:259,262).
Pure NumPy; deterministic per seed.
"""
from __future__ import annotations

import numpy as np


class HostBatch:
    """A collated disjoint batch on the host, already in CSR (int32) form."""

    def __init__(self, x, rowptr, colidx, vals, graph_ptr, y):
        self.x, self.rowptr, self.colidx, self.vals, self.graph_ptr, self.y = x, rowptr, colidx, vals, graph_ptr, y
        self.n, self.f = x.shape
        self.nnz = int(len(colidx))
        self.n_graphs = int(len(graph_ptr) - 1)

    def indices(self):
        """DisjointLoader form: [nnz, 2] int64 (row, col), row-major."""
        rows = np.repeat(np.arange(self.n, dtype=np.int64), np.diff(self.rowptr))
        return np.stack([rows, self.colidx.astype(np.int64)], axis=1)

    def ids(self):
        return np.repeat(np.arange(self.n_graphs, dtype=np.int64), np.diff(self.graph_ptr))

    def slice_graphs(self, g0, g1):
        """Rows/edges of graphs [g0, g1) re-based to local indices (block-diagonal => no halo)."""
        r0, r1 = int(self.graph_ptr[g0]), int(self.graph_ptr[g1])
        e0, e1 = int(self.rowptr[r0]), int(self.rowptr[r1])
        return HostBatch(self.x[r0:r1], (self.rowptr[r0:r1 + 1] - e0).astype(np.int32),
                         (self.colidx[e0:e1] - r0).astype(np.int32),
                         None if self.vals is None else self.vals[e0:e1],
                         (self.graph_ptr[g0:g1 + 1] - r0).astype(np.int32), self.y[g0:g1])


def _csr_from_pairs(n, u, v):
    """Symmetric 0/1 CSR with self-loops from undirected pairs (u != v, duplicates allowed)."""
    diag = np.arange(n, dtype=np.int64)
    rows = np.concatenate([u, v, diag])
    cols = np.concatenate([v, u, diag])
    key = np.unique(rows * n + cols)          # sorts row-major and removes duplicates
    rows, cols = key // n, key % n
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=rowptr[1:])
    return rowptr.astype(np.int32), cols.astype(np.int32)


def _labels(rng, b, n_labels=2):
    lab = rng.integers(0, n_labels, size=b)
    y = np.zeros((b, n_labels), dtype=np.float32)
    y[np.arange(b), lab] = 1.0
    return y


def gcn_norm_host(rowptr, colidx, mode="spektral"):
    """fp32 values of Spektral gcn_filter on a CSR that stores every diagonal (host helper for
    building inputs; the device equivalent is gcnx_gcn_norm)."""
    n = len(rowptr) - 1
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    v = np.ones(len(colidx), dtype=np.float64)
    if mode == "spektral":
        v[rows == colidx] += 1.0
    deg = np.bincount(rows, weights=v, minlength=n)
    dinv = np.where(deg > 0, 1.0 / np.sqrt(np.maximum(deg, 1e-300)), 0.0)
    return (v * dinv[rows] * dinv[colidx]).astype(np.float32)


# ---- config 1: 16 tiny random graphs (plumbing) ------------------------------------------------

def tiny_graphs(n_graphs=16, f=32, seed=0, p=0.15, n_min=8, n_max=64):
    """List of (x float64 [n,F], a scipy CSR int64 with self-loops, y one-hot int64) -- the
    per-graph data contract of MyDataset (gcn.py:153-157)."""
    import scipy.sparse as sp

    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_graphs):
        n = int(rng.integers(n_min, n_max + 1))
        upper = np.triu(rng.random((n, n)) < p, 1)
        a = (upper | upper.T | np.eye(n, dtype=bool)).astype(np.int64)
        x = rng.standard_normal((n, f))
        y = np.zeros(2, dtype=np.int64)
        y[int(rng.integers(0, 2))] = 1
        out.append((x, sp.csr_matrix(a), y))
    return out


# ---- config 2: E. coli-shaped batch ---------------------------------------------------------------

def ecoli_graph_pairs(rng, band=3, long_range=4, bridges=20):
    """One inter-protein graph: two chains, lengths ~ clip(lognormal(ln 300, 0.5), 50, 1500);
    sequence-neighbour contacts within +-band, ~2*long_range random intra-chain contacts per
    residue (3-D contacts), `bridges` inter-chain DCA edges.  Returns (n, u, v)."""
    l1, l2 = (int(np.clip(rng.lognormal(np.log(300.0), 0.5), 50, 1500)) for _ in range(2))
    n = l1 + l2
    us, vs = [], []
    for lo, ln in ((0, l1), (l1, l2)):
        idx = np.arange(lo, lo + ln)
        for d in range(1, band + 1):
            us.append(idx[:-d]); vs.append(idx[d:])
        a = np.repeat(idx, long_range)
        b = lo + rng.integers(0, ln, size=a.size)
        keep = a != b
        us.append(a[keep]); vs.append(b[keep])
    bu = rng.integers(0, l1, size=bridges)
    bv = l1 + rng.integers(0, l2, size=bridges)
    us.append(bu); vs.append(bv)
    return n, np.concatenate(us).astype(np.int64), np.concatenate(vs).astype(np.int64)


def ecoli_batch(n_graphs=32, f=128, seed=1):
    """BASELINE config 2: B=32 graphs, ~600 nodes each, mean degree ~15 incl. self-loop."""
    rng = np.random.default_rng(seed)
    sizes, us, vs, off = [], [], [], 0
    for _ in range(n_graphs):
        n, u, v = ecoli_graph_pairs(rng)
        us.append(u + off); vs.append(v + off)
        sizes.append(n); off += n
    rowptr, colidx = _csr_from_pairs(off, np.concatenate(us), np.concatenate(vs))
    gp = np.zeros(n_graphs + 1, dtype=np.int32)
    np.cumsum(sizes, out=gp[1:])
    x = rng.standard_normal((off, f), dtype=np.float32)
    return HostBatch(x, rowptr, colidx, None, gp, _labels(rng, n_graphs))


# ---- config 3: 1M nodes / 10M entries, block-diagonal ----------------------------------------------

def _lognormal_sizes(rng, total, mean_size, lo, hi):
    b = max(1, int(round(total / mean_size)))
    if b * lo > total or b * hi < total:
        raise ValueError(f"{b} graphs of {lo}..{hi} nodes cannot hold {total} nodes (mean_size={mean_size})")
    s = np.clip(rng.lognormal(np.log(mean_size), 0.5, size=b), lo, hi)
    s = np.maximum(lo, np.floor(s * (total / s.sum()))).astype(np.int64)
    diff = int(total - s.sum())
    # spread the rounding remainder one node at a time
    i = 0
    while diff != 0:
        j = i % b
        if diff > 0:
            s[j] += 1; diff -= 1
        elif s[j] > lo:
            s[j] -= 1; diff += 1
        i += 1
    return s


def block_diag_batch(n=1_000_000, nnz=10_000_000, f=256, seed=2, mean_size=600, with_x=True):
    """BASELINE config 3: disjoint batch of ~n/600 graphs; column indices uniform *within the
    row's own graph block*; symmetric, self-loops, exactly `nnz` stored entries."""
    rng = np.random.default_rng(seed)
    sizes = _lognormal_sizes(rng, n, mean_size, 50, 3000)
    b = len(sizes)
    gp = np.zeros(b + 1, dtype=np.int64)
    np.cumsum(sizes, out=gp[1:])
    graph_of = np.repeat(np.arange(b), sizes)
    m_target = (nnz - n) // 2                      # undirected off-diagonal pairs
    pairs = np.empty(0, dtype=np.int64)
    while pairs.size < m_target:
        need = int((m_target - pairs.size) * 1.05) + 1024
        u = rng.integers(0, n, size=need)
        g = graph_of[u]
        lo, sz = gp[g], sizes[g]
        off = (rng.random(need) * (sz - 1)).astype(np.int64)
        v = lo + off
        v += (v >= u)
        a, c = np.minimum(u, v), np.maximum(u, v)
        pairs = np.unique(np.concatenate([pairs, a * n + c]))
    if pairs.size > m_target:
        pairs = np.sort(rng.choice(pairs, size=m_target, replace=False))
    rowptr, colidx = _csr_from_pairs(n, pairs // n, pairs % n)
    x = rng.standard_normal((n, f), dtype=np.float32) if with_x else np.zeros((n, f), np.float32)
    return HostBatch(x, rowptr, colidx, None, gp.astype(np.int32), _labels(rng, b))


# ---- shardable generators: graph g comes from its own random stream ---------------------------------
# bench.py (every N, including 1) uses these: a rank builds ONLY the graphs of its shard -- sizes of all graphs are
# cheap and generated everywhere (the partition needs them), edges and features only for the own range -- and the
# global batch is the same whatever the number of ranks (graph g depends on (seed, g) alone).

def _graph_rng(seed, g, stream=0):
    return np.random.default_rng([int(seed), int(g), int(stream)])


def ecoli_sizes(n_graphs, seed=1):
    """(n_g, nnz_g upper estimate) of every graph of the per-graph-seeded E. coli batch (cheap: two draws per graph)."""
    sizes = np.empty(n_graphs, np.int64)
    for g in range(n_graphs):
        rng = _graph_rng(seed, g)
        sizes[g] = sum(int(np.clip(rng.lognormal(np.log(300.0), 0.5), 50, 1500)) for _ in range(2))
    return sizes


def ecoli_shard(g0, g1, f=128, seed=1):
    """Graphs [g0, g1) of the per-graph-seeded BASELINE config 2 batch as one HostBatch (local indices)."""
    sizes, us, vs, xs, off = [], [], [], [], 0
    for g in range(g0, g1):
        rng = _graph_rng(seed, g)
        n, u, v = ecoli_graph_pairs(rng)               # its first two draws are the chain lengths (= ecoli_sizes)
        us.append(u + off); vs.append(v + off)
        sizes.append(n); off += n
        xs.append(rng.standard_normal((n, f), dtype=np.float32))
    rowptr, colidx = _csr_from_pairs(off, np.concatenate(us), np.concatenate(vs))
    gp = np.zeros(g1 - g0 + 1, dtype=np.int32)
    np.cumsum(sizes, out=gp[1:])
    lab = np.array([int(_graph_rng(seed, g, 1).integers(0, 2)) for g in range(g0, g1)])
    y = np.zeros((g1 - g0, 2), np.float32)
    y[np.arange(g1 - g0), lab] = 1.0
    return HostBatch(np.concatenate(xs) if xs else np.zeros((0, f), np.float32), rowptr, colidx, None, gp, y)


def block_diag_plan(n=1_000_000, nnz=10_000_000, seed=2, mean_size=600):
    """Sizes and off-diagonal pair counts of every graph of the per-graph-seeded BASELINE config 3 batch: sizes as
    block_diag_batch, pairs split in proportion to the graph sizes (largest remainder), so that the batch holds
    exactly `nnz` entries whichever ranks build which graphs."""
    rng = np.random.default_rng(seed)
    sizes = _lognormal_sizes(rng, n, mean_size, 50, 3000)
    m_target = (nnz - n) // 2
    cap = sizes * (sizes - 1) // 2
    quota = m_target * sizes / sizes.sum()
    m = np.minimum(np.floor(quota).astype(np.int64), cap)
    order = np.argsort(-(quota - np.floor(quota)), kind="stable")
    i = 0
    while m.sum() < m_target:
        j = order[i % len(order)]
        if m[j] < cap[j]:
            m[j] += 1
        i += 1
    return sizes, m


def block_diag_shard(g0, g1, sizes, pairs, f=256, seed=2, with_x=True):
    """Graphs [g0, g1) of that batch: column indices uniform within the row's own graph, symmetric, self-loops,
    exactly pairs[g] off-diagonal pairs in graph g."""
    us, vs, off = [], [], 0
    for g in range(g0, g1):
        rng = _graph_rng(seed, g)
        n, m = int(sizes[g]), int(pairs[g])
        keys = np.empty(0, np.int64)
        while keys.size < m:
            need = int((m - keys.size) * 1.1) + 16
            u = rng.integers(0, n, size=need)
            v = rng.integers(0, n - 1, size=need)
            v += (v >= u)
            keys = np.unique(np.concatenate([keys, np.minimum(u, v) * n + np.maximum(u, v)]))
        if keys.size > m:
            keys = np.sort(rng.choice(keys, size=m, replace=False))
        us.append(keys // n + off); vs.append(keys % n + off)
        off += n
    rowptr, colidx = _csr_from_pairs(off, np.concatenate(us), np.concatenate(vs))
    gp = np.zeros(g1 - g0 + 1, dtype=np.int32)
    np.cumsum(sizes[g0:g1], out=gp[1:])
    if with_x:
        x = np.concatenate([_graph_rng(seed, g, 2).standard_normal((int(sizes[g]), f), dtype=np.float32) for g in range(g0, g1)])
    else:
        x = np.zeros((off, f), np.float32)
    lab = np.array([int(_graph_rng(seed, g, 1).integers(0, 2)) for g in range(g0, g1)])
    y = np.zeros((g1 - g0, 2), np.float32)
    y[np.arange(g1 - g0), lab] = 1.0
    return HostBatch(x, rowptr, colidx, None, gp, y)


# ---- config 5: power-law degrees, max degree 4096 ---------------------------------------------------

def power_law_batch(n_graphs=122, graph_size=8192, f=256, seed=3, alpha=2.0, max_deg=4096, with_x=True, first_graph=None):
    """BASELINE config 5: per-node target degree ~ Zipf(alpha) truncated to [1, max_deg];
    Chung-Lu pairing inside each 8192-node graph; symmetric + self-loops.
    first_graph (bench.py, sharded runs): graphs [first_graph, first_graph + n_graphs) of the per-graph-seeded batch --
    every graph from its own stream, the explicitly wired max-degree row in global graph 0."""
    if first_graph is not None:
        parts = []
        for g in range(first_graph, first_graph + n_graphs):
            hb = power_law_batch(1, graph_size, f, seed=[int(seed), g], alpha=alpha, max_deg=max_deg, with_x=with_x)
            if g != 0:      # only global graph 0 carries the explicit max-degree row: rebuild without it
                hb = _power_law_one(np.random.default_rng([int(seed), g]), graph_size, f, alpha, max_deg, with_x, wire=False)
            parts.append(hb)
        return concat_batches(parts)
    rng = np.random.default_rng(seed)
    n = n_graphs * graph_size
    k = np.arange(1, max_deg + 1, dtype=np.float64)
    pmf = k ** (-alpha)
    pmf /= pmf.sum()
    us, vs = [], []
    for g in range(n_graphs):
        deg = rng.choice(max_deg, size=graph_size, p=pmf) + 1
        if g == 0:
            deg[0] = 1                            # row 0 is wired explicitly below
        w = deg / deg.sum()
        m = int(deg.sum() // 2)
        u = rng.choice(graph_size, size=m, p=w)
        v = rng.choice(graph_size, size=m, p=w)
        keep = u != v
        us.append(u[keep] + g * graph_size); vs.append(v[keep] + g * graph_size)
        if g == 0:  # duplicates collapse in Chung-Lu pairing: wire one row to exactly max_deg entries
            nb = 1 + rng.choice(graph_size - 1, size=max_deg - 1, replace=False)
            us.append(np.zeros(max_deg - 1, dtype=np.int64)); vs.append(nb.astype(np.int64))
    rowptr, colidx = _csr_from_pairs(n, np.concatenate(us).astype(np.int64), np.concatenate(vs).astype(np.int64))
    gp = (np.arange(n_graphs + 1) * graph_size).astype(np.int32)
    x = rng.standard_normal((n, f), dtype=np.float32) if with_x else np.zeros((n, f), np.float32)
    return HostBatch(x, rowptr, colidx, None, gp, _labels(rng, n_graphs))


def _power_law_one(rng, graph_size, f, alpha, max_deg, with_x, wire):
    k = np.arange(1, max_deg + 1, dtype=np.float64)
    pmf = k ** (-alpha)
    pmf /= pmf.sum()
    deg = rng.choice(max_deg, size=graph_size, p=pmf) + 1
    if wire:
        deg[0] = 1
    w = deg / deg.sum()
    m = int(deg.sum() // 2)
    u = rng.choice(graph_size, size=m, p=w)
    v = rng.choice(graph_size, size=m, p=w)
    keep = u != v
    us, vs = [u[keep]], [v[keep]]
    if wire:
        nb = 1 + rng.choice(graph_size - 1, size=max_deg - 1, replace=False)
        us.append(np.zeros(max_deg - 1, dtype=np.int64)); vs.append(nb.astype(np.int64))
    rowptr, colidx = _csr_from_pairs(graph_size, np.concatenate(us).astype(np.int64), np.concatenate(vs).astype(np.int64))
    x = rng.standard_normal((graph_size, f), dtype=np.float32) if with_x else np.zeros((graph_size, f), np.float32)
    return HostBatch(x, rowptr, colidx, None, np.array([0, graph_size], np.int32), _labels(rng, 1))


def concat_batches(parts):
    """Disjoint union of HostBatches (block_diag of the adjacencies)."""
    noff = np.cumsum([0] + [p.n for p in parts])
    eoff = np.cumsum([0] + [p.nnz for p in parts])
    rowptr = np.concatenate([[0]] + [p.rowptr[1:].astype(np.int64) + eoff[i] for i, p in enumerate(parts)]).astype(np.int32)
    colidx = np.concatenate([p.colidx.astype(np.int64) + noff[i] for i, p in enumerate(parts)]).astype(np.int32)
    gp = np.concatenate([[0]] + [p.graph_ptr[1:].astype(np.int64) + noff[i] for i, p in enumerate(parts)]).astype(np.int32)
    vals = None if parts[0].vals is None else np.concatenate([p.vals for p in parts])
    return HostBatch(np.concatenate([p.x for p in parts]), rowptr, colidx, vals, gp, np.concatenate([p.y for p in parts]))


def spmm_algorithmic_bytes(n, nnz, f, weighted, elem=4):
    """4(N+1) + 4 nnz (+4 nnz weighted) + s N F (read H once) + s N F (write)."""
    return 4 * (n + 1) + 4 * nnz + (4 * nnz if weighted else 0) + 2 * elem * n * f
