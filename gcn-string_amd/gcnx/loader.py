"""Host mirror of the Spektral data surface the reference uses: Graph, Dataset, DisjointLoader.

Reference call sites: ``MyDataset(Dataset)`` src/scripts/gcn.py:66-197 builds
``Graph(x=, a=, y=)`` objects (gcn.py:153-157); ``DisjointLoader(data, batch_size=, epochs=,
shuffle=)`` gcn.py:316-317 is iterated at gcn.py:367 and stepped with ``__next__`` at
gcn.py:350; ``steps_per_epoch`` is read at gcn.py:348,372.  Semantics restated from Spektral
1.x; pure NumPy/SciPy, as upstream.
"""
from __future__ import annotations

import math
from collections import namedtuple

import numpy as np

# What tf.SparseTensor exposes and the model consumes: indices [nnz,2] int64 (row, col)
# row-major sorted, values [nnz], dense_shape (N, N).
SparseTensor = namedtuple("SparseTensor", ["indices", "values", "dense_shape"])


class Graph:
    """spektral.data.Graph: node features x [n,F], adjacency a (scipy sparse / dense [n,n]),
    optional edge features e, label y."""

    def __init__(self, x=None, a=None, e=None, y=None, **kwargs):
        self.x, self.a, self.e, self.y = x, a, e, y
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def n_nodes(self):
        return self.x.shape[0] if self.x is not None else self.a.shape[0]

    @property
    def n_node_features(self):
        return self.x.shape[-1]

    @property
    def n_labels(self):
        y = np.asarray(self.y)
        return y.shape[-1] if y.ndim else 1

    def numpy(self):
        return tuple(v for v in (self.x, self.a, self.e, self.y) if v is not None)


class Dataset:
    """spektral.data.Dataset: subclass and implement read() -> list[Graph] (gcn.py:84-102);
    supports len, integer / slice / index-array / boolean-mask indexing (gcn.py:282-294)."""

    def __init__(self, transforms=None, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)
        self.graphs = self.read()
        if transforms:
            for t in (transforms if isinstance(transforms, (list, tuple)) else [transforms]):
                self.apply(t)

    def read(self):
        raise NotImplementedError

    def apply(self, transform):
        self.graphs = [transform(g) for g in self.graphs]

    def __len__(self):
        return len(self.graphs)

    def __getitem__(self, key):
        if isinstance(key, (int, np.integer)):
            return self.graphs[int(key)]
        sub = object.__new__(type(self))
        sub.__dict__.update({k: v for k, v in self.__dict__.items() if k != "graphs"})
        if isinstance(key, slice):
            sub.graphs = self.graphs[key]
        else:
            key = np.asarray(key)
            idx = np.nonzero(key)[0] if key.dtype == bool else key
            sub.graphs = [self.graphs[int(i)] for i in idx]
        return sub

    def __iter__(self):
        return iter(self.graphs)

    @property
    def n_labels(self):
        return self.graphs[0].n_labels

    @property
    def n_node_features(self):
        return self.graphs[0].n_node_features


class ListDataset(Dataset):
    """A Dataset over an in-memory list of Graphs."""

    def __init__(self, graphs, **kwargs):
        self._graphs = list(graphs)
        super().__init__(**kwargs)

    def read(self):
        return self._graphs


def format_graph(graph):
    """MyDataset.format_graph (gcn.py:187-197): node labels -> integers 0..n-1 in the graph's own node order
    (nx.convert_node_labels_to_integers), and the ``weight`` edge attribute removed -- so that the adjacency the model
    sees is the 0/1 contact pattern, not the distances / DCA scores the offline builder stored on the edges."""
    import networkx as nx
    f = nx.convert_node_labels_to_integers(graph)
    for _, _, d in f.edges(data=True):
        d.pop("weight", None)
    return f


def from_networkx(graph, label, use_edge_data=False, feature_name="x"):
    """One networkx graph of the reference's pickles -> ``Graph(x, a, y)`` exactly as MyDataset builds it
    (gcn.py:104-128, 153-181, 187-197), without reading any file:
      a  nx.adjacency_matrix in node order, scipy CSR int64, 0/1 (weights stripped), the builder's self-loops kept
         (gcn_utills.py:224-227: the 0-Angstrom diagonal counts as a contact), symmetric for undirected graphs;
      x  np.vstack of every node's ``feature_name`` attribute in node order (float64 as stored);
      y  np.array(label) (the one-hot pair of gcn.py:259,262);
      e  the edge attributes as an [n_edges, n_attrs] array, attached only with use_edge_data (gcn.py:173-180 computes
         them either way and drops them by default)."""
    import networkx as nx
    import scipy.sparse as sp
    g = format_graph(graph)
    a = sp.csr_matrix(nx.adjacency_matrix(g)).astype(np.int64)
    a.sort_indices()
    x = np.vstack([feat for _, feat in g.nodes.data(feature_name)])
    y = np.array(label)
    if not use_edge_data:
        return Graph(x=x, a=a, y=y)
    attrs = [d for _, _, d in g.edges(data=True)]
    names = list(attrs[0].keys()) if attrs else []
    e = np.array([[d[k] for d in attrs] for k in names]).T if names else np.zeros((len(attrs), 0))
    return Graph(x=x, a=a, y=y, e=e)


class NetworkxDataset(Dataset):
    """MyDataset (gcn.py:66-197) over in-memory networkx graphs: ``NetworkxDataset(graphs, labels, n_samples=None)``
    reads the first n_samples graphs through from_networkx.  (The reference unpickles them from disk with
    nx.read_gpickle, which networkx 3 no longer has; loading pickles is the caller's business -- nothing here does.)"""

    def __init__(self, graphs, labels, n_samples=None, use_edge_data=False, **kwargs):
        self._nx = list(graphs)
        self.labels = list(labels)
        self.n_samples = len(self._nx) if n_samples is None else int(n_samples)
        self.use_edge_data = use_edge_data
        super().__init__(**kwargs)

    def read(self):
        return [from_networkx(self._nx[i], self.labels[i], self.use_edge_data) for i in range(self.n_samples)]


def to_disjoint(x_list, a_list):
    """spektral.data.utils.to_disjoint: x = vstack, a = block_diag, i = repeat(arange(B), n)."""
    import scipy.sparse as sp

    x = np.vstack(x_list)
    a = sp.block_diag(list(a_list))
    n_nodes = np.array([x_.shape[0] for x_ in x_list], dtype=np.int64)
    i = np.repeat(np.arange(len(n_nodes), dtype=np.int64), n_nodes)
    return x, a, i


def sp_matrix_to_sp_tensor(a):
    """spektral.layers.ops.sp_matrix_to_sp_tensor + tf.sparse.reorder: (row, col, val) of the
    non-zero entries in row-major order."""
    a = a.tocoo()
    keep = a.data != 0                      # sp.find drops explicit zeros
    row, col, val = a.row[keep], a.col[keep], a.data[keep]
    order = np.lexsort((col, row))
    idx = np.stack([row[order].astype(np.int64), col[order].astype(np.int64)], axis=1)
    return SparseTensor(idx, val[order], (int(a.shape[0]), int(a.shape[1])))


def _disjoint_coo(a_list):
    """block_diag + sp.find + tf.sparse.reorder of a list of scipy matrices in one pass: the (row, col, value) triples of
    the disjoint union in row-major order.  A graph's CSR rows are already in that order once its column indices are
    sorted, and the blocks follow each other, so the union is the concatenation of the per-graph triples shifted by the
    node offset -- no block_diag matrix, no 350 k-element lexsort (30 ms of the 50 a 32-graph E. coli batch took)."""
    import scipy.sparse as sp

    rows, cols, vals, off = [], [], [], 0
    for a in a_list:
        c = a if sp.isspmatrix_csr(a) else sp.csr_matrix(a)
        if not c.has_sorted_indices:
            c = c.sorted_indices()
        n = c.shape[0]
        r = np.repeat(np.arange(off, off + n, dtype=np.int64), np.diff(c.indptr))
        k = c.indices.astype(np.int64) + off
        v = c.data
        if v.size and not v.all():             # sp.find drops explicitly stored zeros
            keep = v != 0
            r, k, v = r[keep], k[keep], v[keep]
        rows.append(r); cols.append(k); vals.append(v)
        off += n
    if not rows:
        return SparseTensor(np.zeros((0, 2), np.int64), np.zeros(0), (0, 0))
    idx = np.stack([np.concatenate(rows), np.concatenate(cols)], axis=1)
    return SparseTensor(idx, np.concatenate(vals), (off, off))


def collate_disjoint(graphs, node_level=False):
    """DisjointLoader.collate: the same ((x, a, i), y) as to_disjoint + sp_matrix_to_sp_tensor, built directly.
    Graphs that carry edge features (Graph(e=...): the reference computes them either way and attaches them only with
    use_edge_data, gcn.py:173-180) yield ((x, a, e, i), y) as Spektral does -- e = vstack of the graphs' [n_edges, S]
    arrays, a dense [n, n, S] array reduced to the adjacency's stored entries first (to_disjoint).  The models of this
    package take (x, a, i), like spektral.models.GeneralGNN: a batch with e is for the caller's own layers."""
    x_list = [g.x for g in graphs]
    x = np.vstack(x_list)
    n_nodes = np.array([x_.shape[0] for x_ in x_list], dtype=np.int64)
    i = np.repeat(np.arange(len(n_nodes), dtype=np.int64), n_nodes)
    y = np.vstack([g.y for g in graphs]) if node_level else np.array([g.y for g in graphs])
    a = _disjoint_coo([g.a for g in graphs])
    if graphs and all(getattr(g, "e", None) is not None for g in graphs):
        import scipy.sparse as sp
        e_list = [np.asarray(g.e) for g in graphs]
        if e_list[0].ndim == 3:                  # dense [n, n, S] -> the rows of the stored entries (sp.find order, as upstream)
            e_list = [e[sp.find(g.a)[:-1]] for e, g in zip(e_list, graphs)]
        return (x, a, np.vstack(e_list), i), y
    return (x, a, i), y


class DisjointLoader:
    """spektral.data.DisjointLoader(dataset, node_level=False, batch_size=1, epochs=None,
    shuffle=True): an iterator that yields ``((x, a, i), y)`` per batch.

    Each epoch optionally shuffles the graph order, then yields consecutive slices of
    batch_size graphs (the last one may be smaller); epochs=None iterates forever, as the
    reference's evaluate() relies on (gcn.py:317,348-350).
    """

    def __init__(self, dataset, node_level=False, batch_size=1, epochs=None, shuffle=True, seed=None):
        self.dataset, self.node_level = dataset, node_level
        self.batch_size, self.epochs, self.shuffle = int(batch_size), epochs, shuffle
        self._rng = np.random.default_rng(seed) if seed is not None else np.random
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
                yield [self.dataset[int(j)] for j in order[s:s + self.batch_size]]
            epoch += 1

    def __iter__(self):
        return self

    def __next__(self):
        return collate_disjoint(next(self._gen), self.node_level)

    def collate(self, batch):
        return collate_disjoint(batch, self.node_level)

    def load(self):
        return self

    def tf_signature(self):
        """Shape/dtype description of one batch (the reference feeds it to tf.function,
        gcn.py:328); plain tuples here since there is no TensorFlow."""
        g = self.dataset[0]
        f = g.n_node_features
        inputs = [("x", (None, f), np.float64), ("a", (None, None), "sparse")]
        if getattr(g, "e", None) is not None:
            inputs.append(("e", (None, np.asarray(g.e).shape[-1]), np.asarray(g.e).dtype))
        inputs.append(("i", (None,), np.int64))
        return (tuple(inputs), ("y", (None, g.n_labels), np.asarray(g.y).dtype))
